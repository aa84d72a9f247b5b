"""fp64 layer (wrmf_f64.hip) and wide ranks (wrmf_wide.hip): what the parity / coverage paths cost -- WRMF fits on a synthetic
matrix at the given size, iterations/s per (precision, rank, solver).  python tools/gpu_f64_time.py [--users N --items M]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import scipy.sparse as sp
import torch

from rsparse_amd import WRMF, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=200_000)
    ap.add_argument("--items", type=int, default=50_000)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    d = synth.make_dataset(a.users, a.items, device="cpu", feedback="implicit")
    p, i, x = (t.numpy() for t in d["c_iu"])          # columns = users
    m = sp.csc_matrix((x.astype(np.float64), i, p), shape=(a.items, a.users)).T.tocsr()
    out = {"what": "WRMF.fit_transform, implicit feedback, synthetic %d x %d, %d nnz" % (a.users, a.items, m.nnz), "runs": []}
    for precision, rank, solver in (("float", 32, "conjugate_gradient"), ("double", 32, "conjugate_gradient"),
                                    ("float", 32, "cholesky"), ("double", 32, "cholesky"),
                                    ("float", 128, "conjugate_gradient"), ("float", 160, "conjugate_gradient"),
                                    ("float", 256, "conjugate_gradient"), ("float", 256, "cholesky")):
        model = WRMF(rank=rank, lambda_=0.1, feedback="implicit", solver=solver, precision=precision, rng=1)
        model.fit_transform(m, n_iter=1, convergence_tol=-1)      # upload, schedules, first touch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit_transform(m, n_iter=a.iters, convergence_tol=-1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # (a fit is n_iter iterations + the final exact user half-iteration + the upload: the figure is the whole call)
        out["runs"].append({"precision": precision, "rank": rank, "solver": solver, "seconds_per_fit": round(dt, 3),
                            "iterations": a.iters, "fits_iterations_per_s": round(a.iters / dt, 3)})
        print(out["runs"][-1], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
