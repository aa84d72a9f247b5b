"""with_user_item_bias fits: the solves run at rank + 1 (the reference's rank + 2 minus the dropped bias row).
python tools/gpu_bias_time.py  -> seconds per fit for ranks whose rank + 1 is / is not a multiple of 4"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

from rsparse_amd import WRMF, synth


def main():
    users, items = 1_000_000, 100_000
    d = synth.make_dataset(users, items, device="cpu", feedback="explicit")
    p, i, x = (t.numpy() for t in d["c_iu"])
    m = sp.csc_matrix((x.astype(np.float64), i, p), shape=(items, users)).T.tocsr()
    for feedback, solver in (("explicit", "conjugate_gradient"), ("explicit", "cholesky")):
        for rank in (62, 63, 64, 30, 31):
            for bias in (False, True):
                model = WRMF(rank=rank, lambda_=0.1, feedback=feedback, solver=solver, precision="float", rng=1,
                             with_user_item_bias=bias, with_global_bias=bias)
                model.fit_transform(m, n_iter=1, convergence_tol=-1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model.fit_transform(m, n_iter=3, convergence_tol=-1)
                torch.cuda.synchronize()
                print("%s %-18s rank %3d bias %-5s  %.3f s per fit (3 iterations + final solve)" % (feedback, solver, rank, bias, time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    main()
