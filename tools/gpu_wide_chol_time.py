"""the exact solver (and one NNLS case) at the wide system orders (129..256): ms per ALS iteration inside WRMF.fit_transform at
200k x 50k (1e7 non-zeros) -- rank 128 with user/item biases (a system of order 129, padded to 132), ranks 160 and 256.
RSPARSE_HIP_LIB picks the library build.   python tools/gpu_wide_chol_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

from rsparse_amd import WRMF, synth

NU, NI = 200_000, 50_000
mats = {}
for fb in ("implicit", "explicit"):
    d = synth.make_dataset(NU, NI, device="cpu", feedback=fb)
    p, i, x = (t.numpy() for t in d["c_iu"])
    mats[fb] = sp.csc_matrix((x.astype(np.float64), i, p), shape=(NI, NU)).T.tocsr()
cases = (("implicit", 128, True, "cholesky"), ("explicit", 128, True, "cholesky"), ("implicit", 160, False, "cholesky"),
         ("implicit", 256, False, "cholesky"), ("explicit", 128, True, "conjugate_gradient"), ("implicit", 128, False, "cholesky"))
if len(sys.argv) > 1:   # a filter: indices of the cases, e.g. "0" or "0,2"
    cases = tuple(cases[int(c)] for c in sys.argv[1].split(","))
for fb, rank, bias, solver in cases:
    model = WRMF(rank=rank, lambda_=0.1, feedback=fb, solver=solver, precision="float", rng=1, with_user_item_bias=bias)
    model.fit_transform(mats[fb], n_iter=1, convergence_tol=-1)
    ts = []
    for n_iter in (1, 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit_transform(mats[fb], n_iter=n_iter, convergence_tol=-1)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("%-8s %-18s rank %3d bias %-5s  %8.1f ms per iteration (fit of 3 iterations %.2f s)  loss %.6f"
          % (fb, solver, rank, bias, 500 * (ts[1] - ts[0]), ts[1], model.losses[-1][1]), flush=True)
