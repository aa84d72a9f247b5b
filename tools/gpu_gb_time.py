"""implicit feedback with a global bias (cg_solver_implicit_global_bias): ms per ALS iteration inside WRMF.fit_transform at 1M x 100k,
ranks 64 and 128, next to the same fit without the global bias.   python tools/gpu_gb_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

from rsparse_amd import WRMF, synth

d = synth.make_dataset(1_000_000, 100_000, device="cpu", feedback="implicit")
p, i, x = (t.numpy() for t in d["c_iu"])
m = sp.csc_matrix((x.astype(np.float64), i, p), shape=(100_000, 1_000_000)).T.tocsr()
for rank in (64, 128):
    for gb in (False, True):
        model = WRMF(rank=rank, lambda_=0.1, feedback="implicit", solver="conjugate_gradient", precision="float", rng=1, with_global_bias=gb)
        model.fit_transform(m, n_iter=1, convergence_tol=-1)
        ts = []
        for n_iter in (1, 11):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.fit_transform(m, n_iter=n_iter, convergence_tol=-1)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print("rank %3d global bias %-5s  %.1f ms per iteration" % (rank, gb, 100 * (ts[1] - ts[0])), flush=True)
