"""where a WRMF.fit_transform call spends its time outside the solves (host-side conversions, upload, schedules): cProfile of one
call at 1M x 100k, 5e7 non-zeros.  python tools/gpu_fit_profile.py"""
import cProfile
import os
import pstats
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
model = WRMF(rank=int(os.environ.get("RANK", "64")), lambda_=0.1, feedback="implicit", solver="conjugate_gradient", precision="float", rng=1)
model.fit_transform(m, n_iter=1, convergence_tol=-1)
torch.cuda.synchronize()
for n_iter in (1, 10):
    t0 = time.perf_counter()
    model.fit_transform(m, n_iter=n_iter, convergence_tol=-1)
    torch.cuda.synchronize()
    print("fit_transform n_iter=%d: %.3f s" % (n_iter, time.perf_counter() - t0))
pr = cProfile.Profile()
pr.enable()
model.fit_transform(m, n_iter=1, convergence_tol=-1)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
