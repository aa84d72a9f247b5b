"""NNLS corners: seconds per ALS iteration inside WRMF.fit_transform at 200k x 20k for ranks either side of 64, with and without
user/item biases (the solves run at rank + 1 with biases).   python tools/gpu_nnls_corner.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

from rsparse_amd import WRMF, synth

d = synth.make_dataset(200_000, 20_000, device="cpu", feedback="implicit")
p, i, x = (t.numpy() for t in d["c_iu"])
m = sp.csc_matrix((x.astype(np.float64), i, p), shape=(20_000, 200_000)).T.tocsr()
for rank, bias in ((64, False), (68, False), (63, True), (64, True), (128, False)):
    model = WRMF(rank=rank, lambda_=0.1, feedback="implicit", solver="nnls", precision="float", rng=1, with_user_item_bias=bias)
    t0 = time.perf_counter()
    model.fit_transform(m, n_iter=1, convergence_tol=-1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    model.fit_transform(m, n_iter=3, convergence_tol=-1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rank %3d bias %-5s  first fit of 1 iteration %.2f s, fit of 3 iterations %.2f s, losses %s" % (rank, bias, t1 - t0, t2 - t1, [round(l[1], 4) for l in model.losses]), flush=True)
