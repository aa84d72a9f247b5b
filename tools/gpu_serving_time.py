"""latency of the serving calls after a fit: WRMF.transform / WRMF.predict of a batch of new users (1M x 100k model, rank 128)."""
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
for rank, precision in ((128, "float"), (10, "double")):
    model = WRMF(rank=rank, lambda_=0.1, feedback="implicit", solver="conjugate_gradient", precision=precision, rng=1)
    model.fit_transform(m, n_iter=2, convergence_tol=-1)
    for nb in (1, 1000, 100_000):
        xb = m[:nb]
        for what in ("transform", "predict"):
            f = (lambda: model.transform(xb)) if what == "transform" else (lambda: model.predict(xb, k=10))
            f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            print("rank %3d %-6s %-9s of %6d users: %.2f ms per call" % (rank, precision, what, nb, 1e3 * (time.perf_counter() - t0) / reps), flush=True)
