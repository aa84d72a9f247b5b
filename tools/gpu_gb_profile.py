"""one WRMF fit with the implicit global bias (rank given, 1M x 100k, 5 iterations) -- run under rocprofv3 --kernel-trace --stats"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from rsparse_amd import WRMF, synth

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = synth.make_dataset(1_000_000, 100_000, device="cpu", feedback="implicit")
p, i, x = (t.numpy() for t in d["c_iu"])
m = sp.csc_matrix((x.astype(np.float64), i, p), shape=(100_000, 1_000_000)).T.tocsr()
model = WRMF(rank=rank, lambda_=0.1, feedback="implicit", solver="conjugate_gradient", precision="float", rng=1, with_global_bias=True)
model.fit_transform(m, n_iter=5, convergence_tol=-1)
print("losses", model.losses[-1])
