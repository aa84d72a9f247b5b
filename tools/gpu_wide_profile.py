"""one WRMF fit at a wide system order (explicit CG, rank 128 + biases = order 129 -> 132; or `rank` given), 1M x 100k, 3 iterations -- under rocprofv3"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from rsparse_amd import WRMF, synth

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bias = rank == 128
fb = "explicit"
d = synth.make_dataset(1_000_000, 100_000, device="cpu", feedback=fb)
p, i, x = (t.numpy() for t in d["c_iu"])
m = sp.csc_matrix((x.astype(np.float64), i, p), shape=(100_000, 1_000_000)).T.tocsr()
model = WRMF(rank=rank, lambda_=0.1, feedback=fb, solver="conjugate_gradient", precision="float", rng=1, with_user_item_bias=bias)
model.fit_transform(m, n_iter=3, convergence_tol=-1)
print("losses", model.losses[-1])
