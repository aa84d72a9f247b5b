"""one precision = "double" fit (rank argv[1], n_iter argv[2]) of the 1M x 100k implicit matrix: run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split of an fp64 iteration (users' side / items' side / Gramian)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from rsparse_amd import WRMF, synth

rank, n_iter = int(sys.argv[1]), int(sys.argv[2])
d = synth.make_dataset(1_000_000, 100_000, device="cpu", feedback="implicit")
p, i, x = (t.numpy() for t in d["c_iu"])
m = sp.csc_matrix((x.astype(np.float64), i, p), shape=(100_000, 1_000_000)).T.tocsr()
model = WRMF(rank=rank, lambda_=0.1, feedback="implicit", solver="conjugate_gradient", precision="double", rng=1)
model.fit_transform(m, n_iter=n_iter, convergence_tol=-1)
print("ok")
