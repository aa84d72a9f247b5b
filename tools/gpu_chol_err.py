import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_hip_parity import _problem, _oracle64
from conftest import rel_fro
from rsparse_amd import als
for k in (64, 128):
    for seed in (90 + k, 7, 8):
        csc, X, Y0 = _problem(700, 500, k, seed=seed, scale=0.3)
        Yref, lref = _oracle64(csc, X, Y0, 0.1, 0, 3, True)
        errs = []
        for rep in range(3):
            Y = Y0.copy(order="F")
            als.als_implicit(csc, X, Y, 0.1, 1, 0, 3, "float", False, False)
            err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
            errs.append((rel_fro(Y, Yref), float(err.max()), int((err > 1e-4).sum())))
        print("k", k, "seed", seed, [("%.2e" % a, "%.2e" % b, c) for a, b, c in errs], flush=True)
