"""per-row error of the Cholesky half-iteration against the fp64 oracle, by row length (debug aid for wrmf_chol_lr.hip)"""
import sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_hip_parity as T
from rsparse_amd import als
for (n_u, n_i) in [(700, 500), (3000, 200)]:
    csc, X, Y0 = T._problem(n_u, n_i, 128, seed=90 + 128, feedback="implicit", scale=0.3)
    lam = 0.1
    cnt = np.diff(csc[2])
    Yref, lref = T._oracle64(csc, X, Y0, lam, 0, 3, True, True, cnt.astype(np.float64))
    Y = Y0.copy(order="F")
    loss = als.als_implicit(csc, X, Y, lam, 1, 0, 3, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    print("problem", n_u, n_i, "cols", Y.shape, "max err", err.max(), "loss", loss, lref)
    order = np.argsort(-cnt, kind="stable")
    for lo, hi in [(0, 1), (1, 17), (17, 33), (33, 49), (49, 65), (65, 10**9)]:
        m = (cnt >= lo) & (cnt < hi)
        if m.any():
            print("  len [%d,%d): rows %d, max err %.3e, bad rows %d" % (lo, hi, m.sum(), err[m].max(), (err[m] > 1e-4).sum()))
    bad = np.where(err > 1e-4)[0]
    pos = {r: i for i, r in enumerate(order)}
    print("  bad rows (row, len, position in the longest-first order, err):", [(int(r), int(cnt[r]), pos[r], float("%.2e" % err[r])) for r in bad[:24]])
