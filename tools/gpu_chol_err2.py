import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_hip_parity import _problem, _oracle64
from rsparse_amd import als
k = 128
csc, X, Y0 = _problem(700, 500, k, seed=218, scale=0.3)
lens = np.diff(csc[2])
Yref, _ = _oracle64(csc, X, Y0, 0.1, 0, 3, True)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    Y = Y0.copy(order="F")
    als.als_implicit(csc, X, Y, 0.1, 1, 0, 3, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    bad = np.nonzero(err > 1e-4)[0]
    # which elements of the bad rows are wrong
    info = []
    for b in bad[:4]:
        d = np.abs(Y[:, b] - Yref[:, b]) / np.abs(Yref[:, b]).max()
        info.append((int(b), int(lens[b]), "%.1e" % err[b], "first bad elem %d, n bad elems %d" % (int(np.argmax(d > 1e-4)), int((d > 1e-4).sum()))))
    if len(bad): print(rep, len(bad), info, flush=True)
print("done")
