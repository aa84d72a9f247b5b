"""dev: the global-bias CG half-iteration on tests/test_bias.py's long-row problem, per long row: error against the fp64 oracle, with
and without the global bias, ranks 64 / 128 (a row of 2300 non-zeros is cut into two segments since the fine lists take the machine's slots)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from rsparse_amd import als
from oracle import wrmf_oracle as O
from test_bias import _long_row_problem

for k in (64, 128):
    for gb in (0.037, 0.0):
        m, X32, Y32 = _long_row_problem(900 + k, k)
        p, i, x = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
        lam = 0.1
        X64, Y64 = np.asfortranarray(X32, dtype=np.float64), np.asfortranarray(Y32, dtype=np.float64).copy(order="F")
        lref = O.als_implicit(p, i, x, X64, Y64, O.gramian(X64, lam), lam, 1, 3, global_bias=gb, n_threads=8)
        Y = Y32.copy(order="F")
        base = np.zeros(k - 1, dtype=np.float32)
        csc = (m.shape[0], m.shape[1], p, i, x)
        loss = als.als_implicit(csc, X32, Y, lam, 1, 1, 3, "float", False, True, initialize_bias_base=True, global_bias=gb, global_bias_base=base)
        norm = np.maximum(np.linalg.norm(Y64, axis=0), 1e-30)
        err = np.linalg.norm(Y - Y64, axis=0) / norm
        lens = np.diff(p)
        print("k", k, "gb", gb, "loss rel", abs(loss - lref) / abs(lref), "long rows:", [(int(lens[c]), float("%.2e" % err[c])) for c in np.where(lens > 512)[0]], "max other", float(err[lens <= 512].max()), flush=True)
