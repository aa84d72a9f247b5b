"""dev: which split rows go wrong at rank 64 -- variants of the long-row problem (no global bias), error of every row beyond 512 non-zeros"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
from rsparse_amd import als
from oracle import wrmf_oracle as O


def problem(lens, k, n_rows, seed=1):
    rng = np.random.default_rng(seed)
    cols, rows = [], []
    for c, n in enumerate(lens):
        rows.append(np.sort(rng.choice(n_rows, size=int(n), replace=False)))
        cols.append(np.full(int(n), c))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = 1.0 + rng.geometric(0.5, size=rows.size).astype(np.float64)
    m = sp.csc_matrix((vals, (rows, cols)), shape=(n_rows, len(lens)))
    m.sort_indices()
    X = np.asfortranarray((rng.standard_normal((k, n_rows)) * 0.1).astype(np.float32))
    Y = np.asfortranarray((rng.standard_normal((k, len(lens))) * 0.1).astype(np.float32))
    return m, X, Y


rng0 = np.random.default_rng(0)
short = list(rng0.integers(1, 90, 240))
variants = {
    "base 513,700,1100,2300": [0, 1, 2, 31] + [513, 700, 1100, 2300] + short,
    "2304": [513, 700, 1100, 2304] + short,
    "4096": [513, 700, 1100, 4096] + short,
    "2300 first": [2300, 513, 700, 1100] + short,
    "2300 alone": [2300] + short,
    "two of 2300": [2300, 2300, 600] + short,
    "2300 + 600 ordinary": [2300] + list(rng0.integers(520, 900, 600)) + short,
    "6000 (3 parts?)": [6000, 513] + short,
}
k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, lens in variants.items():
    m, X32, Y32 = problem(lens, k, 8000)
    p, i, x = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
    lam = 0.1
    X64, Y64 = np.asfortranarray(X32, dtype=np.float64), np.asfortranarray(Y32, dtype=np.float64).copy(order="F")
    lref = O.als_implicit(p, i, x, X64, Y64, O.gramian(X64, lam), lam, 1, 3, n_threads=8)
    Y = Y32.copy(order="F")
    loss = als.als_implicit((m.shape[0], m.shape[1], p, i, x), X32, Y, lam, 1, 1, 3, "float", False, False)
    err = np.linalg.norm(Y - Y64, axis=0) / np.maximum(np.linalg.norm(Y64, axis=0), 1e-30)
    ln = np.diff(p)
    print("%-24s loss rel %.1e  long rows:" % (name, abs(loss - lref) / abs(lref)), [(int(ln[c]), float("%.1e" % err[c])) for c in np.where(ln > 512)[0]][:8], "max other %.1e" % err[ln <= 512].max(), flush=True)
