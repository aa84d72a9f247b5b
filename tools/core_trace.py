#!/usr/bin/env python3
"""Per-iteration trace (GPU box) for the cells of tests/test_wrmf_core.py whose error exceeds 1e-4: after n = 1..5
iterations, the device fit and the oracle-in-float fit against the fp64 oracle (item factors, relative Frobenius) --
shows WHERE the fp32 arithmetics part from fp64 and that the device parts no earlier or faster than the fp32 oracle."""
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import warnings  # noqa: E402

from conftest import csc_take_rows, load_movielens, rel_fro  # noqa: E402
from oracle import wrmf_oracle as O  # noqa: E402
from rsparse_amd import WRMF  # noqa: E402

warnings.simplefilter("ignore")
n_user_all, n_item, p, i, x = load_movielens()
tp, ti, tx = csc_take_rows(900, p, i, x)
n_user = 900
train = sp.csc_matrix((tx, ti, tp), shape=(n_user, n_item))
CELLS = [("implicit", "nnls", 0.1, True, "double"), ("implicit", "nnls", 0.0, True, "double"),
         ("implicit", "nnls", 0.0, True, "float"), ("explicit", "conjugate_gradient", 0.1, True, "float"),
         ("explicit", "conjugate_gradient", 1000.0, False, "double"), ("explicit", "nnls", 0.1, False, "float")]
for feedback, solver, lam, bias, precision in CELLS:
    seed = sum(ord(c) for c in feedback + solver + precision) + int(lam * 10) + 7 * bias
    rng = np.random.default_rng(seed)
    rank0, K = int(rng.integers(4, 11)), int(rng.integers(4, 11))
    rank = rank0 + 2 * bias
    U0 = (rng.standard_normal((n_user, rank)) * 0.01).astype(np.float32)
    V0 = (rng.standard_normal((rank, n_item)) * 0.01).astype(np.float32)
    print("%s | %s | lambda %g | biases %d | %s | rank %d" % (feedback, solver, lam, bias, precision, rank0))
    print("  iterations   device: items  users(emb)     fp32 oracle: items  users(emb)")
    for n_iter in range(1, 6):
        out = {}
        for dt in (np.float64, np.float32):
            ref = O.OracleWRMF(rank0, lam=lam, feedback=feedback, solver=solver, dtype=dt, n_threads=8, with_user_item_bias=bias)
            emb = ref.fit_transform(n_user, n_item, tp, ti, tx, U0.T.astype(dt), n_iter=n_iter, convergence_tol=-1,
                                    init_components=None if solver == "conjugate_gradient" else V0.astype(dt))
            out[dt] = (ref.components.copy(), emb.copy())
        init = None if solver == "conjugate_gradient" else V0.astype(np.float64 if precision == "double" else np.float32)
        m = WRMF(rank=rank0, lambda_=lam, feedback=feedback, solver=solver, with_user_item_bias=bias, precision=precision, init=init)
        m._init_user_factors = U0
        emb = m.fit_transform(train, n_iter=n_iter, convergence_tol=-1)
        a, b = out[np.float64], out[np.float32]
        print("  %d            %.2e  %.2e                  %.2e  %.2e" % (n_iter, rel_fro(m.components, a[0]), rel_fro(emb, a[1]),
                                                                           rel_fro(b[0], a[0]), rel_fro(b[1], a[1])))
