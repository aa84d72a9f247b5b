"""the fp64 conjugate-gradient wave kernel (f64_cg_wave_kernel) on rows of 0..700 non-zeros (several 64-non-zero chunks: the
cross-chunk prefetch) at ranks 3..64, implicit and explicit, against the fp64 oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from oracle import wrmf_oracle as O
from rsparse_amd import als

worst = 0.0
for trial in range(24):
    rng = np.random.default_rng(500 + trial)
    implicit = trial % 2 == 0
    k = int(rng.choice([3, 10, 16, 17, 24, 32, 33, 50, 64]))
    n_rows = int(rng.integers(1, 120))
    lens = rng.integers(0, int(rng.choice([5, 70, 130, 260, 700])), size=n_rows)
    n_item = 900
    p = np.zeros(n_rows + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
    idx = np.concatenate([np.sort(rng.choice(n_item, size=int(n), replace=False)) for n in lens] or [np.zeros(0)]).astype(np.int32)
    x = (1.0 + rng.gamma(1.0, 2.0, size=idx.size)) if implicit else np.round(1.0 + 4.0 * rng.random(idx.size))
    X = np.asfortranarray(rng.standard_normal((k, n_item)) * 0.3)
    Y0 = np.asfortranarray(rng.standard_normal((k, n_rows)) * 0.3)
    csc = (n_item, n_rows, p, idx, x)
    lam, dyn, steps = 0.1, bool(trial % 3 == 0), int(rng.choice([0, 1, 3, 5]))
    cnt = np.bincount(idx, minlength=n_item).astype(np.float64)
    Yr = Y0.copy(order="F"); Y = Y0.copy(order="F")
    if implicit:
        lref = O.als_implicit(p, idx, x, X, Yr, O.gramian(X, lam), lam, 1, steps)
        loss = als.als_implicit(csc, X, Y, lam, 1, 1, steps, "double", False, False)
    else:
        lref = O.als_explicit(p, idx, x, X, Yr, cnt, lam, 1, steps, dyn)
        loss = als.als_explicit(csc, X, Y, cnt, lam, 1, 1, steps, dyn, "double", False, False)
    err = np.linalg.norm(Y - Yr, axis=0) / np.maximum(np.linalg.norm(Yr, axis=0), 1e-300)
    e = float(err.max()); le = abs(loss - lref) / max(abs(lref), 1e-300)
    worst = max(worst, e)
    print("trial %2d %s k %2d rows %3d max len %3d steps %d: row err %.2e loss err %.2e%s" % (
        trial, "implicit" if implicit else "explicit", k, n_rows, int(lens.max(initial=0)), steps, e, le, "" if e < 1e-9 and le < 1e-9 else "   <-- CHECK"), flush=True)
print("worst", worst)
