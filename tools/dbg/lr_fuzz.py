"""random shapes through the wave-per-pass low-rank kernels (implicit rank 128, explicit ranks 64 / 128) against the fp64 oracle:
row counts that are not multiples of the packing, empty classes, confidences at 1, tiny and large factor scales."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np

from oracle import wrmf_oracle as O
from rsparse_amd import als

worst = 0.0
for trial in range(40):
    rng = np.random.default_rng(1000 + trial)
    implicit = trial % 2 == 0
    k = 128 if implicit else (64 if trial % 4 == 1 else 128)
    n_rows = int(rng.integers(1, 400))
    hi = int(rng.choice([5, 17, 33, 49, 66, 90]))
    lens = rng.integers(0, hi, size=n_rows)
    n_item = 300
    p = np.zeros(n_rows + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
    idx = np.concatenate([np.sort(rng.choice(n_item, size=int(n), replace=False)) for n in lens] or [np.zeros(0)]).astype(np.int32)
    if implicit:
        x = (1.0 + rng.gamma(1.0, 2.0, size=idx.size)).astype(np.float32).astype(np.float64)
        x[rng.random(x.size) < rng.choice([0.0, 0.5, 0.9])] = 1.0
    else:
        x = np.round(1.0 + 4.0 * rng.random(idx.size))
    scale = float(rng.choice([1e-3, 0.1, 1.0, 10.0]))
    X = np.asfortranarray((rng.standard_normal((k, n_item)) * scale).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, n_rows)) * scale).astype(np.float32))
    csc = (n_item, n_rows, p, idx, x)
    lam = float(rng.choice([0.01, 0.1, 10.0]))
    dyn = bool(trial % 3 == 0)
    cnt = np.bincount(idx, minlength=n_item).astype(np.float64)
    X64 = np.asfortranarray(X, dtype=np.float64)
    Yr = np.asfortranarray(Y0, dtype=np.float64).copy(order="F")
    Y = Y0.copy(order="F")
    if implicit:
        lref = O.als_implicit(p, idx, x, X64, Yr, O.gramian(X64, lam), lam, 0, 3)
        loss = als.als_implicit(csc, X, Y, lam, 1, 0, 3, "float", False, False)
    else:
        lref = O.als_explicit(p, idx, x, X64, Yr, cnt, lam, 0, 3, dyn)
        loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), lam, 1, 0, 3, dyn, "float", False, False)
    # the fp32 oracle on the same system: what plain float arithmetic does with its conditioning
    Y32 = Y0.copy(order="F")
    if implicit:
        O.als_implicit(p, idx, x, X, Y32, O.gramian(X, lam), lam, 0, 3)
    else:
        O.als_explicit(p, idx, x, X, Y32, cnt.astype(np.float32), lam, 0, 3, dyn)
    nz = lens > 0
    err = np.linalg.norm(Y - Yr, axis=0) / np.maximum(np.linalg.norm(Yr, axis=0), 1e-30)
    err32 = np.linalg.norm(Y32 - Yr, axis=0) / np.maximum(np.linalg.norm(Yr, axis=0), 1e-30)
    e = float(err[nz].max()) if nz.any() else 0.0
    e32 = float(err32[nz].max()) if nz.any() else 0.0
    wr = int(np.argmax(np.where(nz, err, -1)))
    le = abs(loss - lref) / max(abs(lref), 1e-30)
    worst = max(worst, e)
    flag = "" if (e < max(1e-4, 3 * e32) and le < 1e-4 and np.isfinite(Y).all()) else "   <-- CHECK"
    print("trial %2d %s k %3d rows %3d max len %2d scale %g lambda %g dyn %d: row err %.2e (fp32 oracle %.2e; worst row has %d nnz) loss err %.2e%s" % (
        trial, "implicit" if implicit else "explicit", k, n_rows, hi - 1, scale, lam, dyn, e, e32, int(lens[wr]), le, flag), flush=True)
print("worst row error", worst)
