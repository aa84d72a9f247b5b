"""Random shapes through the kernels added at the end of round 4, against the oracle (seeded: the cases are fixed).

* the wave-per-pass low-rank exact solves (wrmf_chol_lr.hip: implicit rank 128, explicit ranks 64 / 128) -- row counts that are
  not multiples of the packing, empty classes, confidences at exactly 1, factor scales 1e-3..10, lambda 0.01..10, rows on both
  sides of the 64-non-zero boundary.  Bound: max(1e-4, 3 x the fp32 oracle's own distance from the fp64 oracle) per row -- with
  lambda = 0.01 and factors of size 10 plain fp32 arithmetic on the k x k system is off by 1e-1, the device is not;
* the fp64 conjugate-gradient wave kernel (wrmf_f64.hip): rows of 0..700 non-zeros (several 64-non-zero chunks), ranks 3..128 (65..128: two coordinates per lane, round 5),
  0..5 CG steps.  Bound: 1e-9 per row.
"""
import numpy as np
import pytest

from oracle import wrmf_oracle as O
from rsparse_amd import als

pytestmark = pytest.mark.gpu


def _rows(rng, n_rows, hi, n_item):
    lens = rng.integers(0, hi, size=n_rows)
    p = np.zeros(n_rows + 1, dtype=np.int32)
    p[1:] = np.cumsum(lens)
    idx = np.concatenate([np.sort(rng.choice(n_item, size=int(n), replace=False)) for n in lens] or [np.zeros(0)]).astype(np.int32)
    return lens, p, idx


@pytest.mark.parametrize("trial", range(16))
def test_low_rank_wave_kernels_on_random_shapes(trial):
    rng = np.random.default_rng(1000 + trial)
    implicit = trial % 2 == 0
    k = 128 if implicit else (64 if trial % 4 == 1 else 128)
    n_rows, n_item = int(rng.integers(1, 400)), 300
    lens, p, idx = _rows(rng, n_rows, int(rng.choice([5, 17, 33, 49, 66, 90])), n_item)
    if implicit:
        x = (1.0 + rng.gamma(1.0, 2.0, size=idx.size)).astype(np.float32).astype(np.float64)
        x[rng.random(x.size) < rng.choice([0.0, 0.5, 0.9])] = 1.0
    else:
        x = np.round(1.0 + 4.0 * rng.random(idx.size))
    scale = float(rng.choice([1e-3, 0.1, 1.0, 10.0]))
    X = np.asfortranarray((rng.standard_normal((k, n_item)) * scale).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, n_rows)) * scale).astype(np.float32))
    lam, dyn = float(rng.choice([0.01, 0.1, 10.0])), trial % 3 == 0
    cnt = np.bincount(idx, minlength=n_item).astype(np.float64)
    X64 = np.asfortranarray(X, dtype=np.float64)
    Yr, Y32, Y = np.asfortranarray(Y0, dtype=np.float64).copy(order="F"), Y0.copy(order="F"), Y0.copy(order="F")
    csc = (n_item, n_rows, p, idx, x)
    if implicit:
        lref = O.als_implicit(p, idx, x, X64, Yr, O.gramian(X64, lam), lam, 0, 3)
        O.als_implicit(p, idx, x, X, Y32, O.gramian(X, lam), lam, 0, 3)
        loss = als.als_implicit(csc, X, Y, lam, 1, 0, 3, "float", False, False)
    else:
        lref = O.als_explicit(p, idx, x, X64, Yr, cnt, lam, 0, 3, dyn)
        O.als_explicit(p, idx, x, X, Y32, cnt.astype(np.float32), lam, 0, 3, dyn)
        loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), lam, 1, 0, 3, dyn, "float", False, False)
    assert np.isfinite(Y).all()
    nrm = np.maximum(np.linalg.norm(Yr, axis=0), 1e-30)
    err, err32 = np.linalg.norm(Y - Yr, axis=0) / nrm, np.linalg.norm(Y32 - Yr, axis=0) / nrm
    nz = lens > 0
    if nz.any():
        assert err[nz].max() < max(1e-4, 3.0 * float(err32[nz].max())), (float(err[nz].max()), float(err32[nz].max()))
    assert abs(loss - lref) <= 1e-4 * abs(lref)


@pytest.mark.parametrize("trial", range(12))
def test_f64_cg_wave_kernel_on_random_shapes(trial):
    rng = np.random.default_rng(500 + trial)
    implicit = trial % 2 == 0
    k = int(rng.choice([3, 10, 16, 17, 24, 32, 33, 50, 64, 65, 96, 100, 127, 128]))
    n_rows, n_item = int(rng.integers(1, 120)), 900
    lens, p, idx = _rows(rng, n_rows, int(rng.choice([5, 70, 130, 260, 700])), n_item)
    x = (1.0 + rng.gamma(1.0, 2.0, size=idx.size)) if implicit else np.round(1.0 + 4.0 * rng.random(idx.size))
    X = np.asfortranarray(rng.standard_normal((k, n_item)) * 0.3)
    Y0 = np.asfortranarray(rng.standard_normal((k, n_rows)) * 0.3)
    lam, dyn, steps = 0.1, trial % 3 == 0, int(rng.choice([0, 1, 3, 5]))
    cnt = np.bincount(idx, minlength=n_item).astype(np.float64)
    Yr, Y = Y0.copy(order="F"), Y0.copy(order="F")
    csc = (n_item, n_rows, p, idx, x)
    if implicit:
        lref = O.als_implicit(p, idx, x, X, Yr, O.gramian(X, lam), lam, 1, steps)
        loss = als.als_implicit(csc, X, Y, lam, 1, 1, steps, "double", False, False)
    else:
        lref = O.als_explicit(p, idx, x, X, Yr, cnt, lam, 1, steps, dyn)
        loss = als.als_explicit(csc, X, Y, cnt, lam, 1, 1, steps, dyn, "double", False, False)
    err = np.linalg.norm(Y - Yr, axis=0) / np.maximum(np.linalg.norm(Yr, axis=0), 1e-300)
    assert err.max(initial=0.0) < 1e-9
    assert abs(loss - lref) <= 1e-9 * max(abs(lref), 1e-300)
