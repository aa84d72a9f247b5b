"""The reference's own test grid for this path, tests/testthat/test-wrmf.R:9-71 ("test WRMF core") and :73-90
("test WRMF FLOAT"), run through the Python mirror of the R6 class on the GPU, plus what the reference cannot
assert for lack of golden values: agreement of every fit with the CPU oracle driven from the same initial factors.

Grid (test-wrmf.R:10-27): implicit x {cholesky, nnls} x lambda {0, 0.1, 1000}; implicit x conjugate_gradient x
lambda {0, 0.1, 1000}; explicit x {conjugate_gradient, cholesky, nnls} x lambda {0.1, 1000}; precision {double, float};
with_user_item_bias {TRUE, FALSE} wherever the reference's grid has it -- i.e. the whole grid.  rank and K are drawn
from 4:10 as in the reference (:30-31)."""
import itertools
import json
import os
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN, csc_drop_rows, rel_fro
from oracle import wrmf_oracle as O

pytestmark = pytest.mark.gpu



def _record(cell, errs):
    """errors of this run (device and fp32-oracle yardstick, both against the fp64 oracle) -> gpurun_out/
    wrmf_core_errors.jsonl; tools/make_core_table.py turns it into profiles/r03/wrmf_core_parity_table.md.  A report only:
    nothing in this test reads it back."""
    out = Path(os.environ.get("GRAFT_REPO_ROOT", ".")) / "gpurun_out"
    if out.is_dir():
        with open(out / "wrmf_core_errors.jsonl", "a") as f:
            f.write(json.dumps({"cell": cell, **errs}) + "\n")


def _fit_errors(comp, emb, losses, ref, ref_emb):
    return {"components": rel_fro(comp, ref.components), "user_emb": rel_fro(emb, ref_emb),
            "loss": float(np.max(np.abs(np.array([l[1] for l in losses]) / np.array([l[1] for l in ref.losses]) - 1.0)))}

GRID = ([("implicit", s, l, b) for s in ("cholesky", "nnls") for l in (0.0, 0.1, 1000.0) for b in (False, True)] +
        [("implicit", "conjugate_gradient", l, False) for l in (0.0, 0.1, 1000.0)] +
        [("explicit", s, l, b) for s in ("conjugate_gradient", "cholesky", "nnls") for l in (0.1, 1000.0) for b in (False, True)])


def _data(movielens, ml_train):
    n_user_all, n_item, p, i, x = movielens
    n_user, _, tp, ti, tx = ml_train
    train = sp.csc_matrix((tx, ti, tp), shape=(n_user, n_item))
    cp, ci, cx = csc_drop_rows(900, p, i, x)
    cv = sp.csc_matrix((cx, ci, cp), shape=(n_user_all - 900, n_item)).tocsr()
    return train, cv, (n_user, n_item, tp, ti, tx)


@pytest.mark.parametrize("feedback,solver,lam,bias", GRID)
@pytest.mark.parametrize("precision", ["double", "float"])
def test_wrmf_core(movielens, ml_train, feedback, solver, lam, bias, precision):
    from rsparse_amd import WRMF
    train, cv, (n_user, n_item, tp, ti, tx) = _data(movielens, ml_train)
    seed = sum(ord(c) for c in feedback + solver + precision) + int(lam * 10) + 7 * bias
    rng = np.random.default_rng(seed)
    rank0, K = int(rng.integers(4, 11)), int(rng.integers(4, 11))                 # test-wrmf.R:30-31
    rank = rank0 + 2 * bias                                                        # rank_with_bias, :38
    U0 = (rng.standard_normal((n_user, rank)) * 0.01).astype(np.float32)
    V0 = (rng.standard_normal((rank, n_item)) * 0.01).astype(np.float32)
    init = None if solver == "conjugate_gradient" else V0.astype(np.float64 if precision == "double" else np.float32)
    model = WRMF(rank=rank0, lambda_=lam, feedback=feedback, solver=solver, with_user_item_bias=bias,
                 precision=precision, init=init)
    model._init_user_factors = U0
    user_emb = model.fit_transform(train, n_iter=5, convergence_tol=-1)            # :48
    want_dtype = np.float64 if precision == "double" else np.float32
    assert user_emb.shape == (train.shape[0], rank) and user_emb.dtype == want_dtype          # :51, FLOAT :86-87
    assert model.components.shape == (rank, train.shape[1]) and model.components.dtype == want_dtype   # :53
    assert np.array_equal(user_emb, model.transform(train))                       # :57 fit_transform == transform
    preds = model.predict(cv, K)                                                   # :59-61
    assert preds.shape == (cv.shape[0], K)
    cv_emb = model.transform(cv)                                                   # :63-64
    assert cv_emb.shape == (cv.shape[0], rank)
    if solver == "nnls":                                                           # :66-69
        assert cv_emb.min() >= 0 and user_emb.min() >= 0 and model.components.min() >= 0
    assert np.all(np.isfinite(user_emb)) and np.all(np.isfinite(model.components))

    # the same fit on the CPU oracle (fp64), same initial factors
    ref = O.OracleWRMF(rank0, lam=lam, feedback=feedback, solver=solver, dtype=np.float64, n_threads=8,
                       with_user_item_bias=bias)
    ref_emb = ref.fit_transform(n_user, n_item, tp, ti, tx, U0.T.astype(np.float64), n_iter=5, convergence_tol=-1,
                                init_components=None if solver == "conjugate_gradient" else V0.astype(np.float64))
    # precision = "double" cells: the device computes in double like the reference -> 1e-4 flat (below; they sit at 1e-11).
    # precision = "float" cells: ONE rule, and every number in it comes from the oracle: five ALS iterations compound
    # the per-solve fp32 differences, so the yardstick is the reference-shaped arithmetic in float -- the SAME fit on the
    # oracle in fp32 (the reference's precision = "float" build) against the oracle in fp64:
    #     err(device vs fp64 oracle)  <=  max(1e-4, 2 x err(fp32 oracle vs fp64 oracle))      (precision = "float" cells)
    # (err = the largest of: relative Frobenius error of the item factors, of the user embeddings, relative error of the
    # user-side loss sequence).  1e-4 is the north star's tolerance; the cells whose yardstick exceeds it are the ones
    # where ANY fp32 arithmetic departs from fp64 (lambda = 1000: the factors shrink to ~1e-26, fp32 underflow; NNLS: the
    # solver squares the per-row system and stops at 1e-4 relative steps; explicit CG with biases: 3 CG steps from a warm
    # start amplify the rounding of the previous iterate) -- profiles/r04/wrmf_core_parity_table.md lists both columns
    # (device / fp32 oracle between 0.1 and 1.25 there, hence the factor 2: round 3's was 3).
    # In those cells the fp32 fit is a noisy trajectory: one-ulp changes of the initial factors move its distance from
    # the fp64 fit by a factor of two (implicit NNLS with biases: 1.4e-2 ... 4.7e-2 over six such fits), so a single
    # fp32 fit is a fragile yardstick.  Wherever the first one is above 3e-5 the yardstick is therefore the largest
    # distance over FIVE fp32-oracle fits: the given initial user factors and four copies of them with every entry
    # scaled by 1 +- 2^-22 (seeded signs).
    cell = "%s|%s|%g|%d|%s" % (feedback, solver, lam, bias, precision)
    errs = _fit_errors(model.components, user_emb, model.losses, ref, ref_emb)
    if precision == "double":
        # the reference computes these cells in double (R/model_WRMF.R:82) and so does the device (the fp64 layer of the
        # library, rsparse_amd/csrc/wrmf_f64.hip): the north star's bound, flat, no yardstick
        _record(cell, {"rank": rank0, "device": errs, "fp32_oracle": None, "fp32_fits": 0, "bound": 1e-4})
        assert max(errs.values()) <= 1e-4, (cell, errs)
        return
    yard, prng = None, np.random.default_rng(12345)
    for trial in range(5):
        Up = U0 if trial == 0 else (U0 * (1 + np.float32(2.0 ** -22) * prng.choice([-1, 1], size=U0.shape).astype(np.float32))).astype(np.float32)
        ref32 = O.OracleWRMF(rank0, lam=lam, feedback=feedback, solver=solver, dtype=np.float32, n_threads=8,
                             with_user_item_bias=bias)
        emb32 = ref32.fit_transform(n_user, n_item, tp, ti, tx, Up.T.copy(), n_iter=5, convergence_tol=-1,
                                    init_components=None if solver == "conjugate_gradient" else V0.copy())
        y = _fit_errors(ref32.components, emb32, ref32.losses, ref, ref_emb)
        yard = y if yard is None else {q: max(yard[q], y[q]) for q in y}
        if max(yard.values()) <= 3e-5:
            break
    tol = max(1e-4, 2.0 * max(yard.values()))
    _record(cell, {"rank": rank0, "device": errs, "fp32_oracle": yard, "fp32_fits": trial + 1, "bound": tol})
    assert max(errs.values()) <= tol, (cell, errs, yard, tol)


# the seven precision = "float" cells whose five-iteration yardstick exceeds 1e-4 (profiles/r04/wrmf_core_parity_table.md)
NOISY = [("implicit", "nnls", 0.1, True), ("implicit", "nnls", 0.0, True), ("explicit", "conjugate_gradient", 1000.0, False),
         ("explicit", "nnls", 0.1, False), ("explicit", "nnls", 0.1, True), ("explicit", "conjugate_gradient", 0.1, True),
         ("implicit", "nnls", 0.1, False)]


@pytest.mark.parametrize("feedback,solver,lam,bias", NOISY)
def test_noisy_float_cells_after_one_iteration(movielens, ml_train, feedback, solver, lam, bias):
    """in the cells above a regression of tens of percent would pass the five-iteration rule (2 x a
    yardstick that is itself 3e-4 ... 4e-2).  After ONE iteration -- item half, user half, the exact solve fit_transform ends
    with: three solves from inputs that have not yet drifted apart -- the fp32 trajectories have not diverged, and the device
    has to be within max(1e-4, 1.5 x) of what the reference-shaped fp32 arithmetic itself loses there (both against the fp64
    oracle; the yardstick of four of the seven cells is below 1e-4 at this point, so for them the bound IS 1e-4)."""
    from rsparse_amd import WRMF
    train, cv, (n_user, n_item, tp, ti, tx) = _data(movielens, ml_train)
    rng = np.random.default_rng(int(lam * 10) + 7 * bias + len(solver))
    rank0 = 8
    rank = rank0 + 2 * bias
    U0 = (rng.standard_normal((n_user, rank)) * 0.01).astype(np.float32)
    V0 = (rng.standard_normal((rank, n_item)) * 0.01).astype(np.float32)
    init = None if solver == "conjugate_gradient" else V0.copy()
    model = WRMF(rank=rank0, lambda_=lam, feedback=feedback, solver=solver, with_user_item_bias=bias, precision="float", init=init)
    model._init_user_factors = U0
    emb = model.fit_transform(train, n_iter=1, convergence_tol=-1)
    fits = {}
    for dt in (np.float64, np.float32):
        ref = O.OracleWRMF(rank0, lam=lam, feedback=feedback, solver=solver, dtype=dt, n_threads=8, with_user_item_bias=bias)
        ref_emb = ref.fit_transform(n_user, n_item, tp, ti, tx, U0.T.astype(dt), n_iter=1, convergence_tol=-1,
                                    init_components=None if solver == "conjugate_gradient" else V0.astype(dt))
        fits[dt] = (ref, ref_emb)
    ref, ref_emb = fits[np.float64]
    errs = _fit_errors(model.components, emb, model.losses, ref, ref_emb)
    yard = _fit_errors(fits[np.float32][0].components, fits[np.float32][1], fits[np.float32][0].losses, ref, ref_emb)
    mult = 1.5
    if solver == "nnls":
        # NNLS stops where a sweep's largest relative step falls below 1e-4: which sweep that is flips with the last bits of the
        # squared system, so ONE fp32 fit is a fragile yardstick even after one iteration (measured: device 8.2e-3 against
        # 2.0e-3 on the user embeddings of implicit / lambda 0.1, both with item factors at 4e-6).  As in tests/test_nnls.py the
        # bound is 3 x the reference-shaped fp32 arithmetic, here the largest of three fits from one-ulp-scale perturbations
        mult, prng = 3.0, np.random.default_rng(99)
        for _ in range(2):
            Up = (U0 * (1 + np.float32(2.0 ** -22) * prng.choice([-1, 1], size=U0.shape).astype(np.float32))).astype(np.float32)
            r32 = O.OracleWRMF(rank0, lam=lam, feedback=feedback, solver=solver, dtype=np.float32, n_threads=8, with_user_item_bias=bias)
            e32 = r32.fit_transform(n_user, n_item, tp, ti, tx, Up.T.copy(), n_iter=1, convergence_tol=-1, init_components=V0.copy())
            y = _fit_errors(r32.components, e32, r32.losses, ref, ref_emb)
            yard = {q: max(yard[q], y[q]) for q in y}
    tol = max(1e-4, mult * max(yard.values()))
    _record("one-iteration|%s|%s|%g|%d" % (feedback, solver, lam, bias), {"rank": rank0, "device": errs, "fp32_oracle": yard,
                                                                         "fp32_fits": 1, "bound": tol})
    assert max(errs.values()) <= tol, (errs, yard, tol)


def test_wrmf_implicit_cg_with_biases_is_rejected():
    """Outside the reference's grid (test-wrmf.R:16-21 keeps with_user_item_bias = FALSE for conjugate_gradient): the
    reference drops a row of the warm start twice on that path (wrmf_implicit.hpp:189,197) and cannot run it; the
    device path answers UNSUPPORTED.  The global bias goes with every solver (tests/test_bias.py)."""
    from rsparse_amd import WRMF, _lib
    with pytest.raises(_lib.UnsupportedOnDevice):
        WRMF(rank=6, lambda_=0.1, feedback="implicit", solver="conjugate_gradient", with_user_item_bias=True)
    WRMF(rank=6, lambda_=0.1, feedback="implicit", solver="conjugate_gradient", with_global_bias=True)   # the default solver
    WRMF(rank=6, lambda_=0.1, feedback="implicit", solver="cholesky", with_global_bias=True)
