"""Pins the CPU oracle (oracle/) -- the reference has no numeric goldens for this path
(tests/testthat/test-wrmf.R asserts shapes only), so the restatement is pinned by closed-form
identities on the reference's own data set (movielens100k, train = rows 1:900, test-wrmf.R:6):

  * Cholesky branch == dense numpy.linalg.solve of the stated normal equations
    (inst/include/wrmf_implicit.hpp:207-208,231,236; wrmf_explicit.hpp:103-108)
  * CG with many steps -> the Cholesky answer (wrmf_implicit.hpp:8-32)
  * fit_transform(train) == transform(train) (test-wrmf.R:57)
  * float build within 1e-4 relative Frobenius of the double build (north-star tolerance)
  * committed goldens (tests/golden/make_goldens.py) reproduce.
"""
import numpy as np
import pytest

from conftest import GOLDEN, csc_drop_rows, rel_fro
from oracle import wrmf_oracle as O


def _rand_factors(k, n, seed, dtype=np.float64):
    rng = np.random.default_rng(seed)
    return np.asfortranarray((rng.standard_normal((k, n)) * 0.01).astype(dtype))


@pytest.mark.parametrize("lam", [0.0, 0.1, 1000.0])
def test_implicit_cholesky_matches_dense_solve(ml_train, lam):
    n_user, n_item, p, i, x = ml_train
    k = 8
    U = _rand_factors(k, n_user, 1)
    if lam == 0.0:
        U = np.asfortranarray(U * 100.0)  # keep XtX well conditioned without a ridge
    Y = np.zeros((k, n_item), order="F")
    loss = O.als_implicit(p, i, x, U, Y, O.gramian(U, lam), lam, O.CHOLESKY)
    Yn, lossn = O.np_half_iteration_implicit(p, i, x, U, lam)
    assert rel_fro(Y, Yn) < 1e-8
    assert abs(loss - lossn) <= 1e-9 * abs(lossn)
    # empty columns (items nobody in `train` rated) stay exactly zero (wrmf_implicit.hpp:281)
    empty = np.diff(p) == 0
    assert empty.any() and np.all(Y[:, empty] == 0)


@pytest.mark.parametrize("dynamic_lambda", [True, False])
def test_explicit_cholesky_matches_dense_solve(ml_train, dynamic_lambda):
    n_user, n_item, p, i, x = ml_train
    k, lam = 6, 0.1
    U = _rand_factors(k, n_user, 2) * 30
    U = np.asfortranarray(U)
    cnt = np.diff(O.csc_transpose(n_user, n_item, p, i, x)[0]).astype(np.float64)
    Y = np.zeros((k, n_item), order="F")
    loss = O.als_explicit(p, i, x, U, Y, cnt, lam, O.CHOLESKY, dynamic_lambda=dynamic_lambda)
    Yn, lossn = O.np_half_iteration_explicit(p, i, x, U, lam, dynamic_lambda, cnt)
    assert rel_fro(Y, Yn) < 1e-8
    assert abs(loss - lossn) <= 1e-9 * abs(lossn)


def test_cg_many_steps_converges_to_cholesky(ml_train):
    n_user, n_item, p, i, x = ml_train
    k, lam = 6, 0.1
    U = np.asfortranarray(_rand_factors(k, n_user, 3) * 50)
    XtX = O.gramian(U, lam)
    Yc = np.zeros((k, n_item), order="F")
    O.als_implicit(p, i, x, U, Yc, XtX, lam, O.CHOLESKY)
    Yg = np.zeros((k, n_item), order="F")
    O.als_implicit(p, i, x, U, Yg, XtX, lam, O.CONJUGATE_GRADIENT, cg_steps=60)
    assert rel_fro(Yg, Yc) < 1e-4  # CG_TOL=1e-10 on ||r||^2 stops it slightly early
    # explicit
    cnt = np.diff(O.csc_transpose(n_user, n_item, p, i, x)[0]).astype(np.float64)
    Yc[:] = 0
    Yg[:] = 0
    O.als_explicit(p, i, x, U, Yc, cnt, lam, O.CHOLESKY)
    O.als_explicit(p, i, x, U, Yg, cnt, lam, O.CONJUGATE_GRADIENT, cg_steps=60)
    assert rel_fro(Yg, Yc) < 1e-4


def test_cg_zero_steps_keeps_warm_start(ml_train):
    n_user, n_item, p, i, x = ml_train
    k, lam = 5, 0.1
    U = _rand_factors(k, n_user, 4)
    Y0 = _rand_factors(k, n_item, 5)
    Y = Y0.copy(order="F")
    O.als_implicit(p, i, x, U, Y, O.gramian(U, lam), lam, O.CONJUGATE_GRADIENT, cg_steps=0)
    nonempty = np.diff(p) > 0
    assert np.array_equal(Y[:, nonempty], Y0[:, nonempty])
    assert np.all(Y[:, ~nonempty] == 0)


def test_gramian_ridge_is_fp32_rounded():
    X = _rand_factors(7, 333, 6)
    lam = 0.1
    G = O.gramian(X, lam)
    ref = X @ X.T + float(np.float32(lam)) * np.eye(7)   # R/model_WRMF.R:476 fl(diag(lambda))
    assert np.allclose(G, ref, rtol=1e-13, atol=1e-15)
    assert not np.allclose(np.diag(G) - np.diag(X @ X.T), lam, rtol=1e-12, atol=0)


@pytest.mark.parametrize("solver", ["conjugate_gradient", "cholesky"])
@pytest.mark.parametrize("feedback", ["implicit", "explicit"])
def test_fit_transform_equals_transform_and_float_close(ml_train, solver, feedback):
    n_user, n_item, p, i, x = ml_train
    k, lam = 8, 0.1
    U0 = _rand_factors(k, n_user, 7)
    C0 = None if solver == "conjugate_gradient" else _rand_factors(k, n_item, 8)
    out = {}
    for dt in (np.float64, np.float32):
        m = O.OracleWRMF(k, lam, feedback, solver, dtype=dt)
        emb = m.fit_transform(n_user, n_item, p, i, x, U0, n_iter=5, convergence_tol=-1,
                              init_components=C0)
        assert emb.shape == (n_user, k) and m.components.shape == (k, n_item)
        assert len(m.losses) == 5
        again = m.transform(*m.c_iu)                       # test-wrmf.R:57
        assert np.array_equal(emb, again)
        out[dt] = (emb, m.components.copy(), [l[1] for l in m.losses])
    assert rel_fro(out[np.float32][0], out[np.float64][0]) < 1e-4
    assert rel_fro(out[np.float32][1], out[np.float64][1]) < 1e-4
    if feedback == "implicit":
        # the observed-entry loss the reference logs decreases on this data set
        lu = out[np.float64][2]
        assert all(b < a for a, b in zip(lu, lu[1:]))


def test_transform_on_heldout_users(movielens, ml_train):
    n_user_all, n_item, p, i, x = movielens
    n_user, _, tp, ti, tx = ml_train
    m = O.OracleWRMF(6, 0.1, "implicit", "conjugate_gradient")
    m.fit_transform(n_user, n_item, tp, ti, tx, _rand_factors(6, n_user, 9), n_iter=3, convergence_tol=-1)
    cp, ci, cx = csc_drop_rows(900, p, i, x)                  # cv = movielens100k[901:943, ]
    emb = m.transform(*O.csc_transpose(n_user_all - 900, n_item, cp, ci, cx))
    assert emb.shape == (43, 6) and np.isfinite(emb).all()


def test_goldens_reproduce(ml_train):
    g = np.load(GOLDEN / "wrmf_movielens_goldens.npz")
    n_user, n_item, p, i, x = ml_train
    for feedback in ("implicit", "explicit"):
        for solver in ("conjugate_gradient", "cholesky"):
            tag = "%s_%s" % (feedback, solver)
            k, lam = int(g[tag + "_rank"]), float(g[tag + "_lambda"])
            C0 = g[tag + "_init_components"] if solver == "cholesky" else None
            m = O.OracleWRMF(k, lam, feedback, solver, dtype=np.float64)
            emb = m.fit_transform(n_user, n_item, p, i, x, g["init_U_k%d" % k], n_iter=5,
                                  convergence_tol=-1, init_components=C0)
            assert np.allclose([l[0] for l in m.losses], g[tag + "_loss_items"], rtol=1e-9)
            assert np.allclose([l[1] for l in m.losses], g[tag + "_loss_users"], rtol=1e-9)
            assert rel_fro(emb, g[tag + "_user_emb"]) < 1e-9
            assert rel_fro(m.components, g[tag + "_components"]) < 1e-9
