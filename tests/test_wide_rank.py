"""Ranks 129..256 (rsparse_amd/csrc/wrmf_wide.hip): the reference has no rank limit (arma::Mat<T>,
inst/include/wrmf_implicit.hpp:103); through round 3 the device path answered RSPARSE_HIP_ERR_UNSUPPORTED above 128.  Every
solver and operand set through the C ABI against the fp64 oracle on the same fp32 inputs, 1e-4 per row (NNLS: the yardstick of
tests/test_nnls.py -- its fp32 arithmetic squares the system)."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import rel_fro
from oracle import wrmf_oracle as O
from rsparse_amd import als, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _problem(n_user, n_item, k, seed, feedback="implicit", scale=0.1, mean_deg=14, d_max=500):
    d = synth.make_dataset(n_user, n_item, seed=seed, mean_deg=mean_deg, d_max=d_max, feedback=feedback, device="cpu")
    p, i, x = d["c_iu"]
    p, i, x = p.numpy(), i.numpy(), x.numpy().astype(np.float64)
    rng = np.random.default_rng(seed)
    X = np.asfortranarray((rng.standard_normal((k, n_item)) * scale).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, n_user)) * scale).astype(np.float32))
    return (n_item, n_user, p, i, x), X, Y0


def _row_err(Y, Yref):
    return np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)


@pytest.mark.parametrize("k,n", [(129, 700), (160, 5000), (200, 333), (256, 3001)])
def test_gramian_wide(k, n):
    rng = np.random.default_rng(k + n)
    X = np.asfortranarray(rng.standard_normal((k, n)).astype(np.float32))
    X[0, :] += 3.0
    G = als.gramian(X, 0.1, "float")
    ref = X.astype(np.float64) @ X.astype(np.float64).T + float(np.float32(0.1)) * np.eye(k)
    assert rel_fro(G, ref) < 3e-6 and np.array_equal(G, G.T)


@pytest.mark.parametrize("k", [130, 160, 256])
@pytest.mark.parametrize("solver", [0, 1, 2])
def test_implicit_wide_half_iteration(k, solver):
    csc, X, Y0 = _problem(400, 300, k, seed=k + solver)
    n_rows, n_cols, p, i, x = csc
    if solver == 2:
        X, Y0 = np.abs(X), np.abs(Y0)
    X64 = np.asfortranarray(X, dtype=np.float64)
    G64 = O.gramian(X64, 0.1)
    Yref = np.asfortranarray(Y0, dtype=np.float64).copy(order="F")
    lref = O.als_implicit(p, i, x, X64, Yref, G64, 0.1, solver, 3, n_threads=8)
    Y = Y0.copy(order="F")
    loss = als.als_implicit(csc, X, Y, 0.1, 1, solver, 3, "float", False, False)
    err = _row_err(Y, Yref)
    if solver == 2:   # fp32 NNLS: yardstick = the oracle in float on the same inputs
        Y32 = Y0.copy(order="F")
        O.als_implicit(p, i, x, X, Y32, O.gramian(X, 0.1), 0.1, 2, 3, n_threads=8)
        assert rel_fro(Y, Yref) <= max(TOL, 3 * rel_fro(Y32, Yref)), (rel_fro(Y, Yref), rel_fro(Y32, Yref))
        assert Y.min() >= 0
    else:
        assert err.max() < TOL, (int(err.argmax()), float(err.max()))
        assert abs(loss - lref) <= TOL * abs(lref)
    assert np.all(Y[:, np.diff(p) == 0] == 0)


@pytest.mark.parametrize("k", [144, 256])
@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("dynamic_lambda", [True, False])
def test_explicit_wide_half_iteration(k, solver, dynamic_lambda):
    csc, X, Y0 = _problem(300, 250, k, seed=9 + k + solver, feedback="explicit", scale=0.3)
    n_rows, n_cols, p, i, x = csc
    cnt = np.diff(sp.csc_matrix((x, i, p), shape=(n_rows, n_cols)).tocsr().indptr).astype(np.float64)
    X64 = np.asfortranarray(X, dtype=np.float64)
    Yref = np.asfortranarray(Y0, dtype=np.float64).copy(order="F")
    lref = O.als_explicit(p, i, x, X64, Yref, cnt, 0.1, solver, 3, dynamic_lambda, n_threads=8)
    Y = Y0.copy(order="F")
    loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), 0.1, 1, solver, 3, dynamic_lambda, "float", False, False)
    if solver == 1:   # three fp32 CG steps from a warm start on lambda_use = 0.1 n systems: the oracle in float as the yardstick
        Y32 = Y0.copy(order="F")
        O.als_explicit(p, i, x, X, Y32, cnt.astype(np.float32), 0.1, 1, 3, dynamic_lambda, n_threads=8)
        assert rel_fro(Y, Yref) <= max(TOL, 3 * rel_fro(Y32, Yref))
    else:
        err = _row_err(Y, Yref)
        assert err.max() < TOL, (int(err.argmax()), float(err.max()))
    assert abs(loss - lref) <= 2e-4 * abs(lref)


@pytest.mark.parametrize("bias_last", [True, False])
def test_wide_with_biases_and_global_bias(bias_last):
    """rank 131 with user/item biases (a 130 x 130 system), explicit and implicit; implicit global bias with every solver at 140"""
    k = 131
    csc, X, Y0 = _problem(200, 150, k, seed=3, feedback="explicit", scale=0.3)
    n_rows, n_cols, p, i, x = csc
    X[0 if bias_last else k - 1, :] = 1.0
    Y0[k - 1 if bias_last else 0, :] = 1.0
    cnt = np.diff(sp.csc_matrix((x, i, p), shape=(n_rows, n_cols)).tocsr().indptr).astype(np.float64)
    X64 = np.asfortranarray(X, dtype=np.float64)
    Yref = np.asfortranarray(Y0, dtype=np.float64).copy(order="F")
    lref = O.als_explicit(p, i, x, X64, Yref, cnt, 0.1, 0, 3, True, with_biases=True, is_x_bias_last_row=bias_last)
    Y = Y0.copy(order="F")
    loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), 0.1, 1, 0, 3, True, "float", True, bias_last)
    assert _row_err(Y, Yref).max() < TOL and abs(loss - lref) <= TOL * abs(lref)
    csc, X, Y0 = _problem(200, 150, k, seed=4)
    n_rows, n_cols, p, i, x = csc
    X[0 if bias_last else k - 1, :] = 1.0
    Y0[k - 1 if bias_last else 0, :] = 1.0
    X64 = np.asfortranarray(X, dtype=np.float64)
    G64 = O.gramian(np.asfortranarray(X64[:-1] if bias_last else X64[1:]), 0.1)
    Yref = np.asfortranarray(Y0, dtype=np.float64).copy(order="F")
    lref = O.als_implicit(p, i, x, X64, Yref, G64, 0.1, 0, 3, with_biases=True, is_x_bias_last_row=bias_last, global_bias=0.02)
    Y = Y0.copy(order="F")
    loss = als.als_implicit(csc, X, Y, 0.1, 1, 0, 3, "float", True, bias_last, global_bias=0.02)
    assert _row_err(Y, Yref).max() < TOL and abs(loss - lref) <= TOL * abs(lref)
    if bias_last:
        k2 = 140
        csc, X, Y0 = _problem(200, 150, k2, seed=5)
        n_rows, n_cols, p, i, x = csc
        X64 = np.asfortranarray(X, dtype=np.float64)
        for solver in (0, 1):
            Yref = np.asfortranarray(Y0, dtype=np.float64).copy(order="F")
            lref = O.als_implicit(p, i, x, X64, Yref, O.gramian(X64, 0.1), 0.1, solver, 3, global_bias=0.03)
            Y = Y0.copy(order="F")
            loss = als.als_implicit(csc, X, Y, 0.1, 1, solver, 3, "float", False, False, global_bias=0.03)
            bound = TOL if solver == 0 else 5e-4      # (the global-bias CG is the reference's "very poor numerical precision" variant)
            assert _row_err(Y, Yref).max() < bound, solver
            assert abs(loss - lref) <= bound * abs(lref)


def test_wide_cholesky_falls_back_to_the_general_solver():
    k = 150
    csc, X, Y0 = _problem(120, 100, k, seed=4, scale=0.3, mean_deg=30)
    n_rows, n_cols, p, i, x = csc
    rng = np.random.default_rng(1)
    x = np.where(rng.random(x.size) < 0.5, 0.25, 3.0)
    X64 = X.astype(np.float64)
    G = np.asfortranarray(0.05 * (X64 @ X64.T) + 0.1 * np.eye(k))
    Yref = np.asfortranarray(Y0, dtype=np.float64).copy(order="F")
    O.als_implicit(p, i, x, np.asfortranarray(X64), Yref, G, 0.1, 0, 3)
    n_bad, cond = 0, np.ones(n_cols)
    for c in range(n_cols):
        idx, val = i[p[c]:p[c + 1]], x[p[c]:p[c + 1]]
        ev = np.linalg.eigvalsh(G + (X64[:, idx] * (val - 1.0)) @ X64[:, idx].T)
        cond[c] = np.abs(ev).max() / np.abs(ev).min()
        n_bad += ev.min() < -1e-4
    assert n_bad >= 3
    Y = Y0.copy(order="F")
    als.als_implicit((n_rows, n_cols, p, i, x), X, Y, 0.1, 1, 0, 3, "float", False, False, XtX=np.asfortranarray(G, dtype=np.float32))
    assert np.all(np.isfinite(Y))
    err = _row_err(Y, Yref)
    bound = np.maximum(1e-4, 20.0 * cond * 6e-8)
    assert np.all(err <= bound), (int(np.argmax(err / bound)), float(err.max()))


def test_wrmf_at_rank_160(ml_train):
    """the class end to end at a rank the reference accepts and the device path used to refuse: fit (CG), the final exact
    solve, transform and predict (the top-k kernel's rank-256 instantiation)"""
    from rsparse_amd import WRMF
    n_user, n_item, p, i, x = ml_train
    train = sp.csc_matrix((x, i, p), shape=(n_user, n_item))
    rng = np.random.default_rng(2)
    U0 = (rng.standard_normal((n_user, 160)) * 0.01).astype(np.float32)
    model = WRMF(rank=160, lambda_=0.1, feedback="implicit", solver="conjugate_gradient", precision="float")
    model._init_user_factors = U0
    emb = model.fit_transform(train, n_iter=2, convergence_tol=-1)
    ref = O.OracleWRMF(160, lam=0.1, feedback="implicit", solver="conjugate_gradient", dtype=np.float64, n_threads=8)
    ref_emb = ref.fit_transform(n_user, n_item, p, i, x, U0.T.astype(np.float64), n_iter=2, convergence_tol=-1)
    assert rel_fro(model.components, ref.components) < TOL and rel_fro(emb, ref_emb) < TOL
    assert np.allclose([l[1] for l in model.losses], [l[1] for l in ref.losses], rtol=TOL)
    assert np.array_equal(emb, model.transform(train))
    top = model.predict(train[:80], 9)
    sc = emb[:80].astype(np.float64) @ model.components.astype(np.float64)
    sc[train[:80].toarray() != 0] = -np.inf
    best = np.sort(sc, axis=1)[:, ::-1][:, :9]
    assert np.allclose(top.scores, best, rtol=1e-4, atol=1e-5)
