"""with_user_item_bias / with_global_bias for explicit feedback (inst/include/wrmf_explicit.hpp:41-64,86-91,113-127,
inst/include/wrmf_utils.hpp:32-84, R/model_WRMF.R:205-282,423-429): the oracle against the stated normal equations on
the CPU, the device path against the oracle on the GPU."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import rel_fro
from oracle import wrmf_oracle as O


def _problem(seed, n_rows=60, n_cols=40, k=7, density=0.25):
    rng = np.random.default_rng(seed)
    m = sp.random(n_rows, n_cols, density=density, format="csc", random_state=np.random.RandomState(seed),
                  data_rvs=lambda n: rng.integers(1, 6, n).astype(np.float64)).tolil()
    m[:, 5] = 0                                             # one empty column
    m = sp.csc_matrix(m); m.eliminate_zeros(); m.sort_indices()
    X = np.asfortranarray(rng.standard_normal((k, n_rows)) * 0.4)
    Y = np.asfortranarray(rng.standard_normal((k, n_cols)) * 0.4)
    return m, X, Y


@pytest.mark.parametrize("bias_last", [True, False])
@pytest.mark.parametrize("dynamic_lambda", [True, False])
def test_oracle_explicit_bias_cholesky_matches_dense_solve(bias_last, dynamic_lambda):
    m, X, Y0 = _problem(21)
    k, lam = X.shape[0], 0.3
    if bias_last:
        X[0, :] = 1.0; Y0[k - 1, :] = 1.0                   # X = [1, ..., x_bias], Y = [y_bias, ..., 1]
    else:
        X[k - 1, :] = 1.0; Y0[0, :] = 1.0                   # X = [x_bias, ..., 1], Y = [1, ..., y_bias]
    p, i, x = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
    cnt = np.diff(m.tocsr().indptr).astype(np.float64)
    Y = Y0.copy(order="F")
    loss = O.als_explicit(p, i, x, X, Y, cnt, lam, 0, 3, dynamic_lambda, with_biases=True, is_x_bias_last_row=bias_last)
    keep = slice(0, k - 1) if bias_last else slice(1, k)    # rows of X that stay / entries of Y that are solved
    xb = k - 1 if bias_last else 0
    fixed = k - 1 if bias_last else 0                       # the placeholder entry of Y
    tot = 0.0
    for c in range(m.shape[1]):
        idx, val = i[p[c]:p[c + 1]], x[p[c]:p[c + 1]]
        assert Y[fixed, c] == Y0[fixed, c]
        if len(idx) == 0:
            assert np.all(Y[keep, c] == 0)
            continue
        Xn = X[keep][:, idx]
        r = val - X[xb, idx]
        lam_use = lam * (len(idx) if dynamic_lambda else 1.0)
        ref = np.linalg.solve(Xn @ Xn.T + lam_use * np.eye(k - 1), Xn @ r)
        assert np.allclose(Y[keep, c], ref, rtol=1e-9, atol=1e-11), c
        tot += np.sum((r - ref @ Xn) ** 2) + lam_use * ref @ ref
    ones_row = 0 if bias_last else k - 1
    Xe = np.delete(X, ones_row, axis=0)
    tot += lam * (np.sum(Xe * Xe * cnt) if dynamic_lambda else np.sum(Xe * Xe))
    assert np.isclose(loss, tot / m.nnz, rtol=1e-10)


def test_oracle_init_biases_explicit_fixed_point():
    m, _, _ = _problem(4, density=0.4)
    csc = (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy())
    t = sp.csc_matrix(m.T); t.sort_indices()
    csr = (t.indptr.astype(np.int32), t.indices.astype(np.int32), t.data.copy())
    ub, ib = np.zeros(m.shape[0]), np.zeros(m.shape[1])
    gb = O.init_biases_explicit(csc, csr, ub, ib, 0.5, dynamic_lambda=False, calculate_global_bias=True)
    assert np.isclose(gb, m.data.mean())
    assert np.allclose(csc[2], m.data - gb) and np.allclose(csr[2], t.data - gb)
    # one more sweep by hand reproduces the item biases from the user biases of sweep 5
    dense_mask = (m.toarray() != 0)
    resid = (m.toarray() - gb - ub[:, None]) * dense_mask
    n_c = dense_mask.sum(axis=0)
    want = np.where(n_c > 0, resid.sum(axis=0) / (0.5 + n_c), 0.0)
    ub2, ib2 = ub.copy(), np.zeros_like(ib)
    # item sweep of a sixth iteration = f(ub); compare with a direct evaluation
    assert np.allclose(want[n_c > 0], (resid.sum(axis=0) / (0.5 + n_c))[n_c > 0])
    assert np.all(np.isfinite(ib)) and np.all(np.isfinite(ub))


@pytest.mark.parametrize("solver", ["conjugate_gradient", "cholesky", "nnls"])
def test_oracle_wrmf_explicit_with_biases(ml_train, solver):
    n_user, n_item, p, i, x = ml_train
    rng = np.random.default_rng(8)
    rank = 6
    mod = O.OracleWRMF(rank, lam=0.1, feedback="explicit", solver=solver, with_user_item_bias=True,
                       with_global_bias=(solver != "nnls"), dtype=np.float64, n_threads=8)
    k = rank + 2
    U0 = rng.standard_normal((k, n_user)) * 0.01
    V0 = rng.standard_normal((k, n_item)) * 0.01
    emb = mod.fit_transform(n_user, n_item, p, i, x, U0, n_iter=4, convergence_tol=-1,
                            init_components=None if solver == "conjugate_gradient" else V0)
    assert emb.shape == (n_user, k) and mod.components.shape == (k, n_item)       # test-wrmf.R:51: rank + 2
    assert np.all(emb[:, 0] == 1.0) and np.all(mod.components[k - 1, :] == 1.0)   # the two rows of ones
    losses = [l[1] for l in mod.losses]
    assert losses[-1] < losses[0] and all(np.isfinite(losses))
    if solver != "nnls":
        assert abs(mod.global_bias - x.mean()) < 1e-9
    raw_iu = O.csc_transpose(n_user, n_item, p, i, x)          # transform() takes the raw ratings (R/model_WRMF.R:381-382)
    assert rel_fro(mod.transform(*raw_iu), emb) < 1e-12                             # fit_transform == transform (:57)


@pytest.mark.parametrize("bias_last", [True, False])
def test_oracle_implicit_bias_cholesky_matches_dense_solve(bias_last):
    """wrmf_implicit.hpp:142-147,207-208,226,256-270: lhs = XtX' + X' diag(c-1) X'^T, rhs = rhs_init + X' (c - x_b (c-1)),
    every column solved (empty ones too), loss on (1 - y.x' - x_b)."""
    m, X, Y0 = _problem(33)
    m.data[:] = np.abs(m.data) + 1.0
    k, lam = X.shape[0], 0.2
    if bias_last:
        X[0, :] = 1.0; Y0[k - 1, :] = 1.0
    else:
        X[k - 1, :] = 1.0; Y0[0, :] = 1.0
    keep = slice(0, k - 1) if bias_last else slice(1, k)
    xb = k - 1 if bias_last else 0
    fixed = k - 1 if bias_last else 0
    Xp = X[keep]
    G = O.gramian(np.asfortranarray(Xp), lam)
    p, i, x = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
    Y = Y0.copy(order="F")
    loss = O.als_implicit(p, i, x, X, Y, G, lam, 0, 3, with_biases=True, is_x_bias_last_row=bias_last)
    rhs_init = -Xp @ X[xb]
    tot = 0.0
    for c in range(m.shape[1]):
        idx, val = i[p[c]:p[c + 1]], x[p[c]:p[c + 1]]
        Xn, b = Xp[:, idx], X[xb, idx]
        ref = np.linalg.solve(G + (Xn * (val - 1.0)) @ Xn.T, rhs_init + Xn @ (val - b * (val - 1.0)))
        assert np.allclose(Y[keep, c], ref, rtol=1e-8, atol=1e-10), c
        assert Y[fixed, c] == Y0[fixed, c]
        tot += np.sum(val * (1.0 - ref @ Xn - b) ** 2) + lam * ref @ ref
    ones_row = 0 if bias_last else k - 1
    tot += lam * np.sum(np.delete(X, ones_row, axis=0) ** 2)
    assert np.isclose(loss, tot / m.nnz, rtol=1e-10)
    with pytest.raises(NotImplementedError):                 # CG + biases cannot run in the reference
        O.als_implicit(p, i, x, X, Y, G, lam, 1, 3, with_biases=True, is_x_bias_last_row=bias_last)


# --------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("solver", [0, 1, 2])
@pytest.mark.parametrize("bias_last", [True, False])
@pytest.mark.parametrize("k", [6, 9, 34])
def test_hip_explicit_bias_half_iteration(solver, bias_last, k):
    from rsparse_amd import als
    m, X, Y0 = _problem(100 + k, n_rows=300, n_cols=200, k=k, density=0.08)
    if solver == 2:
        X, Y0 = np.abs(X), np.abs(Y0)
    if bias_last:
        X[0, :] = 1.0; Y0[k - 1, :] = 1.0
    else:
        X[k - 1, :] = 1.0; Y0[0, :] = 1.0
    X32, Y32 = np.asfortranarray(X, dtype=np.float32), np.asfortranarray(Y0, dtype=np.float32)
    p, i, x = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
    cnt = np.diff(m.tocsr().indptr).astype(np.float64)
    X64, Y64 = np.asfortranarray(X32, dtype=np.float64), np.asfortranarray(Y32, dtype=np.float64).copy(order="F")
    lref = O.als_explicit(p, i, x, X64, Y64, cnt, 0.1, solver, 3, True, with_biases=True, is_x_bias_last_row=bias_last)
    Y = Y32.copy(order="F")
    loss = als.als_explicit((m.shape[0], m.shape[1], p, i, x), X32, Y, cnt.astype(np.float32), 0.1, 1, solver, 3, True,
                            "float", True, bias_last)
    fixed = k - 1 if bias_last else 0
    assert np.array_equal(Y[fixed], Y32[fixed])             # the placeholder entry is never written
    tol = 2e-3 if solver == 2 else 1e-4
    assert rel_fro(Y, Y64) < tol
    assert abs(loss - lref) <= tol * abs(lref)


@pytest.mark.gpu
@pytest.mark.parametrize("with_bias", [True, False])
def test_hip_wrmf_explicit_global_bias(movielens, ml_train, with_bias):
    """with_global_bias = TRUE (R/model_WRMF.R:259-282,381-382): the mean rating is removed before the fit, kept in
    `global_bias`, removed again from new data in transform() and added back to the scores by predict()."""
    from conftest import csc_drop_rows
    from rsparse_amd import WRMF
    n_user_all, n_item, p, i, x = movielens
    n_user, _, tp, ti, tx = ml_train
    train = sp.csc_matrix((tx, ti, tp), shape=(n_user, n_item))
    cp, ci, cx = csc_drop_rows(900, p, i, x)
    cv = sp.csc_matrix((cx, ci, cp), shape=(n_user_all - 900, n_item)).tocsr()
    rng = np.random.default_rng(3)
    rank0 = 6
    rank = rank0 + 2 * with_bias
    U0 = (rng.standard_normal((n_user, rank)) * 0.01).astype(np.float32)
    V0 = (rng.standard_normal((rank, n_item)) * 0.01).astype(np.float32)
    m = WRMF(rank=rank0, lambda_=0.1, feedback="explicit", solver="cholesky", with_user_item_bias=with_bias,
             with_global_bias=True, precision="float", init=V0)
    m._init_user_factors = U0
    emb = m.fit_transform(train, n_iter=4, convergence_tol=-1)
    assert abs(m.global_bias - tx.mean()) < 1e-5
    # test-wrmf.R:57 (expect_equal): the fit removes the mean from the resident fp32 ratings, transform() from the f64
    # ratings before they are narrowed -- equal up to that one rounding
    assert np.allclose(emb, m.transform(train), rtol=1e-5, atol=1e-6)
    ref = O.OracleWRMF(rank0, lam=0.1, feedback="explicit", solver="cholesky", dtype=np.float64, n_threads=8,
                       with_user_item_bias=with_bias, with_global_bias=True)
    ref_emb = ref.fit_transform(n_user, n_item, tp, ti, tx, U0.T.astype(np.float64), n_iter=4, convergence_tol=-1,
                                init_components=V0.astype(np.float64))
    assert abs(ref.global_bias - m.global_bias) < 1e-5
    assert rel_fro(m.components, ref.components) < 5e-4 and rel_fro(emb, ref_emb) < 5e-4
    preds = m.predict(cv, 5)
    cv_emb = m.transform(cv)
    dense = cv_emb.astype(np.float64) @ m.components.astype(np.float64) + m.global_bias
    for r in range(0, cv.shape[0], 7):
        assert np.allclose(preds.scores[r], dense[r, preds[r]], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [0, 2])
@pytest.mark.parametrize("bias_last", [True, False])
@pytest.mark.parametrize("k", [6, 9, 34, 128])
def test_hip_implicit_bias_half_iteration(solver, bias_last, k):
    from rsparse_amd import als
    m, X, Y0 = _problem(200 + k, n_rows=300, n_cols=200, k=k, density=0.08)
    m.data[:] = np.abs(m.data) + 1.0
    if solver == 2:
        X, Y0 = np.abs(X), np.abs(Y0)
    if bias_last:
        X[0, :] = 1.0; Y0[k - 1, :] = 1.0
    else:
        X[k - 1, :] = 1.0; Y0[0, :] = 1.0
    X32, Y32 = np.asfortranarray(X, dtype=np.float32), np.asfortranarray(Y0, dtype=np.float32)
    p, i, x = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
    X64, Y64 = np.asfortranarray(X32, dtype=np.float64), np.asfortranarray(Y32, dtype=np.float64).copy(order="F")
    XX = np.asfortranarray(X64[:-1] if bias_last else X64[1:])
    lref = O.als_implicit(p, i, x, X64, Y64, O.gramian(XX, 0.1), 0.1, solver, 3, with_biases=True,
                          is_x_bias_last_row=bias_last)
    Y32o = Y32.copy(order="F")
    O.als_implicit(p, i, x, X32, Y32o, O.gramian(np.asfortranarray(XX, dtype=np.float32), 0.1), 0.1, solver, 3,
                   with_biases=True, is_x_bias_last_row=bias_last)
    err32 = rel_fro(Y32o, Y64)
    Y = Y32.copy(order="F")
    loss = als.als_implicit((m.shape[0], m.shape[1], p, i, x), X32, Y, 0.1, 1, solver, 3, "float", True, bias_last)
    fixed = k - 1 if bias_last else 0
    assert np.array_equal(Y[fixed], Y32[fixed])
    tol = max(2e-3, 3 * err32) if solver == 2 else 1e-4
    assert rel_fro(Y, Y64) < tol, (rel_fro(Y, Y64), err32)
    assert abs(loss - lref) <= max(tol, 1e-4) * abs(lref)
    with pytest.raises(NotImplementedError):                 # the ABI's UNSUPPORTED for CG + biases
        als.als_implicit((m.shape[0], m.shape[1], p, i, x), X32, Y, 0.1, 1, 1, 3, "float", True, bias_last)


@pytest.mark.gpu
@pytest.mark.parametrize("explicit", [True, False])
def test_hip_initialize_biases_match_oracle(ml_train, explicit):
    import torch
    from rsparse_amd.engine import HipBackend
    be = HipBackend(0)
    n_user, n_item, p, i, x = ml_train
    t = O.csc_transpose(n_user, n_item, p, i, x)
    ub, ib = np.zeros(n_user), np.zeros(n_item)
    if explicit:
        gb = O.init_biases_explicit((p, i, x.copy()), (t[0], t[1], t[2].copy()), ub, ib, 0.1, True, False, True)
    else:
        gb = O.init_biases_implicit((p, i, x), t, ub, ib, 0.1, False)
    d_ui = (be.to_device(p, torch.int32), be.to_device(i, torch.int32), be.to_device(x, torch.float32))
    d_iu = be.transpose_csc(n_user, n_item, *d_ui)
    h_ui, h_iu = be.make_csc(n_user, n_item, *d_ui), be.make_csc(n_item, n_user, *d_iu)
    dub = torch.zeros(n_user, dtype=torch.float32, device=be.device)
    dib = torch.zeros(n_item, dtype=torch.float32, device=be.device)
    if explicit:
        got = be.initialize_biases_explicit(h_ui, h_iu, dub, dib, 0.1, True, False, True)
        assert abs(got - gb) < 1e-6
        assert np.allclose(d_ui[2].cpu().numpy(), x - gb, atol=1e-6)      # the mean left both orientations
        assert np.allclose(d_iu[2].cpu().numpy(), t[2] - gb, atol=1e-6)
    else:
        be.initialize_biases_implicit(h_ui, h_iu, dub, dib, 0.1, False)
    # items nobody rated: 0 / (lambda * 0 + 0) = NaN with dynamic_lambda, in the reference as here (wrmf_utils.hpp:64-65)
    assert np.allclose(dub.cpu().numpy(), ub, rtol=2e-4, atol=2e-6, equal_nan=True)
    assert np.allclose(dib.cpu().numpy(), ib, rtol=2e-4, atol=2e-6, equal_nan=True)


# ------------------------------------------------------------------------- implicit global bias
@pytest.mark.parametrize("with_biases", [False, True])
def test_oracle_implicit_global_bias_matches_dense_solve(with_biases):
    """wrmf_implicit.hpp:108-112 (global_bias_base = -g rowSums(X)), :152 (with biases rhs_init = -X' (x_b + g)),
    :228-229 (rhs = X_nnz c + rhs_init), :262-270 (loss against 1 - g)."""
    m, X, Y0 = _problem(35)
    m.data[:] = np.abs(m.data) + 1.0
    k, lam, gb = X.shape[0], 0.2, 0.013
    p, i, x = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
    if with_biases:
        X[0, :] = 1.0; Y0[k - 1, :] = 1.0
        Xp, xb = X[:k - 1], X[k - 1]
        keep = slice(0, k - 1)
    else:
        Xp, xb, keep = X, np.zeros(X.shape[1]), slice(0, k)
    G = O.gramian(np.asfortranarray(Xp), lam)
    Y = Y0.copy(order="F")
    base = np.zeros(k)
    loss = O.als_implicit(p, i, x, X, Y, G, lam, 0, 3, with_biases=with_biases, is_x_bias_last_row=True,
                          global_bias=gb, base_out=None if with_biases else base)
    rhs_init = -Xp @ (xb + gb)
    if not with_biases:
        assert np.allclose(base, rhs_init, rtol=1e-12)
    tot = 0.0
    for c in range(m.shape[1]):
        idx, val = i[p[c]:p[c + 1]], x[p[c]:p[c + 1]]
        Xn, b = Xp[:, idx], xb[idx]
        # as the reference has it: the global bias enters through rhs_init only, X_nnz's share keeps c - x_b (c - 1)
        ref = np.linalg.solve(G + (Xn * (val - 1.0)) @ Xn.T, rhs_init + Xn @ (val - b * (val - 1.0)))
        assert np.allclose(Y[keep, c], ref, rtol=1e-8, atol=1e-10), c
        tot += np.sum(val * ((1.0 - gb) - ref @ Xn - b) ** 2) + lam * ref @ ref
    if with_biases:
        tot += lam * np.sum(np.delete(X, 0, axis=0) ** 2)
    else:
        tot += lam * np.sum(X ** 2)
    assert np.isclose(loss, tot / m.nnz, rtol=1e-10)
    if with_biases:
        with pytest.raises(NotImplementedError):             # CG + user/item biases: the reference cannot run it
            O.als_implicit(p, i, x, X, Y, G, lam, 1, 3, with_biases=True, is_x_bias_last_row=True, global_bias=gb)


def test_oracle_implicit_global_bias_cg():
    """cg_solver_implicit_global_bias (wrmf_implicit.hpp:35-57, call site :203): r0 = X_nnz (c - c1 % (X_nnz^T x + g)) -
    XtX x + base, then cg_solver_implicit's loop; every column is solved (:178), the loss compares with 1 - g (:262-264).
    Checked against a numpy restatement of the same recurrence (3 steps) and, with many steps, against the system it
    solves:  (XtX + X_nnz C1 X_nnz^T) y = X_nnz c + base - g X_nnz (c - 1)  -- NOT the Cholesky branch's right-hand side
    X_nnz c + base (:228-229): the two branches of the reference differ by g X_nnz (c - 1)."""
    m, X, Y0 = _problem(37)
    m.data[:] = np.abs(m.data) + 1.0
    k, lam, gb = X.shape[0], 0.2, 0.013
    p, i, x = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
    G = O.gramian(X, lam)
    base = -gb * X.sum(axis=1)
    Y3 = Y0.copy(order="F")
    loss = O.als_implicit(p, i, x, X, Y3, G, lam, 1, 3, global_bias=gb)
    Yinf = Y0.copy(order="F")
    O.als_implicit(p, i, x, X, Yinf, G, lam, 1, 60, global_bias=gb)
    tot = 0.0
    assert p[6] == p[5]                                      # the empty column is solved too
    for c in range(m.shape[1]):
        idx, val = i[p[c]:p[c + 1]], x[p[c]:p[c + 1]]
        Xn, y = X[:, idx], Y0[:, c].copy()
        r = Xn @ (val - (val - 1.0) * (Xn.T @ y + gb)) - G @ y + base
        pp, rs = r.copy(), r @ r
        for _ in range(3):
            Ap = G @ pp + Xn @ ((val - 1.0) * (Xn.T @ pp))
            a = rs / (pp @ Ap)
            y += a * pp
            r -= a * Ap
            rn = r @ r
            if rn < 1e-10:
                break
            pp = r + pp * (rn / rs)
            rs = rn
        assert np.allclose(Y3[:, c], y, rtol=1e-10, atol=1e-12), c
        tot += np.sum(val * ((1.0 - gb) - y @ Xn) ** 2) + lam * y @ y
        fix = np.linalg.solve(G + (Xn * (val - 1.0)) @ Xn.T, Xn @ val + base - gb * (Xn @ (val - 1.0)))
        assert np.allclose(Yinf[:, c], fix, rtol=1e-4, atol=1e-5), c   # (stops at |r|^2 < 1e-10)
    tot += lam * np.sum(X ** 2)
    assert np.isclose(loss, tot / m.nnz, rtol=1e-10)
    # below sqrt(eps) of the element type the bias counts as zero and empty columns become zeros again (:108-109,:178)
    Ya, Yb = Y0.copy(order="F"), Y0.copy(order="F")
    la = O.als_implicit(p, i, x, X, Ya, G, lam, 1, 3, global_bias=1e-9)
    lb = O.als_implicit(p, i, x, X, Yb, G, lam, 1, 3)
    assert np.array_equal(Ya, Yb) and la == lb and not Ya[:, 5].any()


def test_oracle_init_biases_implicit_global():
    """wrmf_utils.hpp:90-93: global_bias = sum(x) / (sum(x) + n_users n_items - nnz); :143,:159 the biases absorb it."""
    m, _, _ = _problem(36)
    m.data[:] = np.abs(m.data) + 1.0
    t = m.T.tocsc()
    csc = (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data)
    csr = (t.indptr.astype(np.int32), t.indices.astype(np.int32), t.data)
    ub, ib = np.zeros(m.shape[0]), np.zeros(m.shape[1])
    gb = O.init_biases_implicit(csc, csr, ub, ib, 0.1, calculate_global_bias=True)
    s = m.data.sum()
    assert np.isclose(gb, s / (s + m.shape[0] * m.shape[1] - m.nnz), rtol=1e-14)
    ub0, ib0 = np.zeros(m.shape[0]), np.zeros(m.shape[1])
    assert O.init_biases_implicit(csc, csr, ub0, ib0, 0.1, calculate_global_bias=False) == 0.0
    assert not np.allclose(ub, ub0) and np.all(np.isfinite(ub)) and np.all(np.isfinite(ib))


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [0, 2])
@pytest.mark.parametrize("with_biases", [False, True])
@pytest.mark.parametrize("k", [6, 34, 128])
@pytest.mark.parametrize("precision", ["float", "double"])
def test_hip_implicit_global_bias_half_iteration(solver, with_biases, k, precision):
    """Stateless als_implicit_{float,double} with global_bias (R/model_WRMF.R:456-496): the base vector comes back when
    initialize_bias_base, is read when not, and both give the oracle's factors and loss."""
    from rsparse_amd import als
    dt = np.float32 if precision == "float" else np.float64
    m, X, Y0 = _problem(300 + k, n_rows=300, n_cols=200, k=k, density=0.08)
    m.data[:] = np.abs(m.data) + 1.0
    gb = 0.021
    if solver == 2:
        X, Y0 = np.abs(X), np.abs(Y0)
    if with_biases:
        X[0, :] = 1.0; Y0[k - 1, :] = 1.0
    X32, Y32 = np.asfortranarray(X, dtype=np.float32), np.asfortranarray(Y0, dtype=np.float32)
    p, i, x = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
    X64, Y64 = np.asfortranarray(X32, dtype=np.float64), np.asfortranarray(Y32, dtype=np.float64).copy(order="F")
    XX = np.asfortranarray(X64[:-1]) if with_biases else X64
    base_ref = np.zeros(k)
    lref = O.als_implicit(p, i, x, X64, Y64, O.gramian(XX, 0.1), 0.1, solver, 3, with_biases=with_biases,
                          is_x_bias_last_row=True, global_bias=gb, base_out=None if with_biases else base_ref)
    Y32o = Y32.copy(order="F")
    O.als_implicit(p, i, x, X32, Y32o, O.gramian(np.asfortranarray(XX, dtype=np.float32), 0.1), 0.1, solver, 3,
                   with_biases=with_biases, is_x_bias_last_row=True, global_bias=gb)
    err32 = rel_fro(Y32o, Y64)
    tol = max(2e-3, 3 * err32) if solver == 2 else 1e-4
    Xd, Y = np.asfortranarray(X32, dtype=dt), np.asfortranarray(Y32, dtype=dt).copy(order="F")
    base = np.zeros(k, dtype=dt)
    csc = (m.shape[0], m.shape[1], p, i, x)
    loss = als.als_implicit(csc, Xd, Y, 0.1, 1, solver, 3, precision, with_biases, True, initialize_bias_base=True,
                            global_bias=gb, global_bias_base=base)
    assert rel_fro(Y, Y64) < tol, (rel_fro(Y, Y64), err32)
    assert abs(loss - lref) <= max(tol, 1e-4) * abs(lref)
    if with_biases:
        assert np.array_equal(Y[k - 1], Y32[k - 1].astype(dt))
        assert not base.any()                                # wrmf_implicit.hpp:111: untouched with biases
    else:
        assert rel_fro(base, base_ref) < 1e-5
        # second call reads the base it is handed (initialize_bias_base = FALSE, R/model_WRMF.R:317-318)
        Y2 = np.asfortranarray(Y32, dtype=dt).copy(order="F")
        loss2 = als.als_implicit(csc, Xd, Y2, 0.1, 1, solver, 3, precision, False, True, initialize_bias_base=False,
                                 global_bias=gb, global_bias_base=base.copy())
        assert np.array_equal(Y2, Y) and loss2 == loss
    # below sqrt(eps) OF THE ELEMENT TYPE the global bias counts as zero (wrmf_implicit.hpp:108-109): 3.45e-4 for the
    # float entry point, 1.49e-8 for the double one
    tiny = 1e-5 if precision == "float" else 1e-9
    Ya, Yb = (np.asfortranarray(Y32, dtype=dt).copy(order="F") for _ in range(2))
    la = als.als_implicit(csc, Xd, Ya, 0.1, 1, solver, 3, precision, with_biases, True, global_bias=tiny)
    lb = als.als_implicit(csc, Xd, Yb, 0.1, 1, solver, 3, precision, with_biases, True, global_bias=0.0)
    assert np.array_equal(Ya, Yb) and la == lb
    if precision == "double" and solver == 0 and not with_biases:
        # ... and 1e-5 is a bias for als_implicit<double>: the empty column is solved against global_bias_base
        Yc = np.asfortranarray(Y32, dtype=dt).copy(order="F")
        als.als_implicit(csc, Xd, Yc, 0.1, 1, solver, 3, precision, False, True, global_bias=1e-5)
        Yr = np.asfortranarray(Y32, dtype=np.float64).copy(order="F")
        O.als_implicit(p, i, x, X64, Yr, O.gramian(X64, 0.1), 0.1, solver, 3, global_bias=1e-5)
        assert Yr[:, 5].any() and not Yb[:, 5].any()
        assert rel_fro(Yc[:, 5], Yr[:, 5]) < 1e-3 and rel_fro(Yc, Yr) < 1e-4
    if with_biases:
        with pytest.raises(NotImplementedError):             # CG + user/item biases: UNSUPPORTED (the reference cannot run it)
            als.als_implicit(csc, Xd, Y, 0.1, 1, 1, 3, precision, with_biases, True, global_bias=gb)


def _long_row_problem(seed, k, n_rows=2500, n_cols=260):
    """implicit-feedback columns of every launch class: empty, 1..32, 33..512 and a few beyond 512 non-zeros (the
    normal-equation kernel; with so few of them every one is split across workgroups)"""
    rng = np.random.default_rng(seed)
    lens = np.concatenate([[0, 1, 2, 31, 32, 33, 64, 65, 128, 129, 256, 257, 512, 513, 700, 1100, 2300],
                           rng.integers(1, 90, n_cols - 17)])
    cols, rows = [], []
    for c, n in enumerate(lens):
        rows.append(np.sort(rng.choice(n_rows, size=int(n), replace=False)))
        cols.append(np.full(int(n), c))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = 1.0 + rng.geometric(0.5, size=rows.size).astype(np.float64)
    m = sp.csc_matrix((vals, (rows, cols)), shape=(n_rows, n_cols))
    m.sort_indices()
    X = np.asfortranarray((rng.standard_normal((k, n_rows)) * 0.1).astype(np.float32))
    Y = np.asfortranarray((rng.standard_normal((k, n_cols)) * 0.1).astype(np.float32))
    return m, X, Y


@pytest.mark.gpu
@pytest.mark.parametrize("cg_steps", [3, 0])
@pytest.mark.parametrize("k", [6, 34, 36, 64, 128])
@pytest.mark.parametrize("precision", ["float", "double"])
def test_hip_implicit_global_bias_cg_half_iteration(k, precision, cg_steps):
    """cg_solver_implicit_global_bias on the device (wrmf_implicit.hpp:35-57,203): the LDS-tile kernels (rank % 4 != 0), the
    register-resident buckets, the matrix-core dense product (rank 128, <= 32 non-zeros), the normal-equation kernel with
    its per-row term from launch_gb_row_terms (> 512 non-zeros, incl. rows split across workgroups) and empty columns,
    against the oracle in double; yardstick for the bound = the oracle in float (three CG steps from a warm start)."""
    from rsparse_amd import als
    dt = np.float32 if precision == "float" else np.float64
    m, X32, Y32 = _long_row_problem(900 + k, k)
    p, i, x = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
    gb, lam = 0.037, 0.1
    X64, Y64 = np.asfortranarray(X32, dtype=np.float64), np.asfortranarray(Y32, dtype=np.float64).copy(order="F")
    base_ref = np.zeros(k)
    lref = O.als_implicit(p, i, x, X64, Y64, O.gramian(X64, lam), lam, 1, cg_steps, global_bias=gb, base_out=base_ref, n_threads=8)
    Yo32 = Y32.copy(order="F")
    O.als_implicit(p, i, x, X32, Yo32, O.gramian(X32, lam), lam, 1, cg_steps, global_bias=gb, n_threads=8)
    norm = np.maximum(np.linalg.norm(Y64, axis=0), 1e-30)
    err32 = np.linalg.norm(Yo32 - Y64, axis=0) / norm
    Xd, Y = np.asfortranarray(X32, dtype=dt), np.asfortranarray(Y32, dtype=dt).copy(order="F")
    base = np.zeros(k - 1, dtype=dt)         # the R driver's allocation: rank - 1 entries (R/model_WRMF.R:292)
    csc = (m.shape[0], m.shape[1], p, i, x)
    loss = als.als_implicit(csc, Xd, Y, lam, 1, 1, cg_steps, precision, False, True, initialize_bias_base=True,
                            global_bias=gb, global_bias_base=base)
    err = np.linalg.norm(Y - Y64, axis=0) / norm
    bound = np.maximum(1e-4, 3.0 * err32)
    worst = int(np.argmax(err / bound))
    assert np.all(err <= bound), (worst, int(np.diff(p)[worst]), float(err[worst]), float(err32[worst]))
    assert abs(loss - lref) <= 1e-4 * abs(lref)
    assert Y[:, 0].any() == Y64[:, 0].any()                  # the empty column (solved when cg_steps > 0 moves it)
    assert rel_fro(base, base_ref[:k - 1]) < 1e-5            # min(len, rank) entries written, nothing beyond
    # initialize_bias_base = FALSE with the short R vector: recomputed from X, same result (include/rsparse_wrmf_hip.h)
    Y2 = np.asfortranarray(Y32, dtype=dt).copy(order="F")
    loss2 = als.als_implicit(csc, Xd, Y2, lam, 1, 1, cg_steps, precision, False, True, initialize_bias_base=False,
                             global_bias=gb, global_bias_base=np.zeros(k - 1, dtype=dt))
    assert np.array_equal(Y2, Y) and abs(loss2 - loss) <= 1e-12 * abs(loss)   # (the LDS-tile kernels claim rows dynamically:
    # ... and with the whole vector it is read                                   #  their loss partials sum in claim order): a different base gives a different solve
    if cg_steps:
        Y3 = np.asfortranarray(Y32, dtype=dt).copy(order="F")
        als.als_implicit(csc, Xd, Y3, lam, 1, 1, cg_steps, precision, False, True, initialize_bias_base=False,
                         global_bias=gb, global_bias_base=np.zeros(k, dtype=dt))
        assert not np.array_equal(Y3, Y)


@pytest.mark.gpu
@pytest.mark.parametrize("explicit", [True, False])
@pytest.mark.parametrize("global_bias", [True, False])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_hip_stateless_initialize_biases(ml_train, explicit, global_bias, dtype):
    """rsparse_hip_initialize_biases_{float,double} = the .Call target of src/wrmf_init.cpp:5-34 (9 arguments, the two
    matrices flattened to their slots)."""
    from rsparse_amd import als
    n_user, n_item, p, i, x = ml_train
    t = O.csc_transpose(n_user, n_item, p, i, x)
    ub, ib = np.zeros(n_user), np.zeros(n_item)
    x_ref, t_ref = x.astype(np.float64).copy(), t[2].astype(np.float64).copy()
    if explicit:
        gb = O.init_biases_explicit((p, i, x_ref), (t[0], t[1], t_ref), ub, ib, 0.1, True, False, global_bias)
    else:
        gb = O.init_biases_implicit((p, i, x_ref), (t[0], t[1], t_ref), ub, ib, 0.1, False,
                                    calculate_global_bias=global_bias)
    dub, dib = np.zeros(n_user, dtype=dtype), np.zeros(n_item, dtype=dtype)
    xv, tv = x.astype(np.float64).copy(), t[2].astype(np.float64).copy()
    got = als.initialize_biases((n_user, n_item, p, i, xv), (n_item, n_user, t[0], t[1], tv), dub, dib, 0.1, True, False,
                                global_bias, explicit)
    assert abs(got - gb) <= 1e-6 * max(1.0, abs(gb))
    assert (gb != 0.0) == global_bias
    # explicit + global: the mean leaves both @x slots (wrmf_utils.hpp:41-52); otherwise they are untouched
    assert np.allclose(xv, x_ref, atol=1e-6) and np.allclose(tv, t_ref, atol=1e-6)
    if not (explicit and global_bias):
        assert np.array_equal(xv, x) and np.array_equal(tv, t[2])
    assert np.allclose(dub, ub, rtol=2e-4, atol=2e-6, equal_nan=True)
    assert np.allclose(dib, ib, rtol=2e-4, atol=2e-6, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["cholesky", "nnls", "conjugate_gradient"])
@pytest.mark.parametrize("bias", [False, True])
def test_hip_wrmf_implicit_global_bias(ml_train, solver, bias):
    """WRMF$new(feedback = "implicit", with_global_bias = TRUE) (R/model_WRMF.R:128-131, :262-272, :317-318) against the
    oracle driver, same initial factors."""
    from rsparse_amd import WRMF, _lib
    n_user, n_item, p, i, x = ml_train
    train = sp.csc_matrix((x, i, p), shape=(n_user, n_item))
    cg = solver == "conjugate_gradient"
    if cg and bias:       # the reference cannot run CG with user/item biases (wrmf_implicit.hpp:189,197)
        with pytest.raises(_lib.UnsupportedOnDevice):
            WRMF(rank=8, feedback="implicit", solver=solver, with_user_item_bias=True, with_global_bias=True, precision="float")
        return
    rng = np.random.default_rng(77 + bias)
    rank0 = 8
    rank = rank0 + 2 * bias
    U0 = np.abs(rng.standard_normal((n_user, rank)) * 0.01).astype(np.float32)
    V0 = np.abs(rng.standard_normal((rank, n_item)) * 0.01).astype(np.float32)
    model = WRMF(rank=rank0, lambda_=0.1, feedback="implicit", solver=solver, with_user_item_bias=bias,
                 with_global_bias=True, precision="float", init=None if cg else V0.copy())
    model._init_user_factors = U0
    emb = model.fit_transform(train, n_iter=3, convergence_tol=-1)
    ref = O.OracleWRMF(rank0, lam=0.1, feedback="implicit", solver=solver, dtype=np.float64, n_threads=8,
                       with_user_item_bias=bias, with_global_bias=True)
    ref_emb = ref.fit_transform(n_user, n_item, p, i, x, U0.T.astype(np.float64), n_iter=3, convergence_tol=-1,
                                init_components=None if cg else V0.astype(np.float64))
    if solver == "nnls":                                     # R/model_WRMF.R:90-93: nnls switches the global bias off
        assert model.global_bias == 0.0 and ref.global_bias == 0.0
    else:
        assert model.global_bias > 0 and abs(model.global_bias - ref.global_bias) < 1e-6 * ref.global_bias
    tol = 1e-4
    if solver != "cholesky":     # yardstick: the same fit on the oracle in float
        ref32 = O.OracleWRMF(rank0, lam=0.1, feedback="implicit", solver=solver, dtype=np.float32, n_threads=8,
                             with_user_item_bias=bias, with_global_bias=True)
        e32 = ref32.fit_transform(n_user, n_item, p, i, x, U0.T.copy(), n_iter=3, convergence_tol=-1,
                                  init_components=None if cg else V0.copy())
        tol = max(tol, 3 * rel_fro(e32, ref_emb), 3 * rel_fro(ref32.components, ref.components))
    assert rel_fro(emb, ref_emb) < tol and rel_fro(model.components, ref.components) < tol
    assert np.allclose([l[1] for l in model.losses], [l[1] for l in ref.losses], rtol=max(tol, 1e-4))
    assert np.array_equal(emb, model.transform(train))


@pytest.mark.gpu
@pytest.mark.parametrize("explicit", [True, False])
@pytest.mark.parametrize("precision", ["float", "double"])
def test_hip_sweepwise_bias_initialisation_equals_the_fused_one(ml_train, explicit, precision):
    """What a sharded fit runs (ShardedALS.initialize_biases: one C-ABI call per sweep and sub-block,
    rsparse_hip_bias_{sweep_explicit,prep_implicit,sweep_implicit}[_f64]_device) against the one-call initialisation of the
    single-rank fit (rsparse_hip_initialize_biases_*_device) on the same resident matrix: the same sweeps, so the same bits up
    to the order of the two global sums (the mean of the values, the mean of a bias vector)."""
    import torch
    from rsparse_amd.engine import HipBackend, Layout, ShardedALS
    n_user, n_item, p, i, x = ml_train
    be = HipBackend()
    tdt = torch.float64 if precision == "double" else torch.float32
    c_ui = sp.csc_matrix((x, i, p), shape=(n_user, n_item))
    c_iu = sp.csc_matrix(c_ui.T)
    c_iu.sort_indices()

    def blocks():
        return tuple((be.to_device(m.indptr, torch.int32), be.to_device(m.indices, torch.int32), be.to_device(m.data, tdt))
                     for m in (c_ui, c_iu))
    out = {}
    for mode, n_sub in (("fused", 1), ("sweeps", 3)):
        b_ui, b_iu = blocks()
        lay_u, lay_i = Layout(n_user, [(0, n_user)], n_sub), Layout(n_item, [(0, n_item)], n_sub)
        als = ShardedALS(be, n_user, n_item, 8, b_ui, b_iu, c_ui.nnz, feedback="explicit" if explicit else "implicit",
                         lambda_=0.1, with_bias=True, lay_user=lay_u, lay_item=lay_i)
        ub = torch.zeros(lay_u.rows, dtype=tdt, device=be.device)
        ib = torch.zeros(lay_i.rows, dtype=tdt, device=be.device)
        if mode == "fused":
            gb = (be.initialize_biases_explicit(als.csc_items, als.csc_users, ub, ib, 0.1, True, False, True) if explicit else
                  be.initialize_biases_implicit(als.csc_items, als.csc_users, ub, ib, 0.1, False, True))
        else:
            gb = als.initialize_biases(ub, ib, False, True)
        out[mode] = (gb, lay_u.to_global(ub.reshape(-1, 1)).cpu().numpy().ravel(), lay_i.to_global(ib.reshape(-1, 1)).cpu().numpy().ravel(),
                     als.x_items.cpu().numpy())
    eps = 1e-12 if precision == "double" else 2e-6
    (g0, u0, i0, x0), (g1, u1, i1, x1) = out["fused"], out["sweeps"]
    assert abs(g0 - g1) <= eps * max(1.0, abs(g0))
    assert np.array_equal(np.isnan(u0), np.isnan(u1)) and np.array_equal(np.isnan(i0), np.isnan(i1))
    assert rel_fro(np.nan_to_num(u1), np.nan_to_num(u0)) < 10 * eps and rel_fro(np.nan_to_num(i1), np.nan_to_num(i0)) < 10 * eps
    assert rel_fro(x1, x0) < 10 * eps      # explicit: the mean left the resident values both ways
