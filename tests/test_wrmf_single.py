"""The one-rank driver of `WRMF.fit_transform / transform` (rsparse_amd/wrmf.py, the path a single GPU runs) on CPU: the
numerics are the stand-in backend (tests/oracle_backend.py), what is under test is the host logic around the half-iterations
-- R/model_WRMF.R:173-360: preprocessing, the initial factors and their bias rows, the second orientation, the global
bias in its three forms, the order of the two half-iterations, the loss pair and the stopping rule, the Gramian kept for
`transform`, the final exact solve -- against the one-process oracle driver (oracle/wrmf_oracle.py: OracleWRMF), which
follows the same lines of the reference independently.  The GPU suite runs the same comparisons through the HIP backend
(tests/test_wrmf_core.py, tests/test_bias.py, tests/test_f64.py); this file is what catches a host-side regression without a GPU."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import rel_fro


def _problem(seed=11, n_user=173, n_item=59):
    rng = np.random.default_rng(seed)
    lens = np.clip(rng.lognormal(1.5, 1.0, n_user).astype(int), 0, 40)
    rows = np.repeat(np.arange(n_user), lens)
    cols = np.concatenate([rng.choice(n_item, size=l, replace=False) for l in lens])
    vals = 1.0 + rng.geometric(0.5, size=rows.size)
    m = sp.csr_matrix((vals, (rows, cols)), shape=(n_user, n_item))
    new = sp.csr_matrix((rng.random((19, n_item)) < 0.15) * 2.0)
    return m, new


def _pair(feedback, solver, bias, gb, precision, n_iter=3, tol=-1, preprocess=None, lam=0.1, k=6, dynamic_lambda=True):
    """(model, embeddings, embeddings of new users), (oracle driver, its embeddings, its embeddings of the new users)"""
    from oracle import wrmf_oracle as O
    from oracle_backend import OracleBackend
    from rsparse_amd import WRMF
    m, new = _problem()
    dt = np.float32 if precision == "float" else np.float64
    kk = k + (2 if bias else 0)
    rng = np.random.default_rng(5)
    U0 = (rng.standard_normal((m.shape[0], kk)) * 0.01).astype(dt)
    V0 = None if solver == "conjugate_gradient" else (rng.standard_normal((kk, m.shape[1])) * 0.01).astype(dt)
    kw = {} if preprocess is None else {"preprocess": preprocess}
    model = WRMF(rank=k, lambda_=lam, feedback=feedback, solver=solver, with_user_item_bias=bias, with_global_bias=gb,
                 precision=precision, dynamic_lambda=dynamic_lambda, backend=OracleBackend(), rng=123, **kw)
    model._init_user_factors = U0
    if V0 is not None:
        model.components = V0.copy()
    emb = model.fit_transform(m, n_iter=n_iter, convergence_tol=tol)
    emb_new = model.transform(new)

    c = sp.csc_matrix(m, dtype=np.float64)
    if preprocess is not None:
        c = preprocess(c)
    c.sort_indices()
    ref = O.OracleWRMF(k, lam=lam, feedback=feedback, solver=solver, dtype=dt, n_threads=4, with_user_item_bias=bias,
                       with_global_bias=gb, dynamic_lambda=dynamic_lambda)
    ref_emb = ref.fit_transform(m.shape[0], m.shape[1], c.indptr.astype(np.int32), c.indices.astype(np.int32),
                                c.data.astype(np.float64), U0.T.copy(), n_iter=n_iter, convergence_tol=tol,
                                init_components=V0)
    nt = sp.csc_matrix(new.T, dtype=np.float64)
    if preprocess is not None:
        nt = preprocess(nt)
    nt.sort_indices()
    ref_new = ref.transform(nt.indptr.astype(np.int32), nt.indices.astype(np.int32), nt.data.astype(np.float64))
    return (model, emb, emb_new), (ref, ref_emb, ref_new)


CASES = [  # feedback, solver, user/item biases, global bias, precision
    ("implicit", "conjugate_gradient", False, False, "float"),
    ("implicit", "conjugate_gradient", False, False, "double"),      # the constructor's defaults
    ("implicit", "conjugate_gradient", False, True, "double"),
    ("implicit", "cholesky", False, True, "float"),
    ("implicit", "cholesky", True, False, "double"),
    ("implicit", "cholesky", True, True, "float"),
    ("implicit", "nnls", False, False, "double"),
    ("explicit", "cholesky", False, True, "double"),
    ("explicit", "cholesky", True, True, "float"),
    ("explicit", "conjugate_gradient", True, False, "double"),
    ("explicit", "conjugate_gradient", False, False, "float"),
    ("explicit", "nnls", True, False, "double"),
]


@pytest.mark.parametrize("feedback,solver,bias,gb,precision", CASES)
def test_one_rank_driver_matches_the_oracle_driver(feedback, solver, bias, gb, precision):
    (model, emb, emb_new), (ref, ref_emb, ref_new) = _pair(feedback, solver, bias, gb, precision)
    # the same arithmetic underneath (the stand-in backend calls the oracle's solves): what may differ is the rounding of
    # the values that the driver hands over in float where the oracle driver keeps the dgCMatrix doubles
    tol = 1e-9 if precision == "double" else (2e-3 if solver == "nnls" else 1e-4)
    assert emb.dtype == (np.float64 if precision == "double" else np.float32)
    assert model.components.shape == ref.components.shape and model.components.flags.f_contiguous
    assert abs(model.global_bias - ref.global_bias) <= (1e-12 if precision == "double" else 1e-6) * max(1.0, abs(ref.global_bias))
    assert (model.global_bias != 0.0) == (gb and solver != "nnls")
    assert rel_fro(model.components, ref.components) < tol
    assert rel_fro(emb, ref_emb) < tol
    assert emb_new.shape == ref_new.shape and rel_fro(emb_new, ref_new) < tol
    assert len(model.losses) == len(ref.losses) == 3
    # (users without ratings start from a 0 / 0 bias under dynamic lambda, in the reference too: the first item-half loss
    # carries that NaN through the regulariser; the user half solves those users and it is gone)
    assert np.allclose([l[1] for l in model.losses], [l[1] for l in ref.losses], rtol=max(tol, 1e-9))
    assert np.allclose([l[0] for l in model.losses], [l[0] for l in ref.losses], rtol=max(tol, 1e-9), equal_nan=True)
    if solver == "nnls":
        assert model.components.min() >= 0 and emb.min() >= 0
    if bias:   # the rows of ones stay ones (R/model_WRMF.R:208-245, :427-429)
        assert np.all(model.components[-1] == 1.0) and np.all(emb[:, 0] == 1.0) and np.all(emb_new[:, 0] == 1.0)


@pytest.mark.parametrize("feedback,tol", [("implicit", None), ("explicit", None), ("implicit", 0.05)])
def test_stopping_rule_stops_where_the_reference_stops(feedback, tol):
    """R/model_WRMF.R:173 (the defaults 0.005 / 0.001) and :332-335: loss_prev / loss - 1 < convergence_tol ends the fit
    after the iteration that measured it."""
    (model, emb, _), (ref, ref_emb, _) = _pair(feedback, "conjugate_gradient", False, False, "double", n_iter=40, tol=tol)
    assert 1 < len(ref.losses) < 40, "the problem should converge inside the iteration budget"
    assert len(model.losses) == len(ref.losses)
    assert rel_fro(emb, ref_emb) < 1e-9
    lu = [l[1] for l in model.losses]
    eff = tol if tol is not None else (0.005 if feedback == "implicit" else 0.001)
    assert lu[-2] / lu[-1] - 1 < eff and all(lu[i - 1] / lu[i] - 1 >= eff for i in range(1, len(lu) - 1))


def test_preprocess_applies_to_fit_and_transform():
    """R/model_WRMF.R:184-188 and :376-379: the user's function sees the matrix before anything else does, in both calls."""
    def log1p(x):
        x = x.copy()
        x.data = np.log1p(x.data)
        return x
    (model, emb, emb_new), (ref, ref_emb, ref_new) = _pair("implicit", "cholesky", False, False, "double", preprocess=log1p)
    assert rel_fro(emb, ref_emb) < 1e-9 and rel_fro(emb_new, ref_new) < 1e-9
    (_, emb_raw, _), _ = _pair("implicit", "cholesky", False, False, "double")
    assert rel_fro(emb, emb_raw) > 1e-3      # and it made a difference


def test_static_lambda_and_zero_lambda():
    for lam, dyn in ((0.1, False), (0.0, True)):
        (model, emb, emb_new), (ref, ref_emb, ref_new) = _pair("explicit", "cholesky", False, False, "double", lam=lam,
                                                              dynamic_lambda=dyn, k=4)
        # lambda = 0: users without ratings have a singular system; the reference's solve() returns what it returns
        # for them, the comparison is over the users that have data
        m, _ = _problem()
        has = np.diff(m.indptr) > 0
        assert rel_fro(emb[has], ref_emb[has]) < 1e-8
        assert np.allclose([l[1] for l in model.losses], [l[1] for l in ref.losses], rtol=1e-9)


def test_fit_checks_its_input_like_the_reference():
    from oracle_backend import OracleBackend
    from rsparse_amd import WRMF
    m, _ = _problem()
    neg = m.copy().astype(np.float64)
    neg.data[3] = -1.0
    with pytest.raises(ValueError, match="c_ui@x >= 0"):                           # R/model_WRMF.R:195-197
        WRMF(rank=4, feedback="implicit", precision="float", backend=OracleBackend()).fit_transform(neg, n_iter=1)
    with pytest.raises(ValueError, match="c_ui@x >= 0"):
        WRMF(rank=4, feedback="explicit", solver="nnls", precision="float", backend=OracleBackend()).fit_transform(neg, n_iter=1)
    # explicit feedback without the non-negative solver takes negative ratings
    WRMF(rank=4, lambda_=0.1, feedback="explicit", solver="cholesky", precision="float", backend=OracleBackend(),
         rng=1).fit_transform(neg, n_iter=1)
    bad = WRMF(rank=4, feedback="implicit", precision="float", backend=OracleBackend(),
               init=np.zeros((4, m.shape[1] + 1), dtype=np.float32))
    with pytest.raises(ValueError, match="rank x n_item"):                         # :246-248
        bad.fit_transform(m, n_iter=1)
    fitted = WRMF(rank=4, lambda_=0.1, feedback="implicit", precision="float", backend=OracleBackend(), rng=1)
    fitted.fit_transform(m, n_iter=1)
    with pytest.raises(ValueError):                                                # :371: ncol(x) == ncol(components)
        fitted.transform(sp.csr_matrix((3, m.shape[1] + 2)))


def test_refit_warm_starts_from_components():
    """A second fit_transform starts from `components` (R/model_WRMF.R:246-250): two fits of n iterations == the factors a
    model initialised with the first fit's components reaches."""
    from oracle_backend import OracleBackend
    from rsparse_amd import WRMF
    m, _ = _problem()
    rng = np.random.default_rng(2)
    # (user factors of unit scale: with the usual 0.01 the first item systems are lambda I to three digits, three CG steps
    # solve them from any start and the warm start would leave no trace)
    U0 = rng.standard_normal((m.shape[0], 5))
    a = WRMF(rank=5, lambda_=0.1, feedback="implicit", precision="double", backend=OracleBackend(), rng=7)
    a._init_user_factors = U0
    a.fit_transform(m, n_iter=2, convergence_tol=-1)
    first = a.components.copy()
    a.fit_transform(m, n_iter=2, convergence_tol=-1)
    b = WRMF(rank=5, lambda_=0.1, feedback="implicit", precision="double", backend=OracleBackend(), rng=7, init=first)
    b._init_user_factors = U0
    b.fit_transform(m, n_iter=2, convergence_tol=-1)
    assert rel_fro(a.components, b.components) < 1e-12       # (the oracle's threaded sums are not ordered: not bit-equal)
    c = WRMF(rank=5, lambda_=0.1, feedback="implicit", precision="double", backend=OracleBackend(), rng=7)
    c._init_user_factors = U0
    c.fit_transform(m, n_iter=2, convergence_tol=-1)
    assert rel_fro(c.components, first) < 1e-12 and rel_fro(a.components, first) > 1e-4   # the start mattered


def test_csr_input_is_uploaded_as_it_stands_and_gives_the_same_device_arrays():
    """A canonical CSR matrix is the item-user orientation already (R/model_WRMF.R:190): fit_transform uploads it and has the
    device produce the user-item one, instead of converting on the host.  Both routes must hand the SAME arrays to the
    solver (bit for bit: index and value work), whatever the index / value types of the input; anything that is not a
    canonical CSR matrix, and any model with a `preprocess`, keeps the conversion."""
    from oracle_backend import OracleBackend
    from rsparse_amd import WRMF

    class Spy(OracleBackend):
        def __init__(self):
            self.made, self.transposed = [], []

        def make_csc(self, n_rows, n_cols, p, i, x):
            self.made.append((n_rows, n_cols, p.numpy().copy(), i.numpy().copy(), x.numpy().copy()))
            return super().make_csc(n_rows, n_cols, p, i, x)

        def transpose_csc(self, n_rows, n_cols, p, i, x):
            self.transposed.append((n_rows, n_cols))
            return super().transpose_csc(n_rows, n_cols, p, i, x)

    def uploads(x, **kw):
        be = Spy()
        model = WRMF(rank=4, lambda_=0.1, feedback="implicit", precision="float", backend=be, rng=3, **kw)
        model.fit_transform(x, n_iter=1, convergence_tol=-1)
        return be

    m, _ = _problem()
    m = sp.lil_matrix(m, dtype=np.float64)
    m[5, :] = 0                      # a user and an item without data
    m[:, 7] = 0
    m = sp.csr_matrix(m)
    m.eliminate_zeros()
    m.sort_indices()
    n_user, n_item = m.shape
    want = uploads(sp.csc_matrix(m))
    assert want.transposed == [(n_user, n_item)]                       # the CSC route: c_iu made on the device
    as_f32 = m.astype(np.float32)
    as_i64 = m.copy()
    as_i64.indptr, as_i64.indices = as_i64.indptr.astype(np.int64), as_i64.indices.astype(np.int64)
    for name, x in (("f64", m), ("f32", as_f32), ("int64 indices", as_i64)):
        assert x.format == "csr" and x.has_canonical_format
        got = uploads(x)
        assert got.transposed == [(n_item, n_user)], name                # the CSR route: c_ui made on the device
        assert len(got.made) == len(want.made) > 0
        for g, w in zip(got.made, want.made):
            assert g[:2] == w[:2], name
            for a, b in zip(g[2:], w[2:]):
                assert a.dtype == b.dtype and np.array_equal(a, b), name
    # not canonical: unsorted indices, duplicates -> the host conversion, as before
    rng = np.random.default_rng(0)
    shuffled = m.copy()
    for r in range(n_user):
        lo, hi = shuffled.indptr[r], shuffled.indptr[r + 1]
        perm = rng.permutation(hi - lo)
        shuffled.indices[lo:hi], shuffled.data[lo:hi] = shuffled.indices[lo:hi][perm], shuffled.data[lo:hi][perm]
    shuffled.has_sorted_indices = False
    shuffled.has_canonical_format = False
    got = uploads(shuffled)
    assert got.transposed == [(n_user, n_item)]
    for g, w in zip(got.made, want.made):
        assert all(np.array_equal(a, b) for a, b in zip(g[2:], w[2:]))
    # a model with a preprocess function sees the CsparseMatrix, whatever came in
    seen = []

    def pre(c):
        seen.append(c.format)
        return c
    got = uploads(m, preprocess=pre)
    assert seen == ["csc"] and got.transposed == [(n_user, n_item)]
    # the input is not modified -- neither its types nor, when the global mean leaves the resident values (:278-282), its data
    assert as_f32.dtype == np.float32 and as_i64.indices.dtype == np.int64
    before = m.data.copy()
    model = WRMF(rank=4, lambda_=0.1, feedback="explicit", solver="cholesky", with_global_bias=True, precision="double",
                 backend=OracleBackend(), rng=3)
    model.fit_transform(m, n_iter=1, convergence_tol=-1)
    assert model.global_bias == pytest.approx(before.mean(), rel=1e-12) and np.array_equal(m.data, before)


def test_fit_logs_the_loss_of_every_half_iteration_like_the_reference(ml_train, caplog):
    """R/model_WRMF.R:324,330,333: `logger$info("iter %d (items) loss = %.4f", i, loss)` after the item half, the same for the
    user half, "Converged after %d iterations" when the tolerance is met (VERDICT r05 missing #6: the losses existed only as
    `model.losses`).  Python's `logging`, logger "rsparse_amd"; silent unless configured, as lgr's threshold makes the reference's."""
    import logging
    import scipy.sparse as sp
    from rsparse_amd import WRMF
    from oracle_backend import OracleBackend
    n_user, n_item, p, i, x = ml_train
    train = sp.csc_matrix((x, i, p), shape=(n_user, n_item))
    model = WRMF(rank=8, lambda_=0.1, feedback="implicit", solver="conjugate_gradient", precision="double", rng=3,
                 backend=OracleBackend())
    with caplog.at_level(logging.INFO, logger="rsparse_amd"):
        model.fit_transform(train, n_iter=3, convergence_tol=-1)
    lines = [r.getMessage() for r in caplog.records if r.name == "rsparse_amd"]
    assert lines == [m for it, (li, lu) in enumerate(model.losses, 1)
                     for m in ("iter %d (items) loss = %.4f" % (it, li), "iter %d (users) loss = %.4f" % (it, lu))]
    caplog.clear()
    with caplog.at_level(logging.INFO, logger="rsparse_amd"):
        model.fit_transform(train, n_iter=50, convergence_tol=0.5)
    assert caplog.records[-1].getMessage() == "Converged after %d iterations" % len(model.losses)
