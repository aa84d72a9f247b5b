"""Layer (4) of the C ABI: the multi-GPU context (csrc/wrmf_ctx.cpp; SURVEY.md 8b "callable from one host thread, internally
drives 1-8 GPUs", 8e).  On the one GPU of a test box the ranks are threads of the library with streams of their own
(RSPARSE_HIP_COMM_SHARED: a collective is a host barrier + device copies); ownership, sub-block storage, exchange points and
summation orders are the production code.  Checked: the context at 1 rank == the single-GPU device-resident layer through
engine.py bit for bit; 1 / 2 / 4 / 8 ranks against the fp64 oracle at the north star's tolerance (1e-4, or 3 x the fp32 oracle's own
distance where three fp32 iterations of the matrix are further than that) and against each other (losses 1e-5); ranks
without rows; explicit feedback with the dynamic regulariser; the exact solve; argument errors without a device."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp

from rsparse_amd import _lib

gpu = pytest.mark.gpu


def _matrix(n_user, n_item, seed, mean_deg=30, heavy=False):
    rng = np.random.default_rng(seed)
    deg = np.clip(np.round(rng.lognormal(np.log(mean_deg) - 0.5, 1.0, n_user)), 0, n_item).astype(np.int64)
    if heavy:
        deg[0] = n_item // 2          # one user with a large share of the entries: blocks of very different row counts
    rows = np.repeat(np.arange(n_user), deg)
    cols = np.concatenate([rng.choice(n_item, size=int(d), replace=False) for d in deg]) if deg.sum() else np.zeros(0, np.int64)
    if heavy:
        cols[rng.random(cols.size) < 0.4] = 3   # ... and one item in most rows (duplicates are summed below)
    vals = 1.0 + rng.geometric(0.5, size=rows.size).astype(np.float64)
    x = sp.csr_matrix((vals, (rows, cols)), shape=(n_user, n_item))
    x.sum_duplicates()
    return x


def _fit(ctx, x, U0, V0, feedback, solver, n_iter=3, lam=0.1, n_sub=(0, 0)):
    ctx.set_matrix(x, n_sub=n_sub)
    ctx.set_factors(U0, V0)
    losses = []
    for _ in range(n_iter):
        li = ctx.half_iteration("items", feedback, lam, solver)
        lu = ctx.half_iteration("users", feedback, lam, solver)
        losses.append((li, lu))
    U, V = ctx.get_factors()
    return U, V, np.asarray(losses)


def _rows_err(a, b):
    return float((np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)).max())


def _fro(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _oracle_fit(x, U0, V0, feedback, solver, dtype, n_iter=3, lam=0.1):
    """the same half-iterations on the CPU oracle (fp64: the reference's arithmetic; fp32: the yardstick of what single precision does
    to this fit)"""
    from oracle import wrmf_oracle as O
    c = sp.csc_matrix(x); c.sort_indices()
    ct = sp.csc_matrix(c.T); ct.sort_indices()
    Ur, Vr = np.array(U0.T, dtype=dtype, order="F", copy=True), np.array(V0.T, dtype=dtype, order="F", copy=True)   # (copies: the oracle solves in place)
    cnt_u, cnt_i = np.diff(ct.indptr).astype(dtype), np.diff(c.indptr).astype(dtype)
    sc = {"cholesky": 0, "conjugate_gradient": 1}[solver]
    for _ in range(n_iter):
        if feedback == "implicit":
            O.als_implicit(c.indptr, c.indices, c.data, Ur, Vr, O.gramian(Ur, lam), lam, sc, 3, n_threads=8)
            O.als_implicit(ct.indptr, ct.indices, ct.data, Vr, Ur, O.gramian(Vr, lam), lam, sc, 3, n_threads=8)
        else:
            O.als_explicit(c.indptr, c.indices, c.data, Ur, Vr, cnt_u, lam, sc, 3, True, n_threads=8)
            O.als_explicit(ct.indptr, ct.indices, ct.data, Vr, Ur, cnt_i, lam, sc, 3, True, n_threads=8)
    return Ur.T.astype(np.float64), Vr.T.astype(np.float64)


def _check_fit(tag, U, V, U1, V1, ref64, ref32):
    """A sharded fit is held to what the single-GPU fit is held to: the fp64 oracle at the north star's 1e-4 -- or, where three
    fp32 CG iterations of this matrix are themselves further than that from fp64, 3 x the fp32 ORACLE's own distance (the rule
    of tests/test_hip_parity.py); and it must be the same fit as the one-rank run up to what a differently ordered Gramian sum
    does to such a trajectory (1e-3: a sanity bound, the oracle bound is the claim)."""
    (Uo, Vo), (Uf, Vf) = ref64, ref32
    yard = max(1e-4, 3.0 * max(_fro(Uf, Uo), _fro(Vf, Vo)))
    got = max(_fro(U, Uo), _fro(V, Vo))
    assert got <= yard, (tag, got, yard)
    assert _fro(U, U1) < 1e-3 and _fro(V, V1) < 1e-3, (tag, _fro(U, U1), _fro(V, V1))


@gpu
@pytest.mark.parametrize("k,feedback,solver", [(16, "implicit", "conjugate_gradient"), (128, "implicit", "conjugate_gradient"),
                                               (64, "implicit", "cholesky"), (32, "explicit", "conjugate_gradient"),
                                               (128, "implicit", "cholesky")])
def test_ranks_agree_with_one_rank_and_with_the_oracle(k, feedback, solver):
    from rsparse_amd.ctx import MultiGpuALS
    n_user, n_item = 3000, 700
    x = _matrix(n_user, n_item, seed=k)
    rng = np.random.default_rng(1)
    U0 = (rng.standard_normal((n_user, k)) * 0.01).astype(np.float32)
    V0 = (rng.standard_normal((n_item, k)) * 0.01).astype(np.float32) if solver == "cholesky" else np.zeros((n_item, k), np.float32)
    one = MultiGpuALS(1, comm="shared")
    U1, V1, L1 = _fit(one, x, U0, V0, feedback, solver)
    one.close()
    ref64 = _oracle_fit(x, U0, V0, feedback, solver, np.float64)
    ref32 = _oracle_fit(x, U0, V0, feedback, solver, np.float32)
    _check_fit("1 rank", U1, V1, U1, V1, ref64, ref32)
    for n_ranks, n_sub in ((2, (0, 0)), (4, (3, 2)), (8, (8, 4))):
        ctx = MultiGpuALS(n_ranks, comm="shared")
        U, V, L = _fit(ctx, x, U0, V0, feedback, solver, n_sub=n_sub)
        info = ctx.info()
        ctx.close()
        assert info["ranks"] == n_ranks and info["nnz"] == x.nnz and info["users_rank0"] < n_user
        _check_fit("%d ranks" % n_ranks, U, V, U1, V1, ref64, ref32)
        assert np.allclose(L, L1, rtol=2e-4, atol=0), (n_ranks, L, L1)   # (third-iteration losses of two fp32 trajectories)


@gpu
def test_one_rank_is_the_single_gpu_path_bit_for_bit():
    """ws = 1, one sub-block: the context issues exactly the calls engine.ShardedALS issues on one GPU"""
    import torch
    from rsparse_amd.ctx import MultiGpuALS
    from rsparse_amd.engine import HipBackend, ShardedALS
    n_user, n_item, k = 2500, 600, 64
    x = _matrix(n_user, n_item, seed=5)
    rng = np.random.default_rng(2)
    U0 = (rng.standard_normal((n_user, k)) * 0.01).astype(np.float32)
    V0 = np.zeros((n_item, k), np.float32)
    one = MultiGpuALS(1, comm="shared")
    U1, V1, L1 = _fit(one, x, U0, V0, "implicit", "conjugate_gradient", n_iter=2)
    one.close()
    be = HipBackend(0)
    c_ui = sp.csc_matrix(x); c_ui.sort_indices()
    c_iu = sp.csc_matrix(c_ui.T); c_iu.sort_indices()
    dev = lambda m: (be.to_device(m.indptr, torch.int32), be.to_device(m.indices, torch.int32), be.to_device(m.data, torch.float32))
    als = ShardedALS(be, n_user, n_item, k, dev(c_ui), dev(c_iu), x.nnz, lambda_=0.1)
    U, V = be.to_device(U0, torch.float32), be.to_device(V0, torch.float32)
    L = []
    for _ in range(2):
        L.append((als.half_iteration("items", U, V, 1), als.half_iteration("users", U, V, 1)))
    assert np.array_equal(U.cpu().numpy(), U1) and np.array_equal(V.cpu().numpy(), V1)
    assert np.allclose(np.asarray(L), L1, rtol=1e-12)


@gpu
def test_ranks_without_rows_and_very_unequal_blocks():
    """eight ranks on a matrix whose nnz-balanced cut leaves ranks WITHOUT users and ranks without items (one user with a large
    share of the entries, one item in most rows): every collective is still entered by every rank"""
    from rsparse_amd.ctx import MultiGpuALS
    n_user, n_item, k = 400, 90, 32
    x = _matrix(n_user, n_item, seed=9, mean_deg=6, heavy=True)
    rng = np.random.default_rng(3)
    U0 = (rng.standard_normal((n_user, k)) * 0.01).astype(np.float32)
    V0 = np.zeros((n_item, k), np.float32)
    one = MultiGpuALS(1, comm="shared")
    U1, V1, L1 = _fit(one, x, U0, V0, "implicit", "conjugate_gradient")
    one.close()
    ctx = MultiGpuALS(8, comm="shared")
    U, V, L = _fit(ctx, x, U0, V0, "implicit", "conjugate_gradient", n_sub=(8, 4))
    ctx.close()
    ref64 = _oracle_fit(x, U0, V0, "implicit", "conjugate_gradient", np.float64)
    ref32 = _oracle_fit(x, U0, V0, "implicit", "conjugate_gradient", np.float32)
    _check_fit("1 rank", U1, V1, U1, V1, ref64, ref32)
    _check_fit("8 ranks", U, V, U1, V1, ref64, ref32)
    assert np.allclose(L, L1, rtol=2e-4)


@gpu
def test_rccl_context_on_one_device_and_its_errors():
    """RCCL with one rank (no communicator is needed: the library must not even be loaded); two ranks on one device are refused"""
    from rsparse_amd.ctx import MultiGpuALS
    one = MultiGpuALS(1, comm="rccl")
    assert one.info()["rccl"] == 0
    one.close()
    with pytest.raises(_lib.RsparseHipError, match="different device"):
        MultiGpuALS(2, comm="rccl", device_ids=[0, 0])


def test_context_argument_errors_without_a_device():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.rsparse_hip_ctx_create(0, None, 0, ctypes.byref(h)) == _lib.ERR_INVALID
    assert lib.rsparse_hip_ctx_create(2, None, 7, ctypes.byref(h)) == _lib.ERR_INVALID
    assert b"comm_kind" in lib.rsparse_hip_last_error()
    assert lib.rsparse_hip_ctx_destroy(None) == _lib.OK
    loss = ctypes.c_double(0)
    assert lib.rsparse_hip_ctx_half_iteration(None, 0, 1, 0.1, 1, 3, 1, ctypes.byref(loss)) == _lib.ERR_INVALID
    assert lib.rsparse_hip_ctx_set_factors(None, 8, None, None) == _lib.ERR_INVALID
