"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on identical inputs.

Tolerance (north star): 1e-4 relative Frobenius on factor matrices vs the reference-shaped CPU
arithmetic; device arithmetic is fp32, so the per-half-iteration checks compare against the fp64
oracle run on the same fp32 inputs.  Losses: 1e-4 relative.
Nothing here reads /root/reference; inputs are seeded synthetic data, the movielens fixture and the
committed goldens.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import GOLDEN, rel_fro
from oracle import wrmf_oracle as O
from rsparse_amd import _lib, als, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _problem(n_user, n_item, k, seed, mean_deg=12, d_max=400, feedback="implicit", scale=0.1):
    d = synth.make_dataset(n_user, n_item, seed=seed, mean_deg=mean_deg, d_max=d_max, feedback=feedback, device="cpu")
    p, i, x = d["c_iu"]          # columns = users
    p, i, x = p.numpy(), i.numpy(), x.numpy().astype(np.float64)
    rng = np.random.default_rng(seed)
    X = np.asfortranarray((rng.standard_normal((k, n_item)) * scale).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, n_user)) * scale).astype(np.float32))
    return (n_item, n_user, p, i, x), X, Y0


def _oracle64(csc, X, Y0, lam, solver, cg_steps, implicit, dynamic_lambda=True, cnt=None):
    n_rows, n_cols, p, i, x = csc
    X64 = np.asfortranarray(X, dtype=np.float64)
    Y64 = np.asfortranarray(Y0, dtype=np.float64).copy(order="F")
    if implicit:
        G = O.gramian(X64, lam)
        loss = O.als_implicit(p, i, x, X64, Y64, G, lam, solver, cg_steps)
    else:
        loss = O.als_explicit(p, i, x, X64, Y64, cnt, lam, solver, cg_steps, dynamic_lambda)
    return Y64, loss


@pytest.mark.parametrize("k,n", [(8, 1), (10, 7), (16, 1000), (64, 5000), (128, 20011), (100, 333)])
def test_gramian(k, n):
    rng = np.random.default_rng(k * 1000 + n)
    X = np.asfortranarray(rng.standard_normal((k, n)).astype(np.float32))
    X[0, :] += 3.0   # asymmetric, non-zero-mean: catches transposed / mirrored tiles
    for lam in (0.0, 0.1):
        G = als.gramian(X, lam, "float")
        ref = X.astype(np.float64) @ X.astype(np.float64).T + float(np.float32(lam)) * np.eye(k)
        assert rel_fro(G, ref) < 2e-6
        assert np.array_equal(G, G.T)


@pytest.mark.parametrize("k", [8, 10, 16, 32, 64, 100, 128])
@pytest.mark.parametrize("cg_steps", [0, 1, 3])
def test_implicit_cg_half_iteration(k, cg_steps):
    csc, X, Y0 = _problem(1500, 400, k, seed=k + cg_steps)      # rows of 1..400 nnz: short, resident-long, streamed-long
    lam = 0.1
    Yref, lref = _oracle64(csc, X, Y0, lam, 1, cg_steps, True)
    Y = Y0.copy(order="F")
    loss = als.als_implicit(csc, X, Y, lam, 1, 1, cg_steps, "float", False, False)
    assert rel_fro(Y, Yref) < TOL
    assert abs(loss - lref) <= TOL * abs(lref)
    if cg_steps == 0:
        assert np.array_equal(Y, Y0)        # warm start returned untouched


@pytest.mark.parametrize("k", [6, 16, 64, 128])
@pytest.mark.parametrize("dynamic_lambda", [True, False])
def test_explicit_cg_half_iteration(k, dynamic_lambda):
    csc, X, Y0 = _problem(1200, 300, k, seed=50 + k, feedback="explicit", scale=0.3)
    lam = 0.1
    cnt = np.diff(sp.csc_matrix((csc[4], csc[3], csc[2]), shape=(csc[0], csc[1])).tocsr().indptr).astype(np.float64)
    Yref, lref = _oracle64(csc, X, Y0, lam, 1, 3, False, dynamic_lambda, cnt)
    Y = Y0.copy(order="F")
    loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), lam, 1, 1, 3, dynamic_lambda, "float", False, False)
    assert rel_fro(Y, Yref) < TOL
    assert abs(loss - lref) <= TOL * abs(lref)


@pytest.mark.parametrize("k", [5, 16, 40, 64, 128])
@pytest.mark.parametrize("implicit", [True, False])
def test_cholesky_half_iteration(k, implicit):
    csc, X, Y0 = _problem(700, 500, k, seed=90 + k, feedback="implicit" if implicit else "explicit", scale=0.3)
    lam = 0.1
    cnt = np.diff(sp.csc_matrix((csc[4], csc[3], csc[2]), shape=(csc[0], csc[1])).tocsr().indptr).astype(np.float64)
    Yref, lref = _oracle64(csc, X, Y0, lam, 0, 3, implicit, True, cnt)
    Y = Y0.copy(order="F")
    if implicit:
        loss = als.als_implicit(csc, X, Y, lam, 1, 0, 3, "float", False, False)
    else:
        loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), lam, 1, 0, 3, True, "float", False, False)
    assert rel_fro(Y, Yref) < TOL
    assert abs(loss - lref) <= TOL * abs(lref)


def _rows_of_lengths(lengths, n_item, k, seed, scale=0.3):
    """a CSC (columns = the rows to solve) whose column j has lengths[j] distinct random items, confidences >= 1"""
    rng = np.random.default_rng(seed)
    p = np.zeros(len(lengths) + 1, dtype=np.int32)
    p[1:] = np.cumsum(lengths)
    idx = np.concatenate([np.sort(rng.choice(n_item, size=int(n), replace=False)) for n in lengths] or [np.zeros(0)]).astype(np.int32)
    x = (1.0 + rng.gamma(1.0, 2.0, size=idx.size)).astype(np.float32).astype(np.float64)
    X = np.asfortranarray((rng.standard_normal((k, n_item)) * scale).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, len(lengths))) * scale).astype(np.float32))
    return (n_item, len(lengths), p, idx, x), X, Y0


@pytest.mark.parametrize("k", [128, 100])
@pytest.mark.parametrize("classes", ["short5", "mid3", "short+long", "all", "one"])
def test_cholesky_short_rows_share_a_pass(k, classes):
    """wrmf_chol_lr.hip packs four rows of <= 16 non-zeros / two of <= 32 into one pass: class sizes that are not multiples
    of the packing, empty classes, the class boundaries (16 / 17, 32 / 33, 64 / 65) and empty rows in between"""
    rng = np.random.default_rng(7)
    lens = {
        "short5": [16, 1, 9, 16, 3],
        "mid3": [17, 32, 25],
        "short+long": [16, 15, 2, 1, 1, 1, 40, 64, 33, 65, 130, 0],
        "all": list(rng.integers(0, 70, size=301)) + [16, 17, 32, 33, 64, 65],
        "one": [7],
    }[classes]
    csc, X, Y0 = _rows_of_lengths(np.asarray(lens, dtype=np.int64), 400, k, seed=11 + k)
    lam = 0.1
    Yref, lref = _oracle64(csc, X, Y0, lam, 0, 3, True)
    Y = Y0.copy(order="F")
    loss = als.als_implicit(csc, X, Y, lam, 1, 0, 3, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    assert err.max() < TOL, (int(np.argmax(err)), lens[int(np.argmax(err))])
    assert abs(loss - lref) <= TOL * abs(lref)


@pytest.mark.parametrize("k", [64, 128, 60])
@pytest.mark.parametrize("dynamic_lambda", [True, False])
def test_cholesky_short_rows_explicit_push_through(k, dynamic_lambda):
    """Explicit feedback, exact solver, rows of <= 64 ratings at rank 64 / 128 (wrmf_chol_lr.hip, als_chol_lrx_kernel):
    y = X_nnz (lambda I + X_nnz^T X_nnz)^-1 r on one wave per pass, rows of <= 16 / <= 32 ratings sharing a pass -- every class
    (49..64, 33..48, 17..32, <= 16), class sizes that are not multiples of the packing, empty rows and longer rows (the k x k /
    wave-per-row kernels' share) in between, lambda_use = lambda * n and plain lambda; rank 60 keeps the k x k path."""
    rng = np.random.default_rng(17)
    lens = np.asarray(list(rng.integers(0, 70, size=260)) + [64, 49, 48, 33, 32, 17, 16, 1, 0, 65, 130, 64, 50, 40, 20, 10, 3],
                      dtype=np.int64)
    (n_item, n_rows, p_, idx, x), X, Y0 = _rows_of_lengths(lens, 400, k, seed=23 + k)
    x = np.round(1.0 + 4.0 * rng.random(x.size))                 # ratings 1..5
    csc = (n_item, n_rows, p_, idx, x)
    cnt = np.bincount(idx, minlength=n_item).astype(np.float64)
    Yref, lref = _oracle64(csc, X, Y0, 0.1, 0, 3, False, dynamic_lambda, cnt)
    Y = Y0.copy(order="F")
    loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), 0.1, 1, 0, 3, dynamic_lambda, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    assert err.max() < TOL, (int(np.argmax(err)), int(lens[int(np.argmax(err))]), float(err.max()))
    assert abs(loss - lref) <= TOL * abs(lref), (loss, lref)


@pytest.mark.parametrize("k", [128, 112])
@pytest.mark.parametrize("conf", ["half_unit", "all_unit", "barely_above_one", "one_normal"])
def test_cholesky_short_rows_with_confidence_exactly_one(k, conf):
    """wrmf_chol_lr.hip, wave-per-pass kernel (rank 128; the workgroup kernel at 112): a slot of confidence exactly 1 has no row in
    W = D^1/2 V' -- its lane rides through the elimination as a right-hand-side row and the forward pass leaves its loss term.
    Rows of every class (49..64, 33..48, 17..32, <= 16 non-zeros) with half of the confidences at 1 (what the bench matrix has),
    all of them at 1 (S = I), confidences 1 + 1e-6 (x_j . y = z_j / sqrt(c_j - 1) with a tiny divisor) and a single one above 1."""
    rng = np.random.default_rng(3)
    lens = np.asarray(list(rng.integers(1, 65, size=150)) + [64, 49, 48, 33, 32, 17, 16, 1, 64, 50, 40, 20, 10], dtype=np.int64)
    (n_item, n_rows, p_, idx, x), X, Y0 = _rows_of_lengths(lens, 400, k, seed=5 + k)
    x = x.copy()
    if conf == "half_unit":
        x[rng.random(x.size) < 0.5] = 1.0
    elif conf == "all_unit":
        x[:] = 1.0
    elif conf == "barely_above_one":
        m = rng.random(x.size) < 0.5
        x[m] = np.float32(1.0 + 1e-6)
        x[~m & (rng.random(x.size) < 0.5)] = 1.0
    else:
        x[:] = 1.0
        x[p_[:-1][lens > 0]] = 3.0          # the first non-zero of every row
    csc = (n_item, n_rows, p_, idx, x)
    Yref, lref = _oracle64(csc, X, Y0, 0.1, 0, 3, True)
    Y = Y0.copy(order="F")
    loss = als.als_implicit(csc, X, Y, 0.1, 1, 0, 3, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    assert err.max() < TOL, (conf, int(np.argmax(err)), int(lens[int(np.argmax(err))]), float(err.max()))
    assert abs(loss - lref) <= TOL * abs(lref), (loss, lref)


@pytest.mark.parametrize("k", [128, 96, 64, 20])
@pytest.mark.parametrize("implicit", [True, False])
def test_cg_long_rows_all_buckets(k, implicit):
    """Rows from 1 to ~2500 non-zeros: exercises every launch bucket of the quad-layout CG kernels
    (one wave per row, teams of 2/4/8 waves, and the streamed path beyond 512 non-zeros), with
    rank == KP (128, 64) and rank < KP (96, 20) padding."""
    fb = "implicit" if implicit else "explicit"
    d = synth.make_dataset(260, 3000, seed=11 + k, mean_deg=400, d_max=2500, feedback=fb, device="cpu")
    p, i, x = (t.numpy() for t in d["c_iu"])
    x = x.astype(np.float64)
    lens = np.diff(p)
    assert lens.max() > 1024 and (lens <= 32).any() and ((lens > 128) & (lens <= 256)).any()
    rng = np.random.default_rng(k)
    X = np.asfortranarray((rng.standard_normal((k, 3000)) * 0.05).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, 260)) * 0.05).astype(np.float32))
    csc = (3000, 260, p, i, x)
    cnt = np.bincount(i, minlength=3000).astype(np.float64)
    Yref, lref = _oracle64(csc, X, Y0, 0.1, 1, 3, implicit, True, cnt)
    Y = Y0.copy(order="F")
    if implicit:
        loss = als.als_implicit(csc, X, Y, 0.1, 1, 1, 3, "float", False, False)
    else:
        loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), 0.1, 1, 1, 3, True, "float", False, False)
    assert rel_fro(Y, Yref) < TOL
    assert abs(loss - lref) <= TOL * abs(lref)
    # per-row check so that one bad bucket cannot hide in the Frobenius norm
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    assert err.max() < 5e-4, (int(err.argmax()), int(lens[err.argmax()]), float(err.max()))


def test_double_entry_points_and_empty_columns(ml_train):
    """movielens train (test-wrmf.R:6) has items nobody rated -> zero columns (wrmf_implicit.hpp:281)."""
    n_user, n_item, p, i, x = ml_train
    k, lam = 16, 0.1
    rng = np.random.default_rng(3)
    U = np.asfortranarray(rng.standard_normal((k, n_user)) * 0.1)
    Y0 = np.asfortranarray(rng.standard_normal((k, n_item)) * 0.1)
    for solver in (1, 0):
        Yref = Y0.copy(order="F")
        lref = O.als_implicit(p, i, x, U, Yref, O.gramian(U, lam), lam, solver, 3)
        Y = Y0.copy(order="F")
        loss = als.als_implicit((n_user, n_item, p, i, x), U, Y, lam, 1, solver, 3, "double", False, False)
        assert Y.dtype == np.float64 and rel_fro(Y, Yref) < TOL
        assert abs(loss - lref) <= TOL * abs(lref)
        empty = np.diff(p) == 0
        assert empty.any() and np.all(Y[:, empty] == 0)
    # degenerate shapes: no columns at all; all-empty columns
    z = als.als_implicit((n_user, 0, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0)), U,
                         np.zeros((k, 0), order="F"), lam, 1, 1, 3, "double", False, False, XtX=O.gramian(U, lam))
    Y = np.asfortranarray(np.ones((k, 3)))
    als.als_implicit((n_user, 3, np.zeros(4, np.int32), np.zeros(0, np.int32), np.zeros(0)), U, Y, lam, 1, 1, 3,
                     "double", False, False)
    assert np.all(Y == 0)


def _fixed_init(m, U0):
    m._init_user_factors = np.ascontiguousarray(U0.T, dtype=np.float32)


@pytest.mark.parametrize("feedback", ["implicit", "explicit"])
@pytest.mark.parametrize("solver", ["conjugate_gradient", "cholesky"])
def test_fit_transform_matches_goldens(ml_train, feedback, solver):
    """BASELINE config 1 protocol (test-wrmf.R:6,48-57): 5 iterations on movielens train from the
    committed initial factors; factors and per-iteration losses vs the fp64 oracle goldens."""
    from rsparse_amd import WRMF
    g = np.load(GOLDEN / "wrmf_movielens_goldens.npz")
    n_user, n_item, p, i, x = ml_train
    tag = "%s_%s" % (feedback, solver)
    k, lam = int(g[tag + "_rank"]), float(g[tag + "_lambda"])
    init = g[tag + "_init_components"] if solver == "cholesky" else None
    m = WRMF(rank=k, lambda_=lam, feedback=feedback, solver=solver, precision="float", init=init)
    _fixed_init(m, g["init_U_k%d" % k])
    train = sp.csc_matrix((x, i, p), shape=(n_user, n_item))
    emb = m.fit_transform(train, n_iter=5, convergence_tol=-1)
    assert emb.shape == (n_user, k) and emb.dtype == np.float32
    assert m.components.shape == (k, n_item)
    assert np.allclose([l[0] for l in m.losses], g[tag + "_loss_items"], rtol=TOL)
    assert np.allclose([l[1] for l in m.losses], g[tag + "_loss_users"], rtol=TOL)
    assert rel_fro(m.components, g[tag + "_components"]) < TOL
    assert rel_fro(emb, g[tag + "_user_emb"]) < TOL
    # fit_transform(train) == transform(train)   (test-wrmf.R:57)
    again = m.transform(train)
    assert np.array_equal(emb, again)


def test_rows_are_independent_of_their_neighbours():
    """Size-independent property: a row's solution depends only on that row's data, so solving a
    shuffled subset of the rows reproduces the same vectors bit for bit (short and long rows alike)."""
    csc, X, Y0 = _problem(3000, 600, 64, seed=7, mean_deg=30, d_max=500)
    n_rows, n_cols, p, i, x = csc
    Y = Y0.copy(order="F")
    als.als_implicit(csc, X, Y, 0.1, 1, 1, 3, "float", False, False)
    rng = np.random.default_rng(0)
    pick = rng.permutation(n_cols)[:500]
    lens = np.diff(p)[pick]
    p2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = np.concatenate([np.arange(p[c], p[c + 1]) for c in pick])
    Ysub = np.asfortranarray(Y0[:, pick])
    als.als_implicit((n_rows, len(pick), p2, i[idx], x[idx]), X, Ysub, 0.1, 1, 1, 3, "float", False, False,
                     XtX=als.gramian(X, 0.1, "float"))
    assert np.array_equal(Ysub, Y[:, pick])


def test_config2_scale_properties():
    """BASELINE config 2 shape (1M x 100k, ~50M nnz, k=64) on the device-resident path: Cholesky
    rows satisfy their normal equations, CG loss decreases, replicas of a half-iteration are
    deterministic."""
    from rsparse_amd.engine import HipBackend, ShardedALS
    be = HipBackend()
    n_user, n_item, k, lam = 1_000_000, 100_000, 64, 0.1
    d = synth.make_dataset(n_user, n_item, device=be.device)
    als_ = ShardedALS(be, n_user, n_item, k, d["c_ui"], d["c_iu"], d["nnz"], lambda_=lam)
    g = torch.Generator(device="cpu").manual_seed(1)
    U = (torch.randn(n_user, k, generator=g) * 0.01).to(be.device)
    V = torch.zeros(n_item, k, device=be.device)
    losses = []
    for _ in range(3):
        als_.half_iteration("items", U, V, 1)
        losses.append(als_.half_iteration("users", U, V, 1))
    assert all(np.isfinite(losses)) and losses[2] < losses[1] < losses[0]
    U2 = U.clone()
    als_.half_iteration("users", U2, V, 1, want_loss=False)
    U3 = U.clone()
    als_.half_iteration("users", U3, V, 1, want_loss=False)
    assert torch.equal(U2, U3)                                    # run-to-run deterministic
    # exact solve: residual of the normal equations on sampled users
    G = als_.gramian(V, als_.lay_item).clone()
    Uc = torch.zeros_like(U)
    als_.half_iteration("users", Uc, V, 0, G=G, want_loss=False)
    be.check_numeric()
    p, i, x = [t.cpu() for t in d["c_iu"]]
    Vc, Gc, Ucc = V.cpu().double(), G.cpu().double(), Uc.cpu().double()
    worst = 0.0
    for u in torch.randint(0, n_user, (200,), generator=g).tolist():
        idx = i[p[u]:p[u + 1]].long()
        c = x[p[u]:p[u + 1]].double()
        Xn = Vc[idx]                                             # n_i x k
        A = Gc + (Xn * (c - 1)[:, None]).T @ Xn
        b = Xn.T @ c
        worst = max(worst, float(torch.linalg.norm(A @ Ucc[u] - b) / torch.linalg.norm(b)))
    assert worst < 1e-4


@pytest.mark.parametrize("k", [128, 64, 40])
@pytest.mark.parametrize("implicit", [True, False])
def test_cholesky_long_rows(k, implicit):
    """Rows up to ~6000 non-zeros through the exact solver: rows beyond 4096 non-zeros take the second launch of
    wrmf_chol.hip (two-level accumulation of the rank-one updates), the rest the main one; per-row bound."""
    fb = "implicit" if implicit else "explicit"
    n_fix, n_solve = 9000, 120
    d = synth.make_dataset(n_solve, n_fix, seed=31 + k, mean_deg=900, d_max=6000, feedback=fb, device="cpu")
    p, i, x = (t.numpy() for t in d["c_iu"])
    x = x.astype(np.float64)
    lens = np.diff(p)
    assert lens.max() > 4096 and (lens <= 4096).sum() > 10
    rng = np.random.default_rng(k)
    X = np.asfortranarray((rng.standard_normal((k, n_fix)) * 0.05).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, n_solve)) * 0.05).astype(np.float32))
    csc = (n_fix, n_solve, p, i, x)
    cnt = np.bincount(i, minlength=n_fix).astype(np.float64)
    Yref, lref = _oracle64(csc, X, Y0, 0.1, 0, 3, implicit, True, cnt)
    Y = Y0.copy(order="F")
    if implicit:
        loss = als.als_implicit(csc, X, Y, 0.1, 1, 0, 3, "float", False, False)
    else:
        loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), 0.1, 1, 0, 3, True, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    assert err.max() < TOL, (int(err.argmax()), float(err.max()), int(lens[err.argmax()]))
    assert abs(loss - lref) <= TOL * abs(lref)


@pytest.mark.parametrize("k", [128, 100, 64])
@pytest.mark.parametrize("implicit", [True, False])
def test_cholesky_rows_of_every_length_class(k, implicit):
    """solver == CHOLESKY dispatches by row length (and rank): <= 64 non-zeros in low-rank form (implicit, rank 98..128),
    up to 64 (rank > 64) or 512 (rank <= 64) on wrmf_chol.hip's kernel, beyond that assembled by the normal-equation
    kernel and solved by the blocked LDL^T in its LDS tiles.  Lengths 1..700 with every class well populated, and most
    of the long rows NOT split (enough rows per workgroup list); per-row bound."""
    fb = "implicit" if implicit else "explicit"
    n_fix, n_solve = 4000, 1500
    d = synth.make_dataset(n_solve, n_fix, seed=77 + k, mean_deg=150, d_max=700, feedback=fb, device="cpu")
    p, i, x = (t.numpy() for t in d["c_iu"])
    x = x.astype(np.float64)
    lens = np.diff(p)
    for lo, hi in ((0, 64), (64, 128), (128, 256), (256, 512), (512, 700)):
        assert ((lens > lo) & (lens <= hi)).sum() >= 20, (lo, hi)
    rng = np.random.default_rng(k)
    X = np.asfortranarray((rng.standard_normal((k, n_fix)) * 0.1).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, n_solve)) * 0.1).astype(np.float32))
    csc = (n_fix, n_solve, p, i, x)
    cnt = np.bincount(i, minlength=n_fix).astype(np.float64)
    Yref, lref = _oracle64(csc, X, Y0, 0.1, 0, 3, implicit, True, cnt)
    Y = Y0.copy(order="F")
    if implicit:
        loss = als.als_implicit(csc, X, Y, 0.1, 1, 0, 3, "float", False, False)
    else:
        loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), 0.1, 1, 0, 3, True, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    assert err.max() < TOL, (int(err.argmax()), float(err.max()), int(lens[err.argmax()]))
    assert abs(loss - lref) <= TOL * abs(lref)


def test_cholesky_with_confidence_below_one_takes_the_general_kernels():
    """The low-rank form of the short rows needs sqrt(c - 1) and the fp16 normal-equation products need c >= 1: a matrix
    with some confidence below 1 is detected on the device; the low-rank kernel stands down and wrmf_chol.hip's kernel
    takes the short rows too (its second range of the length-sorted order), the long rows go through the bf16 products."""
    k = 128
    n_fix, n_solve = 4000, 1500
    d = synth.make_dataset(n_solve, n_fix, seed=7, mean_deg=150, d_max=700, feedback="implicit", device="cpu")
    p, i, x = (t.numpy() for t in d["c_iu"])
    x = x.astype(np.float64)
    x[::9] = 0.5                                   # lhs = XtX - 0.5 x x^T for those: still positive definite here
    lens = np.diff(p)
    assert (lens == 0).sum() == 0 and ((lens >= 1) & (lens <= 64)).sum() > 100 and (lens > 512).sum() > 20
    rng = np.random.default_rng(5)
    X = np.asfortranarray((rng.standard_normal((k, n_fix)) * 0.1).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, n_solve)) * 0.1).astype(np.float32))
    csc = (n_fix, n_solve, p, i, x)
    Yref, lref = _oracle64(csc, X, Y0, 0.1, 0, 3, True)
    Y = Y0.copy(order="F")
    loss = als.als_implicit(csc, X, Y, 0.1, 1, 0, 3, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    assert err.max() < TOL, (int(err.argmax()), float(err.max()), int(lens[err.argmax()]))
    assert abs(loss - lref) <= TOL * abs(lref)


@pytest.mark.parametrize("k", [128, 64])
def test_implicit_cg_long_rows_with_confidence_below_one(k):
    """The fp16 normal-equation kernel takes sqrt(c - 1); a matrix with some confidence below 1 is detected on the device
    and its long rows go through the bf16 kernel instead (wrmf_ne.hip) -- same answer, no host round trip."""
    d = synth.make_dataset(200, 2500, seed=5 + k, mean_deg=420, d_max=2200, feedback="implicit", device="cpu")
    p, i, x = (t.numpy() for t in d["c_iu"])
    x = x.astype(np.float64)
    x[::7] = 0.25                      # confidences below 1 in many rows, long ones included
    assert np.diff(p).max() > 1024
    rng = np.random.default_rng(k)
    X = np.asfortranarray((rng.standard_normal((k, 2500)) * 0.05).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, 200)) * 0.05).astype(np.float32))
    csc = (2500, 200, p, i, x)
    Yref, lref = _oracle64(csc, X, Y0, 0.1, 1, 3, True)
    Y = Y0.copy(order="F")
    loss = als.als_implicit(csc, X, Y, 0.1, 1, 1, 3, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    assert err.max() < TOL, (int(err.argmax()), float(err.max()))
    assert abs(loss - lref) <= TOL * abs(lref)


@pytest.mark.parametrize("below_one", [False, True])
def test_cg_wave_per_row_kernel_every_row_length_class(below_one):
    """wrmf_cg_mf.hip (rank 128, rows of 513..16384 non-zeros) streams a row in steps of 16 non-zeros through a ring of three
    LDS slots, its indices in chunks of 64, ping-pongs two operand sets by step parity and zeroes the last step's tail: a row of
    every length 513..658 (all residues mod 16 and mod 64, odd and even step counts, 9..11 chunks), the powers of two around
    1024 / 4096, and the two lengths either side of the hand-over to the split kernel (16384 | 16385).  Per row against the
    fp64 oracle; `below_one`: the instantiation for matrices with a confidence < 1 (no square root, both operand sets)."""
    k, n_items = 128, 20000
    lens = list(range(513, 659)) + [1023, 1024, 1025, 4095, 4096, 4097, 16383, 16384, 16385, 300, 40, 0]
    rng = np.random.default_rng(2026 + below_one)
    p = np.zeros(len(lens) + 1, dtype=np.int32)
    p[1:] = np.cumsum(lens)
    idx = np.concatenate([np.sort(rng.choice(n_items, size=n, replace=False)).astype(np.int32) for n in lens])
    x = 1.0 + rng.geometric(0.5, size=idx.size).astype(np.float64)
    if below_one:
        x[::9] = 0.3
    X = np.asfortranarray((rng.standard_normal((k, n_items)) * 0.05).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, len(lens))) * 0.05).astype(np.float32))
    csc = (n_items, len(lens), p, idx, x)
    Yref, lref = _oracle64(csc, X, Y0, 0.1, 1, 3, True)
    Y = Y0.copy(order="F")
    loss = als.als_implicit(csc, X, Y, 0.1, 1, 1, 3, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    err[np.asarray(lens) == 0] = 0.0
    assert np.all(Y[:, -1] == 0.0)                                   # the empty column (wrmf_implicit.hpp:281)
    bad = int(err.argmax())
    assert err.max() < TOL, (bad, lens[bad], float(err.max()))
    assert abs(loss - lref) <= TOL * abs(lref)


def test_factor_scale_does_not_matter_to_the_fp16_path():
    """The fp16 normal-equation kernel rescales its operands by powers of two taken from max |x| and max c: factors of
    the order 1e-4 or 30 (far outside fp16's comfortable range unscaled) give the same relative accuracy."""
    d = synth.make_dataset(120, 2000, seed=77, mean_deg=600, d_max=1800, feedback="implicit", device="cpu")
    p, i, x = (t.numpy() for t in d["c_iu"])
    x = x.astype(np.float64)
    x[::5] *= 400.0                    # confidences up to several thousand
    csc = (2000, 120, p, i, x)
    lens = np.diff(p)
    for scale in (1e-4, 30.0):
        rng = np.random.default_rng(3)
        X = np.asfortranarray((rng.standard_normal((128, 2000)) * scale).astype(np.float32))
        X[:, 17] *= 50.0               # one outlier vector sets the scale for everybody
        Y0 = np.asfortranarray((rng.standard_normal((128, 120)) * scale).astype(np.float32))
        Yref, lref = _oracle64(csc, X, Y0, 0.1, 1, 3, True)
        # yardstick: the same half-iteration by the oracle in float (three CG steps on systems this badly scaled lose
        # digits in ANY fp32 arithmetic; the claim is that the fp16 operands add nothing to that)
        Y32 = Y0.copy(order="F")
        O.als_implicit(p, i, x, X, Y32, O.gramian(X, 0.1), 0.1, 1, 3)
        Y = Y0.copy(order="F")
        loss = als.als_implicit(csc, X, Y, 0.1, 1, 1, 3, "float", False, False)
        den = np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
        err = np.linalg.norm(Y - Yref, axis=0) / den
        err32 = np.linalg.norm(Y32 - Yref, axis=0) / den
        bound = np.maximum(TOL, 3.0 * err32)
        worst = int(np.argmax(err / bound))
        assert np.all(err <= bound), (scale, worst, int(lens[worst]), float(err[worst]), float(err32[worst]))
        assert (lens > 512).sum() >= 20
        assert abs(loss - lref) <= max(TOL, 3.0 * float(np.max(err32))) * abs(lref)


@pytest.mark.parametrize("k", [128, 100])
def test_short_rows_share_the_dense_product_on_the_matrix_cores(k):
    """Rows of <= 32 non-zeros at rank 65..128 (implicit CG): the four rows a workgroup solves side by side share G v as
    fp16-term MFMAs with per-vector power-of-two scales (wrmf_cgq.hip, DMF).  Rows of every length 0..32 next to each
    other -- empty rows keep in lockstep with their neighbours --, factor scales far outside fp16's range unscaled, and
    a warm start that is exact for some rows (CG stops at its first test there while the neighbours go on)."""
    rng = np.random.default_rng(11)
    n_rows, n_cols = 900, 1203
    lens = np.tile(np.arange(0, 33), n_cols // 33 + 1)[:n_cols]
    rng.shuffle(lens)
    lens[5:9] = 0                                   # a whole workgroup of empty rows
    p = np.zeros(n_cols + 1, np.int64)
    np.cumsum(lens, out=p[1:])
    i = np.concatenate([np.sort(rng.choice(n_rows, size=int(n), replace=False)) for n in lens]).astype(np.int32)
    x = (1.0 + rng.geometric(0.5, size=int(p[-1]))).astype(np.float64)
    csc = (n_rows, n_cols, p.astype(np.int32), i, x)
    for scale in (1e-3, 1.0, 40.0):
        X = np.asfortranarray((rng.standard_normal((k, n_rows)) * scale).astype(np.float32))
        X[:, 3] *= 4.0      # (a 30x outlier makes three fp32 CG steps meaningless for ANY arithmetic: tools/probes/dmf_accuracy_probe.py)
        Y0 = np.asfortranarray((rng.standard_normal((k, n_cols)) * scale).astype(np.float32))
        Yref, lref = _oracle64(csc, X, Y0, 0.1, 1, 3, True)
        Y0[:, ::7] = Yref[:, ::7].astype(np.float32)   # (nearly) solved already
        Yref, lref = _oracle64(csc, X, Y0, 0.1, 1, 3, True)
        Y32 = Y0.copy(order="F")
        O.als_implicit(p.astype(np.int32), i, x, X, Y32, O.gramian(X, 0.1), 0.1, 1, 3)
        Y = Y0.copy(order="F")
        loss = als.als_implicit(csc, X, Y, 0.1, 1, 1, 3, "float", False, False)
        assert np.all(Y[:, lens == 0] == 0.0)
        den = np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
        err = np.linalg.norm(Y - Yref, axis=0) / den
        err32 = np.linalg.norm(Y32 - Yref, axis=0) / den
        bound = np.maximum(TOL, 3.0 * err32)         # yardstick: the fp32 oracle on the same badly scaled systems
        worst = int(np.argmax(err / bound))
        assert np.all(err <= bound), (k, scale, worst, int(lens[worst]), float(err[worst]), float(err32[worst]))
        assert abs(loss - lref) <= max(TOL, 3.0 * float(np.max(err32))) * abs(lref)


@pytest.mark.parametrize("k", [16, 64, 128])
def test_cholesky_falls_back_to_the_general_solver(k):
    """solve(lhs, rhs, fast + likely_sympd) (wrmf_implicit.hpp:236): when the Cholesky factorisation fails the reference
    goes on to a general LU solve behind a warning.  Confidences below 1 against a Gramian that does not dominate them give
    indefinite but regular systems: the device re-solves exactly those rows by Gaussian elimination with partial pivoting
    (wrmf_lu.hip), reports how many, and matches the oracle's gesv branch (oracle/wrmf_oracle.cpp solve_sympd)."""
    import warnings
    from rsparse_amd import _lib
    from rsparse_amd.engine import HipBackend
    import torch
    csc, X, Y0 = _problem(400, 120, k, seed=4, feedback="implicit", scale=0.3)
    n_rows, n_cols, p, i, x = csc
    rng = np.random.default_rng(1)
    x = np.where(rng.random(x.size) < 0.5, 0.25, 3.0)         # c - 1 = -0.75 on half of the entries
    G = np.asfortranarray((0.05 * (X.astype(np.float64) @ X.astype(np.float64).T) + 0.1 * np.eye(k)))
    Yref = np.asfortranarray(Y0, dtype=np.float64).copy(order="F")
    lref = O.als_implicit(p, i, x, np.asfortranarray(X, dtype=np.float64), Yref, G, 0.1, 0, 3)
    # which systems are not positive definite, and how well conditioned each one is (float64 eigenvalues)
    X64 = X.astype(np.float64)
    lo_bad = hi_bad = 0
    cond = np.zeros(n_cols)
    for c in range(n_cols):
        idx, val = i[p[c]:p[c + 1]], x[p[c]:p[c + 1]]
        ev = np.linalg.eigvalsh(G + (X64[:, idx] * (val - 1.0)) @ X64[:, idx].T)
        cond[c] = np.abs(ev).max() / np.abs(ev).min()
        lo_bad += ev.min() < -1e-4           # certainly indefinite in fp32 too
        hi_bad += ev.min() < 1e-4            # ... possibly
    assert lo_bad >= 5
    Y = Y0.copy(order="F")
    loss = als.als_implicit((n_rows, n_cols, p, i, x), X, Y, 0.1, 1, 0, 3, "float", False, False,
                            XtX=np.asfortranarray(G, dtype=np.float32))          # no error: the rows were re-solved
    assert np.all(np.isfinite(Y))
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    bound = np.maximum(1e-4, 20.0 * cond * 6e-8)                                  # fp32 elimination: ~ cond x eps
    worst = int(np.argmax(err / bound))
    assert np.all(err <= bound), (worst, float(err[worst]), float(cond[worst]))
    ok = cond < 1e3
    assert abs(loss - lref) <= 1e-3 * abs(lref) or not ok.all()
    # the device-resident layer reports the count like the reference's warning
    be = HipBackend()
    dev = be.device
    h = be.make_csc(n_rows, n_cols, be.to_device(p, torch.int32), be.to_device(i, torch.int32), be.to_device(x.astype(np.float32), torch.float32))
    Xd, Yd = be.to_device(np.ascontiguousarray(X.T), torch.float32), be.to_device(np.ascontiguousarray(Y0.T), torch.float32)
    Gd = be.to_device(np.ascontiguousarray(G.astype(np.float32)), torch.float32)
    lossd = torch.zeros(1, dtype=torch.float64, device=dev)
    be.half_iteration(h, True, Xd, Yd, Gd, 0.1, 0, 3, True, lossd)
    with pytest.warns(RuntimeWarning, match="general"):
        be.check_numeric()
    assert lo_bad <= be.last_fallback_rows <= hi_bad
    assert np.array_equal(Yd.cpu().numpy().T, Y)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        be.check_numeric()                                    # the counters were taken


def test_singular_systems_are_an_error_or_a_consistent_solution():
    """Explicit feedback, lambda = 0, fewer ratings than factors: lhs = X_nnz X_nnz^T is singular.  Cholesky fails; the
    general solver either meets an exactly zero pivot column -- RSPARSE_HIP_ERR_NUMERIC, the reference's R error -- or
    eliminates through rounding noise and returns one of the solutions of the (consistent) system, which then reproduces
    the ratings.  Either way the call ends, nothing is left non-finite, and the library stays usable."""
    from rsparse_amd import _lib
    csc, X, Y0 = _problem(300, 40, 16, seed=5, feedback="explicit", scale=0.3)
    n_rows, n_cols, p, i, x = csc
    Y = Y0.copy(order="F")
    try:
        als.als_explicit(csc, X, Y, None, 0.0, 1, 0, 3, False, "float", False, False)
        assert np.all(np.isfinite(Y))
        for c in range(n_cols):
            idx, val = i[p[c]:p[c + 1]], x[p[c]:p[c + 1]]
            if 0 < len(idx) < 12:
                assert np.linalg.norm(val - Y[:, c] @ X[:, idx]) <= 5e-2 * np.linalg.norm(val), c
    except _lib.RsparseHipError as e:
        assert e.code == _lib.ERR_NUMERIC and "singular" in str(e)
    Y2 = Y0.copy(order="F")
    als.als_explicit(csc, X, Y2, None, 0.1, 1, 0, 3, False, "float", False, False)
    assert np.all(np.isfinite(Y2))


@pytest.mark.parametrize("k,with_biases", [(10, False), (50, False), (7, False), (12, True), (64, True), (11, True)])
def test_explicit_cholesky_with_lambda_zero_at_ranks_that_get_padded(k, with_biases):
    """lambda = 0 is the reference's default.  Ranks that are not a multiple of 4 (and every biased fit, whose solves run at
    rank - 1) run on zero-padded copies -- whose padded coordinates have lambda_use on the diagonal, i.e. NOTHING here: such a
    fit has to keep the true rank (every row came back as singular).  Rows hold more ratings than factors, so the
    true systems are regular, and the oracle's exact solve is the answer."""
    rng = np.random.default_rng(400 + k)
    n_item = 900
    lens = rng.integers(k + 8, 5 * k + 40, 260)
    csc, X, Y0 = _rows_of_lengths(lens, n_item, k, seed=500 + k, scale=0.4)
    n_rows, n_cols, p, i, x = csc
    x = rng.integers(1, 6, x.size).astype(np.float64)
    csc = (n_rows, n_cols, p, i, x)
    cnt = np.bincount(i, minlength=n_item).astype(np.float64)
    if with_biases:                      # the layout of the driver: ones in the first row of X, the x biases in the last
        X[0, :] = 1.0
        Y0[-1, :] = 1.0
    X64, Y64 = np.asfortranarray(X, dtype=np.float64), np.asfortranarray(Y0, dtype=np.float64)
    lref = O.als_explicit(p, i, x, X64, Y64, cnt, 0.0, 0, 3, False, with_biases=with_biases, is_x_bias_last_row=True)
    Y = Y0.copy(order="F")
    loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), 0.0, 1, 0, 3, False, "float", with_biases, True)
    err = np.linalg.norm(Y - Y64, axis=0) / np.maximum(np.linalg.norm(Y64, axis=0), 1e-30)
    # fp32 arithmetic on an unregularised system: the yardstick is the oracle's own fp32 run (cond ~ 1e2..1e3 here)
    Y32 = Y0.copy(order="F")
    O.als_explicit(p, i, x, X, Y32, cnt.astype(np.float32), 0.0, 0, 3, False, with_biases=with_biases, is_x_bias_last_row=True)
    e32 = np.linalg.norm(Y32 - Y64, axis=0) / np.maximum(np.linalg.norm(Y64, axis=0), 1e-30)
    assert err.max() < max(TOL, 3 * e32.max()), (float(err.max()), float(e32.max()), int(lens[err.argmax()]))
    assert abs(loss - lref) <= 1e-3 * abs(lref) + 1e-7


def test_frozen_values_are_scanned_once_and_thawed_values_again():
    """rsparse_hip_csc_freeze_values: with the promise set the statistics of the values (max confidence, "some confidence < 1":
    the operand scales and the kernel choice of the fp16 normal-equation path) come from one scan per handle -- same bits as
    without it --; withdrawn, a change of the values is seen again (here: confidences below 1 appear, which sends the long rows
    to the bf16 kernel; a stale flag would take sqrt(c - 1) of negative numbers)."""
    from rsparse_amd.engine import HipBackend
    rng = np.random.default_rng(77)
    k, n_fix = 128, 5000
    lens = np.concatenate([rng.integers(600, 1500, 40), rng.integers(1, 200, 300)])
    csc, X, Y0 = _rows_of_lengths(lens, n_fix, k, seed=78, scale=0.1)
    n_rows, n_cols, p, i, x = csc
    be = HipBackend(0)
    dp, di = be.to_device(p, torch.int32), be.to_device(i, torch.int32)
    dx = be.to_device(x.astype(np.float32), torch.float32)
    h = be.make_csc(n_fix, n_cols, dp, di, dx)
    Xd = be.to_device(np.ascontiguousarray(X.T), torch.float32)
    G = torch.zeros((k, k), dtype=torch.float32, device=Xd.device)
    be.gramian(Xd, 0.1, G, None)
    loss = torch.zeros(1, dtype=torch.float64, device=Xd.device)

    def solve():
        Y = be.to_device(np.ascontiguousarray(Y0.T), torch.float32)
        be.half_iteration(h, True, Xd, Y, G, 0.1, 1, 3, False, loss)
        return Y.cpu().numpy(), float(loss)
    y_plain, l_plain = solve()
    h.freeze_values(True)
    y1, l1 = solve()
    y2, l2 = solve()
    assert np.array_equal(y_plain, y1) and np.array_equal(y1, y2) and l_plain == l1 == l2
    h.freeze_values(False)
    dx.mul_(0.25)                               # confidences below 1 now
    y3, l3 = solve()
    Yref, lref = _oracle64((n_fix, n_cols, p, i, x * 0.25), X, Y0, 0.1, 1, 3, True)
    err = np.linalg.norm(y3.T - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    assert err.max() < TOL, float(err.max())


@pytest.mark.parametrize("k,implicit", [(128, True), (64, True), (64, False)])
def test_one_giant_row_is_split_across_workgroups(k, implicit):
    """A row far longer than a workgroup's share of the long rows is cut into segments that different workgroups stream;
    their partial sums meet in the COLLECT launch (wrmf_ne.hip).  600 ordinary long rows + one of 60000 non-zeros, as
    on a rank of a multi-GPU run that owns the most popular item."""
    from rsparse_amd.engine import HipBackend
    rng = np.random.default_rng(7 + k)
    n_fix, n_long, giant = 70000, 600, 60000
    lens = np.concatenate([[giant], rng.integers(520, 900, n_long)])
    p = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    i = np.concatenate([np.sort(rng.choice(n_fix, int(n), replace=False)) for n in lens]).astype(np.int32)
    x = (1.0 + rng.random(i.size) * 4.0) if implicit else rng.integers(1, 6, i.size).astype(np.float64)
    X = np.asfortranarray((rng.standard_normal((k, n_fix)) * 0.05).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, lens.size)) * 0.05).astype(np.float32))
    csc = (n_fix, lens.size, p, i, x)
    be = HipBackend(0)
    h = be.make_csc(n_fix, lens.size, be.to_device(p, torch.int32), be.to_device(i, torch.int32),
                    be.to_device(x.astype(np.float32), torch.float32))
    assert h.info()["ne_segments"] >= 4            # the giant row (at least) was cut
    cnt = np.bincount(i, minlength=n_fix).astype(np.float64)
    Yref, lref = _oracle64(csc, X, Y0, 0.1, 1, 3, implicit, True, cnt)
    Y = Y0.copy(order="F")
    if implicit:
        loss = als.als_implicit(csc, X, Y, 0.1, 1, 1, 3, "float", False, False)
    else:
        loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), 0.1, 1, 1, 3, True, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    assert err.max() < TOL, (int(err.argmax()), float(err.max()), float(err[0]))
    assert abs(loss - lref) <= TOL * abs(lref)


@pytest.mark.parametrize("k", [10, 48, 64, 100, 128])
@pytest.mark.parametrize("scale", [1e-12, 1e-20, 1e-27])
def test_explicit_cholesky_with_factors_that_have_shrunk(k, scale):
    """lambda = 1000 of the reference's grid (test-wrmf.R) drives an explicit fit's factors to 1e-28 within five iterations.  The
    exact solver must follow: through round 6 the explicit low-rank kernel solved (T s^2 + lambda s^2 I) z' = r in the scale s of
    its fp16 operands, lambda s^2 overflowed below max |X| ~ 1e-14 and every row of <= 64 non-zeros came back as zeros at the
    native ranks 64 and 128 (tools/dbg/chol_lambda1000_fit.py; the reference grid's ranks, 4..12, took other kernels)."""
    d = synth.make_dataset(3000, 800, seed=5, mean_deg=40, d_max=600, feedback="explicit", device="cpu")
    p, i, x = (t.numpy() for t in d["c_iu"])
    x = x.astype(np.float64)
    n_fix, n_cols = 800, 3000
    cnt = np.bincount(i, minlength=n_fix).astype(np.float64)
    rng = np.random.default_rng(k)
    X = np.asfortranarray((rng.standard_normal((k, n_fix)) * scale).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, n_cols)) * scale).astype(np.float32))
    csc = (n_fix, n_cols, p, i, x)
    for dyn in (False, True):
        Yref, lref = _oracle64(csc, X, Y0, 1000.0, 0, 3, False, dyn, cnt)
        Y = Y0.copy(order="F")
        loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), 1000.0, 1, 0, 3, dyn, "float", False, False)
        err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-300)
        assert err.max() < TOL, (dyn, int(err.argmax()), float(err.max()))
        assert abs(loss - lref) <= TOL * abs(lref)


@pytest.mark.parametrize("k", [64, 96, 128])
@pytest.mark.parametrize("lens", [(2300,), (513, 700, 1100, 2300), (2300, 2300, 600), (6000, 513)])
def test_a_few_long_rows_one_of_them_cut(k, lens):
    """Fewer long rows than workgroup slots, one of them long enough (>= 2048 non-zeros) to be cut into segments by the rule of the
    fine lists.  The matrix's two list sets (two workgroups per CU at rank 97..128, one below) share one segment table: the cut
    must be the same in both -- at the end of round 6 it was not, the coarse lists held the row whole, and the collecting launch
    overwrote its solution at ranks up to 96 (tools/dbg/gb_split_dbg2.py)."""
    rng = np.random.default_rng(3 * k + len(lens))
    n_fix = 8000
    lens = np.concatenate([np.asarray(lens), rng.integers(1, 90, 200)])
    p = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    i = np.concatenate([np.sort(rng.choice(n_fix, int(n), replace=False)) for n in lens]).astype(np.int32)
    x = 1.0 + rng.geometric(0.5, size=i.size).astype(np.float64)
    X = np.asfortranarray((rng.standard_normal((k, n_fix)) * 0.1).astype(np.float32))
    Y0 = np.asfortranarray((rng.standard_normal((k, lens.size)) * 0.1).astype(np.float32))
    csc = (n_fix, lens.size, p, i, x)
    Yref, lref = _oracle64(csc, X, Y0, 0.1, 1, 3, True)
    Y = Y0.copy(order="F")
    loss = als.als_implicit(csc, X, Y, 0.1, 1, 1, 3, "float", False, False)
    err = np.linalg.norm(Y - Yref, axis=0) / np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    assert err.max() < TOL, (int(err.argmax()), int(lens[err.argmax()]), float(err.max()))
    assert abs(loss - lref) <= TOL * abs(lref)


@pytest.mark.parametrize("solver", [1, 0])
@pytest.mark.parametrize("ratio", [2.0 ** 8, 2.0 ** 12, 2.0 ** 16, 2.0 ** 20])
def test_factor_rows_of_very_different_norms_on_the_fp16_paths(ratio, solver):
    """the matrix-core assemblies (wrmf_ne.hip, wrmf_chol_mf.hip, wrmf_cg_mf.hip, the shared dense product of
    wrmf_cgq.hip) scale their fp16 operand terms by ONE power of two taken from the global max |X| (x sqrt(max c - 1)): a factor
    matrix whose rows differ by `ratio` in norm -- a few huge item vectors, many tiny ones (popular against cold items under a strong
    regulariser) -- pushes the terms of the tiny rows towards fp16's subnormals.  Long rows, rows of 65..512 and short rows made ONLY
    of the tiny vectors (and a few rows that also hold a huge one), conjugate gradient and the exact solver, against the fp64 oracle.
    The claim: no worse than 3 x what the fp32 ORACLE does on the same system.  What the test found: the regime the
    worry is about does not exist for this operator -- the huge vectors are part of XtX, whose condition number grows with ratio^2, so at
    ratio >= 2^14 (where fp16 terms of the tiny rows would start to lose bits) the fp32 reference arithmetic itself has no digit left
    (fp32 oracle: errors of 0.1 .. 3 per row at 2^16).  Up to 2^12 every row must meet the bound; beyond, the rows on which fp32
    arithmetic still means something (oracle-in-float error below 1e-2) must, and nothing may be non-finite."""
    rng = np.random.default_rng(int(np.log2(ratio)) + solver)
    n_fix, k = 6000, 128
    n_huge = 12
    lens = np.concatenate([rng.integers(600, 2200, 24), rng.integers(65, 512, 40), rng.integers(1, 64, 60), [900, 300, 20]])
    tiny_ids = np.arange(n_huge, n_fix)
    cols = []
    for j, n in enumerate(lens):
        ids = rng.choice(tiny_ids, size=int(n), replace=False)
        if j >= len(lens) - 3:        # the last three rows also hold huge vectors
            ids[:3] = rng.choice(n_huge, size=3, replace=False)
        cols.append(np.sort(ids))
    p = np.zeros(len(lens) + 1, dtype=np.int32)
    p[1:] = np.cumsum([c.size for c in cols])
    i = np.concatenate(cols).astype(np.int32)
    x = (1.0 + rng.gamma(1.0, 2.0, size=i.size)).astype(np.float32).astype(np.float64)
    X = np.asfortranarray((rng.standard_normal((k, n_fix)) * 0.05).astype(np.float32))
    X[:, :n_huge] *= np.float32(ratio)
    Y0 = np.asfortranarray((rng.standard_normal((k, len(lens))) * 0.05).astype(np.float32))
    csc = (n_fix, len(lens), p, i, x)
    Yref, lref = _oracle64(csc, X, Y0, 0.1, solver, 3, True)
    Y32 = Y0.copy(order="F")
    O.als_implicit(p, i, x, X, Y32, O.gramian(X, 0.1), 0.1, solver, 3)
    Y = Y0.copy(order="F")
    try:
        loss = als.als_implicit(csc, X, Y, 0.1, 1, solver, 3, "float", False, False)
    except _lib.RsparseHipError as e:   # (ratio 2^20: a system that is singular in fp32 for the general solver too is an ERROR, as in the reference)
        assert e.code == _lib.ERR_NUMERIC and ratio >= 2.0 ** 16, e
        return
    assert np.all(np.isfinite(Y)) and np.isfinite(loss)
    den = np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
    err = np.linalg.norm(Y - Yref, axis=0) / den
    err32 = np.linalg.norm(Y32 - Yref, axis=0) / den
    bound = np.maximum(TOL, 3.0 * err32)
    held = np.ones(len(lens), dtype=bool) if ratio <= 2.0 ** 12 else err32 < 1e-2
    if held.any():
        worst = int(np.argmax(np.where(held, err / bound, 0.0)))
        assert np.all(err[held] <= bound[held]), (ratio, solver, worst, int(lens[worst]), float(err[worst]), float(err32[worst]))
    if ratio <= 2.0 ** 12:
        assert abs(loss - lref) <= max(TOL, 3.0 * float(np.max(err32))) * abs(lref)
