"""$predict / top_product: the reference's own value-level test (tests/testthat/test-top-product.R:3-14:
top-k of one row equals order(decreasing)[1:k]) for the oracle on CPU, and GPU parity of the HIP kernel
against the oracle including exclusions, ties and NA fill."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import wrmf_oracle as O
from rsparse_amd import _lib


def test_oracle_matches_order_decreasing():
    rng = np.random.default_rng(1)
    nr, k, nc = 100, 10, 50                       # test-top-product.R:4-6
    m1 = rng.random((nr, k))
    m2 = rng.random((k, nc))
    m3 = m1 @ m2
    res, scores = O.top_product(m1, m2, k)
    for row in (0, 17, 99):
        expect = np.argsort(-m3[row], kind="stable")[:k] + 1
        assert np.array_equal(res[row], expect)
        assert np.allclose(scores[row], m3[row][expect - 1])


def test_oracle_exclusions_ties_and_na():
    x = np.zeros((2, 3))
    x[1] = [1.0, 0.0, 0.0]
    y = np.arange(18, dtype=np.float64).reshape(3, 6)
    # row 0: all scores 0 -> the heap keeps the first k admissible items, output has the larger index first
    res, sc = O.top_product(x, y, 3)
    assert list(res[0]) == [3, 2, 1] and np.all(sc[0] == 0)
    assert list(res[1]) == [6, 5, 4]
    nr = sp.csr_matrix(np.array([[0, 1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1]]))
    res, sc = O.top_product(x, y, 3, nr.indptr, nr.indices, exclude=[5])
    assert list(res[0]) == [4, 3, 1] and list(res[1]) == [4, 3, 2]
    res, sc = O.top_product(x, y, 5, exclude=[1, 2, 3])
    assert list(res[0][:3]) == [6, 5, 4] and np.all(res[0][3:] == O.NA_INTEGER) and np.isnan(sc[0][3:]).all()


def test_oracle_tie_eviction_order():
    """src/matrix_top_product.cpp:61-86: a min-heap on (score, index) pairs; a newcomer replaces the top only if its score
    is strictly larger, and the top among tied minima is the smallest index."""
    one = np.ones((1, 1))
    res, _ = O.top_product(one, np.array([[1.0, 1.0, 5.0]]), 2)
    assert list(res[0]) == [3, 2]          # (1, item 0) is evicted by the 5, (1, item 1) stays
    res, _ = O.top_product(one, np.array([[1.0, 5.0, 1.0]]), 2)
    assert list(res[0]) == [2, 1]          # the heap is full with minimum 1 when the second 1 arrives: not admitted


def _hip_top_product(x, y, k, nr=None, exclude=(), glob_mean=0.0):
    lib = _lib.load()
    nrow, rank = x.shape
    nc = y.shape[1]
    xf = np.asfortranarray(x, dtype=np.float64)
    yf = np.asfortranarray(y, dtype=np.float64)
    res = np.zeros((nrow, k), dtype=np.int32, order="F")
    sc = np.zeros((nrow, k), dtype=np.float64, order="F")
    vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    p = j = None
    if nr is not None:
        nr = sp.csr_matrix(nr)
        nr.sort_indices()
        p, j = nr.indptr.astype(np.int32), nr.indices.astype(np.int32)
    ex = np.asarray(list(exclude), dtype=np.int32)
    _lib.check(lib.rsparse_hip_top_product(vp(xf), vp(yf), nrow, nc, rank, k, 1, vp(p), vp(j), vp(ex) if ex.size else None,
                                           int(ex.size), float(glob_mean), vp(res), vp(sc)))
    return res, sc


@pytest.mark.gpu
@pytest.mark.parametrize("rank,nr,nc,k", [(10, 100, 50, 10), (16, 70, 3001, 25), (64, 33, 777, 7), (128, 129, 5000, 100),
                                          (20, 5, 40, 12),
                                          # round 4: every launch geometry of wrmf_topk.hip -- four waves sharing the item tile
                                          # with 256 / 128 users per workgroup, one tile per wave with 64 / 32, two waves with
                                          # 32 (k up to 256 at rank 128) -- and the scalar staging path (rank % 4 != 0)
                                          (128, 300, 5000, 10), (128, 100, 3000, 100), (128, 70, 4000, 200),
                                          (64, 300, 2000, 256), (128, 33, 2000, 256), (30, 140, 900, 5), (128, 40, 1000, 120),
                                          # few users over many items: the items are split over the workgroups and the slices'
                                          # lists merged (one user: 64 slices; 1000 users: 16)
                                          (128, 1, 70000, 10), (64, 3, 20000, 100), (128, 1000, 9000, 10), (16, 37, 5000, 256)])
def test_hip_matches_oracle(rank, nr, nc, k):
    rng = np.random.default_rng(rank + nr)
    x = rng.standard_normal((nr, rank)).astype(np.float32).astype(np.float64)
    y = rng.standard_normal((rank, nc)).astype(np.float32).astype(np.float64)
    notrec = sp.random(nr, nc, density=0.05, random_state=3, format="csr")
    excl = [1, 7, nc]                                     # 1-based like R
    for args in (dict(), dict(nr=notrec), dict(nr=notrec, exclude=excl, glob_mean=0.5)):
        ref_i, ref_s = O.top_product(x, y, k, *(None, None) if "nr" not in args else (notrec.indptr, notrec.indices),
                                     exclude=args.get("exclude", ()), glob_mean=args.get("glob_mean", 0.0))
        got_i, got_s = _hip_top_product(x, y, k, **args)
        assert np.allclose(got_s, ref_s, rtol=1e-4, atol=1e-5, equal_nan=True)
        # indices agree wherever the neighbouring scores are separated by more than fp32 noise
        gap_ok = np.ones_like(ref_i, dtype=bool)
        d = np.abs(np.diff(ref_s, axis=1))
        tol = 1e-4 * np.maximum(1.0, np.abs(ref_s[:, :-1]))
        gap_ok[:, :-1] &= d > tol
        gap_ok[:, 1:] &= d > tol
        assert np.array_equal(got_i[gap_ok], ref_i[gap_ok])


@pytest.mark.gpu
def test_hip_ties_and_na_fill():
    x = np.zeros((3, 4))
    x[1] = [1.0, 0, 0, 0]
    y = np.arange(40, dtype=np.float64).reshape(4, 10)
    ref_i, ref_s = O.top_product(x, y, 4)
    got_i, got_s = _hip_top_product(x, y, 4)
    assert np.array_equal(got_i, ref_i) and np.array_equal(got_s, ref_s)       # all-zero rows: ties everywhere
    ref_i, ref_s = O.top_product(x, y, 6, exclude=[1, 2, 3, 4, 5, 6, 7])
    got_i, got_s = _hip_top_product(x, y, 6, exclude=[1, 2, 3, 4, 5, 6, 7])
    assert np.array_equal(got_i, ref_i) and np.array_equal(np.isnan(got_s), np.isnan(ref_s))
    assert (got_i[:, 3:] == O.NA_INTEGER).all()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [2, 7, 40])
def test_hip_tied_scores_follow_the_heap(k):
    """Integer-valued factors: every user has many items tied at its k-th score, interleaved with larger scores; which of
    them survive depends on the arrival order exactly as in the reference's heap (the device replays the arrivals of a
    round through the heap when, and only when, there are more candidates at the k-th score than places)."""
    one = np.ones((1, 1))
    for sc, kk in (([1.0, 1.0, 5.0], 2), ([1.0, 5.0, 1.0], 2)):
        ref_i, ref_s = O.top_product(one, np.array([sc]), kk)
        got_i, got_s = _hip_top_product(one, np.array([sc]), kk)
        assert np.array_equal(got_i, ref_i) and np.array_equal(got_s, ref_s)
    rng = np.random.default_rng(100 + k)
    for n_users in (70, 300, 40):      # (the tile-sharing kernel with 128 / 256 users per workgroup, the tile-per-wave kernel)
        x = rng.integers(0, 3, (n_users, 4)).astype(np.float64)
        y = rng.integers(0, 3, (4, 1500)).astype(np.float64)          # scores are small integers: ties everywhere
        nr = sp.random(n_users, 1500, density=0.02, format="csr", random_state=np.random.RandomState(k))
        nr.sort_indices()
        for kw in ({}, {"nr": nr, "exclude": [3, 77, 1400]}):
            ref_i, ref_s = O.top_product(x, y, k, *((kw["nr"].indptr, kw["nr"].indices) if kw else (None, None)),
                                         exclude=kw.get("exclude", ()))
            got_i, got_s = _hip_top_product(x, y, k, **kw)
            assert np.array_equal(got_i, ref_i) and np.array_equal(got_s, ref_s)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [3, 40])
def test_hip_split_items_with_ties_falls_back_to_the_heap(k):
    """The split launch (few users, many items) merges per-slice lists; membership at the k-th score is ambiguous when more
    candidates sit there than places -- then the user is recomputed by the unsplit kernel, which follows the reference's heap.
    Integer-valued factors over 6000 items (ties everywhere), with and without exclusion lists; plus k larger than the number of
    admissible items (NA fill through the merge)."""
    rng = np.random.default_rng(7 + k)
    for n_users in (1, 9, 70):
        x = rng.integers(0, 3, (n_users, 4)).astype(np.float64)
        y = rng.integers(0, 3, (4, 6000)).astype(np.float64)
        nr = sp.random(n_users, 6000, density=0.01, format="csr", random_state=np.random.RandomState(k))
        nr.sort_indices()
        for kw in ({}, {"nr": nr, "exclude": [3, 77, 5999]}):
            ref_i, ref_s = O.top_product(x, y, k, *((kw["nr"].indptr, kw["nr"].indices) if kw else (None, None)),
                                         exclude=kw.get("exclude", ()))
            got_i, got_s = _hip_top_product(x, y, k, **kw)
            assert np.array_equal(got_i, ref_i) and np.array_equal(got_s, ref_s)
    # almost everything excluded: fewer admissible items than k
    x = rng.standard_normal((2, 8))
    y = rng.standard_normal((8, 5000))
    excl = list(range(1, 5001 - 2))                      # 1-based: all but the last two items
    ref_i, ref_s = O.top_product(x, y, k, exclude=excl)
    got_i, got_s = _hip_top_product(x, y, k, exclude=excl)
    assert np.array_equal(got_i, ref_i) and np.array_equal(np.isnan(got_s), np.isnan(ref_s))
    assert (got_i[:, 2:] == O.NA_INTEGER).all()


@pytest.mark.gpu
@pytest.mark.parametrize("rank,nr,nc,k", [(128, 300, 6000, 10), (128, 130, 6000, 100), (64, 77, 3000, 10), (10, 500, 1682, 100),
                                          (36, 3, 30000, 10), (128, 40, 2500, 250)])
def test_hip_orders_like_the_double_product(rank, nr, nc, k):
    """find_top_product multiplies in double (R/utils.R:35-36): the device's candidates come from an fp32 pass, but scores and
    ORDER are those of the double product -- indices equal to the oracle's everywhere, not only where fp32 separates the scores."""
    rng = np.random.default_rng(7 * rank + k)
    x = rng.standard_normal((nr, rank))
    y = rng.standard_normal((rank, nc))
    notrec = sp.random(nr, nc, density=0.03, random_state=5, format="csr")
    for args in (dict(), dict(nr=notrec, exclude=[2, 9, nc], glob_mean=3.5)):
        ref_i, ref_s = O.top_product(x, y, k, *(None, None) if "nr" not in args else (notrec.indptr, notrec.indices),
                                     exclude=args.get("exclude", ()), glob_mean=args.get("glob_mean", 0.0))
        got_i, got_s = _hip_top_product(x, y, k, **args)
        assert np.array_equal(got_i, ref_i)
        assert np.allclose(got_s, ref_s, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("rank,nr,nc,k", [(128, 700, 20000, 100), (64, 520, 9000, 60), (32, 300, 7000, 200)])
def test_hip_large_k_many_users_candidates_in_global_memory(rank, nr, nc, k):
    """Many users at a k whose candidate buffers do not fit the LDS next to two item tiles: the tile-sharing kernel keeps them in
    a global scratch and settles a user once per batch of arrivals (wrmf_topk.hip GBUF).  Several workgroups, the last one
    partly filled, hundreds of tiles, exclusion lists; indices equal to the oracle's."""
    rng = np.random.default_rng(11 * rank + k)
    x = rng.standard_normal((nr, rank))
    y = rng.standard_normal((rank, nc))
    notrec = sp.random(nr, nc, density=0.01, random_state=9, format="csr")
    for args in (dict(), dict(nr=notrec, exclude=[5, 6, nc - 1], glob_mean=-1.25)):
        ref_i, ref_s = O.top_product(x, y, k, *(None, None) if "nr" not in args else (notrec.indptr, notrec.indices),
                                     exclude=args.get("exclude", ()), glob_mean=args.get("glob_mean", 0.0))
        got_i, got_s = _hip_top_product(x, y, k, **args)
        assert np.array_equal(got_i, ref_i)
        assert np.allclose(got_s, ref_s, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_hip_large_k_user_chunks_share_the_scratch():
    """More users than one launch of the global-buffer geometry takes (131072): the chunks run one after the other on the stream and
    reuse the scratch slots; the first and the last users (the second chunk's) against the oracle."""
    rng = np.random.default_rng(5)
    rank, nc, k, nr = 32, 4200, 60, 131072 + 200
    x = rng.standard_normal((nr, rank))
    y = rng.standard_normal((rank, nc))
    got_i, got_s = _hip_top_product(x, y, k, exclude=[10, 11])
    for rows in (slice(0, 64), slice(131072 - 32, 131072 + 200)):
        ref_i, ref_s = O.top_product(x[rows], y, k, exclude=[10, 11])
        assert np.array_equal(got_i[rows], ref_i)
        assert np.allclose(got_s[rows], ref_s, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_hip_near_ties_resolve_in_double():
    """Pairs of items whose vectors differ by 1e-9 relative: equal scores in fp32 (where the larger index would come first),
    ordered by the double product in the reference."""
    rng = np.random.default_rng(3)
    rank, nc, k = 64, 400, 20
    y = rng.standard_normal((rank, nc))
    for j in range(0, nc, 2):                      # item j + 1 = item j, a hair smaller or larger
        y[:, j + 1] = y[:, j] * (1.0 + (1e-9 if (j // 2) % 2 else -1e-9))
    x = rng.standard_normal((60, rank))
    ref_i, ref_s = O.top_product(x, y, k)
    got_i, got_s = _hip_top_product(x, y, k)
    assert np.array_equal(got_i, ref_i)
    assert np.allclose(got_s, ref_s, rtol=1e-13, atol=0)
    f32 = (x.astype(np.float32) @ y.astype(np.float32))
    assert (f32[:, 0::2] == f32[:, 1::2]).mean() > 0.5          # the case under test: fp32 does not separate the pairs


@pytest.mark.gpu
@pytest.mark.parametrize("k", [10, 100])
def test_wrmf_predict_orders_like_the_reference_on_movielens(movielens, ml_train, k):
    """every movielens user, k = 10 and 100, precision = "double" -- the indices `predict` returns equal
    the oracle's top_product of the model's own double factors."""
    from rsparse_amd import WRMF
    n_user, n_item, tp, ti, tx = ml_train
    train = sp.csc_matrix((tx, ti, tp), shape=(n_user, n_item)).tocsr()
    for precision in ("double", "float"):
        m = WRMF(rank=10, lambda_=0.1, feedback="implicit", solver="conjugate_gradient", precision=precision, rng=1)
        emb = m.fit_transform(train, n_iter=3, convergence_tol=-1)
        preds = m.predict(train, k)
        nr = train.copy(); nr.sort_indices()
        ref_i, ref_s = O.top_product(np.asarray(emb, dtype=np.float64), np.asarray(m.components, dtype=np.float64), k,
                                     nr.indptr, nr.indices)
        ref0 = np.where(ref_i == O.NA_INTEGER, -1, ref_i - 1)
        assert np.array_equal(np.asarray(preds), ref0), precision
        assert np.allclose(preds.scores, np.nan_to_num(ref_s), rtol=1e-12 if precision == "double" else 1e-6)


@pytest.mark.gpu
def test_wrmf_predict(movielens, ml_train):
    """test-wrmf.R:59-61: predict(cv, k = K) has nrow(cv) rows and K columns; never recommends seen items."""
    from conftest import csc_drop_rows
    from rsparse_amd import WRMF
    n_user_all, n_item, p, i, x = movielens
    n_user, _, tp, ti, tx = ml_train
    train = sp.csc_matrix((tx, ti, tp), shape=(n_user, n_item))
    cp, ci, cx = csc_drop_rows(900, p, i, x)
    cv = sp.csc_matrix((cx, ci, cp), shape=(n_user_all - 900, n_item)).tocsr()
    m = WRMF(rank=8, lambda_=0.1, feedback="implicit", solver="conjugate_gradient", precision="float", rng=1)
    m.fit_transform(train, n_iter=3, convergence_tol=-1)
    K = 7
    preds = m.predict(cv, K)
    assert preds.shape == (cv.shape[0], K) and preds.scores.shape == (cv.shape[0], K)
    emb = m.transform(cv)
    dense = emb.astype(np.float64) @ m.components.astype(np.float64)
    for r in range(cv.shape[0]):
        seen = set(cv.indices[cv.indptr[r]:cv.indptr[r + 1]])
        assert not (set(preds[r]) & seen)
        assert np.all(np.diff(preds.scores[r]) <= 1e-6)
        assert np.allclose(preds.scores[r], dense[r, preds[r]], rtol=1e-4, atol=1e-5)
    none = m.predict(cv, K, not_recommend=None, items_exclude=[0, 1, 2])
    assert not (set(none.ravel()) & {0, 1, 2})
