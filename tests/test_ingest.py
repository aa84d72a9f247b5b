"""On-device ingest: the transposed orientation c_iu = t_shallow(as.csr.matrix(c_ui))
(R/model_WRMF.R:184-191) and the f64 -> f32 value conversion are produced in HBM; the result must be
bit-identical to the host-side transpose (integer / byte work: no tolerance)."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_movielens

gpu = pytest.mark.gpu


def host_transpose(n_rows, n_cols, p, i, x):
    m = sp.csc_matrix((x, i, p), shape=(n_rows, n_cols))
    t = sp.csc_matrix(m.T)
    t.sort_indices()
    return t.indptr.astype(np.int32), t.indices.astype(np.int32), t.data.astype(np.float32)


def device_transpose(be, n_rows, n_cols, p, i, x):
    import torch
    d = (be.to_device(p, torch.int32), be.to_device(i, torch.int32), be.to_device(x, torch.float32))
    pt, it, xt = be.transpose_csc(n_rows, n_cols, *d)
    return pt.cpu().numpy(), it.cpu().numpy(), xt.cpu().numpy()


def random_csc(rng, n_rows, n_cols, density, empty_cols=(), empty_rows=()):
    m = sp.random(n_rows, n_cols, density=density, format="csc", random_state=np.random.RandomState(rng.integers(1 << 30)),
                  data_rvs=lambda n: rng.integers(1, 9, n).astype(np.float64))
    m = m.tolil()
    for c in empty_cols:
        m[:, c] = 0
    for r in empty_rows:
        m[r, :] = 0
    m = sp.csc_matrix(m)
    m.eliminate_zeros()
    m.sort_indices()
    return m


@gpu
@pytest.mark.parametrize("shape,density", [((943, 1682), 0.06), ((1, 7), 0.5), ((5000, 3), 0.3), ((300, 70000), 0.002),
                                            ((70000, 300), 0.002), ((17, 17), 1.0)])
def test_transpose_matches_host(shape, density):
    from rsparse_amd.engine import HipBackend
    be = HipBackend(0)
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    m = random_csc(rng, *shape, density, empty_cols=(0,) if shape[1] > 2 else (), empty_rows=(shape[0] - 1,) if shape[0] > 2 else ())
    x32 = m.data.astype(np.float32)
    got = device_transpose(be, shape[0], shape[1], m.indptr.astype(np.int32), m.indices.astype(np.int32), x32)
    want = host_transpose(shape[0], shape[1], m.indptr, m.indices, x32)
    for g, w, name in zip(got, want, ("p", "i", "x")):
        assert g.shape == w.shape, name
        assert np.array_equal(g, w), name


@gpu
def test_transpose_empty_and_involution():
    from rsparse_amd.engine import HipBackend
    be = HipBackend(0)
    # no non-zeros at all
    p = np.zeros(6, np.int32)
    pt, it, xt = device_transpose(be, 4, 5, p, np.zeros(0, np.int32), np.zeros(0, np.float32))
    assert np.array_equal(pt, np.zeros(5, np.int32)) and it.size == 0 and xt.size == 0
    # transpose twice = identity (movielens fixture, the reference's own data set)
    n_rows, n_cols, p, i, x = load_movielens()
    p, i, x = p.astype(np.int32), i.astype(np.int32), x.astype(np.float32)
    t = device_transpose(be, n_rows, n_cols, p, i, x)
    tt = device_transpose(be, n_cols, n_rows, *t)
    assert np.array_equal(tt[0], p) and np.array_equal(tt[1], i) and np.array_equal(tt[2], x)
    want = host_transpose(n_rows, n_cols, p, i, x)
    assert all(np.array_equal(a, b) for a, b in zip(t, want))


@gpu
def test_transpose_rejects_bad_index_and_converts_values():
    import torch
    from rsparse_amd import _lib
    from rsparse_amd.engine import HipBackend
    be = HipBackend(0)
    p = np.array([0, 2, 3], np.int32)
    i = np.array([0, 9, 1], np.int32)          # 9 >= n_rows
    with pytest.raises(_lib.RsparseHipError):
        device_transpose(be, 4, 2, p, i, np.ones(3, np.float32))
    x64 = np.array([1.0, 2.5, 1e-3, 16777217.0, -0.0], np.float64)
    got = be.values_to_float(be.to_device(x64, torch.float64)).cpu().numpy()
    assert np.array_equal(got, x64.astype(np.float32)) and got.dtype == np.float32


@gpu
def test_transpose_config2_scale_properties():
    """At 1M x 100k / 5e7 nnz: sortedness inside columns, column counts = histogram of the input rows, checksum of
    (row, col, value) triples preserved, and the generator's own second orientation reproduced exactly."""
    import torch
    from rsparse_amd import synth
    from rsparse_amd.engine import HipBackend
    be = HipBackend(0)
    d = synth.make_dataset(1_000_000, 100_000, device=be.device)
    p, i, x = d["c_iu"]                      # columns = users, rows = items
    pt, it, xt = be.transpose_csc(100_000, 1_000_000, p, i, x)
    p2, i2, x2 = d["c_ui"]
    assert torch.equal(pt, p2) and torch.equal(it, i2) and torch.equal(xt, x2)
    counts = torch.bincount(i.to(torch.int64), minlength=100_000)
    assert torch.equal(torch.diff(pt).to(torch.int64), counts)
    cols = torch.repeat_interleave(torch.arange(100_000, device=be.device), torch.diff(pt).to(torch.int64))
    same_col = cols[1:] == cols[:-1]
    assert bool(torch.all(it[1:][same_col] > it[:-1][same_col]))


@gpu
def test_csc_create_device_validates_like_the_host_constructor():
    """rsparse_hip_csc_create_device runs the same row-index range check as rsparse_hip_csc_create_host (the kernels
    gather X[row_index] unchecked) and rejects decreasing column pointers."""
    import torch
    from rsparse_amd import _lib
    from rsparse_amd.engine import HipBackend
    be = HipBackend(0)
    ok = (be.to_device(np.array([0, 2, 3], np.int32), torch.int32), be.to_device(np.array([0, 3, 1], np.int32), torch.int32),
          be.to_device(np.ones(3, np.float32), torch.float32))
    be.make_csc(4, 2, *ok)
    for bad_i in ([0, 4, 1], [0, -1, 1]):
        bad = (ok[0], be.to_device(np.array(bad_i, np.int32), torch.int32), ok[2])
        with pytest.raises(_lib.RsparseHipError) as e:
            be.make_csc(4, 2, *bad)
        assert e.value.code == _lib.ERR_INVALID
    with pytest.raises(_lib.RsparseHipError) as e:
        be.make_csc(4, 2, be.to_device(np.array([0, 3, 2], np.int32), torch.int32), ok[1], ok[2])
    assert e.value.code == _lib.ERR_INVALID


@gpu
def test_transpose_moves_payload_bits_untouched():
    """The fp64 layer sends each entry's POSITION through the 32-bit ingest as a bit-cast float (engine.py:transpose_csc):
    the payload must come back bit for bit whatever it looks like as a float -- integers beyond 2^23 (not representable as
    consecutive floats), subnormal patterns (small integers), and patterns just below the NaN range."""
    import torch
    from rsparse_amd.engine import HipBackend
    be = HipBackend(0)
    rng = np.random.default_rng(5)
    m = random_csc(rng, 4000, 900, 0.02)
    nnz = m.nnz
    for base in (0, (1 << 23) - 7, (1 << 24) + 1, (1 << 30) + 12345, 0x7F800000 - nnz - 1):
        bits = (np.arange(nnz, dtype=np.int64) + base).astype(np.int32)
        p = be.to_device(m.indptr.astype(np.int32), torch.int32)
        i = be.to_device(m.indices.astype(np.int32), torch.int32)
        x = be.to_device(bits, torch.int32).view(torch.float32)
        pt, it, xt = be.transpose_csc(4000, 900, p, i, x)
        got = xt.view(torch.int32).cpu().numpy()
        # the host transpose of the same matrix with the payload as int64 data
        t = sp.csc_matrix(sp.csc_matrix((bits.astype(np.int64), m.indices, m.indptr), shape=(4000, 900)).T)
        t.sort_indices()
        assert np.array_equal(it.cpu().numpy(), t.indices.astype(np.int32))
        assert np.array_equal(got, t.data.astype(np.int32)), base
    # and through the fp64 branch itself: doubles that differ only below float precision come back in transposed order
    xd = 1.0 + np.arange(nnz, dtype=np.float64) * 2.0 ** -40
    p = be.to_device(m.indptr.astype(np.int32), torch.int32)
    i = be.to_device(m.indices.astype(np.int32), torch.int32)
    pt, it, xt = be.transpose_csc(4000, 900, p, i, be.to_device(xd, torch.float64))
    t = sp.csc_matrix(sp.csc_matrix((xd, m.indices, m.indptr), shape=(4000, 900)).T)
    t.sort_indices()
    assert np.array_equal(xt.cpu().numpy(), t.data)
