"""`WRMF.fit_transform / transform` under torch.distributed (north star: "users and items shard across the GPUs" behind the
R6 API): world_size-2 gloo run through the WRMF class == the one-process oracle driver.  The numerics are the CPU stand-in
backend (tests/oracle_backend.py) -- what is under test is the class's multi-rank control flow: nnz-balanced blocks cut on
the host, broadcast initial factors, ShardedALS with sub-block-major storage, the sharded final exact solve, the row-sharded
transform of new data and its all-reduce assembly."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_fro

ROOT = Path(__file__).resolve().parent.parent


def _problem():
    rng = np.random.default_rng(11)
    n_user, n_item = 211, 67
    lens = np.clip(rng.lognormal(1.5, 1.0, n_user).astype(int), 0, 50)
    rows = np.repeat(np.arange(n_user), lens)
    cols = np.concatenate([rng.choice(n_item, size=l, replace=False) for l in lens])
    vals = 1.0 + rng.geometric(0.5, size=rows.size)
    m = sp.csr_matrix((vals, (rows, cols)), shape=(n_user, n_item))
    new = sp.csr_matrix((rng.random((23, n_item)) < 0.15) * 2.0)
    return m, new


def _fit(feedback, solver, with_global_bias, group_ready):
    from oracle_backend import OracleBackend
    from rsparse_amd import WRMF
    m, new = _problem()
    rng = np.random.default_rng(5)
    k = 8
    model = WRMF(rank=k, lambda_=0.1, feedback=feedback, solver=solver, with_global_bias=with_global_bias,
                 precision="float", backend=OracleBackend(), rng=123 if group_ready else 123)
    model._init_user_factors = (rng.standard_normal((m.shape[0], k)) * 0.01).astype(np.float32)
    if solver != "conjugate_gradient":
        model.components = (rng.standard_normal((k, m.shape[1])) * 0.01).astype(np.float32)
    emb = model.fit_transform(m, n_iter=3, convergence_tol=-1)
    return model, emb, model.transform(new)


def _worker(rank, ws, port, feedback, solver, gb, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        model, emb, emb_new = _fit(feedback, solver, gb, True)
        be = model._backend()
        torch.save({"emb": emb, "new": emb_new, "components": model.components, "losses": model.losses,
                    "global_bias": model.global_bias, "absmax_seen": getattr(be, "absmax_seen", 0)},
                   os.path.join(out_dir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("feedback,solver,gb", [("implicit", "conjugate_gradient", False), ("implicit", "cholesky", True),
                                                ("implicit", "conjugate_gradient", True), ("explicit", "cholesky", True),
                                                ("implicit", "nnls", False), ("explicit", "nnls", False)])
def test_wrmf_two_ranks_match_the_oracle_driver(tmp_path, feedback, solver, gb):
    import torch.multiprocessing as mp
    from oracle import wrmf_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, feedback, solver, gb, str(tmp_path)), nprocs=2, join=True)
    rs = [torch.load(tmp_path / ("r%d.pt" % r), weights_only=False) for r in range(2)]
    for key in ("emb", "new", "components"):
        assert np.array_equal(rs[0][key], rs[1][key]), key      # every rank returns / holds the same thing
    assert rs[0]["losses"] == rs[1]["losses"]
    if feedback == "implicit":
        assert rs[0]["absmax_seen"] > 0                         # max |F| travelled with the fused Gramian collective
    # one-process oracle driver, same initial factors, float arithmetic
    m, new = _problem()
    rng = np.random.default_rng(5)
    k = 8
    U0 = (rng.standard_normal((m.shape[0], k)) * 0.01).astype(np.float32)
    V0 = None if solver == "conjugate_gradient" else (rng.standard_normal((k, m.shape[1])) * 0.01).astype(np.float32)
    c = sp.csc_matrix(m); c.sort_indices()
    ref = O.OracleWRMF(k, lam=0.1, feedback=feedback, solver=solver, dtype=np.float32, n_threads=4, with_global_bias=gb)
    ref_emb = ref.fit_transform(m.shape[0], m.shape[1], c.indptr.astype(np.int32), c.indices.astype(np.int32),
                                c.data.astype(np.float64), U0.T.copy(), n_iter=3, convergence_tol=-1, init_components=V0)
    assert abs(rs[0]["global_bias"] - ref.global_bias) <= 1e-12 * max(1.0, abs(ref.global_bias))
    # NNLS squares the per-row system and stops at relative steps of 1e-4: the k x k Gramian sums that differ in their last
    # bits between one and two ranks (a different order) move its float trajectory by more than the other solvers'
    tol = 2e-3 if solver == "nnls" else 5e-5
    assert rel_fro(rs[0]["components"], ref.components) < tol
    assert rel_fro(rs[0]["emb"], ref_emb) < tol
    assert np.allclose([l[1] for l in rs[0]["losses"]], [l[1] for l in ref.losses], rtol=tol)
    if solver == "nnls":
        assert rs[0]["components"].min() >= 0 and rs[0]["emb"].min() >= 0
    nt = sp.csc_matrix(new.T); nt.sort_indices()
    ref_new = ref.transform(nt.indptr.astype(np.int32), nt.indices.astype(np.int32), nt.data.astype(np.float64))
    assert rs[0]["new"].shape == (new.shape[0], k) and rel_fro(rs[0]["new"], ref_new) < tol


def _worker_fail(rank, ws, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    from oracle_backend import OracleBackend
    from rsparse_amd import WRMF, _lib
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        m, _ = _problem()

        class OneRankFails(OracleBackend):      # only rank 1's block "had" a singular row
            def numeric_counts(self):
                return (3, 1) if rank == 1 else (0, 0)
        model = WRMF(rank=4, lambda_=0.1, feedback="implicit", solver="cholesky", precision="float", backend=OneRankFails(), rng=1)
        try:
            model.fit_transform(m, n_iter=1, convergence_tol=-1)
            outcome = "returned"
        except _lib.RsparseHipError as e:
            outcome = "raised:%d:%s" % (e.code, "3 per-row" in str(e))
        Path(out_dir, "f%d.txt" % rank).write_text(outcome)
    finally:
        dist.destroy_process_group()


def test_one_ranks_numeric_failure_raises_on_every_rank(tmp_path):
    """a rank that raised alone would leave the others waiting in the next collective; the counts are summed over
    the group first, so both ranks raise the same error (and the spawn below joins instead of hanging)."""
    import torch.multiprocessing as mp
    from rsparse_amd import _lib
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_fail, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert (tmp_path / ("f%d.txt" % r)).read_text() == "raised:%d:True" % _lib.ERR_NUMERIC


def _worker_bias(rank, ws, port, feedback, solver, gb, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    from oracle_backend import OracleBackend
    from rsparse_amd import WRMF
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        m, new = _problem()
        rng = np.random.default_rng(5)
        k = 6
        model = WRMF(rank=k, lambda_=0.1, feedback=feedback, solver=solver, with_user_item_bias=True, with_global_bias=gb,
                     precision="float", backend=OracleBackend(), rng=123, n_sub=3)
        model._init_user_factors = (rng.standard_normal((m.shape[0], k + 2)) * 0.01).astype(np.float32)
        if solver != "conjugate_gradient":
            model.components = (rng.standard_normal((k + 2, m.shape[1])) * 0.01).astype(np.float32)
        emb = model.fit_transform(m, n_iter=3, convergence_tol=-1)
        torch.save({"emb": emb, "new": model.transform(new), "components": model.components, "losses": model.losses,
                    "global_bias": model.global_bias}, os.path.join(out_dir, "b%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("feedback,solver,gb", [("explicit", "cholesky", True), ("explicit", "conjugate_gradient", False),
                                                ("implicit", "cholesky", False), ("implicit", "cholesky", True),
                                                ("explicit", "nnls", False)])
def test_wrmf_two_ranks_with_user_item_biases(tmp_path, feedback, solver, gb):
    """with_user_item_bias under sharding (round 4): the bias initialisation runs sweep by sweep over the ranks' blocks
    (ShardedALS.initialize_biases: every sweep against the full vector of the other side, the swept block all-gathered, the
    means all-reduced), then the half-iterations carry the two extra coordinates -- against the one-process oracle driver."""
    import torch.multiprocessing as mp
    from oracle import wrmf_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_bias, args=(2, port, feedback, solver, gb, str(tmp_path)), nprocs=2, join=True)
    rs = [torch.load(tmp_path / ("b%d.pt" % r), weights_only=False) for r in range(2)]
    for key in ("emb", "new", "components"):
        assert np.array_equal(rs[0][key], rs[1][key]), key
    # (users without ratings get 0 / 0 = NaN as their initial bias under dynamic lambda, in the reference too: the first
    # item-half loss carries it through the regulariser; the user half then solves those users and the NaN is gone)
    assert np.array_equal(np.array(rs[0]["losses"]), np.array(rs[1]["losses"]), equal_nan=True)
    m, new = _problem()
    rng = np.random.default_rng(5)
    k = 6
    U0 = (rng.standard_normal((m.shape[0], k + 2)) * 0.01).astype(np.float32)
    V0 = None if solver == "conjugate_gradient" else (rng.standard_normal((k + 2, m.shape[1])) * 0.01).astype(np.float32)
    c = sp.csc_matrix(m); c.sort_indices()
    ref = O.OracleWRMF(k, lam=0.1, feedback=feedback, solver=solver, dtype=np.float32, n_threads=4, with_user_item_bias=True,
                       with_global_bias=gb)
    ref_emb = ref.fit_transform(m.shape[0], m.shape[1], c.indptr.astype(np.int32), c.indices.astype(np.int32),
                                c.data.astype(np.float64), U0.T.copy(), n_iter=3, convergence_tol=-1, init_components=V0)
    tol = 2e-3 if solver == "nnls" else 1e-4
    assert abs(rs[0]["global_bias"] - ref.global_bias) <= 1e-6 * max(1.0, abs(ref.global_bias))
    assert rel_fro(rs[0]["components"], ref.components) < tol
    assert rel_fro(rs[0]["emb"], ref_emb) < tol
    assert np.allclose([l[1] for l in rs[0]["losses"]], [l[1] for l in ref.losses], rtol=tol)
    nt = sp.csc_matrix(new.T); nt.sort_indices()
    ref_new = ref.transform(nt.indptr.astype(np.int32), nt.indices.astype(np.int32), nt.data.astype(np.float64))
    assert rel_fro(rs[0]["new"], ref_new) < tol


def test_csr_input_gives_the_sharded_fit_the_same_blocks():
    """A canonical CSR input spares the host one of its two conversions (it is c_iu already); the blocks every rank uploads
    must be the arrays the CSC route produces, and the caller's matrix must survive the explicit global mean."""
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT / "tests"))
    from oracle_backend import OracleBackend
    from rsparse_amd import WRMF

    class Spy(OracleBackend):
        def __init__(self):
            self.made = []

        def make_csc(self, n_rows, n_cols, p, i, x):
            self.made.append((n_rows, n_cols, p.numpy().copy(), i.numpy().copy(), x.numpy().copy()))
            return super().make_csc(n_rows, n_cols, p, i, x)

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        m, _ = _problem()
        m = sp.csr_matrix(m, dtype=np.float64)
        m.sort_indices()
        assert m.has_canonical_format
        before = m.data.copy()
        for feedback, gb in (("implicit", False), ("explicit", True)):
            runs = []
            for x in (m, sp.csc_matrix(m), m.astype(np.float32)):
                be = Spy()
                model = WRMF(rank=4, lambda_=0.1, feedback=feedback, solver="cholesky", with_global_bias=gb, precision="float",
                             backend=be, rng=3, n_sub=2)
                model._fit_transform_sharded(x, 1, -1, 1, 0)     # the multi-rank driver on a one-rank group
                runs.append((be.made, model.global_bias))
            for made, g in runs[1:]:
                assert g == runs[0][1] and len(made) == len(runs[0][0]) > 0
                for a, b in zip(made, runs[0][0]):
                    assert a[:2] == b[:2] and all(np.array_equal(u, v) and u.dtype == v.dtype for u, v in zip(a[2:], b[2:]))
            assert (runs[0][1] != 0.0) == gb
        assert np.array_equal(m.data, before)
    finally:
        dist.destroy_process_group()


def _skewed_problem():
    """one user holds a quarter of the entries and one item is in most rows: cut into eight nnz-balanced blocks, some ranks
    own NO users and some NO items (an empty block: no solve, a zero-row slab in every exchange, nothing to send in the
    all-to-all that builds the item blocks)"""
    rng = np.random.default_rng(23)
    n_user, n_item = 90, 81
    lens = rng.integers(1, 3, n_user)
    lens[17] = 78
    rows = np.repeat(np.arange(n_user), lens)
    cols = np.concatenate([rng.choice(np.arange(1, n_item), size=l, replace=False) for l in lens])
    vals = 1.0 + rng.geometric(0.5, size=rows.size)
    m = sp.csr_matrix((vals, (rows, cols)), shape=(n_user, n_item)).tolil()
    m[:, 0] = 2.0                     # item 0: nearly every user
    m[::7, 0] = 0.0
    m = sp.csr_matrix(m)
    m.eliminate_zeros()
    m.sort_indices()
    new = sp.csr_matrix((rng.random((19, n_item)) < 0.2) * 1.0)
    return m, new


def _worker8(rank, ws, port, feedback, solver, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")   # 8 ranks on as many cores
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    from oracle_backend import OracleBackend
    from rsparse_amd import WRMF
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        m, new = _skewed_problem()
        rng = np.random.default_rng(5)
        k = 8
        model = WRMF(rank=k, lambda_=0.1, feedback=feedback, solver=solver, precision="float", backend=OracleBackend(),
                     rng=100 + rank, n_sub=(8, 4))                      # (differently seeded ranks: rank 0's factors win)
        model._init_user_factors = (rng.standard_normal((m.shape[0], k)) * 0.01).astype(np.float32) if rank == 0 else None
        if solver != "conjugate_gradient" and rank == 0:
            model.components = (rng.standard_normal((k, m.shape[1])) * 0.01).astype(np.float32)
        elif solver != "conjugate_gradient":
            model.components = np.zeros((k, m.shape[1]), dtype=np.float32)
        emb = model.fit_transform(m, n_iter=2, convergence_tol=-1)
        top = model.predict(new, 5)
        torch.save({"emb": emb, "new": model.transform(new), "components": model.components, "losses": model.losses,
                    "top": np.asarray(top), "host_s": model.host_seconds_before_first_iteration},
                   os.path.join(out_dir, "e%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("feedback,solver", [("implicit", "conjugate_gradient"), ("explicit", "cholesky")])
def test_wrmf_eight_ranks_with_empty_blocks(tmp_path, feedback, solver):
    """World size 8 through the class: n_sub = (8, 4), ranks whose user block or item block is empty,
    the item blocks built by all-to-all from the user blocks (no rank holds the matrix in both orientations), initial
    factors from rank 0 only -- against the one-process oracle driver."""
    import torch.multiprocessing as mp
    from oracle import wrmf_oracle as O
    from rsparse_amd.engine import balanced_bounds
    m, new = _skewed_problem()
    bu = balanced_bounds(np.diff(m.indptr), 8)
    bi = balanced_bounds(np.diff(sp.csc_matrix(m).indptr), 8)
    assert any(b == a for a, b in bu) and any(b == a for a, b in bi)           # the case under test
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker8, args=(8, port, feedback, solver, str(tmp_path)), nprocs=8, join=True)
    rs = [torch.load(tmp_path / ("e%d.pt" % r), weights_only=False) for r in range(8)]
    for r in range(1, 8):
        for key in ("emb", "new", "components", "top"):
            assert np.array_equal(rs[0][key], rs[r][key]), (key, r)
        assert rs[0]["losses"] == rs[r]["losses"]
    rng = np.random.default_rng(5)
    k = 8
    U0 = (rng.standard_normal((m.shape[0], k)) * 0.01).astype(np.float32)
    V0 = None if solver == "conjugate_gradient" else (rng.standard_normal((k, m.shape[1])) * 0.01).astype(np.float32)
    c = sp.csc_matrix(m); c.sort_indices()
    ref = O.OracleWRMF(k, lam=0.1, feedback=feedback, solver=solver, dtype=np.float32, n_threads=4)
    ref_emb = ref.fit_transform(m.shape[0], m.shape[1], c.indptr.astype(np.int32), c.indices.astype(np.int32),
                                c.data.astype(np.float64), U0.T.copy(), n_iter=2, convergence_tol=-1, init_components=V0)
    assert rel_fro(rs[0]["components"], ref.components) < 5e-5
    assert rel_fro(rs[0]["emb"], ref_emb) < 5e-5
    assert np.allclose([l[1] for l in rs[0]["losses"]], [l[1] for l in ref.losses], rtol=5e-5)
    nt = sp.csc_matrix(new.T); nt.sort_indices()
    ref_new = ref.transform(nt.indptr.astype(np.int32), nt.indices.astype(np.int32), nt.data.astype(np.float64))
    assert rel_fro(rs[0]["new"], ref_new) < 5e-5
    assert rs[0]["top"].shape == (new.shape[0], 5)
    assert all(r["host_s"] < 5.0 for r in rs)
