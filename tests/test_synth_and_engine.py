"""CPU tests of the synthetic generator and of the multi-rank control flow (gloo, world_size 2)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_fro
from rsparse_amd import synth
from rsparse_amd.engine import ShardedALS, block_bounds

ROOT = Path(__file__).resolve().parent.parent


def test_generator_properties():
    d = synth.make_dataset(5000, 800, mean_deg=20, d_max=300, device="cpu", block=1777)
    p, i, x = d["c_iu"]
    pi, ri, xi = d["c_ui"]
    assert p.dtype == torch.int32 and i.dtype == torch.int32 and x.dtype == torch.float32
    A = sp.csr_matrix((x.numpy(), i.numpy(), p.numpy()), shape=(5000, 800))
    B = sp.csc_matrix((xi.numpy(), ri.numpy(), pi.numpy()), shape=(5000, 800))
    assert abs(A - B).sum() == 0                       # both orientations hold the same matrix
    deg = np.diff(p.numpy())
    assert deg.min() >= 1 and deg.max() <= 300 and 15 < deg.mean() < 25
    for r in (0, 17, 4999):                            # sorted, unique inside a row
        seg = i[p[r]:p[r + 1]].numpy()
        assert np.all(np.diff(seg) > 0)
    assert x.min() >= 1 and float((x == 1).float().mean()) > 0.4      # 1 + Geometric(1/2)
    d2 = synth.make_dataset(5000, 800, mean_deg=20, d_max=300, device="cpu", block=5000)
    assert torch.equal(d2["c_iu"][1], i) and torch.equal(d2["c_ui"][2], xi)   # block independent
    e = synth.make_dataset(300, 100, mean_deg=10, d_max=50, feedback="explicit", device="cpu")
    assert set(np.unique(e["c_iu"][2].numpy())) <= {1.0, 2.0, 3.0, 4.0, 5.0}


def test_block_bounds():
    B, b = block_bounds(10, 4)
    assert B == 3 and b == [(0, 3), (3, 6), (6, 9), (9, 10)]
    B, b = block_bounds(2, 4)
    assert B == 1 and b == [(0, 1), (1, 2), (2, 2), (2, 2)]
    B, b = block_bounds(10, 2, multiple=4)          # blocks padded to a multiple of the sub-block count
    assert B == 8 and b == [(0, 8), (8, 10)]
    Bu, ub, Bi, ib, n_sub = ShardedALS.partition(301, 97, 2)
    assert n_sub == 4 and Bu % 4 == 0 and ub[1][1] == 301 and ShardedALS.partition(301, 97, 1)[4] == 1


def _shard_csc(p, i, x, c0, c1):
    p = p.to(torch.int64)
    lo, hi = int(p[c0]), int(p[c1])
    return (p[c0:c1 + 1] - lo).to(torch.int32).contiguous(), i[lo:hi].contiguous(), x[lo:hi].contiguous()


def run_sharded(rank, ws, group, feedback, solver, n_iter=2):
    from oracle_backend import OracleBackend
    n_user, n_item, k, lam = 301, 97, 8, 0.1
    d = synth.make_dataset(n_user, n_item, mean_deg=12, d_max=60, feedback=feedback, device="cpu")
    Bu, ub, Bi, ib, n_sub = ShardedALS.partition(n_user, n_item, ws)
    als = ShardedALS(OracleBackend(), n_user, n_item, k, _shard_csc(*d["c_ui"], *ib[rank]),
                     _shard_csc(*d["c_iu"], *ub[rank]), d["nnz"], feedback=feedback, lambda_=lam,
                     dynamic_lambda=True, cg_steps=3, group=group, world_size=ws, my_rank=rank)
    als.cnt_user = (d["c_iu"][0][1:] - d["c_iu"][0][:-1]).to(torch.float32)
    als.cnt_item = (d["c_ui"][0][1:] - d["c_ui"][0][:-1]).to(torch.float32)
    g = torch.Generator().manual_seed(5)
    U = als.alloc_factors(n_user, Bu, "cpu")
    V = als.alloc_factors(n_item, Bi, "cpu")
    U[:n_user] = torch.randn(n_user, k, generator=g) * 0.01
    if solver == 0:
        V[:n_item] = torch.randn(n_item, k, generator=g) * 0.01
    losses = []
    for _ in range(n_iter):
        li = als.half_iteration("items", U, V, solver)
        lu = als.half_iteration("users", U, V, solver)
        losses.append((li, lu))
    return U[:n_user].clone(), V[:n_item].clone(), losses


def _worker(rank, ws, port, feedback, solver, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        U, V, losses = run_sharded(rank, ws, None, feedback, solver)
        torch.save({"U": U, "V": V, "losses": losses}, os.path.join(out_dir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("feedback,solver", [("implicit", 1), ("implicit", 0), ("explicit", 1)])
def test_two_ranks_match_single_rank(tmp_path, feedback, solver):
    """world_size-2 gloo run == single-process run: the factors are independent of the sharding
    (each row's arithmetic is identical); only the Gramian / loss summation order changes."""
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT / "tests"))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, feedback, solver, str(tmp_path)), nprocs=2, join=True)
    U1, V1, l1 = run_sharded(0, 1, None, feedback, solver)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["U"], r1["U"]) and torch.equal(r0["V"], r1["V"])     # replicas agree exactly
    assert rel_fro(r0["U"].numpy(), U1.numpy()) < 2e-5
    assert rel_fro(r0["V"].numpy(), V1.numpy()) < 2e-5
    for (a, b), (c, d) in zip(r0["losses"], l1):
        assert abs(a - c) <= 1e-5 * abs(c) and abs(b - d) <= 1e-5 * abs(d)
    assert r0["losses"] == r1["losses"]
