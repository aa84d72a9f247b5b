"""Time the Cholesky half-iteration (the final transform of every fit) at several scales."""
import sys, time, json
import torch
sys.path.insert(0, ".")
from rsparse_amd import synth
from rsparse_amd.engine import HipBackend, ShardedALS

be = HipBackend(0)
out = []
for (nu, ni, k) in [(200000, 30000, 128), (1000000, 100000, 64), (1000000, 100000, 128)]:
    d = synth.make_dataset(nu, ni, device=be.device)
    als = ShardedALS(be, nu, ni, k, d["c_ui"], d["c_iu"], d["nnz"], lambda_=0.1)
    g = torch.Generator(device=be.device).manual_seed(1)
    U = torch.randn(nu, k, generator=g, device=be.device) * 0.01
    V = torch.zeros(ni, k, device=be.device)
    als.half_iteration("items", U, V, 1)
    als.half_iteration("users", U, V, 1)
    res = {}
    for side in ("users", "items"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        als.half_iteration(side, U.clone(), V.clone(), 0)
        torch.cuda.synchronize(); res[side] = time.perf_counter() - t0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        als.half_iteration(side, U.clone(), V.clone(), 1)
        torch.cuda.synchronize(); res[side + "_cg"] = time.perf_counter() - t0
    be.check_numeric()
    out.append({"n_users": nu, "n_items": ni, "k": k, "nnz": d["nnz"], **{kk: round(v, 4) for kk, v in res.items()}})
    print(out[-1], flush=True)
