#!/usr/bin/env python3
"""one precision = "double" conjugate-gradient fit at config-2 AND config-3 size, timed: seconds per ALS
iteration on the fp64 layer (engine.ShardedALS with float64 blocks = what WRMF(precision="double") drives) and the exact user
half-iteration that ends the fit, next to the fp32 layer on the same matrix.   python tools/gpu_f64_config_time.py 2|3"""
import json, sys, time
import torch
sys.path.insert(0, ".")
from rsparse_amd import synth
from rsparse_amd.engine import HipBackend, ShardedALS

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nu, ni, k = ((1_000_000, 100_000, 64) if cfg == 2 else (10_000_000, 1_000_000, 128))
be = HipBackend(0)
d = synth.make_dataset(nu, ni, device=be.device)
out = {"config": cfg, "n_users": nu, "n_items": ni, "nnz": d["nnz"], "rank": k}
for name, dt in (("float", torch.float32), ("double", torch.float64)):
    blk = lambda c: (c[0], c[1], c[2].to(dt))
    als = ShardedALS(be, nu, ni, k, blk(d["c_ui"]), blk(d["c_iu"]), d["nnz"], lambda_=0.1)
    als.freeze_values()
    g = torch.Generator(device=be.device).manual_seed(1)
    U = (torch.randn(nu, k, generator=g, device=be.device) * 0.01).to(dt)
    V = torch.zeros(ni, k, device=be.device, dtype=dt)
    als.half_iteration("items", U, V, 1); als.half_iteration("users", U, V, 1)   # warm-up iteration
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n_it = 2
    for _ in range(n_it):
        li = als.half_iteration("items", U, V, 1)
        lu = als.half_iteration("users", U, V, 1)
    torch.cuda.synchronize(); t_it = (time.perf_counter() - t0) / n_it
    Z = torch.zeros_like(U)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    als.half_iteration("users", Z, V, 0, want_loss=False)     # the exact solve that ends fit_transform (R/model_WRMF.R:355-359)
    torch.cuda.synchronize(); t_fin = time.perf_counter() - t0
    be.check_numeric()
    out[name] = {"s_per_iteration": round(t_it, 4), "final_exact_user_half_s": round(t_fin, 4), "loss_users": lu}
    del als, U, V, Z
    torch.cuda.empty_cache()
    print(name, out[name], flush=True)
out["double_over_float_iteration"] = round(out["double"]["s_per_iteration"] / out["float"]["s_per_iteration"], 2)
print(json.dumps(out))
