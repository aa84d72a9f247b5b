#!/usr/bin/env python3
"""bench.py -- ALS iterations/s of the WRMF implicit CG hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched through
torch.distributed.run, one rank per GPU, RCCL).  Prints ONE JSON line on rank 0.

Workload = BASELINE.json's metric configuration (configs[2]): synthetic 10M users x 1M items,
~500M nnz, rank 128, implicit feedback, CG solver with 3 steps, lambda 0.1 -- it fits one GPU
(~20 GB resident), so N=1 runs the full problem and N>1 shards the same problem (strong scaling).
A "step" is one ALS iteration = item half-iteration + user half-iteration, each including its
Gramian (+ all-reduce), the solve kernels, the factor all-gather (N>1) and the loss reduction --
exactly what `private$solver` does twice per iteration in R/model_WRMF.R:318-335.  Inputs are
resident in HBM when the timed region starts.

Other SURVEY.md 8(d) configurations (parity-test scale cases, reported in DESIGN.md, not the bench line):
  --config 2   1M x 100k, rank 64, implicit CG(3)          --config 4   config 3 with the Cholesky solver
  --config 5   5M x 500k, rank 64, explicit feedback, CG(3), dynamic lambda (add --solver cholesky for 5b)
or any mix of --users/--items/--rank/--solver/--feedback.

Extra objects on the line:
  roofline      dominant kernel (the CG launch with the largest share of the iteration), algorithmic bytes / measured duration
                (HIP events on the launch stream, recorded inside the library) vs the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (rsparse-shaped C++/OpenMP port, oracle/) timed on this node's host
                cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from rsparse_amd import synth  # noqa: E402
from rsparse_amd.engine import HipBackend, ShardedALS, block_bounds  # noqa: E402

FP32_PEAK_TFLOPS = 157.3   # MI355X vector fp32 (= matrix fp32) peak
CONFIGS = {
    2: dict(users=1_000_000, items=100_000, rank=64, solver="cg", feedback="implicit"),
    3: dict(users=10_000_000, items=1_000_000, rank=128, solver="cg", feedback="implicit"),
    4: dict(users=10_000_000, items=1_000_000, rank=128, solver="cholesky", feedback="implicit"),
    5: dict(users=5_000_000, items=500_000, rank=64, solver="cg", feedback="explicit"),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def shard_csc(p, i, x, c0, c1):
    p64 = p.to(torch.int64)
    lo, hi = int(p64[c0]), int(p64[c1])
    return (p64[c0:c1 + 1] - lo).to(torch.int32).contiguous(), i[lo:hi].contiguous(), x[lo:hi].contiguous()


def algorithmic_bytes(n_rows, nnz, k, n_empty=0):
    """B_half of SURVEY.md 8(d) / BASELINE.md 3 for a set of rows: every non-zero gathers one k-vector
    (4k B) + its index (4 B) + its value (4 B, fp32 once resident); every row reads its warm start and
    writes its solution (2*4k B) and its row pointer (4 B); one Gramian read (4k^2 B)."""
    return nnz * (4 * k + 8) + (n_rows - n_empty) * 8 * k + n_empty * 4 * k + (n_rows + 1) * 4 + 4 * k * k


def cpu_baseline(data, U, V, k, lam, cg_steps, target_s=12.0, implicit=True, solver=1):
    """Time the oracle on host cores over the leading users / items of the same matrices."""
    from oracle import wrmf_oracle as O
    threads = len(os.sched_getaffinity(0))
    try:
        O.lib(native=True)
        native = True
    except Exception:
        native = False
    Vh = np.asfortranarray(V.cpu().numpy().T)       # k x n_item
    Uh = np.asfortranarray(U.cpu().numpy().T)       # k x n_user
    out = {}

    def run(csc, X, Yfull, n_take):
        p, i, x = csc
        p = p[:n_take + 1].cpu().numpy().astype(np.int32)
        nnz = int(p[-1])
        i = i[:nnz].cpu().numpy().astype(np.int32)
        x = x[:nnz].cpu().numpy().astype(np.float64)
        Y = np.asfortranarray(Yfull[:, :n_take]).copy(order="F")
        G = O.gramian(X, lam, native=native)
        t0 = time.perf_counter()
        if implicit:
            O.als_implicit(p, i, x, X, Y, G, lam, solver, cg_steps, n_threads=threads, native=native)
        else:
            O.als_explicit(p, i, x, X, Y, None, lam, solver, cg_steps, dynamic_lambda=True, n_threads=threads,
                           native=native)
        return time.perf_counter() - t0, nnz

    n_user, n_item, nnz_tot = data["n_users"], data["n_items"], data["nnz"]
    # grow each sample until it costs about target_s/2 of wall time (the first calls include thread start-up)
    def sized(csc, X, Yfull, n_all, start):
        take = min(n_all, start)
        t, z = run(csc, X, Yfull, take)
        for _ in range(4):
            if t >= 0.6 * target_s / 2 or take >= n_all:
                break
            take = int(min(n_all, take * min(8.0, max(1.5, (target_s / 2) / max(t, 1e-3)))))
            t, z = run(csc, X, Yfull, take)
        return take, t, z

    take_u, tu, zu = sized(data["c_iu"], Vh, Uh, n_user, 50000)
    take_i, ti, zi = sized(data["c_ui"], Uh, Vh, n_item, 5000)
    est_iter_s = tu * (nnz_tot / max(zu, 1)) + ti * (nnz_tot / max(zi, 1))
    out = {
        "value": 1.0 / est_iter_s, "unit": "iterations/s", "cores": threads, "kind": "port",
        "sample": "oracle/wrmf_oracle.cpp (C++/OpenMP restatement, fp32, %s) on the first %d of %d users "
                  "(%.2fs, %d nnz) and the first %d of %d items (%.2fs, %d nnz); iteration time extrapolated "
                  "linearly in nnz, Gramians excluded" % ("-march=native" if native else "-march=x86-64-v3", take_u,
                                                           n_user, tu, zu, take_i, n_item, ti, zi),
        "user_rows_per_s": take_u / tu, "item_rows_per_s": take_i / ti,
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="SURVEY.md 8(d) configuration; 3 (default) is the one BASELINE.json's metric is quoted on")
    ap.add_argument("--users", type=int, default=None)
    ap.add_argument("--items", type=int, default=None)
    ap.add_argument("--rank", type=int, default=None)
    ap.add_argument("--solver", choices=("cg", "cholesky", "nnls"), default=None)
    ap.add_argument("--feedback", choices=("implicit", "explicit"), default=None)
    ap.add_argument("--mean-deg", type=float, default=50.0)
    ap.add_argument("--cg-steps", type=int, default=3)
    ap.add_argument("--lambda", dest="lam", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-launches", action="store_true",
                    help="issue the CG bucket kernels back to back instead of overlapping them on side streams "
                         "(use under rocprofv3 so that per-kernel durations are well defined)")
    ap.add_argument("--seed", type=int, default=20250222)
    args = ap.parse_args()
    for key, val in CONFIGS[args.config].items():
        if getattr(args, key) is None:
            setattr(args, key, val)
    implicit = args.feedback == "implicit"
    solver = {"cholesky": 0, "cg": 1, "nnls": 2}[args.solver]          # inst/include/wrmf.hpp:16-20 codes

    if args.serial_launches:
        os.environ["RSPARSE_HIP_CONCURRENT"] = "0"   # read by the library at its first CG launch
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if ws != args.gpus:
        if ws == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d ...`" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (ws, args.gpus))
    # RSPARSE_BENCH_BACKEND=gloo is a dry-run aid: all ranks share cuda:0 and collectives go through gloo, so the
    # N>1 control flow (sharding, padding, in-place all-gather) can be exercised on a single-GPU box.
    dry = os.environ.get("RSPARSE_BENCH_BACKEND", "nccl") == "gloo"
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if ws > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=ws)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=torch.device("cuda", local_rank))
    be = HipBackend(local_rank)
    dev = be.device
    k, lam = args.rank, args.lam

    # ---- synthetic data, generated on the device (identical on every rank: counter-based) ----
    t0 = time.perf_counter()
    data = synth.make_dataset(args.users, args.items, seed=args.seed, mean_deg=args.mean_deg, device=dev,
                              feedback=args.feedback)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    n_user, n_item, nnz = data["n_users"], data["n_items"], data["nnz"]
    Bu, ub, Bi, ib, _ = ShardedALS.partition(n_user, n_item, ws)
    c_ui_blk = shard_csc(*data["c_ui"], *ib[rank]) if ws > 1 else data["c_ui"]
    c_iu_blk = shard_csc(*data["c_iu"], *ub[rank]) if ws > 1 else data["c_iu"]
    als = ShardedALS(be, n_user, n_item, k, c_ui_blk, c_iu_blk, nnz, feedback=args.feedback, lambda_=lam,
                     cg_steps=args.cg_steps, world_size=ws, my_rank=rank)
    if not implicit:   # nnz per user / item: weights of the explicit regulariser (wrmf_explicit.hpp:160-170)
        als.cnt_user = torch.diff(data["c_iu"][0]).to(torch.float32)
        als.cnt_item = torch.diff(data["c_ui"][0]).to(torch.float32)
    if ws > 1:
        data = {"n_users": n_user, "n_items": n_item, "nnz": nnz}   # drop the full copies
        torch.cuda.empty_cache()
    # initial factors: U ~ N(0, 0.01^2); item factors zero for CG, N(0, 0.01^2) otherwise (R/model_WRMF.R:204-231)
    g = torch.Generator(device=dev).manual_seed(args.seed)
    U = als.alloc_factors(n_user, Bu, dev)
    V = als.alloc_factors(n_item, Bi, dev)
    U[:n_user] = torch.randn(n_user, k, generator=g, device=dev) * 0.01
    if solver != 1:
        V[:n_item] = torch.randn(n_item, k, generator=g, device=dev) * 0.01
    if solver == 2:                                   # NNLS: abs() of the initial factors (R/model_WRMF.R:252-255)
        U.abs_()
        V.abs_()

    def step(want_loss=True):
        li = als.half_iteration("items", U, V, solver, want_loss=want_loss)
        lu = als.half_iteration("users", U, V, solver, want_loss=want_loss)
        return li, lu

    def barrier():
        if ws > 1:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for _ in range(args.warmup):
        losses.append(step())
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step())
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if ws > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)

    # ---- per-kernel durations (HIP events inside the library, same stream), separate pass ----
    be.profile(True)
    nb = 6
    kern = {sd: {"bucket": [[] for _ in range(nb)], "gram": []} for sd in ("items", "users")}
    half_ms = {"items": [], "users": []}
    for _ in range(max(1, min(args.steps, 3))):
        for side in ("items", "users"):
            F, nF, BF, bF = (U, n_user, Bu, ub) if side == "items" else (V, n_item, Bi, ib)
            torch.cuda.synchronize()
            th = time.perf_counter()
            G, gm = None, [0.0, 0.0]
            if implicit:
                G = als.gramian(F, nF, BF, bF)
                gm = be.profile_last()
            als.half_iteration(side, U, V, solver, G=G, want_loss=True)
            pm = be.profile_last()
            torch.cuda.synchronize()
            half_ms[side].append(1e3 * (time.perf_counter() - th))
            kern[side]["gram"].append(gm[0] + gm[1])
            for b in range(nb):
                kern[side]["bucket"][b].append(pm[b])
    be.profile(False)
    mean = lambda v: float(np.mean(v)) if len(v) else 0.0
    info = {"users": als.csc_users.info(), "items": als.csc_items.info()}
    # per bucket: launches (one per half-iteration that has rows in it), mean duration, algorithmic bytes
    buckets = []
    kp = 32 if k <= 32 else (64 if k <= 64 else 128)
    tf_flag = "true" if implicit else "false"
    if solver != 1:
        # one kernel per half-iteration; no warm-start read (Cholesky ignores it): drop one N*k*4 from B_half
        ms = [mean(kern[sd]["bucket"][0]) for sd in ("items", "users")]
        by = [algorithmic_bytes(info[sd]["n_cols"], info[sd]["nnz"], k, info[sd]["n_empty"]) -
              (info[sd]["n_cols"] - info[sd]["n_empty"]) * 4 * k for sd in ("items", "users")]
        per_row = (k ** 3 / 3.0 + 2.0 * k * k) if solver == 0 else (2.0 * k ** 3 + 4.0 * k * k)
        fl = [2.0 * k * k * info[sd]["nnz"] + (info[sd]["n_cols"] - info[sd]["n_empty"]) * per_row
              for sd in ("items", "users")]
        buckets.append({"kernel": ("als_chol2_kernel<%d, %s, true>" if solver == 0 else "als_nnls_kernel<%d, %s, true>") % (kp, tf_flag),
                        "what": "one 256-thread workgroup per row: normal equations assembled in registers, " +
                                ("blocked Cholesky, two triangular solves" if solver == 0 else
                                 "squared in LDS, sequential coordinate descent on one wave (flops below count the "
                                 "assembly and the squaring, not the data-dependent sweeps)") +
                                " (compute/LDS bound, see roofline.compute)",
                        "launches_per_iteration": 2, "avg_launch_ms": float(np.mean(ms)),
                        "bytes_per_launch": float(np.mean(by)), "total_ms_per_iteration": float(np.sum(ms)),
                        "flops_per_launch": float(np.mean(fl))})
    for b in range(nb if solver == 1 else 0):
        wpr = int(info["users"]["bucket_wpr"][b])
        if wpr <= 0:
            continue
        ms = [mean(kern[sd]["bucket"][b]) for sd in ("items", "users") if info[sd]["bucket_rows"][b] > 0]
        by = [algorithmic_bytes(info[sd]["bucket_rows"][b], info[sd]["bucket_nnz"][b], k,
                                info[sd]["n_empty"] if b == nb - 1 or info["users"]["bucket_wpr"][min(b + 1, nb - 1)] <= 0 else 0)
              for sd in ("items", "users") if info[sd]["bucket_rows"][b] > 0]
        if ms:
            iu = info["users"]
            buckets.append({"kernel": "als_cgq_kernel<%d, %d, %d, %d, %d, %s>" % (kp, iu["bucket_capq"][b], iu["bucket_waves"][b], wpr, iu["bucket_stream"][b], tf_flag),
                            "what": "rows on teams of %d wave(s)%s" % (wpr, ", streamed (longer than the workgroup's resident capacity)" if iu["bucket_stream"][b] else ", register-resident"),
                            "launches_per_iteration": len(ms), "avg_launch_ms": float(np.mean(ms)),
                            "bytes_per_launch": float(np.mean(by)), "total_ms_per_iteration": float(np.sum(ms))})
    dom = max(buckets, key=lambda d: d["total_ms_per_iteration"]) if buckets else None
    solve_ms = sum(d["total_ms_per_iteration"] for d in buckets)
    solve_bytes = sum(d["bytes_per_launch"] * d["launches_per_iteration"] for d in buckets)
    b_iter = (algorithmic_bytes(info["users"]["n_cols"], info["users"]["nnz"], k, info["users"]["n_empty"]) +
              algorithmic_bytes(info["items"]["n_cols"], info["items"]["nnz"], k, info["items"]["n_empty"]))
    traffic = None
    tf = ROOT / "profiles" / "pmc_traffic.json"
    if tf.exists() and dom:
        try:
            traffic = json.loads(tf.read_text()).get(dom["kernel"])
        except Exception:
            traffic = None
    achieved = dom["bytes_per_launch"] / (dom["avg_launch_ms"] * 1e-3) / 1e9 if dom and dom["avg_launch_ms"] > 0 else 0.0
    roofline = {
        "bound": "hbm", "kernel": dom["kernel"] if dom else None, "achieved": achieved,
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
        "bytes_per_launch": dom["bytes_per_launch"] if dom else None,
        "avg_launch_ms": dom["avg_launch_ms"] if dom else None,
        "all_cg_kernels_GBps": solve_bytes / (solve_ms * 1e-3) / 1e9 if solve_ms > 0 else None,
        "whole_iteration_algorithmic_GBps_per_gpu": b_iter / (ms_per_step * 1e-3) / 1e9,
        "cg_kernels": buckets,
        "gramian_ms": {sd: mean(kern[sd]["gram"]) for sd in kern},
        "half_iteration_ms": {sd: mean(v) for sd, v in half_ms.items()},
    }
    if solver != 1 and dom:
        tfl = dom["flops_per_launch"] / (dom["avg_launch_ms"] * 1e-3) / 1e12
        roofline["compute"] = {"achieved": tfl, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / FP32_PEAK_TFLOPS,
                               "note": "the Cholesky solver is compute/LDS bound (about 2 k^2 n_i + k^3/3 flops per row "
                                       "against ~n_i (4k+8) bytes); the hbm figures above are reported as the contract asks"}

    cpu = None
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(data, U[:n_user], V[:n_item], k, lam, args.cg_steps, implicit=implicit, solver=solver,
                               target_s=12.0 if solver == 1 else 20.0)
        except Exception as e:   # the baseline is a report, never a reason to lose the measurement
            cpu = {"value": None, "unit": "iterations/s", "cores": len(os.sched_getaffinity(0)), "kind": "port",
                   "sample": "failed: %r" % (e,)}

    if rank == 0:
        user_half_ms = mean(half_ms["users"])
        line = {
            "metric": "als_iterations_per_sec", "value": args.steps / elapsed, "unit": "iterations/s",
            "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "WRMF %s ALS, %s, synthetic %dx%d, %d nnz, rank %d, lambda %g%s"
                                   % (args.feedback, "CG(%d)" % args.cg_steps if solver == 1 else ("Cholesky" if solver == 0 else "NNLS"), n_user, n_item,
                                      nnz, k, lam, "" if implicit else ", dynamic_lambda"),
                       "survey_config": args.config, "feedback": args.feedback,
                       "n_users": n_user, "n_items": n_item, "nnz": nnz, "rank": k,
                       "solver": {0: "cholesky", 1: "conjugate_gradient", 2: "nnls"}[solver],
                       "cg_steps": args.cg_steps, "parallelism": "rows sharded x%d, factors replicated" % ws},
            "user_rows_per_sec": n_user / (user_half_ms * 1e-3) if user_half_ms > 0 else None,
            "loss_users_last": losses[-1][1] if losses else None,
            "launch_mode": "serial" if os.environ.get("RSPARSE_HIP_CONCURRENT", "1") == "0" else
                           "overlapped (the per-kernel times under roofline are measured in a serialised pass)",
            "datagen_s": t_gen,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if ws > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
