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
from rsparse_amd.engine import HipBackend, ShardedALS  # noqa: E402

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


def physical_cores():
    """Physical cores this process may run on (SMT siblings counted once); falls back to the affinity count."""
    allowed = os.sched_getaffinity(0)
    seen = set()
    try:
        for c in allowed:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        return max(1, len(seen))
    except OSError:
        return len(allowed)


def cpu_model():
    """CPU model name of the host the baseline ran on (printed with the core count)."""
    try:
        names = {ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.lower().startswith("model name")}
        sockets = {ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.lower().startswith("physical id")}
        if names:
            return "%s%s" % ("%d x " % len(sockets) if len(sockets) > 1 else "", " / ".join(sorted(names)))
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def _sample_rows(csc, n_take, seed):
    """CSC of `n_take` random columns (seeded), p re-based; returns numpy (p, i, x) and the picked ids."""
    p, i, x = csc
    n = p.numel() - 1
    g = torch.Generator(device="cpu").manual_seed(seed)
    pick = torch.sort(torch.randperm(n, generator=g)[:n_take]).values.to(p.device)
    p64 = p.to(torch.int64)
    lens = p64[pick + 1] - p64[pick]
    off = torch.zeros(n_take + 1, dtype=torch.int64, device=p.device)
    torch.cumsum(lens, 0, out=off[1:])
    rid = torch.repeat_interleave(torch.arange(n_take, device=p.device), lens)
    pos = torch.arange(int(off[-1]), device=p.device) - off[rid] + p64[pick][rid]
    return (off.cpu().numpy().astype(np.int32), i[pos].cpu().numpy().astype(np.int32),
            x[pos].double().cpu().numpy(), pick)


def cpu_baseline(data, U, V, k, lam, cg_steps, target_s=12.0, implicit=True, solver=1, seed=1):
    """The CPU oracle (oracle/wrmf_oracle.cpp: the reference-shaped C++/OpenMP restatement) on the node's physical
    cores, as BASELINE.md 2 specifies it: fp64 (the reference's default precision, R/model_WRMF.R:82) and fp32,
    Gramian included, on a seeded RANDOM sample of the users and of the items of the same matrices; the iteration time
    is extrapolated linearly in nnz."""
    from oracle import wrmf_oracle as O
    threads = physical_cores()
    os.environ["OMP_NUM_THREADS"] = str(threads)
    try:
        O.lib(native=True)
        native = True
    except Exception:
        native = False
    n_user, n_item, nnz_tot = data["n_users"], data["n_items"], data["nnz"]
    res = {}
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        Vh = np.asfortranarray(V.cpu().numpy().T.astype(dt))       # k x n_item
        Uh = np.asfortranarray(U.cpu().numpy().T.astype(dt))       # k x n_user
        # Gramians: the item one in full, the user one on the leading 1M users (its cost is linear in the row count)
        t0 = time.perf_counter()
        Gv = O.gramian(Vh, lam, native=native)
        t_gram = time.perf_counter() - t0
        n_g = min(n_user, 1_000_000)
        t0 = time.perf_counter()
        Gu = O.gramian(np.asfortranarray(Uh[:, :n_g]), lam, native=native)
        t_gram += (time.perf_counter() - t0) * (n_user / n_g)
        if n_g < n_user:   # the solve needs the real one: fp64 on the device (plumbing, not the timed baseline)
            Gu = np.asfortranarray(((U.double().T @ U.double()).cpu().numpy() + float(np.float32(lam)) * np.eye(k)).astype(dt))

        def run(csc, X, G, Yfull, n_all, n_take):
            p, i, x, pick = _sample_rows(csc, n_take, seed)
            Y = np.asfortranarray(Yfull[:, pick.cpu().numpy()]).copy(order="F")
            t1 = time.perf_counter()
            if implicit:
                O.als_implicit(p, i, x, X, Y, G, lam, solver, cg_steps, n_threads=threads, native=native)
            else:
                O.als_explicit(p, i, x, X, Y, None, lam, solver, cg_steps, dynamic_lambda=True, n_threads=threads,
                               native=native)
            return time.perf_counter() - t1, int(p[-1])

        def sized(csc, X, G, Yfull, n_all, start):
            take = min(n_all, start)
            t, z = run(csc, X, G, Yfull, n_all, take)
            for _ in range(3):
                if t >= 0.5 * target_s / 4 or take >= n_all:
                    break
                take = int(min(n_all, take * min(8.0, max(1.5, (target_s / 4) / max(t, 1e-3)))))
                t, z = run(csc, X, G, Yfull, n_all, take)
            return take, t, z

        take_u, tu, zu = sized(data["c_iu"], Vh, Gv, Uh, n_user, 20000)
        take_i, ti, zi = sized(data["c_ui"], Uh, Gu, Vh, n_item, 2000)
        est = tu * (nnz_tot / max(zu, 1)) + ti * (nnz_tot / max(zi, 1)) + (t_gram if implicit else 0.0)
        res[name] = dict(value=1.0 / est, take_u=take_u, take_i=take_i, tu=tu, ti=ti, t_gram=t_gram,
                         user_rows_per_s=take_u / tu)
    main = res["f64"]
    return {
        "value": main["value"], "unit": "iterations/s", "cores": threads, "cpu_model": cpu_model(), "kind": "port", "dtype": "f64",
        "sample": "%d random users + %d random items, extrapolated in nnz; Gramians incl. (user one from 1M rows).  The extrapolation "
                  "FLATTERS the CPU: a whole user half-iteration of config 2 measured once on these cores took 1.97 x its sampled "
                  "estimate (the sample's factor rows stay in cache; profiles/r06/r6s/cpu_full_half_config2.json)"
                  % (main["take_u"], main["take_i"]),
        "value_f32": res["f32"]["value"], "user_rows_per_s": main["user_rows_per_s"],
        "user_rows_per_s_f32": res["f32"]["user_rows_per_s"], "gramians_s": main["t_gram"],
        "march": "native" if native else "x86-64-v3",
    }


def parity_check(als, U, V, solver, per_bucket=64, threads=None):
    """Full-size sampled parity of this very run (checker only, after the timed region): one more iteration in which
    ~64 rows per launch bucket and side are re-solved by the fp64 oracle from the same inputs."""
    from oracle import sampled_parity as SP
    threads = threads or physical_cores()
    reps = []
    for side in ("items", "users"):
        _, rep = SP.half_iteration_with_check(als, side, U, V, solver, per_bucket=per_bucket, seed=7, n_threads=threads)
        reps.append(rep)
    worst = max(reps, key=lambda r: r["max_row_err"])
    return {"rows_checked": sum(r["rows_checked"] for r in reps), "nnz_checked": sum(r["nnz_checked"] for r in reps),
            "max_row_err": worst["max_row_err"], "max_fro_err": max(r["fro_err"] for r in reps),
            "worst": {"side": worst["side"], "row_len": worst["worst_len"]}, "tolerance": 1e-4,
            "oracle": "fp64 restatement (oracle/wrmf_oracle.cpp), same inputs; ~%d rows per length class and side" % per_bucket}


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

    if os.environ.get("RSPARSE_BENCH_STACKS_AFTER"):   # debugging aid: every rank's Python stack after N seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["RSPARSE_BENCH_STACKS_AFTER"]), exit=True)
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if ws != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: re-launch the same command line under torch.distributed.run, one rank
            # per GPU of this node (the driver's own N > 1 launch sets WORLD_SIZE and never comes through here)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            os.execv(sys.executable, cmd)
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
    if args.serial_launches:
        be.set_launch_mode(0)
    dev = be.device
    k, lam = args.rank, args.lam

    # ---- synthetic data, generated on the device (counter-based: identical whatever rank generates a piece) ----
    t0 = time.perf_counter()
    if ws == 1:
        data = synth.make_dataset(args.users, args.items, seed=args.seed, mean_deg=args.mean_deg, device=dev,
                                  feedback=args.feedback)
        n_user, n_item, nnz = data["n_users"], data["n_items"], data["nnz"]
        lay_u, lay_i = ShardedALS.layouts(n_user, n_item, 1)
        c_ui_blk, c_iu_blk = data["c_ui"], data["c_iu"]
        cnt_user = torch.diff(data["c_iu"][0]).to(torch.float32)
        cnt_item = torch.diff(data["c_ui"][0]).to(torch.float32)
    else:
        # every rank generates only its own users' rows; the item blocks are exchanged (rsparse_amd/synth.py:make_shard); the blocks
        # are contiguous and balanced by non-zeros, so their row counts differ
        def bounds_fn(cu, ci):
            lu, li = ShardedALS.layouts(args.users, args.items, ws, cu, ci)
            return lu.bounds, (li.bounds if ci is not None else None)
        data = synth.make_shard(args.users, args.items, ws, rank, bounds_fn, seed=args.seed, mean_deg=args.mean_deg,
                                feedback=args.feedback, device=dev, be=be, group=None)
        n_user, n_item, nnz = data["n_users"], data["n_items"], data["nnz"]
        cnt_user, cnt_item = data["cnt_user"].to(torch.float32), data["cnt_item"].to(torch.float32)
        lay_u, lay_i = ShardedALS.layouts(n_user, n_item, ws, data["cnt_user"], data["cnt_item"])
        c_ui_blk, c_iu_blk = data["c_ui"], data["c_iu"]
        data = {"n_users": n_user, "n_items": n_item, "nnz": nnz}
        torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    als = ShardedALS(be, n_user, n_item, k, c_ui_blk, c_iu_blk, nnz, feedback=args.feedback, lambda_=lam,
                     cg_steps=args.cg_steps, world_size=ws, my_rank=rank, lay_user=lay_u, lay_item=lay_i)
    if not implicit:   # nnz per user / item: weights of the explicit regulariser (wrmf_explicit.hpp:160-170)
        als.cnt_user, als.cnt_item = cnt_user, cnt_item
    else:              # the confidences of a fit never change (R/model_WRMF.R:184-191): as WRMF.fit_transform tells the library
        als.freeze_values()
    shard_nnz = [int(c_iu_blk[1].numel()), int(c_ui_blk[1].numel())]
    # initial factors (in storage order; the values of a row do not depend on the number of ranks): U ~ N(0, 0.01^2);
    # item factors zero for CG, N(0, 0.01^2) otherwise (R/model_WRMF.R:204-231)
    g = torch.Generator(device=dev).manual_seed(args.seed)
    U = lay_u.from_global(lay_u.alloc(k, dev), torch.randn(n_user, k, generator=g, device=dev) * 0.01)
    V = lay_i.alloc(k, dev)
    if solver != 1:
        lay_i.from_global(V, torch.randn(n_item, k, generator=g, device=dev) * 0.01)
    if solver == 2:                                   # NNLS: abs() of the initial factors (R/model_WRMF.R:252-255)
        U.abs_()
        V.abs_()
    n_ranks_seen = 1
    if ws > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        n_ranks_seen = int(ones.item())

    def step():
        # both losses stay on the device until the iteration is over: one host sync per iteration (the R driver reads
        # the user-half loss for its convergence test, R/model_WRMF.R:327-335)
        li = als.half_iteration("items", U, V, solver, want_loss="device", defer_exchange=True)
        lu = als.half_iteration("users", U, V, solver, want_loss="device", defer_exchange=True)
        return float(li), float(lu)

    def barrier():
        als.finish()          # (the last exchange of the last half-iteration belongs to the timed region)
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
    # segments of rsparse_hip_profile_last: CG -> one per row-length bucket; Cholesky -> [0] normal-equation launch (long
    # rows, exact solve), [1] low-rank kernel (short rows), [2] k x k kernel, [3] wave-per-row kernel (rank 65..128, rows of 65..512 non-zeros);
    # NNLS -> [0].  The kernel NAMES come from the
    # library too (rsparse_hip_profile_last_names: the runtime's symbol table, i.e. what rocprofv3 prints).
    import re
    be.profile(True)
    nb = 6
    kern = {sd: {"bucket": [[] for _ in range(nb)], "gram": [], "names": [""] * nb} for sd in ("items", "users")}
    half_ms = {"items": [], "users": []}
    short = lambda full: (re.search(r"(\w+<[^()]*>)\s*\(", full) or re.search(r"(\w+)\s*\(", full) or [None, full])[1] if full else ""
    for _ in range(max(1, min(args.steps, 3))):
        for side in ("items", "users"):
            F, layF = (U, lay_u) if side == "items" else (V, lay_i)
            torch.cuda.synchronize()
            th = time.perf_counter()
            G, gm = None, [0.0, 0.0]
            if implicit:
                G = als.gramian(F, layF)
                gm = be.profile_last()
            als.half_iteration(side, U, V, solver, G=G, want_loss=True)
            pm = be.profile_last()
            nm = be.profile_last_names()
            torch.cuda.synchronize()
            half_ms[side].append(1e3 * (time.perf_counter() - th))
            kern[side]["gram"].append(gm[0] + gm[1])
            for b in range(nb):
                kern[side]["bucket"][b].append(pm[b])
                if b < len(nm) and nm[b]:
                    kern[side]["names"][b] = short(nm[b])
    be.profile(False)
    mean = lambda v: float(np.mean(v)) if len(v) else 0.0
    info = {"users": als.csc_users.info(), "items": als.csc_items.info()}
    sides = ("items", "users")
    buckets = []
    if solver == 1:
        # one launch per row-length bucket and half-iteration that has rows in it
        for b in range(nb):
            if int(info["users"]["bucket_wpr"][b]) <= 0:
                continue
            live = [sd for sd in sides if info[sd]["bucket_rows"][b] > 0]
            if not live:
                continue
            last = b == nb - 1 or info["users"]["bucket_wpr"][min(b + 1, nb - 1)] <= 0
            ms = [mean(kern[sd]["bucket"][b]) for sd in live]
            by = [algorithmic_bytes(info[sd]["bucket_rows"][b], info[sd]["bucket_nnz"][b], k, info[sd]["n_empty"] if last else 0)
                  for sd in live]
            name = kern[live[0]]["names"][b]
            ne = name.startswith("als_ne_kernel")
            mf = "cg_mf" in name
            wpr = int(info["users"]["bucket_wpr"][b])
            buckets.append({"kernel": name,
                            "what": "rows beyond 512 non-zeros: one pass, normal equations on the matrix cores (fp32 operands as "
                                    "2 fp16 / 3 bf16 split terms, fp32 accumulation), CG on the k x k system in LDS" if ne else
                                    "rows beyond 512 non-zeros: one WAVE per row, one pass, normal equations in the wave's matrix-core "
                                    "accumulator registers (fp32 operands as 2 fp16 split terms), CG from the tiles (wrmf_cg_mf.hip); the "
                                    "segment also covers the rows beyond 16384 non-zeros, split across workgroups by als_ne_kernel on a "
                                    "second stream, and its bytes count both" if mf else
                                    "rows on teams of %d wave(s), register-resident" % wpr,
                            "launches_per_iteration": len(ms), "avg_launch_ms": float(np.mean(ms)),
                            "bytes_per_launch": float(np.mean(by)), "total_ms_per_iteration": float(np.sum(ms))})
    else:
        # Cholesky / NNLS: which rows a launch takes (wrmf_capi.cpp run_half_iteration).  Length classes of csc_info:
        # [> 512, 257-512, 129-256, 65-128, 33-64, <= 32 (incl. empty)] non-zeros
        def rows_of(sd, seg):
            r, z = info[sd]["bucket_rows"], info[sd]["bucket_nnz"]
            emp = info[sd]["n_empty"]
            if solver == 2:
                return (sum(r), sum(z), emp) if seg == 0 else (0, 0, 0)
            ne_ok = k > 32 and k % 4 == 0
            lr_ok = implicit and 96 < k <= 128 and k % 2 == 0
            mf_ok = ne_ok and 64 < k <= 128         # round 6: the rows of 65..512 non-zeros on the wave-per-row kernel (segment 3)
            cut = 4 if (k > 64 and not mf_ok) else 1   # the normal-equation launch takes the classes [0, cut)
            if seg == 0:
                return (sum(r[:cut]), sum(z[:cut]), 0) if ne_ok else (0, 0, 0)
            if seg == 3:
                return (sum(r[1:4]), sum(z[1:4]), 0) if mf_ok else (0, 0, 0)
            lo = 4 if mf_ok else (cut if ne_ok else 0)
            if seg == 1:
                return (r[4] + r[5] - emp, z[4] + z[5], 0) if lr_ok else (0, 0, 0)
            hi = 4 if lr_ok else 6
            return (sum(r[lo:hi]) + (emp if lr_ok else 0), sum(z[lo:hi]), emp)
        what = {0: "normal-equation launch: rows beyond 512 non-zeros assembled on the matrix cores in one pass (long rows split across workgroups), LDL^T in the waves' registers",
                3: "wave-per-row kernel (rank 65..128): rows of 65..512 non-zeros assembled into the matrix-core accumulators, blocked Cholesky with the trailing updates on v_mfma_f32_32x32x2_f32",
                1: "low-rank form of the exact solve for rows of 1..64 non-zeros (Woodbury on XtX = L L^T; matrix cores)",
                2: "k x k kernel: normal equations in registers, blocked Cholesky, two triangular solves"}
        if solver == 2:
            what = {0: "one 256-thread workgroup per row: normal equations assembled in registers, squared in LDS, sequential "
                       "coordinate descent on one wave (flops count the assembly and the squaring, not the data-dependent sweeps)"}
        for seg in range(4 if solver == 0 else 1):
            live = [sd for sd in sides if rows_of(sd, seg)[0] > 0 and mean(kern[sd]["bucket"][seg]) > 0]
            if not live:
                continue
            ms = [mean(kern[sd]["bucket"][seg]) for sd in live]
            by, fl = [], []
            for sd in live:
                nr, nz, emp = rows_of(sd, seg)
                # no warm-start read (the exact solvers ignore it): drop one N*k*4 from B_half
                by.append(algorithmic_bytes(nr, nz, k, emp) - (nr - emp) * 4 * k)
                # nominal flops of the reference's formulation for these rows: lhs assembly 2 k^2 n_i, rhs 2 k n_i,
                # potrf k^3 / 3 + two triangular solves 2 k^2  (NNLS: the squaring 2 k^3 + 4 k^2 instead)
                per_row = (k ** 3 / 3.0 + 2.0 * k * k) if solver == 0 else (2.0 * k ** 3 + 4.0 * k * k)
                fl.append((2.0 * k * k + 2.0 * k) * nz + (nr - emp) * per_row)
            buckets.append({"kernel": kern[live[0]]["names"][seg], "what": what[seg],
                            "launches_per_iteration": len(ms), "avg_launch_ms": float(np.mean(ms)),
                            "bytes_per_launch": float(np.mean(by)), "total_ms_per_iteration": float(np.sum(ms)),
                            "flops_per_launch": float(np.mean(fl)),
                            "rows_per_launch": float(np.mean([rows_of(sd, seg)[0] for sd in live]))})
    dom = max(buckets, key=lambda d: d["total_ms_per_iteration"]) if buckets else None
    solve_ms = sum(d["total_ms_per_iteration"] for d in buckets)
    solve_bytes = sum(d["bytes_per_launch"] * d["launches_per_iteration"] for d in buckets)
    b_iter = (algorithmic_bytes(info["users"]["n_cols"], info["users"]["nnz"], k, info["users"]["n_empty"]) +
              algorithmic_bytes(info["items"]["n_cols"], info["items"]["nnz"], k, info["items"]["n_empty"]))
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure is the one
    # collected by tools/gpu_pmc_full.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) for THIS workload --
    # the table records the workload it was collected on; for any other workload the field is null
    this_workload = {"users": n_user, "items": n_item, "nnz": nnz, "rank": k, "feedback": args.feedback, "solver": args.solver,
                     "cg_steps": args.cg_steps, "n_gpus": ws}
    traffic, traffic_source = None, "no PMC table (profiles/pmc_traffic.json)"
    tf = ROOT / "profiles" / "pmc_traffic.json"
    if tf.exists() and dom:
        try:
            table = json.loads(tf.read_text())
            if table.get("workload") == this_workload:
                traffic = table.get("kernels", {}).get(dom["kernel"], {}).get("hbm_bytes_per_launch")
                traffic_source = {"file": "profiles/pmc_traffic.json", "collected_on": table["workload"],
                                  "how": table.get("how"), "kernel_found": traffic is not None}
            else:
                traffic_source = {"file": "profiles/pmc_traffic.json", "collected_on": table.get("workload"),
                                  "note": "collected on a different workload than this run: not applicable, traffic = null"}
        except Exception as e:
            traffic_source = "unreadable PMC table: %r" % (e,)
    achieved = dom["bytes_per_launch"] / (dom["avg_launch_ms"] * 1e-3) / 1e9 if dom and dom["avg_launch_ms"] > 0 else 0.0
    roofline = {
        "bound": "hbm", "kernel": dom["kernel"] if dom else None, "achieved": achieved,
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
        "traffic_source": traffic_source,
        "bytes_per_launch": dom["bytes_per_launch"] if dom else None,
        "avg_launch_ms": dom["avg_launch_ms"] if dom else None,
        "all_solve_kernels_GBps": solve_bytes / (solve_ms * 1e-3) / 1e9 if solve_ms > 0 else None,
        "whole_iteration_algorithmic_GBps_per_gpu": b_iter / (ms_per_step * 1e-3) / 1e9,
        "solve_kernels": buckets,
        "gramian_ms": {sd: mean(kern[sd]["gram"]) for sd in kern},
        "half_iteration_ms": {sd: mean(v) for sd, v in half_ms.items()},
    }
    if solver != 1 and dom:
        # the exact solvers are compute / LDS bound (about 2 k^2 n_i + k^3/3 flops per row against ~n_i (4k+8) bytes): the
        # fraction below is THAT launch's nominal flops over THAT launch's time, and the line's whole-iteration figure
        tfl = dom["flops_per_launch"] / (dom["avg_launch_ms"] * 1e-3) / 1e12
        all_fl = sum(d["flops_per_launch"] * d["launches_per_iteration"] for d in buckets)
        roofline["compute"] = {"kernel": dom["kernel"], "achieved": tfl, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tfl / FP32_PEAK_TFLOPS,
                               "whole_iteration_tflops": all_fl / (ms_per_step * 1e-3) / 1e12,
                               "whole_iteration_frac": all_fl / (ms_per_step * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                               "note": "nominal flops of the reference's formulation (assembly 2 k^2 n_i + potrf k^3/3 + solves) "
                                       "for the rows each launch takes, over that launch's own event-timed duration; the "
                                       "hbm figures above are reported as the contract asks"}

    comm_ms = None
    if ws > 1:   # exchange alone: the slab all-gathers of both sides, back to back (the timed steps overlap them with solves)
        comm_ms = {}
        for side, S, lay in (("items", V, lay_i), ("users", U, lay_u)):
            barrier()
            tc = time.perf_counter()
            for j in range(lay.n_sub):
                w = als._gather_slab(S, lay, j)
                if w is not None:
                    w.wait()
            torch.cuda.synchronize()
            comm_ms[side] = 1e3 * (time.perf_counter() - tc)
    first_ref = None
    ref_file = ROOT / "profiles" / "bench_n1_first_losses.json"
    cfg_key = "%dx%d_k%d_%s_%s_cg%d_seed%d" % (args.users, args.items, k, args.feedback, args.solver, args.cg_steps, args.seed)
    if ref_file.exists():
        try:
            first_ref = json.loads(ref_file.read_text()).get(cfg_key)
        except Exception:
            first_ref = None
    cpu, parity = None, None
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(data, U[:n_user], V[:n_item], k, lam, args.cg_steps, implicit=implicit, solver=solver,
                               target_s=12.0 if solver == 1 else 20.0)
        except Exception as e:   # the baseline is a report, never a reason to lose the measurement
            cpu = {"value": None, "unit": "iterations/s", "cores": physical_cores(), "kind": "port",
                   "sample": "failed: %r" % (e,)}
        try:   # checker leg: rows of this run re-solved by the fp64 oracle (after the timed region)
            parity = parity_check(als, U, V, solver)
        except Exception as e:
            parity = {"rows_checked": 0, "max_row_err": None, "error": repr(e)}

    transform = None
    if rank == 0 and ws == 1 and implicit and k <= 128:
        # WRMF$transform / the last step of every fit_transform (R/model_WRMF.R:359,412-452): one EXACT user half-iteration
        # against the final item factors, Gramian precomputed (private$XtX).  Not part of the timed iterations; last,
        # because it overwrites U
        try:
            Gt = als.gramian(V, lay_i).clone()
            tms = []
            for _ in range(2):
                torch.cuda.synchronize()
                tt = time.perf_counter()
                als.half_iteration("users", U, V, 0, G=Gt, want_loss=False)
                torch.cuda.synchronize()
                tms.append(1e3 * (time.perf_counter() - tt))
            be.check_numeric()
            transform = {"rows_per_sec": n_user / (min(tms) * 1e-3), "ms": min(tms), "rows": n_user, "solver": "cholesky",
                         "what": "WRMF$transform of every user: one exact user half-iteration from the final item factors"}
        except Exception as e:
            transform = {"rows_per_sec": None, "error": repr(e)}

    if rank == 0:
        user_half_ms = mean(half_ms["users"])
        line = {
            "metric": "als_iterations_per_sec", "value": args.steps / elapsed, "unit": "iterations/s",
            "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("f32 (Gram products of rows > 512 / <= 32 non-zeros as 2 x fp16 / 3 x bf16 split terms of the fp32 operands, "
                      "fp32 accumulate; everything else fp32)" if (solver == 1 and k > 32 and k % 4 == 0) else
                      ("f32 (normal equations of rows beyond 64 / 512 non-zeros assembled from 2 x fp16 / 3 x bf16 split terms, "
                       "fp32 accumulate; factorisations fp32)" if solver == 0 else "f32")),
            "data": "synthetic",
            "config": {"workload": "WRMF %s ALS, %s, synthetic %dx%d, %d nnz, rank %d, lambda %g%s"
                                   % (args.feedback, "CG(%d)" % args.cg_steps if solver == 1 else ("Cholesky" if solver == 0 else "NNLS"), n_user, n_item,
                                      nnz, k, lam, "" if implicit else ", dynamic_lambda"),
                       "survey_config": args.config, "feedback": args.feedback,
                       "n_users": n_user, "n_items": n_item, "nnz": nnz, "rank": k,
                       "solver": {0: "cholesky", 1: "conjugate_gradient", 2: "nnls"}[solver],
                       "cg_steps": args.cg_steps, "parallelism": "rows sharded x%d, factors replicated" % ws},
            "user_rows_per_sec": n_user / (user_half_ms * 1e-3) if user_half_ms > 0 else None,
            "transform": transform,
            "loss_users_last": losses[-1][1] if losses else None,
            "loss_first_iteration": list(losses[0]) if losses else None,
            # N > 1: the first iteration's losses against the committed N = 1 values of the same configuration (the
            # factors do not depend on the sharding; only the Gramian / loss summation order does)
            "loss_first_vs_n1_rel": (max(abs(a / b - 1.0) for a, b in zip(losses[0], first_ref))
                                     if (first_ref and losses) else None),
            "n_ranks_seen": n_ranks_seen, "shard_nnz_rank0": shard_nnz, "comm_ms": comm_ms,
            "partition": "contiguous blocks balanced by non-zeros, %d sub-blocks per rank and side (in-place slab all-gathers)" % lay_u.n_sub if ws > 1 else "single rank",
            "launch_mode": ("serial (--serial-launches)" if args.serial_launches else
                            "long-row launch alone, then the resident buckets on side streams (library default; the "
                            "per-kernel times under roofline are measured in a serialised pass)"),
            "datagen_s": t_gen,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(line))
    if ws > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
