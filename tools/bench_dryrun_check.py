#!/usr/bin/env python3
"""bench.py's N > 1 control flow on ONE GPU: the same reduced-size workload at N = 1 and at N = 8 with
RSPARSE_BENCH_BACKEND=gloo (all ranks share cuda:0, collectives go through gloo: sharding, sub-block storage, in-place slab
all-gathers, the fused Gramian exchange and the deferred last wait are the production code), then
  * n_ranks_seen == 8,
  * the first iteration's losses of the two runs agree to 1e-9 relative (the factors of a row do not depend on the number of
    ranks; only the order of the Gramian / loss sums does).
Not a scaling measurement: eight processes share one device.   python tools/bench_dryrun_check.py [--ranks 8] [--users N] [--items M]"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--users", type=int, default=400_000)
ap.add_argument("--items", type=int, default=40_000)
ap.add_argument("--rank", type=int, default=128)
ap.add_argument("--timeout", type=int, default=420, help="seconds per bench.py run")
a = ap.parse_args()
common = ["--users", str(a.users), "--items", str(a.items), "--rank", str(a.rank), "--steps", "2", "--warmup", "0", "--no-cpu-baseline"]


def run(n, env_extra):
    env = dict(os.environ, **env_extra)
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n)] + common, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=a.timeout)
    except subprocess.TimeoutExpired as e:
        raise SystemExit("bench.py --gpus %d did not finish in %d s; stderr tail:\n%s" % (n, a.timeout, (e.stderr or b"")[-3000:]))
    print("bench.py --gpus %d: %.0f s" % (n, time.perf_counter() - t0), flush=True)
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise SystemExit("bench.py --gpus %d failed (rc %d):\n%s" % (n, r.returncode, r.stderr[-3000:]))
    return json.loads(lines[-1])


one = run(1, {})
many = run(a.ranks, {"RSPARSE_BENCH_BACKEND": "gloo"})
rel = max(abs(x / y - 1.0) for x, y in zip(many["loss_first_iteration"], one["loss_first_iteration"]))
out = {"ranks": a.ranks, "n_ranks_seen": many["n_ranks_seen"], "loss_first_iteration_n1": one["loss_first_iteration"],
       "loss_first_iteration_nN": many["loss_first_iteration"], "loss_first_rel_diff": rel, "partition": many["partition"],
       "shard_nnz_rank0": many["shard_nnz_rank0"], "workload": many["config"]["workload"]}
print(json.dumps(out))
assert many["n_ranks_seen"] == a.ranks, out
assert rel <= 1e-9, out
print("ok")
