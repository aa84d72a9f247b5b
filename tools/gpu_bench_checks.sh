#!/bin/bash
# bench.py end to end: default N=1 line (cpu_baseline + parity), and the N=2 control flow through the driver's entry
# command (self-launch) with gloo on the single GPU (RSPARSE_BENCH_BACKEND=gloo dry run)
TAG=${1:-bc}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench default rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_full.json").read().strip().splitlines()[-1])
    print("it/s %.3f ms %.1f"%(d["value"],d["ms_per_step"]), "roofline", {k:d["roofline"][k] for k in ("kernel","achieved","frac","traffic")})
    print("cpu_baseline", d["cpu_baseline"]); print("parity", d["parity"]); print("first", d["loss_first_iteration"])
except Exception as e:
    print("parse failed", e); print(open("$OUT/bench_full.err").read()[-2000:])
PY
RSPARSE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err; echo "bench gloo n2 rc=$?"
timeout 600 python bench.py --gpus 1 --config 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_n1_c2.json 2> $OUT/bench_n1_c2.err; echo "bench n1 config2 rc=$?"
python - <<PY
import json
for f in ("$OUT/bench_n2_gloo.json","$OUT/bench_n1_c2.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "n_gpus",d["n_gpus"],"ranks_seen",d["n_ranks_seen"],"it/s %.2f"%d["value"],"first",d["loss_first_iteration"],"last",d["loss_users_last"],"comm",d["comm_ms"],"shard_nnz",d["shard_nnz_rank0"])
    except Exception as e:
        print(f,"parse failed",e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
