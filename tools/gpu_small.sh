#!/bin/bash
# the bench at 1M x 100k (fixed sides of 512 MB / 51 MB: the gathers hit MALL / L2 instead of HBM)
TAG=${1:-small}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python bench.py --users 1000000 --items 100000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/small.json 2> $OUT/small.err
python - <<PY
import json
d=json.loads(open("$OUT/small.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for kk in d["roofline"]["solve_kernels"]: print("  %-52s n=%d %.3f ms  %.2f GB/launch -> %.2f TB/s"%(kk["kernel"],kk["launches_per_iteration"],kk["avg_launch_ms"],kk["bytes_per_launch"]/1e9,kk["bytes_per_launch"]/kk["avg_launch_ms"]/1e9))
PY
