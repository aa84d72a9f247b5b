#!/bin/bash
# per-kernel launch times of the bench line against the number of CG steps: the slope is the cost of a sweep, the intercept
# the gather / row set-up / store
TAG=${1:-cgsteps}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for n in 0 1 2 3; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cg-steps $n > $OUT/b_$n.json 2> $OUT/b_$n.err
  python - $OUT/b_$n.json $n <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("cg_steps %s  ms %.1f  " % (sys.argv[2], d["ms_per_step"]) + "  ".join("%.2f" % c["avg_launch_ms"] for c in d["roofline"]["solve_kernels"]))
except Exception as e:
    print(sys.argv[2], "no json:", e)
PY
done
