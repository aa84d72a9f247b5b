#!/bin/bash
TAG=${1:-abl}
OUT=gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/summary.txt
for ab in 0 1 2 3 4 16 7; do
  RSPARSE_HIP_ABLATE=$ab timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/b$ab.json 2> $OUT/b$ab.err
  python - <<PY >> $OUT/summary.txt 2>&1
import json
try:
    d=json.load(open("$OUT/b$ab.json")); r=d["roofline"]
    print("ablate=$ab ms/step %.1f half %s" % (d["ms_per_step"], {k: round(v,1) for k,v in r["half_iteration_ms"].items()}), " | ".join("%.1f" % c["total_ms_per_iteration"] for c in r["cg_kernels"]))
except Exception as e:
    print("ablate=$ab failed", e)
PY
done
cat $OUT/summary.txt
