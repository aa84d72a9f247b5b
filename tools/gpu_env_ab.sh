#!/bin/bash
# bench the full config under several environment settings: args "NAME=VAL[,NAME=VAL] ..." (use "-" for none)
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/summary.txt
for setting in "$@"; do
  envs=$(echo "$setting" | tr ',' ' ')
  [ "$setting" = "-" ] && envs=""
  name=$(echo "$setting" | tr -c 'A-Za-z0-9\n' '_')
  env $envs timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/b_$name.json 2> $OUT/b_$name.err
  python - <<PY >> $OUT/summary.txt 2>&1
import json
try:
    d=json.load(open("$OUT/b_$name.json")); r=d["roofline"]
    print("[$setting] ms/step %.1f half %s |" % (d["ms_per_step"], {k: round(v,1) for k,v in r["half_iteration_ms"].items()}), " | ".join("%.1f" % c["total_ms_per_iteration"] for c in r["cg_kernels"]))
except Exception as e:
    print("[$setting] failed", e, open("$OUT/b_$name.err").read()[-300:])
PY
done
cat $OUT/summary.txt
