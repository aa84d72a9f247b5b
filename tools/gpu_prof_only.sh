#!/bin/bash
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial-launches > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof_bench.err); echo "rocprof rc=$?"
python tools/rocpd_summary.py $OUT/prof/bench_kernel_stats.csv | cut -c1-160 | head -12
python - <<PY
import json
d=json.load(open("$OUT/prof_bench.json")); r=d["roofline"]
print(d["launch_mode"], d["ms_per_step"]); [print(c["kernel"], round(c["avg_launch_ms"],2)) for c in r["cg_kernels"]]
PY
find $OUT/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
