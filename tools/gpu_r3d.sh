#!/bin/bash
# measurement objects after the bench.py rework: default line (cpu_baseline, parity, transform), rocprof stats, config 4, PMC table
TAG=${1:-r3d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 1200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/bench_full.err >> $OUT/summary.txt
python - $OUT/bench_full.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("it/s %.3f ms %.1f half %s dtype %s" % (d["value"], d["ms_per_step"], r["half_iteration_ms"], d["dtype"][:40]))
print("dominant", r["kernel"], "frac %.3f" % r["frac"], "traffic", r["traffic"], r["traffic_source"] if isinstance(r["traffic_source"], str) else r["traffic_source"].get("note", r["traffic_source"].get("kernel_found")))
for kx in r["solve_kernels"]:
    print("  %-66s %6.2f ms x%d  %.0f GB/s" % (kx["kernel"], kx["avg_launch_ms"], kx["launches_per_iteration"], kx["bytes_per_launch"] / kx["avg_launch_ms"] / 1e6))
print("transform", d.get("transform")); print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k != "sample"}); print("parity", d.get("parity"))
PY
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial-launches > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof_bench.err); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-170 | head -12; done >> $OUT/summary.txt 2>&1
find $OUT/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/config4.json 2> $OUT/config4.err; echo "config4 rc=$?" | tee -a $OUT/summary.txt
python - $OUT/config4.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("it/s %.3f ms %.1f half %s" % (d["value"], d["ms_per_step"], r["half_iteration_ms"]))
for kx in r["solve_kernels"]:
    print("  %-66s %7.2f ms x%d  %.1f TF  rows %.3g" % (kx["kernel"], kx["avg_launch_ms"], kx["launches_per_iteration"], kx["flops_per_launch"] / kx["avg_launch_ms"] / 1e9, kx["rows_per_launch"]))
print("compute", {k: v for k, v in r["compute"].items() if k != "note"}); print("transform", d.get("transform"))
PY
tail -3 $OUT/config4.err >> $OUT/summary.txt
bash tools/gpu_pmc_full.sh $TAG/pmc >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
