#!/bin/bash
# round 5, last session (the library as committed: fp32 path as in r5z, fp64 layer after the long-row / generic-kernel work):
# smoke, the whole GPU suite, fp64 fit timings + the per-kernel split of a rank-128 double fit, then the bench line (default command)
TAG=${1:-r5f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
echo "== smoke" | tee $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -v "Warning\|warnings.warn\|^$\|model = WRMF\|WRMF(rank" $OUT/pytest_gpu.log | tail -4 >> $OUT/summary.txt
echo "== fp64: ms per iteration inside WRMF.fit_transform, 1M x 100k, 5e7 non-zeros" | tee -a $OUT/summary.txt
RSPARSE_TOOL_BUDGET_S=90 timeout 200 python tools/gpu_default_time.py double:128 float:128 double:64 float:64 double:10 float:10 2>&1 | grep "rank" > $OUT/f64_per_iteration.txt
cat $OUT/f64_per_iteration.txt >> $OUT/summary.txt
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof128 -o fit -- python $REPO/tools/gpu_f64_fit.py 128 3 > $REPO/$OUT/fit128.log 2>&1)
echo "== rank 128 double, 3 iterations: kernels" >> $OUT/summary.txt
find $OUT/prof128 -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-200 | head -12; done >> $OUT/summary.txt 2>&1
find $OUT/prof128 -name "*kernel_trace*" -delete 2>/dev/null
echo "== bench (default command)" | tee -a $OUT/summary.txt
timeout 600 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "rc=$?" | tee -a $OUT/summary.txt
python - $OUT/bench_full.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("it/s %.3f ms %.1f half %s" % (d["value"], d["ms_per_step"], r["half_iteration_ms"]))
print("dominant", r["kernel"], "frac %.3f" % r["frac"], "traffic", r["traffic"])
for kx in r["solve_kernels"]:
    print("  %-66s %6.2f ms x%d  %.0f GB/s" % (kx["kernel"], kx["avg_launch_ms"], kx["launches_per_iteration"], kx["bytes_per_launch"] / kx["avg_launch_ms"] / 1e6))
print("parity", d.get("parity"))
PY
echo "== done" | tee -a $OUT/summary.txt
