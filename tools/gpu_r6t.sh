#!/bin/bash
# round 6, mid-round safety pass: smoke, the whole GPU suite, the default bench line.   tools/gpu_r6t.sh TAG
TAG=${1:-r6t}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -f gpurun_out/wrmf_core_errors.jsonl gpurun_out/sampled_parity_*.json
echo "== smoke" | tee $OUT/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 2700 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -v "Warning\|warnings.warn\|^$\|model = WRMF\|WRMF(rank" $OUT/pytest_gpu.log | tail -30 >> $OUT/summary.txt
cp gpurun_out/wrmf_core_errors.jsonl $OUT/ 2>/dev/null; cp gpurun_out/sampled_parity_*.json $OUT/ 2>/dev/null
echo "== bench (default command)" | tee -a $OUT/summary.txt
timeout 1200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "rc=$?" | tee -a $OUT/summary.txt
python - $OUT/bench_full.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("it/s %.3f ms %.1f half %s" % (d["value"], d["ms_per_step"], r["half_iteration_ms"]))
print("dominant", r["kernel"], "frac %.3f" % r["frac"], "traffic", r["traffic"])
for kx in r["solve_kernels"]:
    print("  %-66s %6.2f ms x%d  %.0f GB/s" % (kx["kernel"], kx["avg_launch_ms"], kx["launches_per_iteration"], kx["bytes_per_launch"] / kx["avg_launch_ms"] / 1e6))
print("transform", d.get("transform")); print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k != "sample"}); print("parity", d.get("parity"))
PY
cat $OUT/summary.txt
