#!/bin/bash
# tests (all, or PYTEST_ARGS) + a short bench line: tools/gpu_quick2.sh TAG
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
grep -v "Warning\|warnings.warn\|^$\|model = WRMF\|WRMF(rank" $OUT/pytest_gpu.log | tail -${TAILN:-25} >> $OUT/summary.txt
if [ "${SKIP_BENCH:-0}" != "1" ]; then
timeout 900 python bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - $OUT/bench.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("it/s %.3f ms %.1f half %s" % (d["value"], d["ms_per_step"], r["half_iteration_ms"]))
for kx in r["solve_kernels"]:
    print("  %-60s %6.2f ms x%d  %.0f GB/s" % (kx["kernel"], kx["avg_launch_ms"], kx["launches_per_iteration"], kx["bytes_per_launch"] / kx["avg_launch_ms"] / 1e6))
PY
tail -3 $OUT/bench.err >> $OUT/summary.txt
fi
cat $OUT/summary.txt
