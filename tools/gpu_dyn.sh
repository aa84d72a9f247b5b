#!/bin/bash
# after a scheduling change of the static-list kernels: parity subsets, bench line, config 4
TAG=${1:-dyn}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_sampled_parity.py tests/test_bias.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -2 $OUT/pytest.log >> $OUT/summary.txt
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.txt
python - $OUT/bench.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("it/s %.3f ms %.2f half %s transform %s" % (d["value"], d["ms_per_step"], r["half_iteration_ms"], (d.get("transform") or {}).get("ms")))
for kx in r["solve_kernels"]:
    print("  %-66s %6.2f ms x%d  %.0f GB/s" % (kx["kernel"], kx["avg_launch_ms"], kx["launches_per_iteration"], kx["bytes_per_launch"] / kx["avg_launch_ms"] / 1e6))
print("parity", (d.get("parity") or {}).get("max_row_err"))
PY
bash tools/gpu_configs.sh $TAG/cfg config4 > /dev/null 2>&1
cat $OUT/cfg/summary.txt >> $OUT/summary.txt
cut -c1-230 $OUT/summary.txt
