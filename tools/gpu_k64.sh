#!/bin/bash
# rank-64 geometry check: parity tests that touch k = 64 + configs 2 and 5
TAG=${1:-k64}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_sampled_parity.py tests/test_bias.py tests/test_wrmf_core.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/pytest.log >> $OUT/summary.txt
for c in 2 5; do
  timeout 600 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > $OUT/config$c.json 2> $OUT/config$c.err; echo "config$c rc=$?" >> $OUT/summary.txt
  python - $OUT/config$c.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("it/s %.3f ms %.2f half %s" % (d["value"], d["ms_per_step"], r["half_iteration_ms"]))
for kx in r["solve_kernels"]:
    print("  %-66s %6.2f ms x%d  %.0f GB/s" % (kx["kernel"], kx["avg_launch_ms"], kx["launches_per_iteration"], kx["bytes_per_launch"] / kx["avg_launch_ms"] / 1e6))
print("parity", d.get("parity"))
PY
done
cat $OUT/summary.txt
