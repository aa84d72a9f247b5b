#!/bin/bash
# A/B of the matrix-core dense product for the one-wave rows (RSPARSE_HIP_DENSE_MFMA): parity tests, then the bench line both ways
TAG=${1:-dmf}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_wrmf_core.py -m gpu -q -x --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 $OUT/pytest.log
fi
for m in 1 0; do
  RSPARSE_HIP_DENSE_MFMA=$m timeout 900 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_dmf$m.json 2> $OUT/bench_dmf$m.err; echo "bench dmf=$m rc=$?"
  python - $OUT/bench_dmf$m.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("it/s %.3f  ms %.1f  parity %s" % (d["value"], d["ms_per_step"], json.dumps(d.get("parity", {}))[:300]))
    for c in d["roofline"]["solve_kernels"]:
        print("   %-55s %7.2f ms  %6.0f GB/s" % (c["kernel"], c["avg_launch_ms"], c["bytes_per_launch"] / c["avg_launch_ms"] / 1e6))
except Exception as e:
    print("no json:", e)
PY
  tail -3 $OUT/bench_dmf$m.err
done
