#!/bin/bash
# round 5, session b: the GPU suite on the interleaved quad pass / reduce-scatter group sums / cgp KFULL, then a same-box A/B:
#   main = shipped; noilv = -DRSP_NO_QUAD_ILV; nogrs = -DRSP_NO_GRS; prio = -DRSP_GATHER_PRIO (s_setprio around the gather)
TAG=${1:-r5b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
tail -5 $OUT/pytest.log | cut -c1-400 >> $OUT/summary.txt
run() {  # name lib
  name=$1; lib=$2
  RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/$lib timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/b_$name.json 2> $OUT/b_$name.err
  python - $OUT/b_$name.json $name <<'PY' >> $OUT/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-10s it/s %.3f ms %.1f  " % (sys.argv[2], d["value"], d["ms_per_step"]) + "  ".join("%.2f" % c["avg_launch_ms"] for c in d["roofline"]["solve_kernels"]) + "  parity %s" % (d.get("parity", {}) or {}).get("max_row_err"))
except Exception as e:
    print(sys.argv[2], "no json:", e)
PY
}
for rep in 1 2; do
  for v in "" _noilv _nogrs _prio; do
    run main$v.$rep librsparse_wrmf_hip$v.so
  done
done
cat $OUT/summary.txt
