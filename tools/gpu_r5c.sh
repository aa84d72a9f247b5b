#!/bin/bash
# round 5, session c: the GPU suite on the normal-equation kernel with sqrt(c - 1) travelling with the values and the
# right-hand side on wave 0, then a same-box A/B: main = shipped; rhs2 = -DRSP_NE_RHS_ROLE=2 (round 4's wave)
TAG=${1:-r5c}
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
  for v in "" _rhs2; do
    run main$v.$rep librsparse_wrmf_hip$v.so
  done
done
RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip.so timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/config4.json 2> $OUT/config4.err
python - $OUT/config4.json <<'PY' >> $OUT/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("config4 it/s %.3f ms %.1f" % (d["value"], d["ms_per_step"]), (d.get("parity") or {}).get("max_row_err"), d.get("transform"))
except Exception as e:
    print("config4 no json", e)
PY
cat $OUT/summary.txt
