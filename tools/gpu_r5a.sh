#!/bin/bash
# round 5, session a: the GPU suite on the KFULL / LDS-DMA-prefetch kernels, then a same-box A/B of the bench line
#   main = shipped library; base = -DRSP_AB build with RSPARSE_HIP_KFULL=0 (the round-4 kernels); nodma = KFULL without the DMA prefetch
TAG=${1:-r5a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
tail -5 $OUT/pytest.log | cut -c1-400 >> $OUT/summary.txt
run() {  # name lib env...
  name=$1; lib=$2; shift 2
  env "$@" RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/$lib timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/b_$name.json 2> $OUT/b_$name.err
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
  run main.$rep librsparse_wrmf_hip.so A=1
  run base.$rep librsparse_wrmf_hip_ab.so RSPARSE_HIP_KFULL=0
  run nodma.$rep librsparse_wrmf_hip_nodma.so A=1
done
cat $OUT/summary.txt
