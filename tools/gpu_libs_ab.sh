#!/bin/bash
# same-box A/B of library builds on the bench line: tools/gpu_libs_ab.sh TAG suffix...   ("" = the shipped library)
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/rsparse_amd/lib/librsparse_wrmf_hip${v/main/}.so
  RSPARSE_HIP_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/b_$v.$rep.json 2> $OUT/b_$v.$rep.err
  python - $OUT/b_$v.$rep.json $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-8s it/s %.3f ms %.1f  " % (sys.argv[2], d["value"], d["ms_per_step"]) + "  ".join("%.2f" % c["avg_launch_ms"] for c in d["roofline"]["solve_kernels"]))
except Exception as e:
    print(sys.argv[2], "no json:", e)
PY
done
done
