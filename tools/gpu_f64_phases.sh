#!/bin/bash
# where a row of the generic fp64 kernel (exact solve: WRMF's closing `transform`, solver = "cholesky" in double) spends its time:
# dev build with s_memtime counters per phase (rsparse_amd/csrc/wrmf_f64.hip, RSP_F64_PROF), one fit of 1 iteration at rank 128 / 64
TAG=${1:-f64ph}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for r in 128 64; do
  RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_f64prof.so timeout 300 python tools/gpu_f64_fit.py $r 1 2>&1 | grep "phases\|ok" | sed "s/^/rank $r: /" | tee -a $OUT/summary.txt
done
