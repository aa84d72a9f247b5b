#!/bin/bash
# phase ticks of the wave-per-row NNLS kernel (wrmf_nnls.hip) on config 2: -DRSP_NE_PROF -DRSP_NNLS_PROF build (tools/build_prof.sh with EXTRA=-DRSP_NNLS_PROF)
TAG=${1:-nnlsprof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_prof.so
RSPARSE_NNLS_PROF=1 timeout 900 python bench.py --config 2 --solver nnls --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/prof.txt; echo rc=$?
grep nnls_prof $OUT/prof.txt | tail -8 | tee $OUT/summary.txt
