#!/bin/bash
# phase ticks of the wave-per-row exact solver (wrmf_chol_mf.hip) at 1M x 100k, rank 128: -DRSP_NE_PROF -DRSP_MF_PROF build
TAG=${1:-mfprof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_prof.so
RSPARSE_MF_PROF=1 timeout 600 python tools/gpu_chol_time.py > $OUT/chol_time.txt 2> $OUT/mf_prof.txt; echo rc=$?
tail -1 $OUT/chol_time.txt
grep mf_prof $OUT/mf_prof.txt | tail -6
