#!/bin/bash
# round 6: seconds per ALS iteration of WRMF fits in double and float for the exact solver and NNLS at the
# BASELINE ranks, 1M x 100k (tools/gpu_default_time.py)
TAG=${1:-r6f64}
OUT=gpurun_out/$TAG
mkdir -p $OUT
RSPARSE_TOOL_BUDGET_S=1500 timeout 1800 python tools/gpu_default_time.py float:64:cholesky double:64:cholesky float:128:cholesky double:128:cholesky float:64:nnls double:64:nnls 2>&1 | grep -v Warning | tee $OUT/f64_solvers_per_iteration.txt
