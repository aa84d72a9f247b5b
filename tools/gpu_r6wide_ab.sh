#!/bin/bash
# round 6: the wide family's kernel (als_wide_kernel) with its gathers in flight and a 2-D trailing update -- tests, then the exact
# solver at the orders 132 / 160 / 256, release against `before`
TAG=${1:-r6wide_ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_wide_rank.py tests/test_bias.py -m gpu -q -x --timeout=900 -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/pytest_tail.txt
for v in before rel; do
  if [ $v = rel ]; then unset RSPARSE_HIP_LIB; else export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_$v.so; fi
  echo "== $v" | tee -a $OUT/summary.txt
  timeout 900 python tools/gpu_wide_chol_time.py 2>&1 | grep "rank" | tee -a $OUT/summary.txt
done
