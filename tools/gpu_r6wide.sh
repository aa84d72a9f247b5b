#!/bin/bash
# round 6: conjugate gradient at the wide ranks on the wave-per-row kernel (wrmf_wide_cg.hip): wide-rank and bias tests, then timings per library
TAG=${1:-r6wide}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_wide_rank.py tests/test_bias.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "Warning\|warnings.warn\|^$" $OUT/pytest.log | tail -4 | cut -c1-300
for sfx in rel "$@"; do
  if [ "$sfx" = rel ]; then unset RSPARSE_HIP_LIB; else export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_$sfx.so; fi
  echo "library: $sfx"; timeout 1200 python tools/gpu_wide_time.py 2>&1 | grep "CG rank"
done | tee $OUT/summary.txt
