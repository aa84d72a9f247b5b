#!/bin/bash
# round 6: the global-bias CG kernels take the KFULL instantiations at rank == padded rank: bias tests, then ms per iteration per library
TAG=${1:-r6gb}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_bias.py tests/test_wrmf_core.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "Warning\|warnings.warn\|^$" $OUT/pytest.log | tail -3 | cut -c1-300
for sfx in rel "$@"; do
  if [ "$sfx" = rel ]; then unset RSPARSE_HIP_LIB; else export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_$sfx.so; fi
  echo "library: $sfx"; timeout 900 python tools/gpu_gb_time.py 2>&1 | grep "^rank"
done | tee $OUT/summary.txt
