#!/bin/bash
# round 6: which phase of als_wide_kernel costs what at order 132 -- timing-only builds (-DRSP_WIDE_ABL=bits) against the release
OUT=gpurun_out/${1:-r6wide_abl}; mkdir -p $OUT
for v in rel wabl1 wabl2 wabl4 wabl8 wabl16 wabl32 wabl63; do
  if [ $v = rel ]; then unset RSPARSE_HIP_LIB; else export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_$v.so; fi
  echo -n "$v  " | tee -a $OUT/summary.txt
  timeout 300 python tools/gpu_wide_chol_time.py 0 2>&1 | grep rank | tee -a $OUT/summary.txt
done
