#!/bin/bash
# dev loop of the wave-per-pass low-rank kernel (wrmf_chol_lr.hip: als_chol_lrw_kernel): Cholesky parity tests, then the user half
# of config 4 through one exact half-iteration (tools/gpu_lr_abl.sh) for each library given ("" = the product library);
# SQ=1 adds the SQ counters of one iteration of config 4
TAG=${1:-lrw}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
REPO=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_bias.py -q -m gpu -k "chol or Chol or low_rank or packs or short or dispatch or singular or general_solver" -p no:cacheprovider -x 2>&1 | tail -15 > $OUT/tests.txt
tail -5 $OUT/tests.txt
bash tools/gpu_lr_abl.sh "$@" 2>&1 | grep "users half\|per launch\|lrw prof" | tee $OUT/time.txt
if [ -n "$SQ" ]; then
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $REPO/$OUT/sq -o p -- python $REPO/bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline > $REPO/$OUT/sq.log 2>&1)
  python tools/pmc_summary.py $OUT > $OUT/sq_summary.txt 2>&1
  grep -A10 "als_chol_lrw_kernel\|als_chol_lr_kernel" $OUT/sq_summary.txt | cut -c1-200 | head -30
  find $OUT -name "*kernel_trace.csv" -size +5M -delete
fi
