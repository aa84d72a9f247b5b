#!/bin/bash
# PMC counter passes (rocprofv3 --pmc, own runs, kernel-trace only) over a short bench run.
TAG=${1:-pmc}
CFG=${2:-1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
CMD="python $REPO/bench.py --users 1000000 --items 100000 --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -c "" $OUT/counters_list.txt
pass() {
  name=$1; shift
  (cd /tmp && RSPARSE_HIP_CGQ_CFG=$CFG timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $REPO/$OUT/$name -o p -- $CMD > $REPO/$OUT/$name.log 2>&1)
  echo "pass $name rc=$?"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAVES_EQ_64 SQ_INSTS_FLAT SQ_LDS_UNALIGNED_STALL
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcc2 FETCH_SIZE
pass tcc3 WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | cut -c1-250 | head -80
