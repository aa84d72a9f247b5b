#!/bin/bash
# SQ counters of the Cholesky launches on config 4 (one iteration; kernel-trace only, one --pmc pass)
TAG=${1:-pmcchol}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
CMD="python $REPO/bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline"
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $REPO/$OUT/sq -o p -- $CMD > $REPO/$OUT/sq.log 2>&1); echo "pass sq rc=$?"
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -B1 -A12 "als_chol_lr_kernel\|als_ne_kernel.*true, false>" $OUT/summary.txt | cut -c1-200 | head -60
find $OUT -name "*kernel_trace.csv" -size +5M -delete
