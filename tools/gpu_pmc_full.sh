#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the CG kernels on the full bench configuration (separate --pmc passes, kernel-trace only)
TAG=${1:-pmcfull}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/$OUT/$c -o p -- $CMD > $REPO/$OUT/$c.log 2>&1); echo "pass $c rc=$?"
done
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $REPO/$OUT/sq -o p -- $CMD > $REPO/$OUT/sq.log 2>&1); echo "pass sq rc=$?"
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A14 "als_cgq\|als_ne" $OUT/summary.txt | cut -c1-200 | head -120
find $OUT -name "*kernel_trace.csv" -size +5M -delete
python tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1; tail -12 $OUT/pmc_traffic.txt
