#!/bin/bash
# round 4, first session: NNLS on config 2 WITH the CPU leg; config 5 with Cholesky + the SQ counters of als_chol2_kernel<64>
TAG=${1:-r4a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 600 python bench.py --config 2 --solver nnls --steps 2 --warmup 1 > $OUT/config2_nnls.json 2> $OUT/config2_nnls.err; echo "nnls rc=$?"
timeout 600 python bench.py --config 5 --solver cholesky --steps 2 --warmup 1 --no-cpu-baseline > $OUT/config5_chol.json 2> $OUT/config5_chol.err; echo "c5chol rc=$?"
CMD="python $REPO/bench.py --config 5 --solver cholesky --steps 1 --warmup 0 --no-cpu-baseline"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $REPO/$OUT/sq -o p -- $CMD > $REPO/$OUT/sq.log 2>&1); echo "pass sq rc=$?"
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -B1 -A12 "als_chol2_kernel" $OUT/summary.txt | cut -c1-200 | head -60
find $OUT -name "*kernel_trace.csv" -size +5M -delete
python - <<PY
import json
for n in ("config2_nnls", "config5_chol"):
    try:
        d = json.load(open("gpurun_out/%s/%s.json" % ("$TAG", n)))
        print(n, d["value"], d["ms_per_step"], d.get("cpu_baseline"))
    except Exception as e:
        print(n, "no json", e)
PY
