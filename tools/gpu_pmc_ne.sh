#!/bin/bash
# SQ / TCC counters of the normal-equation kernel on the bench configuration (separate --pmc passes, kernel-trace only)
TAG=${1:-pmcne}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --serial-launches ${BENCH_ARGS:-}"
run() { n=$1; shift; (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "als_ne" --output-format csv -d $REPO/$OUT/$n -o p -- $CMD > $REPO/$OUT/$n.log 2>&1); echo "pass $n rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_WAVES
if [ "${SKIP_TCC:-0}" != "1" ]; then run fetch FETCH_SIZE; run write WRITE_SIZE; fi
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]; short=n[n.find("als_ne"):][:60]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]): print("   %-28s n=%d mean %.4g  vals %s"%(c,len(agg[k][c]),sum(agg[k][c])/len(agg[k][c]),["%.3g"%v for v in agg[k][c][:4]]))
PY
find $OUT -name "*kernel_trace.csv" | head -2 | while read f; do python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    print(r["Kernel_Name"][-70:], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "ms", "vgpr", r.get("VGPR_Count"), "accum", r.get("Accum_VGPR_Count"), "lds", r.get("LDS_Block_Size"), "scratch", r.get("Scratch_Size"), "grid", r.get("Grid_Size"))
PY
done
