#!/bin/bash
# round 4, after the wave-per-pass low-rank kernel: the whole GPU suite, the bench line (transform object), config 4 and 5b,
# kernel stats + SQ counters of one iteration of config 4.   tools/gpu_r4_lrw_final.sh TAG
TAG=${1:-r4w}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
rm -f gpurun_out/wrmf_core_errors.jsonl gpurun_out/sampled_parity_*.json
echo "== smoke" | tee $OUT/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -v "Warning\|warnings.warn\|^$\|model = WRMF\|WRMF(rank" $OUT/pytest_gpu.log | tail -6 >> $OUT/summary.txt
cp gpurun_out/wrmf_core_errors.jsonl $OUT/ 2>/dev/null; cp gpurun_out/sampled_parity_*.json $OUT/ 2>/dev/null
echo "== bench (default command)" | tee -a $OUT/summary.txt
timeout 1200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "rc=$?" | tee -a $OUT/summary.txt
python - $OUT/bench_full.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("it/s %.3f ms %.1f half %s" % (d["value"], d["ms_per_step"], r["half_iteration_ms"]))
print("transform", d.get("transform")); print("parity", d.get("parity"))
PY
echo "== config 4, config 5 with Cholesky" | tee -a $OUT/summary.txt
bash tools/gpu_configs.sh $TAG/cfg config4 config5_chol > /dev/null 2>&1
cat $OUT/cfg/summary.txt >> $OUT/summary.txt
echo "== rocprofv3 kernel stats of one iteration of config 4" | tee -a $OUT/summary.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o c4 -- python $REPO/bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline --serial-launches > $REPO/$OUT/prof_c4.json 2> $REPO/$OUT/prof_c4.err); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-170 | head -14; done >> $OUT/summary.txt 2>&1
find $OUT/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
echo "== SQ counters, config 4" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $REPO/$OUT/sq/sq -o p -- python $REPO/bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline > $REPO/$OUT/sq.log 2>&1); echo "sq rc=$?" | tee -a $OUT/summary.txt
python tools/pmc_summary.py $OUT/sq > $OUT/sq/summary.txt 2>&1
grep -A10 "als_chol_lrw_kernel\|als_ne_kernel.*true, false" $OUT/sq/summary.txt | cut -c1-200 | head -70 >> $OUT/summary.txt
find $OUT/sq -name "*kernel_trace.csv" -size +5M -delete
echo "== done" | tee -a $OUT/summary.txt
