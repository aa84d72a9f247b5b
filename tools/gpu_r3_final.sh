#!/bin/bash
# round 3, final-state evidence: smoke, the whole GPU suite, the bench line (+ rocprof stats of the same command, PMC table),
# the other configurations, $predict.   tools/gpu_r3_final.sh TAG
TAG=${1:-r3z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
rm -f gpurun_out/wrmf_core_errors.jsonl gpurun_out/sampled_parity_*.json
echo "== smoke" | tee $OUT/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -v "Warning\|warnings.warn\|^$\|model = WRMF\|WRMF(rank" $OUT/pytest_gpu.log | tail -12 >> $OUT/summary.txt
cp gpurun_out/wrmf_core_errors.jsonl $OUT/ 2>/dev/null; cp gpurun_out/sampled_parity_*.json $OUT/ 2>/dev/null
echo "== bench (default command)" | tee -a $OUT/summary.txt
timeout 1200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "rc=$?" | tee -a $OUT/summary.txt
python - $OUT/bench_full.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("it/s %.3f ms %.1f half %s" % (d["value"], d["ms_per_step"], r["half_iteration_ms"]))
print("dominant", r["kernel"], "frac %.3f" % r["frac"], "traffic", r["traffic"])
for kx in r["solve_kernels"]:
    print("  %-66s %6.2f ms x%d  %.0f GB/s" % (kx["kernel"], kx["avg_launch_ms"], kx["launches_per_iteration"], kx["bytes_per_launch"] / kx["avg_launch_ms"] / 1e6))
print("transform", d.get("transform")); print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k != "sample"}); print("parity", d.get("parity"))
PY
echo "== rocprofv3 kernel stats of the bench" | tee -a $OUT/summary.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial-launches > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof_bench.err); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-170 | head -14; done >> $OUT/summary.txt 2>&1
find $OUT/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
echo "== PMC" | tee -a $OUT/summary.txt
bash tools/gpu_pmc_full.sh $TAG/pmc > $OUT/pmc.log 2>&1; tail -14 $OUT/pmc/pmc_traffic.txt >> $OUT/summary.txt
echo "== other configurations" | tee -a $OUT/summary.txt
bash tools/gpu_configs.sh $TAG/cfg config2 config5 config5_chol config4 > /dev/null 2>&1
cat $OUT/cfg/summary.txt >> $OUT/summary.txt
echo "== predict" | tee -a $OUT/summary.txt
timeout 600 python tools/gpu_predict.py > $OUT/predict_1Mx1M.json 2> $OUT/predict.err; cat $OUT/predict_1Mx1M.json >> $OUT/summary.txt
timeout 600 python tools/gpu_predict.py --users 200000 --items 100000 --rank 64 > $OUT/predict_200kx100k_k64.json 2>> $OUT/predict.err; cat $OUT/predict_200kx100k_k64.json >> $OUT/summary.txt
timeout 600 python tools/gpu_predict.py --users 100000 --topk 100 --batch 50000 > $OUT/predict_100kx1M_top100.json 2>> $OUT/predict.err; cat $OUT/predict_100kx1M_top100.json >> $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_predict -o predict -- python $REPO/tools/gpu_predict.py --users 200000 > $REPO/$OUT/prof_predict.json 2> $REPO/$OUT/prof_predict.err); echo "rocprof predict rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_predict -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-170 | head -4; done >> $OUT/summary.txt 2>&1
find $OUT/prof_predict -name "*kernel_trace*" -size +5M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt
