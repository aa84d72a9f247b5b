#!/bin/bash
# round 6, end state: smoke, the whole GPU suite, the bench line (default command) + rocprof stats of the same command, the PMC
# passes (HBM bytes per launch -> pmc_traffic.json), configs 2 / 5 / 5b / 4 / 2-NNLS, the 2- and 8-rank gloo dry runs of bench.py,
# $predict with the double re-scoring, fp64 fit timings.   tools/gpu_r6_end.sh TAG
TAG=${1:-r6z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
rm -f gpurun_out/wrmf_core_errors.jsonl gpurun_out/sampled_parity_*.json
echo "== smoke" | tee $OUT/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
grep -v "Warning\|warnings.warn\|^$\|model = WRMF\|WRMF(rank" $OUT/pytest_gpu.log | tail -4 >> $OUT/summary.txt
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
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-170 | head -18; done >> $OUT/summary.txt 2>&1
find $OUT/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
echo "== PMC passes (HBM bytes per launch)" | tee -a $OUT/summary.txt
bash tools/gpu_pmc_full.sh $TAG/pmc > $OUT/pmc.log 2>&1; tail -16 $OUT/pmc.log | cut -c1-200 >> $OUT/summary.txt
echo "== other configurations" | tee -a $OUT/summary.txt
bash tools/gpu_configs.sh $TAG/cfg config2 config5 config5_chol config4 config2_nnls > /dev/null 2>&1
cat $OUT/cfg/summary.txt >> $OUT/summary.txt
echo "== bench.py N = 2 / 8 dry run (gloo, one GPU)" | tee -a $OUT/summary.txt
for n in 2 8; do
  timeout 500 python tools/bench_dryrun_check.py --ranks $n --users 200000 --items 20000 --timeout 240 > $OUT/dryrun$n.txt 2>&1; echo "ranks $n rc=$?" | tee -a $OUT/summary.txt
  tail -3 $OUT/dryrun$n.txt | cut -c1-500 >> $OUT/summary.txt
done
echo "== \$predict" | tee -a $OUT/summary.txt
timeout 300 python tools/gpu_predict.py 2>&1 | grep "^{" > $OUT/predict_top10_fp32_pass.json
timeout 300 python tools/gpu_predict.py --rescore 2>&1 | grep "^{" > $OUT/predict_top10_rescored.json
timeout 300 python tools/gpu_predict.py --rescore --topk 100 --users 200000 2>&1 | grep "^{" > $OUT/predict_top100_rescored.json
cat $OUT/predict_*.json | cut -c1-420 >> $OUT/summary.txt
echo "== fp64 at the BASELINE ranks: ms per iteration inside WRMF.fit_transform, 1M x 100k" | tee -a $OUT/summary.txt
RSPARSE_TOOL_BUDGET_S=200 timeout 400 python tools/gpu_default_time.py double:128 float:128 double:64 float:64 double:10 float:10 2>&1 | grep "rank" > $OUT/f64_per_iteration.txt
cat $OUT/f64_per_iteration.txt >> $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
