#!/bin/bash
# round 3, first session: baseline of this round's box, the reference-grid report (device vs fp32 oracle), perf evidence for
# $predict / NNLS / transform and the k = 64 configurations.   tools/gpu_r3a.sh TAG
TAG=${1:-r3a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
rm -f gpurun_out/wrmf_core_errors.jsonl
echo "== env" | tee $OUT/summary.txt
(rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep "Model name") >> $OUT/summary.txt 2>&1
echo "== core grid" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_wrmf_core.py -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_core.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -15 $OUT/pytest_core.log >> $OUT/summary.txt
cp gpurun_out/wrmf_core_errors.jsonl $OUT/ 2>/dev/null
timeout 600 python tools/core_trace.py > $OUT/core_trace.txt 2>&1; echo "trace rc=$?" | tee -a $OUT/summary.txt
echo "== bench baseline" | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "rc=$?" | tee -a $OUT/summary.txt
python - $OUT/bench_full.json >> $OUT/summary.txt 2>&1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("it/s %.3f ms %.1f half %s" % (d["value"], d["ms_per_step"], r["half_iteration_ms"]))
for kx in r["solve_kernels"]:
    print("  %-60s %6.2f ms x%d  %.0f GB/s" % (kx["kernel"], kx["avg_launch_ms"], kx["launches_per_iteration"], kx["bytes_per_launch"] / kx["avg_launch_ms"] / 1e6))
PY
echo "== predict" | tee -a $OUT/summary.txt
timeout 600 python tools/gpu_predict.py > $OUT/predict_1Mx1M.json 2> $OUT/predict.err; echo "rc=$?" | tee -a $OUT/summary.txt
cat $OUT/predict_1Mx1M.json >> $OUT/summary.txt
timeout 600 python tools/gpu_predict.py --exclude-deg 0 > $OUT/predict_1Mx1M_noexcl.json 2>> $OUT/predict.err
cat $OUT/predict_1Mx1M_noexcl.json >> $OUT/summary.txt
timeout 600 python tools/gpu_predict.py --users 200000 --items 100000 --rank 64 > $OUT/predict_200kx100k_k64.json 2>> $OUT/predict.err
cat $OUT/predict_200kx100k_k64.json >> $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_predict -o predict -- python $REPO/tools/gpu_predict.py --users 200000 > $REPO/$OUT/prof_predict.json 2> $REPO/$OUT/prof_predict.err); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_predict -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-200 | head -6; done >> $OUT/summary.txt 2>&1
find $OUT/prof_predict -name "*kernel_trace*" -size +5M -delete 2>/dev/null
echo "== other configs" | tee -a $OUT/summary.txt
bash tools/gpu_configs.sh $TAG/cfg config2 config5 config2_nnls > /dev/null 2>&1
cat $OUT/cfg/summary.txt >> $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt
