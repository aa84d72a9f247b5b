#!/bin/bash
# round 4, final-state evidence: smoke, the whole GPU suite, the bench line (+ rocprof stats of the same command, PMC table),
# the other configurations (NNLS with its CPU leg), $predict.   tools/gpu_r4_final.sh TAG
TAG=${1:-r4z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
rm -f gpurun_out/wrmf_core_errors.jsonl gpurun_out/sampled_parity_*.json
echo "== smoke" | tee $OUT/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
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
echo "== NNLS on config 2, with the CPU leg" | tee -a $OUT/summary.txt
timeout 600 python bench.py --config 2 --solver nnls --steps 3 --warmup 1 > $OUT/config2_nnls.json 2> $OUT/config2_nnls.err; echo "rc=$?" | tee -a $OUT/summary.txt
python -c "
import json; d=json.load(open('$OUT/config2_nnls.json')); c=d.get('cpu_baseline') or {}
print('  it/s %.3f ms %.1f  cpu f64 %.4f f32 %.4f it/s on %s cores (%s)  ->  %.1fx / %.1fx' % (d['value'], d['ms_per_step'], c.get('value',0), c.get('value_f32',0), c.get('cores'), c.get('cpu_model'), d['value']/max(c.get('value',1e-9),1e-9), d['value']/max(c.get('value_f32',1e-9),1e-9)))" >> $OUT/summary.txt 2>&1
echo "== SQ counters of the wave-per-row kernels (config 5 with Cholesky; config 2 with NNLS)" | tee -a $OUT/summary.txt
for what in "5 cholesky" "2 nnls"; do set -- $what
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $REPO/$OUT/sq_$2/sq -o p -- python $REPO/bench.py --config $1 --solver $2 --steps 1 --warmup 0 --no-cpu-baseline > $REPO/$OUT/sq_$2.log 2>&1); echo "sq $2 rc=$?" | tee -a $OUT/summary.txt
  python tools/pmc_summary.py $OUT/sq_$2 > $OUT/sq_$2/summary.txt 2>&1
  grep -B1 -A12 "als_chol_wave_kernel\|als_nnls_wave_kernel" $OUT/sq_$2/summary.txt | cut -c1-200 | head -32 >> $OUT/summary.txt
  find $OUT/sq_$2 -name "*kernel_trace.csv" -size +5M -delete
done
echo "== predict" | tee -a $OUT/summary.txt
for a in "--users 1000000 --items 1000000 --rank 128 --topk 10:predict_1Mx1M_top10" "--users 100000 --items 1000000 --rank 128 --topk 100:predict_100kx1M_top100" "--users 100000 --items 1000000 --rank 128 --topk 200:predict_100kx1M_top200" "--users 200000 --items 100000 --rank 64 --topk 10:predict_200kx100k_k64" "--users 1000000 --items 1000000 --rank 128 --topk 10 --batch 100000:predict_1Mx1M_top10_batches_of_100k"; do
  timeout 600 python tools/gpu_predict.py ${a%%:*} > $OUT/${a##*:}.json 2>> $OUT/predict.err
  python -c "
import json; d=json.load(open('$OUT/${a##*:}.json')); print('  %-40s users/s %.0f  TF %.1f  frac %.2f  match %.3f' % ('${a##*:}', d['users_per_sec'], d['score_tflops'], d['frac_of_fp32_peak'], d['scores_match_torch_topk_frac']))" >> $OUT/summary.txt 2>&1
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_predict -o predict -- python $REPO/tools/gpu_predict.py --users 262144 > $REPO/$OUT/prof_predict.json 2> $REPO/$OUT/prof_predict.err); echo "rocprof predict rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_predict -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-170 | head -4; done >> $OUT/summary.txt 2>&1
find $OUT/prof_predict -name "*kernel_trace*" -size +5M -delete 2>/dev/null
timeout 120 tools/probes/mfma_f32_rate_probe >> $OUT/summary.txt 2>&1
echo "== done" | tee -a $OUT/summary.txt
