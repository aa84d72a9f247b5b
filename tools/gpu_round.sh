#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, benches, rocprof kernel stats.  Everything is
# wrapped in `timeout`; logs land in gpurun_out/<tag>/ (merged back by gpurun).
TAG=${1:-r1a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== env" | tee $OUT/summary.txt
(rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep "Model name"; free -g | head -2) >> $OUT/summary.txt 2>&1
echo "== smoke" | tee -a $OUT/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/smoke.log >> $OUT/summary.txt
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -40 $OUT/pytest_gpu.log >> $OUT/summary.txt
fi
echo "== bench small (1M x 100k, k=128)" | tee -a $OUT/summary.txt
timeout 600 python bench.py --users 1000000 --items 100000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_small.json 2> $OUT/bench_small.err; echo "rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench_small.json >> $OUT/summary.txt; tail -5 $OUT/bench_small.err >> $OUT/summary.txt
if [ "${SKIP_FULL:-0}" != "1" ]; then
echo "== bench full (default)" | tee -a $OUT/summary.txt
timeout 1200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench_full.json >> $OUT/summary.txt; tail -5 $OUT/bench_full.err >> $OUT/summary.txt
echo "== rocprofv3 kernel stats of the full bench" | tee -a $OUT/summary.txt
REPO=$PWD
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial-launches > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof_bench.err); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*kernel_stats*" | head -3 | while read f; do echo "--- $f"; python tools/rocpd_summary.py "$f" | cut -c1-160 | head -10; done >> $OUT/summary.txt 2>&1
# keep only the small summaries (the trace itself can be large)
find $OUT/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
fi
echo "== done" | tee -a $OUT/summary.txt
