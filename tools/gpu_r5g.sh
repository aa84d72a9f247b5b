#!/bin/bash
# round 5, session g: the fp64 wave kernel with the transposed reduction (tests + ms per iteration), then the bench.py dry run at 2 and 8 ranks
TAG=${1:-r5g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_f64.py tests/test_fuzz.py -m gpu -q -p no:cacheprovider -k "double_half_iteration or baseline_ranks or f64_cg_wave or cg_steps or runs_in_double" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
tail -4 $OUT/pytest.log | cut -c1-300 >> $OUT/summary.txt
RSPARSE_TOOL_BUDGET_S=120 timeout 300 python tools/gpu_default_time.py double:128 double:96 double:72 2>&1 | grep rank >> $OUT/summary.txt
timeout 400 python tools/bench_dryrun_check.py --ranks 2 --users 200000 --items 20000 --timeout 180 > $OUT/dryrun2.txt 2>&1; echo "dryrun2 rc=$?" >> $OUT/summary.txt
tail -4 $OUT/dryrun2.txt | cut -c1-500 >> $OUT/summary.txt
timeout 700 python tools/bench_dryrun_check.py --ranks 8 --users 200000 --items 20000 --timeout 320 > $OUT/dryrun8.txt 2>&1; echo "dryrun8 rc=$?" >> $OUT/summary.txt
tail -6 $OUT/dryrun8.txt | cut -c1-500 >> $OUT/summary.txt
cat $OUT/summary.txt
