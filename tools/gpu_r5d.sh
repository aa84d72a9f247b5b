#!/bin/bash
# round 5, session d: the GPU suite with the double-rescored $predict and the fp64 conjugate-gradient wave kernel at ranks 65..128,
# then fit timings of precision = "double" against "float" at rank 128 / 96 on the config-2 shape
TAG=${1:-r5d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
tail -15 $OUT/pytest.log | cut -c1-600 >> $OUT/summary.txt
RSPARSE_TOOL_BUDGET_S=240 timeout 400 python tools/gpu_default_time.py double:128 float:128 double:96 float:96 > $OUT/f64_rank128_per_iteration.txt 2>&1
cat $OUT/f64_rank128_per_iteration.txt | tail -6 >> $OUT/summary.txt
timeout 300 python tools/gpu_predict.py 2>&1 | tail -3 > $OUT/predict.txt
timeout 300 python tools/gpu_predict.py --rescore 2>&1 | tail -3 >> $OUT/predict.txt
timeout 300 python tools/gpu_predict.py --rescore --topk 100 --users 200000 2>&1 | tail -3 >> $OUT/predict.txt
cat $OUT/predict.txt >> $OUT/summary.txt
cat $OUT/summary.txt
