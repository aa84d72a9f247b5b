#!/bin/bash
# round 6: $predict at a large k with the candidate buffers in global memory (wrmf_topk.hip GBUF): tests, then users/s at top-100 / 50 / 10
TAG=${1:-r6topk}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_top_product.py -m gpu -q -x --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log | cut -c1-300
for spec in "100 200000" "100 1000000" "50 400000" "30 400000" "10 1000000"; do
  set -- $spec
  timeout 300 python tools/gpu_predict.py --rescore --topk $1 --users $2 2>&1 | grep "^{" | cut -c1-460 | tee -a $OUT/predict.jsonl
done
