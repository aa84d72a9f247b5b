#!/bin/bash
# $predict after a change of wrmf_topk.hip: its tests, then throughput at the three reference shapes
TAG=${1:-r4p}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_top_product.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6
timeout 300 python tools/gpu_predict.py --users 1000000 --items 1000000 --rank 128 --topk 10 > $OUT/predict_1Mx1M_top10.json 2> $OUT/p1.err
timeout 300 python tools/gpu_predict.py --users 100000 --items 1000000 --rank 128 --topk 100 > $OUT/predict_100kx1M_top100.json 2> $OUT/p2.err
timeout 300 python tools/gpu_predict.py --users 200000 --items 100000 --rank 64 --topk 10 > $OUT/predict_200kx100k_k64.json 2> $OUT/p3.err
timeout 300 python tools/gpu_predict.py --users 100000 --items 1000000 --rank 128 --topk 200 > $OUT/predict_100kx1M_top200.json 2> $OUT/p4.err
for f in $OUT/predict_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('%-40s users/s %.0f  TF %.1f  frac %.2f  match %.3f' % ('$f'.split('/')[-1], d['users_per_sec'], d['score_tflops'], d['frac_of_fp32_peak'], d['scores_match_torch_topk_frac']))" || tail -3 ${f%.json}.err; done
