#!/bin/bash
# round 6: $predict with the candidate buffers in global memory (wrmf_topk.hip GBUF) -- the release against the round-5 geometry
# (`before`), and the global buffers for EVERY k (`gball`: -DRSP_TOPK_GBUF_ALWAYS) against the LDS buffers of the small k
TAG=${1:-r6topk_ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1
run() {  # label lib-suffix args...
  local label=$1 sfx=$2; shift 2
  if [ -z "$sfx" ]; then unset RSPARSE_HIP_LIB; else export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_$sfx.so; fi
  timeout 300 python tools/gpu_predict.py --rescore "$@" 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s users %7d topk %3d excl %2d  %.3f s  %7.0f users/s  %.1f TFLOP/s' % ('$label', d['users'], d['topk'], d['exclude_per_user'], d['seconds'], d['users_per_sec'], d['score_tflops']))" | tee -a $OUT/summary.txt
}
U=${USERS:-262144}
for k in 100 50 30 10 1; do
  run "release"            ""       --topk $k --users $U
  run "global buffers always" gball --topk $k --users $U
  run "round 5 geometry"   before   --topk $k --users $U
done
run "release, 1M users"  ""       --topk 100 --users 1000000
run "release, 1M users"  ""       --topk 10 --users 1000000
run "always, 1M users"   gball    --topk 10 --users 1000000
run "release, 200k users" ""      --topk 100 --users 200000
run "release, 100k-user calls" "" --topk 100 --users 400000 --batch 100000
