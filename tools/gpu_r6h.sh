#!/bin/bash
# round 6: the multi-GPU context's tests (SHARED transport on the one GPU) + the parity suite after the thread_local change
TAG=${1:-r6h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ctx.py -m gpu -q -x --timeout=600 -p no:cacheprovider > $OUT/pytest_ctx.log 2>&1; echo "ctx rc=$?"
tail -25 $OUT/pytest_ctx.log
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_nccl_single_rank.py tests/test_ingest.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest_parity.log 2>&1; echo "parity rc=$?"
tail -5 $OUT/pytest_parity.log
