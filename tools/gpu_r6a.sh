#!/bin/bash
# round 6, the wave-per-row exact solver of wrmf_chol_mf.hip: its parity tests, then the exact half-iteration timed at 1M x 100k
TAG=${1:-r6a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "chol or Chol or singular or general" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest.log
timeout 600 python tools/gpu_chol_time.py > $OUT/chol_time.txt 2>&1; echo "chol_time rc=$?"
cat $OUT/chol_time.txt | tail -5
