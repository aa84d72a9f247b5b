#!/bin/bash
# Cholesky path after a kernel change: its parity tests, configs 4 / 5b, then the bench line (the CG kernels share the file)
TAG=${1:-chol}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_sampled_parity.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "chol or Chol or long or giant or split" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest.log
tools/gpu_configs.sh $TAG config4 config5_chol | cut -c1-250
SKIP_TESTS=1 STEPS=3 tools/gpu_dmf.sh $TAG 2>&1 | head -12
