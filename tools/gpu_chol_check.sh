#!/bin/bash
# exact-solve (Cholesky) check: parity tests, then configs 4 and 5 with Cholesky
TAG=${1:-cholchk}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_sampled_parity.py tests/test_bias.py tests/test_wrmf_core.py tests/test_abi.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/pytest.log >> $OUT/summary.txt
bash tools/gpu_configs.sh $TAG/cfg config4 config5_chol > /dev/null 2>&1
cat $OUT/cfg/summary.txt >> $OUT/summary.txt
cat $OUT/summary.txt | cut -c1-260
