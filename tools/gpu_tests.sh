#!/bin/bash
# the whole GPU test-suite (or a -k selection): tools/gpu_tests.sh TAG [pytest args]
OUT=gpurun_out/${1:-t}; mkdir -p $OUT; shift
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -${TAILN:-25} $OUT/pytest.log | cut -c1-400
