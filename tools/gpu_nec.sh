#!/bin/bash
# solver == CHOLESKY: length threshold of the normal-equation launch (RSPARSE_HIP_NE_CHOL_MIN) -- parity at a low threshold, then config 4 / 5b against it
TAG=${1:-nec}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
RSPARSE_HIP_NE_CHOL_MIN=96 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_sampled_parity.py -m gpu -q -x --timeout=500 -p no:cacheprovider -k "chol or Chol" 2>&1 | tail -3
for t in "$@"; do
  RSPARSE_HIP_NE_CHOL_MIN=$t timeout 300 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/c4_$t.json 2> $OUT/c4_$t.err
  python - $OUT/c4_$t.json $t <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("min %s: it/s %.3f ms %.1f half %s loss %.6f" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in r["half_iteration_ms"].items()}, d["loss_users_last"]))
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
done
