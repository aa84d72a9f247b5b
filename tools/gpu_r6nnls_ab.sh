#!/bin/bash
# round 6: the NNLS sweep with the moving lane written by number (v_writelane_b32) -- tests, then config 2 with solver = nnls, release against `before`
TAG=${1:-r6nnls_ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_nnls.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
for v in before rel; do
  if [ $v = rel ]; then unset RSPARSE_HIP_LIB; else export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_$v.so; fi
  timeout 600 python bench.py --config 2 --solver nnls --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b_$v.$rep.json 2> $OUT/b_$v.$rep.err
  python - $OUT/b_$v.$rep.json $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-7s it/s %.3f ms %.1f  loss %s" % (sys.argv[2], d["value"], d["ms_per_step"], d.get("loss")))
except Exception as e:
    print(sys.argv[2], "no json:", e)
PY
done
done | tee $OUT/summary.txt
