#!/bin/bash
# same-box A/B of two library builds on the bench line (release against rsparse_amd/lib/librsparse_wrmf_hip_<suffix>.so), CG parity tests first
TAG=${1:-r6ab2}; SFX=${2:-before}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "cg or CG or scale or giant or norms or implicit or every_row" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log | cut -c1-300
for rep in 1 2; do
for v in $SFX rel; do
  if [ $v = rel ]; then unset RSPARSE_HIP_LIB; else export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_$v.so; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b_$v.$rep.json 2> $OUT/b_$v.$rep.err
  python - $OUT/b_$v.$rep.json $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-7s it/s %.3f ms %.1f  " % (sys.argv[2], d["value"], d["ms_per_step"]) + "  ".join("%.2f" % c["avg_launch_ms"] for c in d["roofline"]["solve_kernels"]))
except Exception as e:
    print(sys.argv[2], "no json:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
done | tee $OUT/summary.txt
