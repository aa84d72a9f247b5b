#!/bin/bash
# round 6: chol tests, same-box A/B of the exact half-iteration at 1M x 100k, phase ticks, config 4 (mf on)
TAG=${1:-r6e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_ab.so
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "chol or Chol or singular or general" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log
for mf in 0 1; do
  RSPARSE_HIP_CHOL_MF=$mf timeout 600 python tools/gpu_chol_time.py 2>/dev/null | tail -1 | sed "s/^/mf=$mf /"
done | tee $OUT/chol_time_ab.txt
RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_prof.so RSPARSE_MF_PROF=1 timeout 600 python tools/gpu_chol_time.py 2>&1 >/dev/null | grep mf_prof | tail -3 | tee $OUT/mf_prof.txt
for mf in 1; do
  RSPARSE_HIP_CHOL_MF=$mf timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/config4_mf$mf.json 2> $OUT/config4_mf$mf.err
  python - $OUT/config4_mf$mf.json $mf <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("mf=%s it/s %.3f ms/step %.1f half %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in r["half_iteration_ms"].items()}))
    for c in r["solve_kernels"]:
        print("   %-60s %.2f ms x %d" % (c["kernel"][:60], c["avg_launch_ms"], c["launches_per_iteration"]))
except Exception as e:
    print("mf", sys.argv[2], "(no json)", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done | tee $OUT/config4_ab.txt
