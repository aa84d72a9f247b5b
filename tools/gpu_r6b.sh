#!/bin/bash
# round 6: same-box A/B of the exact half-iteration with and without the wave-per-row kernel (wrmf_chol_mf.hip); -DRSP_AB build
TAG=${1:-r6b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_ab.so
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "chol or Chol or singular or general" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log
for rep in 1 2; do
for mf in 0 1; do
  RSPARSE_HIP_CHOL_MF=$mf timeout 600 python tools/gpu_chol_time.py 2>/dev/null | tail -1 | sed "s/^/mf=$mf /"
done
done | tee $OUT/chol_time_ab.txt
for mf in 0 1; do
  RSPARSE_HIP_CHOL_MF=$mf timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/config4_mf$mf.json 2> $OUT/config4_mf$mf.err
  python - $OUT/config4_mf$mf.json $mf <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("mf=%s it/s %.3f ms/step %.1f half %s parity %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in r["half_iteration_ms"].items()}, (d.get("parity") or {}).get("max_row_err")))
    for c in r["solve_kernels"]:
        print("   %-60s %.2f ms x %d" % (c["kernel"][:60], c["avg_launch_ms"], c["launches_per_iteration"]))
except Exception as e:
    print("mf", sys.argv[2], "(no json)", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done | tee $OUT/config4_ab.txt
