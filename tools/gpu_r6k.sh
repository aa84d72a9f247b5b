#!/bin/bash
# round 6: how fast does the wave-per-row assembly (wrmf_chol_mf.hip) STREAM?  Every row beyond 64 non-zeros of config 3's matrix
# on it (RSPARSE_HIP_CHOL_MF_ALL, dev build), (a) the whole kernel, (b) assembly only (-DRSP_MF_ABL=7: results are garbage)
TAG=${1:-r6k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in ab abl7; do
  RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_$v.so RSPARSE_HIP_CHOL_MF_ALL=${MF_ALL:-1} timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/config4_$v.json 2> $OUT/config4_$v.err
  python - $OUT/config4_$v.json $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("%s it/s %.3f ms/step %.1f half %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in r["half_iteration_ms"].items()}))
    for c in r["solve_kernels"]:
        print("   %-60s %.2f ms x %d  bytes %.1f GB" % (c["kernel"][:60], c["avg_launch_ms"], c["launches_per_iteration"], c["bytes_per_launch"] / 1e9))
except Exception as e:
    print(sys.argv[2], "(no json)", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1200:])
PY
done | tee $OUT/summary.txt
