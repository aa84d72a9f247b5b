#!/bin/bash
# round 6: config 4 (exact solver) with the release library and with a dev library given by suffix (RSPARSE_HIP_LIB), same box
TAG=${1:-r6z}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for sfx in rel "$@"; do
  if [ "$sfx" = rel ]; then unset RSPARSE_HIP_LIB; else export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_$sfx.so; fi
  for rep in 1 2; do
  timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/config4_$sfx.json 2> $OUT/config4_$sfx.err
  python - $OUT/config4_$sfx.json $sfx <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("%s it/s %.3f ms/step %.1f half %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in r["half_iteration_ms"].items()}))
    for kx in r["solve_kernels"]:
        print("   %-60s %6.2f ms x %d" % (kx["kernel"][:60], kx["avg_launch_ms"], kx["launches_per_iteration"]))
except Exception as e:
    print(sys.argv[2], "(no json)", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
  done
done | tee $OUT/summary.txt
