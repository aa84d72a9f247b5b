#!/bin/bash
# timing of the ablation builds (tools/build_abl.sh): per-phase cycle counters of the item half
for n in "$@"; do
  echo "== abl $n"
  RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_abl$n.so RSPARSE_NE_PROF=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 >/dev/null | grep "n_cols 1000000" | head -2 | cut -c1-200
done
