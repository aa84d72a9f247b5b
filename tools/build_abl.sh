#!/bin/bash
# dev builds of the library with ablations of the normal-equation kernel's accumulate step (timing only, results are
# garbage): tools/build_abl.sh 1 2 3 ... -> rsparse_amd/lib/librsparse_wrmf_hip_abl<N>.so   (bits: see RSP_ABL in wrmf_ne.hip)
cd $(dirname $0)/..
for n in "$@"; do
  python -m rsparse_amd.build -DRSP_NE_PROF -DRSP_AB -DRSP_ABL=$n --out rsparse_amd/lib/librsparse_wrmf_hip_abl$n.so
done
