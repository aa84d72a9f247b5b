#!/bin/bash
# dev builds of the library with ablations of the normal-equation kernel's accumulate step (timing only, results are
# garbage): tools/build_abl.sh 1 2 3 ... -> rsparse_amd/lib/librsparse_wrmf_hip_abl<N>.so   (bits: see RSP_ABL in wrmf_ne.hip)
cd $(dirname $0)/..
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRSP_NE_PROF -DRSP_ABL=$n rsparse_amd/csrc/wrmf_kernels.hip rsparse_amd/csrc/wrmf_cgq.hip rsparse_amd/csrc/wrmf_ne.hip rsparse_amd/csrc/wrmf_chol.hip rsparse_amd/csrc/wrmf_chol_lr.hip rsparse_amd/csrc/wrmf_topk.hip rsparse_amd/csrc/wrmf_ingest.hip rsparse_amd/csrc/wrmf_nnls.hip rsparse_amd/csrc/wrmf_bias.hip rsparse_amd/csrc/wrmf_capi.cpp -o rsparse_amd/lib/librsparse_wrmf_hip_abl$n.so &
done
wait
