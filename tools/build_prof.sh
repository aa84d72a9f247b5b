#!/bin/bash
# dev build with in-kernel cycle counters of the normal-equation kernel: RSPARSE_HIP_LIB=.../librsparse_wrmf_hip_prof.so RSPARSE_NE_PROF=1
cd $(dirname $0)/.. && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRSP_NE_PROF ${EXTRA:-} rsparse_amd/csrc/wrmf_kernels.hip rsparse_amd/csrc/wrmf_cgq.hip rsparse_amd/csrc/wrmf_ne.hip rsparse_amd/csrc/wrmf_chol.hip rsparse_amd/csrc/wrmf_topk.hip rsparse_amd/csrc/wrmf_ingest.hip rsparse_amd/csrc/wrmf_nnls.hip rsparse_amd/csrc/wrmf_bias.hip rsparse_amd/csrc/wrmf_capi.cpp -o rsparse_amd/lib/librsparse_wrmf_hip_prof.so
