#!/bin/bash
# build an experimental variant of the library: tools/build_variant.sh NAME [-DFLAG ...] -> rsparse_amd/lib/variants/NAME.so
# (select it at run time with RSPARSE_HIP_LIB=<path>)
NAME=$1; shift
mkdir -p rsparse_amd/lib/variants
C=rsparse_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" $C/wrmf_kernels.hip $C/wrmf_cgq.hip $C/wrmf_chol.hip $C/wrmf_topk.hip $C/wrmf_ingest.hip $C/wrmf_nnls.hip $C/wrmf_bias.hip $C/wrmf_capi.cpp -o rsparse_amd/lib/variants/$NAME.so
