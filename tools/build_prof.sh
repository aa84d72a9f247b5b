#!/bin/bash
# dev build with in-kernel cycle counters of the normal-equation kernel: RSPARSE_HIP_LIB=.../librsparse_wrmf_hip_prof.so RSPARSE_NE_PROF=1
cd $(dirname $0)/.. && python -m rsparse_amd.build -DRSP_NE_PROF -DRSP_AB ${EXTRA:-} --out rsparse_amd/lib/librsparse_wrmf_hip_prof.so
