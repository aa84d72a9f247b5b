#!/bin/bash
# in-kernel cycle counters of the normal-equation kernel's exact solve on config 4 (dev build of tools/build_prof.sh)
TAG=${1:-nechprof}
OUT=gpurun_out/$TAG; mkdir -p $OUT
RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_prof.so RSPARSE_NE_PROF=1 timeout 600 python bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/c4.json 2> $OUT/c4.err
grep ne_prof $OUT/c4.err | cut -c1-600
