#!/bin/bash
# in-kernel cycle counters of the long-row (normal-equation) launch on the bench line (dev build of tools/build_prof.sh)
TAG=${1:-neprof}
OUT=gpurun_out/$TAG; mkdir -p $OUT
RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_prof.so RSPARSE_NE_PROF_DUMP=$PWD/$OUT/wg.txt RSPARSE_NE_PROF=1 timeout 600 python bench.py ${BENCH_ARGS:-} --steps 1 --warmup 0 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
grep ne_prof $OUT/b.err | head -8 | cut -c1-700
