#!/bin/bash
# phase ticks of the wave-per-row CG kernel (wrmf_cg_mf.hip) on the bench line: -DRSP_NE_PROF -DRSP_MF_PROF builds
# (tools/build_prof.sh with EXTRA=-DRSP_MF_PROF; timing-only ablations -DCGM_ABL=1 no matrix instructions, =2 only those)
#   tools/gpu_cgmf_prof.sh TAG [lib suffixes]
TAG=${1:-cgmfprof}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for sfx in "${@:-prof}"; do
  export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_$sfx.so
  RSPARSE_MF_PROF=1 timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_$sfx.json 2> $OUT/prof_$sfx.txt; echo "$sfx rc=$?"
  grep cgmf_prof $OUT/prof_$sfx.txt | tail -2
done | tee $OUT/summary.txt
