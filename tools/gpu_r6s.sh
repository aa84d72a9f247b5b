#!/bin/bash
# round 6: (a) one full-size CPU half-iteration next to its sampled extrapolation, (b) double CG fits at config-2 and
# config-3 size (item 5d)
TAG=${1:-r6s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python tools/gpu_cpu_full_half.py > $OUT/cpu_full_half_config2.json 2> $OUT/cpu_full_half.err; echo "cpu rc=$?"; cat $OUT/cpu_full_half_config2.json | head -20
timeout 600 python tools/gpu_f64_config_time.py 2 > $OUT/f64_config2.txt 2>&1; echo "f64 config2 rc=$?"; tail -1 $OUT/f64_config2.txt | cut -c1-600
timeout 1200 python tools/gpu_f64_config_time.py 3 > $OUT/f64_config3.txt 2>&1; echo "f64 config3 rc=$?"; tail -2 $OUT/f64_config3.txt | cut -c1-600
