#!/bin/bash
# round 6: kernel-level timings of the bench line with the wave-per-row CG kernel (rocprofv3 --kernel-trace --stats), and the
# dynamic-range test
TAG=${1:-r6m}
REPO=$PWD
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout=600 -p no:cacheprovider -k "norms" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -6 $OUT/pytest.log | cut -c1-300
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial-launches > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof_bench.err); echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-170 | head -24; done > $OUT/kernel_stats.txt 2>&1
cat $OUT/kernel_stats.txt
rm -rf $OUT/prof
