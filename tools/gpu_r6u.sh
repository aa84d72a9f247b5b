#!/bin/bash
# round 6: wrmf_cg_mf.hip with the LDS-DMA ring: CG parity tests, the bench line, kernel-level timings (rocprofv3 --kernel-trace --stats)
TAG=${1:-r6u}
REPO=$PWD
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "cg or CG or scale or giant or norms or implicit" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 $OUT/pytest.log | cut -c1-300
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY' | tee $OUT/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("it/s %.3f ms %.1f  " % (d["value"], d["ms_per_step"]) + "  ".join("%.2f" % c["avg_launch_ms"] for c in d["roofline"]["solve_kernels"]) + "  parity %s" % (d.get("parity") or {}).get("max_row_err"))
except Exception as e:
    print("no json:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial-launches > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof_bench.err); echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-170 | head -24; done > $OUT/kernel_stats.txt 2>&1
cat $OUT/kernel_stats.txt
rm -rf $OUT/prof
