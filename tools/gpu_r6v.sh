#!/bin/bash
# round 6: wrmf_cg_mf.hip, pipelined step: CG parity tests, then kernel-level timings (rocprofv3 --kernel-trace --stats) of the bench
# line for each library given:   tools/gpu_r6v.sh TAG [lib suffixes, "" = release]
TAG=${1:-r6v}; shift
REPO=$PWD
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "cg or CG or scale or giant or norms or implicit" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 $OUT/pytest.log | cut -c1-300
export TMPDIR=/tmp
for sfx in "${@:-rel}"; do
  if [ "$sfx" = rel ]; then unset RSPARSE_HIP_LIB; else export RSPARSE_HIP_LIB=$REPO/rsparse_amd/lib/librsparse_wrmf_hip_$sfx.so; fi
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_$sfx -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial-launches > $REPO/$OUT/prof_bench_$sfx.json 2> $REPO/$OUT/prof_bench_$sfx.err); echo "$sfx rocprof rc=$?"
  find $OUT/prof_$sfx -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-170 | head -12; done > $OUT/kernel_stats_$sfx.txt 2>&1
  grep -i "cg_mf\|kernel  " $OUT/kernel_stats_$sfx.txt
  python -c "
import json,sys
d=json.load(open('$OUT/prof_bench_$sfx.json')); print('$sfx serial-launch bench: ms %.1f parity %s' % (d['ms_per_step'], (d.get('parity') or {}).get('max_row_err')))"
  rm -rf $OUT/prof_$sfx
done
unset RSPARSE_HIP_LIB
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY' | tee $OUT/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("release: it/s %.3f ms %.1f  " % (d["value"], d["ms_per_step"]) + "  ".join("%.2f" % c["avg_launch_ms"] for c in d["roofline"]["solve_kernels"]) + "  parity %s" % (d.get("parity") or {}).get("max_row_err"))
except Exception as e:
    print("no json:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
