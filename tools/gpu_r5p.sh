#!/bin/bash
# round 5, session p: the fp64 conjugate-gradient wave kernel with the transposed reduction at ranks 33..64 as well: parity, fit
# timings double / float, and the per-kernel split of a double fit (rocprofv3 kernel stats) at rank 64 and 128
TAG=${1:-r5p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
timeout 900 python -m pytest tests/test_f64.py tests/test_fuzz.py -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
grep -v "Warning\|warnings.warn\|^$\|model = WRMF\|WRMF(rank" $OUT/pytest.log | tail -6 | cut -c1-400 >> $OUT/summary.txt
RSPARSE_TOOL_BUDGET_S=200 timeout 400 python tools/gpu_default_time.py double:64 double:128 double:48 double:32 2>&1 | grep "rank" > $OUT/f64_per_iteration.txt
cat $OUT/f64_per_iteration.txt >> $OUT/summary.txt
for r in 128 64; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof$r -o fit -- python $REPO/tools/gpu_f64_fit.py $r 3 > $REPO/$OUT/fit$r.log 2>&1)
  echo "== rank $r, 3 iterations" >> $OUT/summary.txt
  find $OUT/prof$r -name "*kernel_stats*" | head -1 | while read f; do python tools/rocpd_summary.py "$f" | cut -c1-200 | head -10; done >> $OUT/summary.txt 2>&1
  find $OUT/prof$r -name "*kernel_trace*" | head -1 | while read f; do grep "f64_cg_wave" "$f" | awk -F, '{print $(NF-0)}' | head -0; done
  python - $OUT/prof$r >> $OUT/summary.txt 2>&1 <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace*.csv", recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "f64_cg_wave" in r["Kernel_Name"]]
    print("f64_cg_wave launches (ms):", " ".join("%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows))
PY
  find $OUT/prof$r -name "*kernel_trace*" -delete 2>/dev/null
done
cat $OUT/summary.txt
