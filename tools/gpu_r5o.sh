#!/bin/bash
# round 5, session o: the L2 prefetch of the next row's vectors (one- / two-wave kernels, <= 16 kernel): parity subset, same-box A/B
# against -DRSP_NO_L2PF, and the FETCH_SIZE pass (does the prefetch cost HBM bytes?)
TAG=${1:-r5o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_sampled_parity.py -m gpu -q -p no:cacheprovider -x -k "cg or short or dense or implicit or config3 or config2" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
tail -2 $OUT/pytest.log | cut -c1-200 >> $OUT/summary.txt
run() {
  name=$1; lib=$2
  RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/$lib timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/b_$name.json 2> $OUT/b_$name.err
  python - $OUT/b_$name.json $name <<'PY' >> $OUT/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-10s it/s %.3f ms %.1f  " % (sys.argv[2], d["value"], d["ms_per_step"]) + "  ".join("%.2f" % c["avg_launch_ms"] for c in d["roofline"]["solve_kernels"]))
except Exception as e:
    print(sys.argv[2], "no json:", e)
PY
}
for rep in 1 2; do
  run main.$rep librsparse_wrmf_hip.so
  run nol2.$rep librsparse_wrmf_hip_nol2.so
done
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/FETCH_SIZE -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $REPO/$OUT/FETCH_SIZE.log 2>&1); echo "pmc rc=$?" >> $OUT/summary.txt
python - $OUT <<'PY' >> $OUT/summary.txt 2>&1
import csv, glob, re, sys
from collections import defaultdict
v = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/FETCH_SIZE/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            m = re.search(r"((?:als)\w*<[^()]*>)\s*\(", r.get("Kernel_Name", ""))
            if m: v[m.group(1)].append(float(r["Counter_Value"]))
for k in sorted(v):
    if "cgq" in k or "cgp" in k: print("  FETCH 2x KB -> GB per launch: %-60s %.2f" % (k, 2 * sum(v[k]) / len(v[k]) * 1024 / 1e9))
PY
find $OUT -name "*kernel_trace.csv" -size +5M -delete
cat $OUT/summary.txt
