#!/bin/bash
# quick session: CG parity tests + small & full bench for the default configuration (+ optional env)
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x -k "not config2" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
tail -3 $OUT/pytest.log | cut -c1-300 >> $OUT/summary.txt
timeout 900 python bench.py ${BENCH_ARGS:---no-cpu-baseline} > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?" >> $OUT/summary.txt
python - <<PY >> $OUT/summary.txt 2>&1
import json
try:
    d=json.load(open("$OUT/bench_full.json"))
    r=d["roofline"]
    print("  full: it/s %.3f ms/step %.1f  half %s  frac %.3f dom %s" % (d["value"], d["ms_per_step"], {k: round(v,1) for k,v in r["half_iteration_ms"].items()}, r["frac"], r["kernel"]))
    for c in r.get("solve_kernels", []): print("   ", c["kernel"], "launches", c["launches_per_iteration"], "ms/iter %.2f" % c["total_ms_per_iteration"], "GB/s %.0f" % (c["bytes_per_launch"]/c["avg_launch_ms"]/1e6))
    print("   gram", r["gramian_ms"], "cpu", d.get("cpu_baseline"))
except Exception as e:
    print("  (no json)", e); print(open("$OUT/bench_full.err").read()[-800:])
PY
cat $OUT/summary.txt
