#!/bin/bash
# A/B session for the CG kernel variants: parity tests + small bench per variant, full bench for the quad variants.
TAG=${1:-r1b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run_variant() {
  name=$1; shift
  echo "== variant $name ($*)" | tee -a $OUT/summary.txt
  env "$@" timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x -k "not config2" > $OUT/pytest_$name.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest_$name.log | cut -c1-300 >> $OUT/summary.txt
  env "$@" timeout 600 python bench.py --users 1000000 --items 100000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_small_$name.json 2> $OUT/bench_small_$name.err
  echo "bench small rc=$?" | tee -a $OUT/summary.txt
  python - <<PY >> $OUT/summary.txt 2>&1
import json
try:
    d=json.load(open("$OUT/bench_small_$name.json"))
    r=d["roofline"]
    print("  small: ms/step %.1f  half %s  frac %.3f" % (d["ms_per_step"], {k: round(v,1) for k,v in r["half_iteration_ms"].items()}, r["frac"]))
    for c in r.get("cg_kernels", []): print("   ", c["kernel"], "ms/iter %.2f" % c["total_ms_per_iteration"], "GB/s %.0f" % (c["bytes_per_launch"]/c["avg_launch_ms"]/1e6))
except Exception as e:
    print("  (no json)", e); print(open("$OUT/bench_small_$name.err").read()[-800:])
PY
}
: > $OUT/summary.txt
run_variant q0 RSPARSE_HIP_CGQ_CFG=0
run_variant q1 RSPARSE_HIP_CGQ_CFG=1
run_variant lds RSPARSE_HIP_CG=lds
for v in 0 1; do
  echo "== full bench cfg $v" | tee -a $OUT/summary.txt
  RSPARSE_HIP_CGQ_CFG=$v timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_full_q$v.json 2> $OUT/bench_full_q$v.err; echo "rc=$?" | tee -a $OUT/summary.txt
  python - <<PY >> $OUT/summary.txt 2>&1
import json
try:
    d=json.load(open("$OUT/bench_full_q$v.json"))
    r=d["roofline"]
    print("  full: it/s %.3f ms/step %.1f  half %s  frac %.3f dom %s" % (d["value"], d["ms_per_step"], {k: round(v,1) for k,v in r["half_iteration_ms"].items()}, r["frac"], r["kernel"]))
    for c in r.get("cg_kernels", []): print("   ", c["kernel"], "launches", c["launches_per_iteration"], "ms/iter %.2f" % c["total_ms_per_iteration"], "GB/s %.0f" % (c["bytes_per_launch"]/c["avg_launch_ms"]/1e6))
    print("   gram", r["gramian_ms"])
except Exception as e:
    print("  (no json)", e); print(open("$OUT/bench_full_q$v.err").read()[-800:])
PY
done
echo "== done" | tee -a $OUT/summary.txt
