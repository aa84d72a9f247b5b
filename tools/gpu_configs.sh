#!/bin/bash
# SURVEY.md 8(d) configurations other than the bench line (config 3), one GPU each
TAG=${1:-cfgs}
OUT=gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/summary.txt
run() {
  name=$1; shift
  timeout 900 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?" >> $OUT/summary.txt
  python - $OUT/$name.json <<'PY' >> $OUT/summary.txt 2>&1
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("  %s" % d["config"]["workload"])
    print("  it/s %.3f ms/step %.1f half %s user_rows/s %.3g loss %.5f" % (d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in r["half_iteration_ms"].items()}, d["user_rows_per_sec"], d["loss_users_last"]))
    print("  dominant %s %.2f ms/launch  %.0f GB/s frac %.3f  compute %s" % (r["kernel"], r["avg_launch_ms"], r["achieved"], r["frac"], r.get("compute")))
    print("  cpu", d.get("cpu_baseline"))
except Exception as e:
    print("  (no json)", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}

run config5 --config 5 --steps 3 --warmup 1
run config5_chol --config 5 --solver cholesky --steps 2 --warmup 1 --no-cpu-baseline

cat $OUT/summary.txt
