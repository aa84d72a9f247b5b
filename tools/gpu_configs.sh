#!/bin/bash
# SURVEY.md 8(d) configurations other than the bench line (config 3), one GPU each: tools/gpu_configs.sh TAG [names...]
TAG=${1:-cfgs}; shift
WHAT=${@:-config2 config5 config5_chol config4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/summary.txt
run() {
  name=$1; shift
  case " $WHAT " in *" $name "*) ;; *) return;; esac
  timeout 900 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?" >> $OUT/summary.txt
  python - $OUT/$name.json <<'PY' >> $OUT/summary.txt 2>&1
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("  %s" % d["config"]["workload"])
    print("  it/s %.3f ms/step %.1f half %s user_rows/s %.3g loss %.5f" % (d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in r["half_iteration_ms"].items()}, d["user_rows_per_sec"], d["loss_users_last"]))
    print("  dominant %s %.2f ms/launch  %.0f GB/s frac %.3f  compute %s" % (r["kernel"], r["avg_launch_ms"], r["achieved"], r["frac"], {k: v for k, v in (r.get("compute") or {}).items() if k != "note"}))
    print("  cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k != "sample"})
except Exception as e:
    print("  (no json)", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
run config2 --config 2 --steps 5 --warmup 1
run config5 --config 5 --steps 3 --warmup 1
run config5_chol --config 5 --solver cholesky --steps 2 --warmup 1 --no-cpu-baseline
run config4 --config 4 --steps 2 --warmup 1 --no-cpu-baseline
run config2_nnls --config 2 --solver nnls --steps 2 --warmup 1 --no-cpu-baseline
cat $OUT/summary.txt
