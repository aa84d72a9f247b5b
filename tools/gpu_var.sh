#!/bin/bash
# A/B of library variants (tools/build_variant.sh): full bench per (variant, cfg); CG parity for the candidates; probe.
TAG=${1:-var}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
show() {
python - "$1" <<'PY' >> $OUT/summary.txt 2>&1
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("  it/s %.3f ms/step %.1f half %s" % (d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in r["half_iteration_ms"].items()}))
    print("   " + " | ".join("%s %.1f" % (c["kernel"].replace("als_cgq_kernel", ""), c["total_ms_per_iteration"]) for c in r.get("cg_kernels", [])))
except Exception as e:
    print("  (no json)", e)
PY
}
for spec in ${VARIANTS:-base:0 pd:0 gv:0 pdgv:0 base:3 pdgv:3}; do
  v=${spec%%:*}; c=${spec##*:}
  echo "== $v cfg $c" >> $OUT/summary.txt
  RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/variants/$v.so RSPARSE_HIP_CGQ_CFG=$c timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_${v}_$c.json 2> $OUT/bench_${v}_$c.err
  show $OUT/bench_${v}_$c.json
done
for spec in ${TESTV:-pdgv:3}; do
  v=${spec%%:*}; c=${spec##*:}
  echo "== parity $v cfg $c" >> $OUT/summary.txt
  RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/variants/$v.so RSPARSE_HIP_CGQ_CFG=$c timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x -k "not config2" > $OUT/pytest_${v}_$c.log 2>&1
  echo "pytest rc=$?" >> $OUT/summary.txt; tail -3 $OUT/pytest_${v}_$c.log | cut -c1-300 >> $OUT/summary.txt
done
if [ "${PROBE:-1}" = "1" ]; then
  echo "== probe" >> $OUT/summary.txt
  timeout 600 python tools/gpu_cg_probe.py $OUT/cg_probe.json > $OUT/probe.log 2>&1; echo "probe rc=$?" >> $OUT/summary.txt
  cat $OUT/probe.log >> $OUT/summary.txt
fi
cat $OUT/summary.txt
