#!/bin/bash
# round 6: the wave-per-row conjugate-gradient kernel for the long rows (wrmf_cg_mf.hip): CG parity tests, then the bench line
# with and without it on the same box (-DRSP_AB build, RSPARSE_HIP_CG_MF=0/1)
TAG=${1:-r6l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip_ab.so
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "cg or CG or scale or giant or norms or implicit" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $OUT/pytest.log | cut -c1-300
for rep in 1 2; do
for v in 0 1; do
  RSPARSE_HIP_CG_MF=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b_$v.$rep.json 2> $OUT/b_$v.$rep.err
  python - $OUT/b_$v.$rep.json $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("cg_mf=%s it/s %.3f ms %.1f  " % (sys.argv[2], d["value"], d["ms_per_step"]) + "  ".join("%.2f" % c["avg_launch_ms"] for c in d["roofline"]["solve_kernels"]) + "  parity %s" % (d.get("parity") or {}).get("max_row_err"))
    print("      ", d["roofline"]["solve_kernels"][0]["kernel"][:70])
except Exception as e:
    print(sys.argv[2], "no json:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
done | tee $OUT/summary.txt
