#!/bin/bash
# dev loop for the normal-equation kernel: parity subsets, then the bench line
TAG=${1:-ne}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout=300 -p no:cacheprovider -x -k "${KEXPR:-long_rows or implicit_cg_half or explicit_cg_half or independent}" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -${TAILN:-15} $OUT/pytest.log | cut -c1-300
if [ "${SKIP_SAMPLED:-0}" != "1" ]; then
timeout 900 python -m pytest tests/test_sampled_parity.py -m gpu -q --timeout=600 -p no:cacheprovider -k "${SEXPR:-config3 or config5}" > $OUT/sampled.log 2>&1; echo "sampled rc=$?"
tail -8 $OUT/sampled.log | cut -c1-400
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print("it/s %.3f ms %.1f loss %.6f"%(d["value"],d["ms_per_step"],d["loss_users_last"]))
    for kk in d["roofline"]["solve_kernels"]: print("  %-52s n=%d %.2f ms  %.1f GB/launch -> %.2f TB/s"%(kk["kernel"],kk["launches_per_iteration"],kk["avg_launch_ms"],kk["bytes_per_launch"]/1e9,kk["bytes_per_launch"]/kk["avg_launch_ms"]/1e9))
    print("  half ms", d["roofline"]["half_iteration_ms"], "gram", d["roofline"]["gramian_ms"])
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/bench.err").read()[-1500:])
PY
fi
