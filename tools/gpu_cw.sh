#!/bin/bash
# wave-per-row kernels (wrmf_chol_wave.hip, the wave kernel of wrmf_nnls.hip) after a change: their parity tests, config 5 with
# Cholesky and config 2 with NNLS
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_bias.py tests/test_nnls.py -q -m gpu -k "chol or Chol or general_solver or singular or nnls or NNLS" -p no:cacheprovider -x 2>&1 | tail -4
timeout 600 python -m pytest tests/test_sampled_parity.py -q -m gpu -k "config5 or config2" -p no:cacheprovider 2>&1 | tail -3
for what in "5 cholesky" "2 nnls"; do set -- $what
timeout 600 python bench.py --config $1 --solver $2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('config $1 $2 it/s %.3f ms %.1f'%(d['value'],d['ms_per_step']), r['half_iteration_ms'], [(k['kernel'][:36], round(k['avg_launch_ms'],1)) for k in r['solve_kernels']])"
done
