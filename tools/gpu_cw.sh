timeout 900 python -m pytest tests/test_hip_parity.py tests/test_bias.py tests/test_wrmf_core.py -q -m gpu -k "chol or Chol or general_solver or singular or core" -p no:cacheprovider -x 2>&1 | tail -6
timeout 600 python -m pytest tests/test_sampled_parity.py -q -m gpu -k "config5" -p no:cacheprovider 2>&1 | tail -3
timeout 600 python bench.py --config 5 --solver cholesky --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('config5b it/s %.3f ms %.1f'%(d['value'],d['ms_per_step']), r['half_iteration_ms'], [(k['kernel'][:36], round(k['avg_launch_ms'],1)) for k in r['solve_kernels']])"
