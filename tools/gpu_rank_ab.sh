for lib in "_nopad" ""; do for r in 10 50 126; do
RSPARSE_HIP_LIB=$PWD/rsparse_amd/lib/librsparse_wrmf_hip$lib.so timeout 600 python bench.py --config 2 --rank $r --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib[$lib] config 2 at rank $r: it/s %.2f ms %.2f'%(d['value'],d['ms_per_step']))"
done; done
