#!/bin/bash
# per-row probe (tools/gpu_cg_probe.py) for every library variant under rsparse_amd/lib/variants/
TAG=${1:-pv}
OUT=gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/summary.txt
for f in rsparse_amd/lib/variants/*.so; do
  v=$(basename $f .so)
  echo "== $v" >> $OUT/summary.txt
  RSPARSE_HIP_LIB=$PWD/$f PROBE_L="${PROBE_L:-16 32 64 128 256 512}" PROBE_ROWS="${PROBE_ROWS:-4000000}" timeout 300 python tools/gpu_cg_probe.py $OUT/probe_$v.json 2>&1 | grep "^{" | python -c "
import sys, ast
print('  ' + '  '.join('L%d/%s %.3f' % (d['L'], 'c' if d['n_rows'] < 100000 else 'h', d['us_per_row_per_cu']) for d in map(ast.literal_eval, sys.stdin)))
" >> $OUT/summary.txt
done
cat $OUT/summary.txt
