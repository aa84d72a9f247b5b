#!/bin/bash
# Cholesky kernel variants: parity tests + half-iteration time at 1M x 100k, k = 128 and 64
TAG=${1:-cv}
OUT=gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/summary.txt
for f in rsparse_amd/lib/variants/*.so; do
  v=$(basename $f .so)
  echo "== $v" >> $OUT/summary.txt
  RSPARSE_HIP_LIB=$PWD/$f timeout 600 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x -k "chol or golden or fit_transform" > $OUT/pytest_$v.log 2>&1
  echo "pytest rc=$?  $(tail -1 $OUT/pytest_$v.log)" >> $OUT/summary.txt
  RSPARSE_HIP_LIB=$PWD/$f timeout 600 python tools/gpu_chol_time.py 2>&1 | grep "^{" >> $OUT/summary.txt
done
cat $OUT/summary.txt
