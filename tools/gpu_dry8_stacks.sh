#!/bin/bash
# where do eight bench.py ranks on ONE GPU (gloo dry run) stop?  Python stacks of every rank after 40 s.
OUT=gpurun_out/dry8
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
RSPARSE_BENCH_BACKEND=gloo RSPARSE_BENCH_STACKS_AFTER=40 timeout 75 python bench.py --gpus 8 --users 200000 --items 20000 --rank 128 --steps 2 --warmup 0 --no-cpu-baseline > $OUT/stdout.txt 2> $OUT/stderr.txt
echo "rc=$?" > $OUT/summary.txt
grep -c "Thread\|Current thread" $OUT/stderr.txt >> $OUT/summary.txt
grep -A14 "Current thread\|most recent call first" $OUT/stderr.txt | grep "File" | sed 's/.*File "//' | awk '{print $1, $2, $3, $4, $5}' | sort | uniq -c | sort -rn | head -40 >> $OUT/summary.txt
tail -c 6000 $OUT/stderr.txt > $OUT/stderr_tail.txt
cat $OUT/summary.txt; cat $OUT/stdout.txt | cut -c1-300
