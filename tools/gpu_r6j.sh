#!/bin/bash
# round 6: bench.py --gpus N as a dry run on ONE GPU (gloo), N = 2 and 8, after synth.make_shard stopped generating the whole
# matrix on every rank; one rank's generation at config-3 size alone
TAG=${1:-r6j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python tools/gpu_datagen_rank.py 3 8 > $OUT/datagen_rank3_of_8.json 2> $OUT/datagen.err; echo "datagen rc=$?"; cat $OUT/datagen_rank3_of_8.json
timeout 400 python tools/bench_dryrun_check.py --ranks 2 --users 200000 --items 20000 --timeout 180 > $OUT/dryrun2.txt 2>&1; echo "dryrun2 rc=$?"
tail -3 $OUT/dryrun2.txt | cut -c1-400
timeout 500 python tools/bench_dryrun_check.py --ranks 8 --users 200000 --items 20000 --timeout 240 > $OUT/dryrun8.txt 2>&1; echo "dryrun8 rc=$?"
tail -4 $OUT/dryrun8.txt | cut -c1-600
