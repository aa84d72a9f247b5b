#!/bin/bash
# 2 ranks sharing the one GPU of the box, gloo collectives: control-flow check of bench.py's N>1 path (not a measurement)
TAG=${1:-dry2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
RSPARSE_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --users 200000 --items 30000 > $OUT/bench2.json 2> $OUT/bench2.err
echo "rc=$?"; tail -3 $OUT/bench2.err | cut -c1-300
timeout 600 python bench.py --steps 2 --warmup 1 --users 200000 --items 30000 --no-cpu-baseline > $OUT/bench1.json 2> $OUT/bench1.err
python - <<PY
import json
def last_json(path):   # gloo prints its own connection messages on stdout
    return [json.loads(l) for l in open(path).read().splitlines() if l.startswith('{"metric"')][-1]
a=last_json("$OUT/bench1.json"); b=last_json("$OUT/bench2.json")
print("1 rank loss", a["loss_users_last"], "it/s", a["value"]); print("2 rank loss", b["loss_users_last"], "it/s", b["value"], "n_gpus", b["n_gpus"])
print("loss rel diff", abs(a["loss_users_last"]-b["loss_users_last"])/abs(a["loss_users_last"]))
PY
