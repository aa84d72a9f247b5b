#!/usr/bin/env python3
"""What ONE rank of an N-rank run generates at config-3 size (10M x 1M): its own users' rows and their item counts -- steps 1-2
of synth.make_shard, the part no collective is in -- timed alone on one GPU.  python tools/gpu_datagen_rank.py [rank] [world]"""
import json, sys, time
import torch
sys.path.insert(0, ".")
from rsparse_amd import synth
from rsparse_amd.engine import ShardedALS

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ws = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_users, n_items, dev = 10_000_000, 1_000_000, torch.device("cuda", 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
perm = synth.item_permutation(n_items, 20250222, dev)
cnt_user = synth.degrees(torch.arange(n_users, dtype=torch.int64, device=dev), 20250222, 50.0, 5000, n_items)
lu, _ = ShardedALS.layouts(n_users, n_items, ws, cnt_user, None)
u0, u1 = lu.bounds[rank]
torch.cuda.synchronize(); t1 = time.perf_counter()
cnt_item = torch.zeros(n_items, dtype=torch.int64, device=dev)
nnz = 0
for b0 in range(u0, u1, 2_000_000):
    b1 = min(u1, b0 + 2_000_000)
    ip, it, v = synth.generate_user_block(b0, b1, n_items, device=dev, perm=perm)
    cnt_item += torch.bincount(it, minlength=n_items)
    nnz += int(ip[-1])
torch.cuda.synchronize(); t2 = time.perf_counter()
print(json.dumps({"rank": rank, "world": ws, "users_of_rank": u1 - u0, "nnz_of_rank": nnz, "degrees_and_bounds_s": round(t1 - t0, 3),
                  "own_rows_s": round(t2 - t1, 3), "whole_matrix_one_rank_s_round5": "2 x 2.6 (two passes over all users)"}))
