#!/usr/bin/env python3
"""`$predict` on the device at scale: top_product_kernel, n_users x n_items scores at rank k fused with
the top-`topk` selection and the per-user exclusion list.  Prints one JSON line: users/s, TFLOP/s of the score product
(2 * n_users * n_items * k flops) against the 157.3 TFLOP/s fp32 matrix peak, and bytes if the factors were read once.

  python tools/gpu_predict.py [--users 1000000] [--items 1000000] [--rank 128] [--topk 10] [--exclude-deg 50] [--batch 100000]
"""
import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from rsparse_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=1_000_000)
ap.add_argument("--items", type=int, default=1_000_000)
ap.add_argument("--rank", type=int, default=128)
ap.add_argument("--topk", type=int, default=10)
ap.add_argument("--exclude-deg", type=int, default=50, help="not_recommend entries per user (0 = no exclusion list)")
ap.add_argument("--batch", type=int, default=0, help="users per call (0 = all in one call: a call's last wave of workgroups is "
                "partly empty -- 100k users are 391 workgroups of 256 users on 256 CUs, 1.5 rounds that cost 2 --, which is what the "
                "100k batches of round 3 measured)")
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--rescore", action="store_true", help="the entry WRMF.predict uses since round 5: k + max(8, k/4) candidates from "
                "the fp32 pass, scores and order from the double product (rsparse_hip_top_product_f64_device)")
a = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
U = torch.randn(a.users, a.rank, generator=g, device=dev) * 0.1
V = torch.randn(a.items, a.rank, generator=g, device=dev) * 0.1
nb = min(a.batch, a.users) if a.batch > 0 else a.users
nr_p = nr_j = None
if a.exclude_deg > 0:
    d = a.exclude_deg
    j = torch.randint(0, a.items, (nb, d), generator=g, device=dev, dtype=torch.int64)
    j = torch.sort(j, dim=1).values
    j = j + torch.arange(d, device=dev)          # strictly increasing inside a row
    j = j.clamp_(max=a.items - 1)
    nr_j = j.to(torch.int32).contiguous().view(-1)
    nr_p = (torch.arange(nb + 1, device=dev, dtype=torch.int64) * d).to(torch.int32)
res = torch.empty((nb, a.topk), dtype=torch.int32, device=dev)
sc = torch.empty((nb, a.topk), dtype=torch.float64 if a.rescore else torch.float32, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def run(u0, n):
    if a.rescore:
        _lib.check(lib.rsparse_hip_top_product_f64_device(U[u0:u0 + n].data_ptr(), V.data_ptr(), None, None, n, a.items, a.rank,
                                                          a.topk, -1, None if nr_p is None else nr_p.data_ptr(),
                                                          None if nr_j is None else nr_j.data_ptr(), None, 0, 0.0,
                                                          res.data_ptr(), sc.data_ptr(), stream))
        return
    _lib.check(lib.rsparse_hip_top_product_device(U[u0:u0 + n].data_ptr(), V.data_ptr(), n, a.items, a.rank, a.topk,
                                                  None if nr_p is None else nr_p.data_ptr(), None if nr_j is None else nr_j.data_ptr(),
                                                  None, 0, 0.0, res.data_ptr(), sc.data_ptr(), stream))


run(0, min(nb, 4096))   # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    for u0 in range(0, a.users - nb + 1, nb):
        run(u0, nb)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
done = (a.users // nb) * nb
# spot check of the last batch against torch (fp32 GEMM + topk) on 64 users
u0 = done - nb
S = U[u0:u0 + 64] @ V.T
if nr_p is not None:
    S.scatter_(1, nr_j.view(nb, -1)[:64].to(torch.int64), float("-inf"))
ref = torch.topk(S, a.topk, dim=1)
ok = float((torch.abs(ref.values - sc[:64].to(torch.float32)) <= 1e-4 * ref.values.abs().clamp_min(1e-6)).float().mean())
flops = 2.0 * done * a.items * a.rank
print(json.dumps({"what": "top_product_kernel ($predict)" + (" + double re-scoring" if a.rescore else ""), "users": done, "items": a.items, "rank": a.rank, "topk": a.topk,
                  "exclude_per_user": a.exclude_deg, "users_per_call": nb, "seconds": dt, "users_per_sec": done / dt,
                  "score_tflops": flops / dt / 1e12, "fp32_matrix_peak_tflops": 157.3, "frac_of_fp32_peak": flops / dt / 1e12 / 157.3,
                  "scores_match_torch_topk_frac": ok}))
