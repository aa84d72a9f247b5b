#!/usr/bin/env python3
"""ONE full-size CPU half-iteration, timed, next to the figure that bench.py's `cpu_baseline` extrapolates from a
sample.  User side of config 2 (1M users x 100k items, 5.0e7 non-zeros, rank 64, implicit CG(3)), fp64, every physical core of the
GPU box's host: (a) the whole user half-iteration on the oracle, (b) the same on a random sample of users as bench.py takes it,
extrapolated in nnz.   python tools/gpu_cpu_full_half.py > profiles/r06/..."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench as B
from oracle import wrmf_oracle as O
from rsparse_amd import synth

threads = B.physical_cores()
os.environ["OMP_NUM_THREADS"] = str(threads)
try:
    O.lib(native=True); native = True
except Exception:
    native = False
n_user, n_item, k, lam = 1_000_000, 100_000, 64, 0.1
dev = "cuda" if torch.cuda.is_available() else "cpu"
d = synth.make_dataset(n_user, n_item, device=dev)
g = torch.Generator(device=dev).manual_seed(20250222)
U = torch.randn(n_user, k, generator=g, device=dev) * 0.01
V = torch.randn(n_item, k, generator=g, device=dev) * 0.01
Vh = np.asfortranarray(V.cpu().numpy().T.astype(np.float64))
Uh = np.asfortranarray(U.cpu().numpy().T.astype(np.float64))
G = O.gramian(Vh, lam, native=native)
p, i, x = (t.cpu().numpy() for t in d["c_iu"])
x = x.astype(np.float64)
out = {"workload": "user half-iteration of config 2: %d users x %d items, %d nnz, rank %d, implicit CG(3), fp64" % (n_user, n_item, d["nnz"], k),
       "cores": threads, "cpu_model": B.cpu_model(), "march": "native" if native else "x86-64-v3"}
Y = Uh.copy(order="F")
t0 = time.perf_counter()
O.als_implicit(p, i, x, Vh, Y, G, lam, 1, 3, n_threads=threads, native=native)
out["full_size_s"] = time.perf_counter() - t0
for take in (20000, 160000):
    ps, is_, xs, pick = B._sample_rows(d["c_iu"], take, 1)
    Ys = np.asfortranarray(Uh[:, pick.cpu().numpy()]).copy(order="F")
    t0 = time.perf_counter()
    O.als_implicit(ps, is_, xs, Vh, Ys, G, lam, 1, 3, n_threads=threads, native=native)
    t = time.perf_counter() - t0
    out["sample_%d_users_s" % take] = t
    out["sample_%d_extrapolated_s" % take] = t * d["nnz"] / max(int(ps[-1]), 1)
out["extrapolated_over_measured"] = out["sample_160000_extrapolated_s"] / out["full_size_s"]
print(json.dumps(out, indent=1))
