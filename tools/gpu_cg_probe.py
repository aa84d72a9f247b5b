"""Per-row cost of the CG kernels for uniform row lengths, with the gathered factor matrix (a) small enough
to stay in L2 and (b) far larger than every cache: (b) - (a) is what the gather phase costs."""
import os, sys, time, json
import torch
sys.path.insert(0, ".")
from rsparse_amd.engine import HipBackend

be = HipBackend(0)
dev = be.device
k = 128
G = torch.eye(k, device=dev) * 0.1
res = []
for L in [int(v) for v in os.environ.get('PROBE_L', '16 32 64 128 256 512 1024').split()]:
    n_cols = max(20000, int(6e7 // L))
    for n_rows in [int(v) for v in os.environ.get('PROBE_ROWS', '4096 4000000').split()]:
        g = torch.Generator(device=dev).manual_seed(L)
        X = torch.randn(n_rows, k, generator=g, device=dev) * 0.05
        G = (X[:4096].T @ X[:4096]) * (n_rows / 4096) + 0.1 * torch.eye(k, device=dev)
        p = (torch.arange(n_cols + 1, device=dev, dtype=torch.int64) * L).to(torch.int32)
        i = torch.randint(0, n_rows, (n_cols * L,), generator=g, device=dev, dtype=torch.int32)
        i = i.view(n_cols, L).sort(dim=1).values.reshape(-1).contiguous()
        x = torch.ones(n_cols * L, device=dev) * 2.0
        csc = be.make_csc(n_rows, n_cols, p, i, x)
        Y = torch.zeros(n_cols, k, device=dev)
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            be.half_iteration(csc, True, X, Y, G, 0.1, 1, 3, True, loss)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        byts = n_cols * L * 520 + n_cols * 1024
        res.append(dict(L=L, n_rows=n_rows, n_cols=n_cols, ms=round(dt * 1e3, 3), us_per_row_per_cu=round(dt * 1e6 * 256 / n_cols, 3),
                        GBps=round(byts / dt / 1e9, 1)))
        print(res[-1], flush=True)
        del csc, X, Y, i, x, p
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cg_probe.json", "w"))
