"""How long does ONE very long row take (the most popular item of config 3 has ~350k non-zeros and is solved by a
single 8-wave team)?  Decides whether the multi-GPU item half needs a split-row path."""
import sys, time
import torch
sys.path.insert(0, ".")
from rsparse_amd.engine import HipBackend

be = HipBackend(0); dev = be.device; k = 128
n_rows = 4000000
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn(n_rows, k, generator=g, device=dev) * 0.05
G = (X[:4096].T @ X[:4096]) * (n_rows / 4096) + 0.1 * torch.eye(k, device=dev)
for n_cols, L in ((1, 350000), (8, 350000), (1, 100000), (64, 30000), (256, 8000)):
    p = (torch.arange(n_cols + 1, device=dev, dtype=torch.int64) * L).to(torch.int32)
    i = torch.randint(0, n_rows, (n_cols * L,), generator=g, device=dev, dtype=torch.int32)
    i = i.view(n_cols, L).sort(dim=1).values.reshape(-1).contiguous()
    x = torch.ones(n_cols * L, device=dev) * 2.0
    csc = be.make_csc(n_rows, n_cols, p, i, x)
    Y = torch.zeros(n_cols, k, device=dev); loss = torch.zeros(1, dtype=torch.float64, device=dev)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        be.half_iteration(csc, True, X, Y, G, 0.1, 1, 3, True, loss)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(dict(n_cols=n_cols, L=L, ms=round(dt * 1e3, 3), ns_per_nnz=round(dt * 1e9 / L, 2)), flush=True)
