import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, scipy.sparse as sp, torch
from rsparse_amd import WRMF, synth
d = synth.make_dataset(1_000_000, 100_000, device="cpu", feedback="implicit")
p, i, x = (t.numpy() for t in d["c_iu"])
m = sp.csc_matrix((x.astype(np.float64), i, p), shape=(100_000, 1_000_000)).T.tocsr()
for rank in (10, 20, 32, 48):
    model = WRMF(rank=rank, lambda_=0.1, feedback="implicit", solver="cholesky", precision="float", rng=1)
    model.fit_transform(m, n_iter=1, convergence_tol=-1)
    ts = []
    for n_iter in (1, 6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.fit_transform(m, n_iter=n_iter, convergence_tol=-1)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("pad %s rank %d: %.1f ms per iteration" % (os.environ.get("RSPARSE_HIP_CHOL_PAD", "-"), rank, 200 * (ts[1] - ts[0])), flush=True)
