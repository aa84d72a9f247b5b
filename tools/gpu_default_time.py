"""the reference's DEFAULT configuration (rank 10, implicit, conjugate gradient, precision double) and neighbours: seconds per
ALS iteration inside WRMF.fit_transform at 1M x 100k, 5e7 non-zeros (difference of a 11-iteration and a 1-iteration fit)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

from rsparse_amd import WRMF, synth

d = synth.make_dataset(1_000_000, 100_000, device="cpu", feedback="implicit")
p, i, x = (t.numpy() for t in d["c_iu"])
m = sp.csc_matrix((x.astype(np.float64), i, p), shape=(100_000, 1_000_000)).T.tocsr()
CASES = (("double", 10, "conjugate_gradient"), ("float", 10, "conjugate_gradient"),
         ("double", 32, "conjugate_gradient"), ("float", 32, "conjugate_gradient"),
         ("double", 60, "conjugate_gradient"), ("float", 60, "conjugate_gradient"),
         ("double", 10, "cholesky"), ("float", 10, "cholesky"))
if len(sys.argv) > 1:   # precision:rank[:solver] ...   e.g.  double:128 float:128   (round 5: the BASELINE ranks in double)
    CASES = tuple((a.split(":")[0], int(a.split(":")[1]), (a.split(":") + ["conjugate_gradient"])[2]) for a in sys.argv[1:])
BUDGET_S = float(os.environ.get("RSPARSE_TOOL_BUDGET_S", "600"))
T_START = time.perf_counter()
for precision, rank, solver in CASES:
    if time.perf_counter() - T_START > BUDGET_S:
        print("budget of %.0f s spent: stopping before %s rank %d" % (BUDGET_S, precision, rank), flush=True)
        break
    model = WRMF(rank=rank, lambda_=0.1, feedback="implicit", solver=solver, precision=precision, rng=1)
    model.fit_transform(m, n_iter=1, convergence_tol=-1)
    ts = []
    for n_iter in (1, 11):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit_transform(m, n_iter=n_iter, convergence_tol=-1)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("%-6s rank %3d %-18s  %.1f ms per iteration (fit of 1 iteration %.2f s, of 11 %.2f s)" % (precision, rank, solver, 100 * (ts[1] - ts[0]), ts[0], ts[1]), flush=True)
