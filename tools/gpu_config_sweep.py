"""ms per ALS iteration inside WRMF.fit_transform (difference of a 6- and a 1-iteration fit) over the constructor's argument space
at 1M x 100k, 5e7 non-zeros: where the slow corners are.  python tools/gpu_config_sweep.py [budget_seconds = 300]
NOT run to completion in round 4: the first attempt had no time budget, was piped through a block-buffered grep and used up the
round's last 19 GPU-minutes without leaving a line (the NNLS corners at rank 100 / 128 take seconds per iteration).  It now stops
at its budget and prints every line unbuffered; run it with `| grep --line-buffered` or not at all."""
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

from rsparse_amd import WRMF, synth

mats = {}
for fb in ("implicit", "explicit"):
    d = synth.make_dataset(1_000_000, 100_000, device="cpu", feedback=fb)
    p, i, x = (t.numpy() for t in d["c_iu"])
    mats[fb] = sp.csc_matrix((x.astype(np.float64), i, p), shape=(100_000, 1_000_000)).T.tocsr()
cases = []
for fb, solver, rank, bias in itertools.product(("implicit", "explicit"), ("conjugate_gradient", "cholesky", "nnls"), (10, 64, 100, 128),
                                                (False, True)):
    if fb == "implicit" and solver == "conjugate_gradient" and bias:
        continue   # (the reference cannot run it either)
    cases.append((fb, solver, rank, bias, "float"))
cases += [("implicit", "cholesky", 10, False, "double"), ("implicit", "nnls", 10, False, "double"), ("explicit", "conjugate_gradient", 10, True, "double")]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
if len(sys.argv) > 2:   # a filter: feedback[,solver,...] e.g. "explicit" or "explicit,cholesky"; "-nnls" drops a solver
    for w in sys.argv[2].split(","):
        cases = [c for c in cases if (w[1:] not in c if w.startswith("-") else w in c)]
t_start = time.perf_counter()
for fb, solver, rank, bias, prec in cases:
    if time.perf_counter() - t_start > budget:
        print("budget of %.0f s used: stopping before %s %s rank %d" % (budget, fb, solver, rank), flush=True)
        break
    model = WRMF(rank=rank, lambda_=0.1, feedback=fb, solver=solver, precision=prec, rng=1, with_user_item_bias=bias)
    m = mats[fb]
    try:
        model.fit_transform(m, n_iter=1, convergence_tol=-1)
        ts = []
        for n_iter in (1, 6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.fit_transform(m, n_iter=n_iter, convergence_tol=-1)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print("%-8s %-18s rank %3d bias %-5s %-6s  %8.1f ms per iteration" % (fb, solver, rank, bias, prec, 200 * (ts[1] - ts[0])), flush=True)
    except Exception as e:   # noqa: BLE001
        print("%-8s %-18s rank %3d bias %-5s %-6s  FAILED: %s" % (fb, solver, rank, bias, prec, str(e)[:80]), flush=True)
