"""dev: the exact solver with a large lambda (the reference grid has lambda = 1000 at small ranks only): per-row error against the fp64
oracle at ranks that take the rank-64 / rank-128 kernels natively or through zero padding"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rsparse_amd import als, synth
from oracle import wrmf_oracle as O

for implicit in (False,):
    d = synth.make_dataset(3000, 800, seed=5, mean_deg=40, d_max=600, feedback="implicit" if implicit else "explicit", device="cpu")
    p, i, x = (t.numpy() for t in d["c_iu"])
    x = x.astype(np.float64)
    n_fix, n_cols = 800, 3000
    cnt = np.bincount(i, minlength=n_fix).astype(np.float64)
    for lam, scale, dyn in ((1000.0, 1e-12, False), (1000.0, 1e-20, False), (1000.0, 1e-27, False), (1000.0, 1e-27, True)):
        for k in (8, 64, 128):
            rng = np.random.default_rng(k)
            X = np.asfortranarray((rng.standard_normal((k, n_fix)) * scale).astype(np.float32))
            Y0 = np.asfortranarray((rng.standard_normal((k, n_cols)) * scale).astype(np.float32))
            X64 = np.asfortranarray(X, dtype=np.float64); Yr = np.asfortranarray(Y0, dtype=np.float64).copy(order="F")
            csc = (n_fix, n_cols, p, i, x)
            if implicit:
                O.als_implicit(p, i, x, X64, Yr, O.gramian(X64, lam), lam, 0, 3, n_threads=8)
                Y = Y0.copy(order="F"); als.als_implicit(csc, X, Y, lam, 1, 0, 3, "float", False, False)
            else:
                lref = O.als_explicit(p, i, x, X64, Yr, cnt, lam, 0, 3, dyn, n_threads=8)
                Y = Y0.copy(order="F"); loss = als.als_explicit(csc, X, Y, cnt.astype(np.float32), lam, 1, 0, 3, dyn, "float", False, False)
            err = np.linalg.norm(Y - Yr, axis=0) / np.maximum(np.linalg.norm(Yr, axis=0), 1e-30)
            print("%s lambda %6.1f scale %g dyn %s rank %3d  max row err %.2e  (rows > 1e-4: %d)  loss rel %.1e" % ("implicit" if implicit else "explicit", lam, scale, dyn, k, err.max(), int((err > 1e-4).sum()), abs(loss - lref) / abs(lref)), flush=True)
