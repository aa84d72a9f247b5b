#!/usr/bin/env python3
"""Generate tests/golden/wrmf_movielens_goldens.npz with the CPU oracle (float64).

These are ORACLE outputs, not reference outputs: the reference cannot run in the build container
(no R / Armadillo) and its own tests hold no numeric vectors for this path.  They pin (a) the
oracle against silent regressions and (b) the HIP path on the GPU box, where only fixtures travel.

Protocol = the reference's own WRMF test loop (tests/testthat/test-wrmf.R:6,48-49): train =
movielens100k[1:900, ], n_iter = 5, convergence_tol = -1; lambda = 0.1 (:12); rank 16 / cg_steps 3
for CG (BASELINE config 1), rank 8 for the Cholesky runs.  Initial factors come from a seeded numpy
generator (N(0, 0.01^2) like src/utils.cpp:139-140) and are stored in the file.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_movielens, csc_take_rows  # noqa: E402
from oracle import wrmf_oracle as O  # noqa: E402


def main():
    n_user_all, n_item, p, i, x = load_movielens()
    tp, ti, tx = csc_take_rows(900, p, i, x)
    n_user = 900
    rng = np.random.default_rng(20250222)
    out = {}
    ranks = {"conjugate_gradient": 16, "cholesky": 8}
    for k in sorted(set(ranks.values())):
        out["init_U_k%d" % k] = np.asfortranarray(rng.standard_normal((k, n_user)) * 0.01)
    for feedback in ("implicit", "explicit"):
        for solver in ("conjugate_gradient", "cholesky"):
            tag = "%s_%s" % (feedback, solver)
            k, lam = ranks[solver], 0.1
            C0 = None
            if solver == "cholesky":
                C0 = np.asfortranarray(rng.standard_normal((k, n_item)) * 0.01)
                out[tag + "_init_components"] = C0
            m = O.OracleWRMF(k, lam, feedback, solver, dtype=np.float64)
            emb = m.fit_transform(n_user, n_item, tp, ti, tx, out["init_U_k%d" % k], n_iter=5,
                                  convergence_tol=-1, init_components=C0)
            out[tag + "_rank"] = np.int32(k)
            out[tag + "_lambda"] = np.float64(lam)
            out[tag + "_loss_items"] = np.array([l[0] for l in m.losses])
            out[tag + "_loss_users"] = np.array([l[1] for l in m.losses])
            out[tag + "_user_emb"] = emb
            out[tag + "_components"] = m.components
            print(tag, "user-half losses", np.round(out[tag + "_loss_users"], 5))
    dst = Path(__file__).with_name("wrmf_movielens_goldens.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, dst.stat().st_size)


if __name__ == "__main__":
    main()
