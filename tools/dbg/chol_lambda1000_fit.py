"""dev: WRMF fits with lambda = 1000 (explicit, exact solver; the reference grid's shrinking regime: the factors reach 1e-26 in five
iterations) at ranks that take the rank-64 / rank-128 kernels natively or through zero padding, against the fp64 oracle fit"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp
from rsparse_amd import WRMF
from oracle import wrmf_oracle as O

z = np.load(os.path.join(ROOT, "tests", "golden", "movielens100k_csc.npz"))
n_user, n_item = (int(v) for v in z["Dim"])
p, i, x = z["p"], z["i"], z["x"]
train = sp.csc_matrix((x, i, p), shape=(n_user, n_item))
tp, ti, tx = train.indptr.astype(np.int32), train.indices.astype(np.int32), train.data
FB = sys.argv[1] if len(sys.argv) > 1 else "explicit"
SOLVER = sys.argv[2] if len(sys.argv) > 2 else "cholesky"
for lam in (1000.0,):
    for rank in (8, 32, 64, 100, 128):
        for n_iter in (1, 5):
            rng = np.random.default_rng(rank)
            U0 = (rng.standard_normal((n_user, rank)) * 0.01).astype(np.float32)
            V0 = (rng.standard_normal((rank, n_item)) * 0.01).astype(np.float32)
            model = WRMF(rank=rank, lambda_=lam, feedback=FB, solver=SOLVER, precision="float", init=(None if SOLVER == "conjugate_gradient" else V0.copy()))
            model._init_user_factors = U0
            emb = model.fit_transform(train, n_iter=n_iter, convergence_tol=-1)
            ref = O.OracleWRMF(rank, lam=lam, feedback=FB, solver=SOLVER, dtype=np.float64, n_threads=8)
            ref_emb = ref.fit_transform(n_user, n_item, tp, ti, tx, U0.T.astype(np.float64), n_iter=n_iter, convergence_tol=-1, init_components=(None if SOLVER == "conjugate_gradient" else V0.astype(np.float64)))
            eu = np.linalg.norm(emb - ref_emb) / max(np.linalg.norm(ref_emb), 1e-300)
            print(FB, SOLVER, "lambda %6.1f rank %3d n_iter %d  user_emb rel err %.2e  |emb| %.2e" % (lam, rank, n_iter, eu, np.abs(ref_emb).max()), flush=True)
