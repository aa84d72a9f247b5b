import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, torch
from rsparse_amd import WRMF, synth
d = synth.make_dataset(1_000_000, 100_000, device="cpu", feedback="implicit")
p, i, x = (t.numpy() for t in d["c_iu"])
m = sp.csc_matrix((x.astype(np.float64), i, p), shape=(100_000, 1_000_000)).T.tocsr()
model = WRMF(rank=128, lambda_=0.1, feedback="implicit", solver="conjugate_gradient", precision="float", rng=1)
model.fit_transform(m, n_iter=1, convergence_tol=-1)
xb = m[:1000]
model.predict(xb, k=10); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): model.predict(xb, k=10)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
