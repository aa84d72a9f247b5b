import os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import wrmf_oracle as O
from rsparse_amd import als
k = 128
rng = np.random.default_rng(11)
n_rows, n_cols = 900, 1203
lens = np.tile(np.arange(0, 33), n_cols // 33 + 1)[:n_cols]
rng.shuffle(lens); lens[5:9] = 0
p = np.zeros(n_cols + 1, np.int64); np.cumsum(lens, out=p[1:]); p = p.astype(np.int32)
i = np.concatenate([np.sort(rng.choice(n_rows, size=int(n), replace=False)) for n in lens]).astype(np.int32)
x = (1.0 + rng.geometric(0.5, size=int(p[-1]))).astype(np.float64)
csc = (n_rows, n_cols, p, i, x)
for outlier in (1.0, 30.0):
  for scale in (1e-3, 1.0, 40.0):
    rng2 = np.random.default_rng(5)
    X = np.asfortranarray((rng2.standard_normal((k, n_rows)) * scale).astype(np.float32)); X[:, 3] *= outlier
    Y0 = np.asfortranarray((rng2.standard_normal((k, n_cols)) * scale).astype(np.float32))
    def ref(Y0):
        Y64 = np.asfortranarray(Y0, dtype=np.float64).copy(order="F"); X64 = np.asfortranarray(X, dtype=np.float64)
        O.als_implicit(p, i, x, X64, Y64, O.gramian(X64, 0.1), 0.1, 1, 3); return Y64
    for warm in (False, True):
        Y0w = Y0.copy(order="F")
        if warm: Y0w[:, ::7] = ref(Y0)[:, ::7].astype(np.float32)
        Yref = ref(Y0w)
        Y32 = Y0w.copy(order="F"); O.als_implicit(p, i, x, X, Y32, O.gramian(X, 0.1), 0.1, 1, 3)
        Y = Y0w.copy(order="F"); als.als_implicit(csc, X, Y, 0.1, 1, 1, 3, "float", False, False)
        den = np.maximum(np.linalg.norm(Yref, axis=0), 1e-30)
        err = np.linalg.norm(Y - Yref, axis=0) / den; err32 = np.linalg.norm(Y32 - Yref, axis=0) / den
        m = lens > 0
        ratio = err[m] / np.maximum(err32[m], 1e-9)
        sel = np.zeros(n_cols, bool); sel[::7] = True
        print("dmf=%s outlier %4.0f scale %6g warm %d: max err %.2e (fp32 oracle %.2e)  median ratio %.2f  max ratio %.1f  [warm cols: max err %.2e, fp32 %.2e]" % (
            os.environ.get("RSPARSE_HIP_DENSE_MFMA", "1"), outlier, scale, warm, err[m].max(), err32[m].max(), np.median(ratio), ratio.max(),
            err[m & sel].max(), err32[m & sel].max()))
