"""Test-only stand-in for rsparse_amd.engine.HipBackend that runs the CPU oracle on CPU tensors,
so the multi-rank control flow of ShardedALS (sharding, Gramian all-reduce, factor all-gather,
loss all-reduce) can run under gloo without a GPU, and so can the one-rank driver of WRMF.fit_transform
(tests/test_wrmf_single.py).  Never imported by the product."""
import numpy as np
import torch

from oracle import wrmf_oracle as O


class _Csc:
    def __init__(self, n_rows, n_cols, p, i, x):
        self.n_rows, self.n_cols = n_rows, n_cols
        self.p = p.numpy().astype(np.int32)
        self.i = i.numpy().astype(np.int32)
        self._x = x   # a VIEW of the caller's values, like the device handle: in-place changes (the global mean) are seen

    @property
    def x(self):
        return self._x.numpy().astype(np.float64)

    def info(self):
        return dict(n_rows=self.n_rows, n_cols=self.n_cols, nnz=int(self.p[-1]), n_long=0, max_len=0)


class OracleBackend:
    name = "oracle"

    def to_device(self, a, dtype):
        # a COPY, as an upload is: what the driver then changes in place (the global mean leaving the values) must not
        # reach the caller's arrays
        return torch.as_tensor(a, dtype=dtype).clone().contiguous()

    def make_csc(self, n_rows, n_cols, p, i, x):
        return _Csc(n_rows, n_cols, p, i, x)

    # ---- what only the one-rank path of WRMF.fit_transform asks of its backend (HipBackend: kernels of wrmf_ingest.hip /
    # wrmf_bias.hip behind the C ABI) -------------------------------------------------------------------------------------
    def values_to_float(self, x64):
        return x64.to(torch.float32)

    def transpose_csc(self, n_rows, n_cols, p, i, x):
        tp, ti, tx = O.csc_transpose(n_rows, n_cols, p.numpy(), i.numpy(), x.numpy())
        return torch.from_numpy(tp), torch.from_numpy(ti), torch.from_numpy(tx)

    def top_product(self, U, V, k, nr_p, nr_j, exclude0, glob_mean):
        res, sc = O.top_product(U.numpy().astype(np.float64), V.numpy().astype(np.float64).T, k,
                                None if nr_p is None else nr_p.numpy(), None if nr_j is None else nr_j.numpy(),
                                () if exclude0 is None else (exclude0.numpy() + 1).tolist(), glob_mean)
        return torch.from_numpy(res), torch.from_numpy(np.nan_to_num(sc, nan=0.0))

    def subtract_mean(self, x, x_other=None):
        m = float(x.to(torch.float64).mean()) if x.numel() else 0.0
        x -= m
        if x_other is not None:
            x_other -= m
        return m

    def _init_biases(self, fn, csc_ui, csc_iu, user_bias, item_bias, *args, **kw):
        x1, x2 = csc_ui.x, csc_iu.x                      # float64 copies; the oracle removes the mean from them in place
        ub, ib = user_bias.numpy(), item_bias.numpy()    # views: filled in place
        gb = fn((csc_ui.p, csc_ui.i, x1), (csc_iu.p, csc_iu.i, x2), ub, ib, *args, **kw)
        csc_ui._x.copy_(torch.from_numpy(x1))
        csc_iu._x.copy_(torch.from_numpy(x2))
        return gb

    def initialize_biases_explicit(self, csc_ui, csc_iu, user_bias, item_bias, lambda_, dynamic_lambda, non_negative,
                                   calculate_global_bias):
        return self._init_biases(O.init_biases_explicit, csc_ui, csc_iu, user_bias, item_bias, lambda_, dynamic_lambda,
                                 non_negative, calculate_global_bias)

    def initialize_biases_implicit(self, csc_ui, csc_iu, user_bias, item_bias, lambda_, non_negative,
                                   calculate_global_bias=False):
        return self._init_biases(O.init_biases_implicit, csc_ui, csc_iu, user_bias, item_bias, lambda_, non_negative,
                                 calculate_global_bias=calculate_global_bias)

    @staticmethod
    def _f(t):  # (n, k) row-major tensor -> (k, n) column-major numpy view
        return t.numpy().T

    def gramian(self, F, lambda_, out, sumsq_out, absmax_inout=None):
        Ff = np.asfortranarray(self._f(F))
        if absmax_inout is not None and Ff.size:
            absmax_inout[0] = max(float(absmax_inout[0]), float(np.abs(Ff).max()))
        G = O.gramian(Ff, lambda_) if Ff.shape[1] else np.float32(lambda_) * np.eye(Ff.shape[0], dtype=np.float32)
        out.copy_(torch.from_numpy(np.ascontiguousarray(G)))
        if sumsq_out is not None:
            sumsq_out[0] = float((Ff.astype(np.float64) ** 2).sum())

    def half_iteration(self, csc, implicit, F, S_block, G, lambda_, solver, cg_steps, dynamic_lambda, loss_out,
                       bias_last_row=None, absmax=None, global_bias=0.0):
        # absmax = what the HIP backend passes on to the library as max |F|; here: checked against the truth (the
        # multi-rank tests thereby verify that every rank holds the GLOBAL maximum when it solves, and that it is current)
        if absmax is not None:
            self.absmax_seen = getattr(self, "absmax_seen", 0) + 1
            assert abs(float(absmax[0]) - float(F.abs().max())) <= 1e-6 * max(1.0, float(absmax[0])), "stale or local absmax"
        X = np.asfortranarray(self._f(F))
        Y = np.asfortranarray(self._f(S_block)).copy(order="F")
        if csc.n_cols == 0:
            loss_out[0] = 0.0
            return
        if bias_last_row is not None:   # user/item biases (wrmf_explicit.hpp:41-64,86-91,113-127; wrmf_implicit.hpp:114-154,186-252)
            blr = bool(bias_last_row)
            if implicit:
                O.als_implicit(csc.p, csc.i, csc.x, X, Y, np.asfortranarray(G.numpy().T), lambda_, solver, cg_steps,
                               with_biases=True, is_x_bias_last_row=blr, global_bias=global_bias)
            else:
                cnt = np.zeros(X.shape[1], dtype=X.dtype)
                O.als_explicit(csc.p, csc.i, csc.x, X, Y, cnt, lambda_, solver, cg_steps, dynamic_lambda, with_biases=True,
                               is_x_bias_last_row=blr)
            S_block.copy_(torch.from_numpy(np.ascontiguousarray(Y.T)))
            loss_out[0] = _row_loss_bias(csc, X, Y, implicit, lambda_, dynamic_lambda, blr, global_bias if implicit else 0.0)
            return
        if implicit:
            Gn = np.asfortranarray(G.numpy().T)
            O.als_implicit(csc.p, csc.i, csc.x, X, Y, Gn, lambda_, solver, cg_steps, global_bias=global_bias)
        else:
            cnt = np.zeros(X.shape[1], dtype=X.dtype)   # regulariser on X is added by the engine
            O.als_explicit(csc.p, csc.i, csc.x, X, Y, cnt, lambda_, solver, cg_steps, dynamic_lambda)
        rows = _row_loss(csc, X, Y, implicit, lambda_, dynamic_lambda, 1.0 - global_bias if implicit else 1.0)
        S_block.copy_(torch.from_numpy(np.ascontiguousarray(Y.T)))
        loss_out[0] = rows

    def weighted_sumsq(self, F, w, out):
        Ff = self._f(F).astype(np.float64)
        s = (Ff ** 2).sum(axis=0)
        out[0] = float((s * w.numpy().astype(np.float64)).sum() if w is not None else s.sum())

    # the bias initialisation, one sweep over one column block (wrmf_utils.hpp:32-165; HipBackend has the same three)
    def bias_sweep_explicit(self, csc, other_bias, lambda_, dynamic_lambda, non_negative, out):
        o = other_bias.numpy()
        dt = o.dtype.type
        for c in range(csc.n_cols):
            p1, p2 = csc.p[c], csc.p[c + 1]
            cnt = dt(p2 - p1)
            lam_use = dt(lambda_) * (cnt if dynamic_lambda else dt(1))
            with np.errstate(invalid="ignore", divide="ignore"):
                b = dt(np.sum(csc.x[p1:p2] - o[csc.i[p1:p2]].astype(np.float64))) / (lam_use + cnt)
            out[c] = float(max(b, 0) if non_negative else b)

    def bias_prep_implicit(self, csc, n_other, lambda_, means, adj):
        for c in range(csc.n_cols):
            p1, p2 = csc.p[c], csc.p[c + 1]
            cnt = p2 - p1
            if cnt > 0:
                a = float(csc.x[p1:p2].sum())
                means[c] = a / (a + (n_other - cnt))
                a += n_other - cnt
                adj[c] = a / (a + lambda_)
            else:
                means[c] = 0.0
                adj[c] = n_other / (n_other + lambda_)

    def bias_sweep_implicit(self, csc, other_bias, n_other, other_sum, means, adj, non_negative, global_bias, out):
        o = other_bias.numpy().astype(np.float64)
        mean0 = 0.0 if other_sum is None else float(other_sum[0]) / n_other
        for c in range(csc.n_cols):
            wsum, bias_this = float(n_other), mean0
            for e in range(csc.p[c], csc.p[c + 1]):
                w = csc.x[e] - 1.0
                wsum += w
                bias_this += (w * (o[csc.i[e]] - bias_this)) / wsum
            b = (float(means[c]) - bias_this - global_bias) * float(adj[c])
            out[c] = float(max(b, 0.0) if non_negative else b)

    def check_numeric(self):
        self.report_numeric(*self.numeric_counts())

    # the two halves of check_numeric, as HipBackend has them: a sharded WRMF sums the counts over its ranks in between
    fake_counts = (0, 0)

    def numeric_counts(self):
        c, self.fake_counts = self.fake_counts, (0, 0)
        return c

    def report_numeric(self, bad, fell):
        self.last_fallback_rows = fell
        if bad:
            from rsparse_amd import _lib
            raise _lib.RsparseHipError(_lib.ERR_NUMERIC, "%d per-row systems were singular" % bad)


def _row_loss_bias(csc, X, Y, implicit, lambda_, dynamic_lambda, blr, global_bias):
    """the same with user/item biases: X = [1, ..., x_bias] / Y = [y_bias, ..., 1] when blr, the other way round otherwise;
    every column is solved with implicit feedback (wrmf_implicit.hpp:178,256-270; wrmf_explicit.hpp:131-132)"""
    X = X.astype(np.float64)
    Y = Y.astype(np.float64)
    k = X.shape[0]
    xs, xb, ys = (slice(0, k - 1), k - 1, slice(0, k - 1)) if blr else (slice(1, k), 0, slice(1, k))
    tot = 0.0
    for c in range(csc.n_cols):
        p1, p2 = csc.p[c], csc.p[c + 1]
        y = Y[ys, c]
        yy = float(y @ y)
        if p1 == p2:
            if implicit:
                tot += lambda_ * yy
            continue
        idx, v = csc.i[p1:p2], csc.x[p1:p2]
        t = y @ X[xs][:, idx]
        if implicit:
            tot += float((((1.0 - global_bias) - t - X[xb, idx]) ** 2) @ v) + lambda_ * yy
        else:
            tot += float(((v - X[xb, idx] - t) ** 2).sum()) + lambda_ * ((p2 - p1) if dynamic_lambda else 1.0) * yy
    return tot


def _row_loss(csc, X, Y, implicit, lambda_, dynamic_lambda, target=1.0):
    """un-normalised row part of the loss (wrmf_implicit.hpp:259-261 / wrmf_explicit.hpp:131-132), float64"""
    X = X.astype(np.float64)
    Y = Y.astype(np.float64)
    tot = 0.0
    for c in range(csc.n_cols):
        p1, p2 = csc.p[c], csc.p[c + 1]
        if p1 == p2:
            if implicit and target != 1.0:      # with a global bias empty columns are solved and regularised (:178, :257)
                tot += lambda_ * float(Y[:, c] @ Y[:, c])
            continue
        t = Y[:, c] @ X[:, csc.i[p1:p2]]
        v = csc.x[p1:p2]
        yy = float(Y[:, c] @ Y[:, c])
        if implicit:
            tot += float(((target - t) ** 2) @ v) + lambda_ * yy
        else:
            tot += float(((v - t) ** 2).sum()) + lambda_ * ((p2 - p1) if dynamic_lambda else 1.0) * yy
    return tot
