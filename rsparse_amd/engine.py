"""Device-resident ALS driver: one process per GPU, factors replicated, rows sharded.

This is the part of the reference's R driver that sits between `WRMF$fit_transform()` and the
`.Call` boundary -- `private$solver` and the two R wrappers (R/model_WRMF.R:111-147, 456-515) --
re-cut for data that lives in HBM for the whole fit:

  per half-iteration (solve side S given fixed side F):
    1. Gramian   G = F F^T + fl(lambda) I                (R/model_WRMF.R:474-486)
       each rank reduces only the block of F it owns (MFMA kernel), then ONE all-reduce of the
       k x k partial (+ the scalar sum(F^2) the loss needs);
    2. solve     every rank solves its own rows of S against the full replica of F
                 (als_implicit / als_explicit kernels, inst/include/wrmf_implicit.hpp:160-283);
    3. exchange  in-place all-gather of the solved blocks so every replica of S is current;
    4. loss      all-reduce of one double                (wrmf_implicit.hpp:286-304).

With world_size == 1 steps 1/3/4 degenerate to local calls and no collective is issued.

torch is used for device memory, streams and torch.distributed (RCCL) only; all numerics go through
the C ABI (`HipBackend`).  The backend is an explicit object so that the multi-rank control flow
can be exercised on CPU with gloo in tests (which plug in the CPU oracle); the product default is
the HIP backend and it raises if the extension or a GPU is missing -- there is no CPU fallback.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib

SOLVER_CODES = {"cholesky": 0, "conjugate_gradient": 1, "nnls": 2}   # R/model_WRMF.R:99-100


def block_bounds(n, world_size, multiple=1):
    """Equal row blocks (the last ones padded): rank r owns rows [r*B, min(n, (r+1)*B)); B is a multiple
    of `multiple` (sub-blocks for communication/compute overlap)."""
    B = max(1, math.ceil(n / world_size)) if n > 0 else 1
    B = -(-B // multiple) * multiple
    return B, [(min(n, r * B), min(n, (r + 1) * B)) for r in range(world_size)]


def default_user_subblocks(world_size):
    """Sub-blocks per rank for the user half: with several ranks the all-gather of the solved user block
    (5 GB at 10M x 128) costs about as much as the solve, so it is pipelined sub-block by sub-block."""
    import os
    if world_size <= 1:
        return 1
    return max(1, int(os.environ.get("RSPARSE_USER_SUBBLOCKS", "4")))


class HipBackend:
    """Numerics on the current HIP device through librsparse_wrmf_hip.so."""

    name = "hip"

    def __init__(self, device=None):
        self.lib = _lib.load()
        if torch.cuda.device_count() < 1 or self.lib.rsparse_hip_device_count() < 1:
            raise RuntimeError("rsparse_amd: no HIP device visible (the device path has no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        torch.cuda.set_device(self.device)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def to_device(self, a, dtype):
        return torch.as_tensor(a, dtype=dtype).to(self.device).contiguous()

    def make_csc(self, n_rows, n_cols, p, i, x):
        """p/i/x: torch tensors already on the device (int32, int32, float32)."""
        assert p.dtype == torch.int32 and i.dtype == torch.int32 and x.dtype == torch.float32
        h = ctypes.c_void_p()
        _lib.check(self.lib.rsparse_hip_csc_create_device(int(n_rows), int(n_cols), p.data_ptr(), i.data_ptr(),
                                                          x.data_ptr(), ctypes.byref(h)))
        return _CscHandle(self.lib, h, (p, i, x), n_rows, n_cols)

    def transpose_csc(self, n_rows, n_cols, p, i, x):
        """CSC (p, i, x) of an n_rows x n_cols matrix, on the device -> CSC of its transpose, row indices ascending
        inside every column (the second orientation of a fit, R/model_WRMF.R:190)."""
        assert p.dtype == torch.int32 and i.dtype == torch.int32 and x.dtype == torch.float32
        nnz = int(i.numel())
        pt = torch.empty(n_rows + 1, dtype=torch.int32, device=self.device)
        it = torch.empty(max(nnz, 1), dtype=torch.int32, device=self.device)[:nnz]
        xt = torch.empty(max(nnz, 1), dtype=torch.float32, device=self.device)[:nnz]
        _lib.check(self.lib.rsparse_hip_csc_transpose_device(int(n_rows), int(n_cols), p.data_ptr(), i.data_ptr(),
                                                             x.data_ptr(), pt.data_ptr(), it.data_ptr(), xt.data_ptr(),
                                                             self._stream()))
        return pt, it, xt

    def values_to_float(self, x64):
        """f64 values as they arrive from a dgCMatrix -> f32 resident values (wrmf_implicit.hpp:182-183)."""
        assert x64.dtype == torch.float64 and x64.is_cuda
        out = torch.empty(x64.shape, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.rsparse_hip_values_to_float_device(int(x64.numel()), x64.data_ptr(), out.data_ptr(),
                                                               self._stream()))
        return out

    def gramian(self, F, lambda_, out, sumsq_out):
        n, k = F.shape
        _lib.check(self.lib.rsparse_hip_gramian_device(F.data_ptr(), k, n, float(lambda_), out.data_ptr(),
                                                       None if sumsq_out is None else sumsq_out.data_ptr(),
                                                       self._stream()))

    def half_iteration(self, csc, implicit, F, S_block, G, lambda_, solver, cg_steps, dynamic_lambda, loss_out,
                       bias_last_row=None):
        """bias_last_row: None = no user/item biases; True/False = als_explicit's is_x_bias_last_row with
        with_biases = TRUE (explicit feedback only)."""
        k = F.shape[1]
        if bias_last_row is not None and implicit:
            # G: (k-1) x (k-1) Gramian of F without its bias row, ridge included (R/model_WRMF.R:463-486)
            _lib.check(self.lib.rsparse_hip_als_implicit_bias_device(csc.h, F.data_ptr(), S_block.data_ptr(), G.data_ptr(),
                                                                     k, float(lambda_), int(solver),
                                                                     int(bool(bias_last_row)), loss_out.data_ptr(),
                                                                     self._stream()))
        elif bias_last_row is not None:
            _lib.check(self.lib.rsparse_hip_als_explicit_bias_device(csc.h, F.data_ptr(), S_block.data_ptr(), k,
                                                                     float(lambda_), int(solver), int(cg_steps),
                                                                     int(bool(dynamic_lambda)), int(bool(bias_last_row)),
                                                                     loss_out.data_ptr(), self._stream()))
        elif implicit:
            _lib.check(self.lib.rsparse_hip_als_implicit_device(csc.h, F.data_ptr(), S_block.data_ptr(), G.data_ptr(),
                                                                k, float(lambda_), int(solver), int(cg_steps),
                                                                loss_out.data_ptr(), self._stream()))
        else:
            _lib.check(self.lib.rsparse_hip_als_explicit_device(csc.h, F.data_ptr(), S_block.data_ptr(), k,
                                                                float(lambda_), int(solver), int(cg_steps),
                                                                int(bool(dynamic_lambda)), loss_out.data_ptr(),
                                                                self._stream()))

    def initialize_biases_explicit(self, csc_ui, csc_iu, user_bias, item_bias, lambda_, dynamic_lambda, non_negative,
                                   calculate_global_bias):
        """wrmf_utils.hpp:32-84 on the device; with calculate_global_bias the resident values of both handles
        lose their mean in place.  Returns the global bias."""
        gb = ctypes.c_double(0.0)
        _lib.check(self.lib.rsparse_hip_initialize_biases_explicit_device(
            csc_ui.h, csc_iu.h, user_bias.data_ptr(), item_bias.data_ptr(), float(lambda_), int(bool(dynamic_lambda)),
            int(bool(non_negative)), int(bool(calculate_global_bias)), ctypes.byref(gb), self._stream()))
        return gb.value

    def initialize_biases_implicit(self, csc_ui, csc_iu, user_bias, item_bias, lambda_, non_negative):
        """wrmf_utils.hpp:86-165 (no global bias) on the device."""
        _lib.check(self.lib.rsparse_hip_initialize_biases_implicit_device(
            csc_ui.h, csc_iu.h, user_bias.data_ptr(), item_bias.data_ptr(), float(lambda_), int(bool(non_negative)),
            self._stream()))

    def subtract_mean(self, x, x_other=None):
        """global_bias = mean(x), removed in place from x (and from the other orientation's values) -- R/model_WRMF.R:278-282"""
        m = ctypes.c_double(0.0)
        _lib.check(self.lib.rsparse_hip_values_subtract_mean_device(int(x.numel()), x.data_ptr(),
                                                                    None if x_other is None else x_other.data_ptr(),
                                                                    ctypes.byref(m), self._stream()))
        return m.value

    def weighted_sumsq(self, F, w, out):
        n, k = F.shape
        _lib.check(self.lib.rsparse_hip_weighted_sumsq_device(F.data_ptr(), k, n,
                                                              None if w is None else w.data_ptr(),
                                                              out.data_ptr(), self._stream()))

    def profile(self, on):
        _lib.check(self.lib.rsparse_hip_profile_enable(int(bool(on))))

    def profile_last(self):
        """milliseconds of the kernels of the last device-layer call (see rsparse_hip_profile_last)"""
        buf = (ctypes.c_double * 8)()
        _lib.check(self.lib.rsparse_hip_profile_last(buf))
        return list(buf)

    def check_numeric(self):
        c = ctypes.c_int64(0)
        _lib.check(self.lib.rsparse_hip_take_numeric_failures(ctypes.byref(c)))
        if c.value:
            raise _lib.RsparseHipError(_lib.ERR_NUMERIC, "%d per-row systems were not positive definite" % c.value)


class _CscHandle:
    def __init__(self, lib, h, keep, n_rows, n_cols):
        self.lib, self.h, self.keep = lib, h, keep
        self.n_rows, self.n_cols = n_rows, n_cols

    def info(self):
        buf = (ctypes.c_int64 * 40)()
        _lib.check(self.lib.rsparse_hip_csc_info(self.h, buf))
        return dict(n_rows=buf[0], n_cols=buf[1], nnz=buf[2], n_long=buf[3], max_len=buf[4], nnz_long=buf[5],
                    n_empty=buf[6], tile_nnz=buf[7], bucket_rows=list(buf[8:14]), bucket_nnz=list(buf[14:20]),
                    cgq_cfg=buf[20], bucket_wpr=list(buf[22:28]), bucket_capq=list(buf[28:34]),
                    bucket_waves=[abs(v) for v in buf[34:40]], bucket_stream=[int(v < 0) for v in buf[34:40]])

    def __del__(self):
        try:
            if self.h:
                self.lib.rsparse_hip_csc_destroy(self.h)
                self.h = None
        except Exception:
            pass


class ShardedALS:
    """Both orientations of one interaction matrix, row-sharded over the ranks of `group`.

    c_ui : CSC of the users x items matrix -> columns = items  (solved in the item half)
    c_iu : CSC of its transpose            -> columns = users  (solved in the user half)
    Each rank is given the column block it owns, with `p` re-based to 0.
    """

    def __init__(self, backend, n_user, n_item, rank_k, c_ui_block, c_iu_block, total_nnz, feedback="implicit",
                 lambda_=0.0, dynamic_lambda=True, cg_steps=3, group=None, world_size=1, my_rank=0, with_bias=False):
        self.be, self.k = backend, int(rank_k)
        self.n_user, self.n_item, self.total_nnz = int(n_user), int(n_item), int(total_nnz)
        self.implicit = feedback == "implicit"
        self.with_bias = bool(with_bias)     # rank_k counts the row of ones and the bias row (R/model_WRMF.R:160)
        self.lambda_, self.dynamic_lambda, self.cg_steps = float(lambda_), bool(dynamic_lambda), int(cg_steps)
        self.group, self.ws, self.me = group, int(world_size), int(my_rank)
        self.Bu, self.ub, self.Bi, self.ib, self.n_sub = self.partition(n_user, n_item, self.ws)
        u0, u1 = self.ub[self.me]
        i0, i1 = self.ib[self.me]
        # item half: fixed side = users (n_user rows of X), solved = my items
        self.csc_items = backend.make_csc(n_user, i1 - i0, *c_ui_block)
        # user half: fixed side = items, solved = my users
        self.csc_users = backend.make_csc(n_item, u1 - u0, *c_iu_block)
        dev = c_ui_block[0].device
        # sub-blocks of my user block, each with its own schedule (only used when world_size > 1)
        self.Bs = self.Bu // self.n_sub
        self.sub_users = []
        if self.n_sub > 1:
            p, i, x = c_iu_block
            p64 = p.to(torch.int64)
            n_my = u1 - u0
            for j in range(self.n_sub):
                c0, c1 = min(n_my, j * self.Bs), min(n_my, (j + 1) * self.Bs)
                lo, hi = int(p64[c0]), int(p64[c1])
                sp_ = (p64[c0:c1 + 1] - lo).to(torch.int32).contiguous()
                self.sub_users.append((c0, c1, backend.make_csc(n_item, c1 - c0, sp_, i[lo:hi].contiguous(),
                                                                x[lo:hi].contiguous())))
        self.scal_sub = torch.zeros(max(1, self.n_sub), dtype=torch.float64, device=dev)
        self.G = torch.zeros((self.k, self.k), dtype=torch.float32, device=dev)
        self.scal = torch.zeros(4, dtype=torch.float64, device=dev)   # [0] sumsq, [1] loss rows, [2] reg
        self.cnt_user = None   # nnz per user (weights of the explicit regulariser on U)
        self.cnt_item = None

    @staticmethod
    def partition(n_user, n_item, world_size):
        """(Bu, user bounds, Bi, item bounds, user sub-blocks per rank) -- callers shard their CSC blocks with
        exactly these bounds."""
        n_sub = default_user_subblocks(world_size)
        Bu, ub = block_bounds(n_user, world_size, multiple=n_sub)
        Bi, ib = block_bounds(n_item, world_size)
        return Bu, ub, Bi, ib, n_sub

    # -- factor storage: (n_pad, k) row-major == k x n_pad column-major, padded to world_size * B rows
    def alloc_factors(self, n, B, dev):
        return torch.zeros((B * self.ws, self.k), dtype=torch.float32, device=dev)

    def _all_reduce(self, t):
        if self.ws > 1:
            torch.distributed.all_reduce(t, group=self.group)

    def _all_gather_blocks(self, S, B):
        if self.ws > 1:
            if S.is_cuda and torch.distributed.get_backend(self.group) == "gloo":
                # dry-run configuration only (several ranks sharing one GPU): gloo has no device all-gather
                host = S.cpu()
                torch.distributed.all_gather_into_tensor(host, host[self.me * B:(self.me + 1) * B].clone(), group=self.group)
                S.copy_(host)
                return
            # the input is a copy of this rank's block (not a view of the output): no reliance on the
            # backend's in-place all-gather semantics, for the price of one on-device block copy
            mine = S[self.me * B:(self.me + 1) * B].clone()
            torch.distributed.all_gather_into_tensor(S, mine, group=self.group)

    def _all_gather_sub(self, S, B, j):
        """Start the all-gather of sub-block j of every rank's block of S; returns a work handle (or None)."""
        Bs = self.Bs
        mine = S[self.me * B + j * Bs:self.me * B + (j + 1) * Bs]
        if S.is_cuda and torch.distributed.get_backend(self.group) == "gloo":
            host = [torch.empty(mine.shape, dtype=S.dtype) for _ in range(self.ws)]   # dry-run configuration only
            torch.distributed.all_gather(host, mine.cpu(), group=self.group)
            for r in range(self.ws):
                S[r * B + j * Bs:r * B + (j + 1) * Bs].copy_(host[r])
            return None
        outs = [S[r * B + j * Bs:r * B + (j + 1) * Bs] for r in range(self.ws)]
        return torch.distributed.all_gather(outs, mine.clone(), group=self.group, async_op=True)

    def gramian(self, F, n, B, bounds):
        """G = F[:n] F[:n]^T + fl(lambda) I, reduced over the ranks' blocks; scal[0] = sum(F^2)."""
        r0, r1 = bounds[self.me] if self.ws > 1 else (0, n)
        blk = F[r0:r1]
        if self.ws == 1:
            self.be.gramian(blk, self.lambda_, self.G, self.scal[0:1])
        else:
            self.be.gramian(blk, 0.0, self.G, self.scal[0:1])
            self._all_reduce(self.G)
            self._all_reduce(self.scal[0:1])
            self.G.diagonal().add_(float(np.float32(self.lambda_)))   # fl(diag(lambda)), R/model_WRMF.R:476
        return self.G

    def gramian_bias(self, F, n, bounds, bias_last_row):
        """(k-1) x (k-1) Gramian of F without its bias row + fl(lambda) I (R/model_WRMF.R:463-486, 345-351)."""
        k1 = self.k - 1
        if getattr(self, "Gb", None) is None:
            self.Gb = torch.zeros((k1, k1), dtype=torch.float32, device=F.device)
        r0, r1 = bounds[self.me] if self.ws > 1 else (0, n)
        blk = (F[r0:r1, :k1] if bias_last_row else F[r0:r1, 1:]).contiguous()
        if self.ws == 1:
            self.be.gramian(blk, self.lambda_, self.Gb, None)
        else:
            self.be.gramian(blk, 0.0, self.Gb, None)
            self._all_reduce(self.Gb)
            self.Gb.diagonal().add_(float(np.float32(self.lambda_)))
        return self.Gb

    def half_iteration(self, side, U, V, solver, G=None, want_loss=True):
        """side 'items': solve V (item factors) given U; side 'users': solve U given V.
        Returns loss/nnz as the reference reports it (python float) or None."""
        if side == "items":
            F, nF, BF, bF, S, BS, bS, csc, cnt_F = U, self.n_user, self.Bu, self.ub, V, self.Bi, self.ib, self.csc_items, self.cnt_user
        else:
            F, nF, BF, bF, S, BS, bS, csc, cnt_F = V, self.n_item, self.Bi, self.ib, U, self.Bu, self.ub, self.csc_users, self.cnt_item
        # user/item biases: solving the items means X = U = [1, ..., user_bias] (is_bias_last_row = TRUE), solving the
        # users X = components = [item_bias, ..., 1] (FALSE)  -- R/model_WRMF.R:321-329
        blr = (side == "items") if self.with_bias else None
        if self.implicit and G is None:
            G = self.gramian_bias(F, nF, bF, blr) if self.with_bias else self.gramian(F, nF, BF, bF)
        s0, s1 = bS[self.me]
        S_block = S[s0:s1]
        if side == "users" and self.ws > 1 and self.n_sub > 1:
            # pipelined: solve sub-block j, start its all-gather, solve sub-block j+1 meanwhile
            works = []
            self.scal_sub.zero_()
            for j, (c0, c1, sub) in enumerate(self.sub_users):
                if c1 > c0:
                    self.be.half_iteration(sub, self.implicit, F[:nF], S_block[c0:c1], G, self.lambda_, solver,
                                           self.cg_steps, self.dynamic_lambda, self.scal_sub[j:j + 1], blr)
                works.append(self._all_gather_sub(S, BS, j))
            for w in works:
                if w is not None:
                    w.wait()
            self.scal[1:2] = self.scal_sub.sum()
        else:
            self.be.half_iteration(csc, self.implicit, F[:nF], S_block, G, self.lambda_, solver, self.cg_steps,
                                   self.dynamic_lambda, self.scal[1:2], blr)
            self._all_gather_blocks(S, BS)
        if not want_loss:
            return None
        # regulariser on the fixed side (wrmf_implicit.hpp:286-301, wrmf_explicit.hpp:146-173)
        reg = 0.0
        if self.lambda_ > 0:
            if self.implicit and G is self.G:
                pass  # scal[0] already holds sum(F^2) from the Gramian pass
            else:
                r0, r1 = bF[self.me] if self.ws > 1 else (0, nF)
                w = None
                if (not self.implicit) and self.dynamic_lambda:
                    w = cnt_F[r0:r1]
                Freg = F[r0:r1]
                if self.with_bias:   # every row of X but the ones (wrmf_explicit.hpp:147-159): ones first when the
                    Freg = (Freg[:, 1:] if blr else Freg[:, :self.k - 1]).contiguous()   # x bias is last, else last
                self.be.weighted_sumsq(Freg, w, self.scal[0:1])
                self._all_reduce(self.scal[0:1])
        self._all_reduce(self.scal[1:2])
        vals = self.scal[0:2].tolist()   # synchronises
        if self.lambda_ > 0:
            reg = self.lambda_ * vals[0]
        return (vals[1] + reg) / float(self.total_nnz)
