"""Python face of layer (4) of the C ABI: the multi-GPU context of librsparse_wrmf_hip.so (csrc/wrmf_ctx.cpp) -- ONE process,
one host thread per device inside the library, RCCL between them.  This is what the R shim of INTEGRATION.md calls; the
torch.distributed driver of engine.py (one process per GPU, the launch contract of bench.py) is the other multi-GPU path."""
import ctypes

import numpy as np
import scipy.sparse as sp

from . import _lib

COMM_RCCL, COMM_SHARED = 0, 1
SIDE_ITEMS, SIDE_USERS = 0, 1
SOLVER_CODES = {"cholesky": 0, "conjugate_gradient": 1, "nnls": 2}


def _vp(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class MultiGpuALS:
    """n_ranks devices (comm='rccl': rank r on device_ids[r], default r) or n_ranks rank threads on one device
    (comm='shared': tests and single-GPU dry runs).  x: users x items scipy sparse matrix."""

    def __init__(self, n_ranks, comm="rccl", device_ids=None):
        self.lib = _lib.load()
        self.h = ctypes.c_void_p()
        ids = None if device_ids is None else np.ascontiguousarray(device_ids, dtype=np.int32)
        _lib.check(self.lib.rsparse_hip_ctx_create(int(n_ranks), _vp(ids), COMM_SHARED if comm == "shared" else COMM_RCCL,
                                                  ctypes.byref(self.h)))
        self.n_ranks, self.shape, self.rank = int(n_ranks), None, None

    def close(self):
        if self.h:
            self.lib.rsparse_hip_ctx_destroy(self.h)
            self.h = ctypes.c_void_p()

    __del__ = close

    def set_matrix(self, x, n_sub=(0, 0)):
        c_ui = sp.csc_matrix(x)          # columns = items, row indices = users
        c_ui.sort_indices()
        c_iu = sp.csc_matrix(c_ui.T)     # columns = users, row indices = items  (R/model_WRMF.R:190)
        c_iu.sort_indices()
        self._keep = [np.ascontiguousarray(a, dtype=t) for a, t in (
            (c_ui.indptr, np.int32), (c_ui.indices, np.int32), (c_ui.data, np.float64),
            (c_iu.indptr, np.int32), (c_iu.indices, np.int32), (c_iu.data, np.float64))]
        n_user, n_item = c_ui.shape
        _lib.check(self.lib.rsparse_hip_ctx_set_matrix(self.h, int(n_user), int(n_item), *[_vp(a) for a in self._keep],
                                                       int(n_sub[0]), int(n_sub[1])))
        self._keep = None
        self.shape = (int(n_user), int(n_item))

    def set_factors(self, U, V):
        """U: (n_user, rank), V: (n_item, rank) -- every entity's vector contiguous (= rank x n column-major)"""
        U = np.ascontiguousarray(U, dtype=np.float32)
        V = np.ascontiguousarray(V, dtype=np.float32)
        assert U.shape[0] == self.shape[0] and V.shape[0] == self.shape[1] and U.shape[1] == V.shape[1]
        self.rank = int(U.shape[1])
        _lib.check(self.lib.rsparse_hip_ctx_set_factors(self.h, self.rank, _vp(U), _vp(V)))

    def get_factors(self):
        U = np.empty((self.shape[0], self.rank), dtype=np.float32)
        V = np.empty((self.shape[1], self.rank), dtype=np.float32)
        _lib.check(self.lib.rsparse_hip_ctx_get_factors(self.h, _vp(U), _vp(V)))
        return U, V

    def half_iteration(self, side, feedback="implicit", lambda_=0.1, solver="conjugate_gradient", cg_steps=3,
                       dynamic_lambda=True):
        loss = ctypes.c_double(0.0)
        _lib.check(self.lib.rsparse_hip_ctx_half_iteration(
            self.h, SIDE_ITEMS if side == "items" else SIDE_USERS, int(feedback == "implicit"), float(lambda_),
            SOLVER_CODES[solver], int(cg_steps), int(bool(dynamic_lambda)), ctypes.byref(loss)))
        return loss.value

    def numeric_counts(self):
        bad, fell = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(self.lib.rsparse_hip_ctx_take_numeric_failures(self.h, ctypes.byref(bad), ctypes.byref(fell)))
        return bad.value, fell.value

    def info(self):
        buf, tm = (ctypes.c_int64 * 16)(), (ctypes.c_double * 2)()
        _lib.check(self.lib.rsparse_hip_ctx_info(self.h, buf, tm))
        keys = ("ranks", "comm_kind", "n_user", "n_item", "nnz", "rank", "n_sub_users", "n_sub_items", "rows_per_sub_users",
                "rows_per_sub_items", "users_rank0", "items_rank0", "rccl")
        d = {k: int(buf[i]) for i, k in enumerate(keys)}
        d["last_half_ms"], d["last_comm_ms"] = float(tm[0]), float(tm[1])
        return d
