"""ORACLE -- TEST INFRASTRUCTURE ONLY (see wrmf_oracle.cpp header; PARITY UNPINNED).

ctypes binding of oracle/liboracle_wrmf.so plus

  * `np_half_iteration_*`: an independent dense-numpy statement of the same normal equations
    (used to pin the C++ oracle: numpy.linalg.solve of  (XtX + X_nnz diag(c-1) X_nnz^T) y = X_nnz c
    -- inst/include/wrmf_implicit.hpp:207-208,231,236 -- and of the explicit system
    -- inst/include/wrmf_explicit.hpp:103-108);
  * `OracleWRMF`: the R6 driver semantics (R/model_WRMF.R:173-360, 412-452) on top of the C++
    oracle, used to generate the committed goldens and to check the product's host driver.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
CHOLESKY, CONJUGATE_GRADIENT = 0, 1

_i32p = ctypes.POINTER(ctypes.c_int32)
_f64p = ctypes.POINTER(ctypes.c_double)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(native=False, force=False):
    """Compile the C++ oracle.  native=True -> -march=native into a separate file (CPU-baseline
    timing on the GPU box's own host cores)."""
    name = "liboracle_wrmf_native.so" if native else "liboracle_wrmf.so"
    out = _HERE / name
    src = _HERE / "wrmf_oracle.cpp"
    if force or not out.exists() or out.stat().st_mtime < src.stat().st_mtime:
        arch = "-march=native" if native else "-march=x86-64-v3"
        cmd = ["g++", "-O3", "-std=c++17", "-fopenmp", "-fPIC", arch, "-shared", "-o", str(out), str(src)]
        subprocess.check_call(cmd)
    return out


def _load(native=False):
    path = build(native=native)
    lib = ctypes.CDLL(str(path))
    for suf, fp in (("f32", _f32p), ("f64", _f64p)):
        f = getattr(lib, "wrmf_oracle_als_implicit_" + suf)
        f.restype = ctypes.c_double
        f.argtypes = [ctypes.c_int, ctypes.c_int, _i32p, _i32p, _f64p, fp, fp, fp, ctypes.c_int,
                      ctypes.c_double, ctypes.c_int, ctypes.c_uint, ctypes.c_uint,
                      ctypes.POINTER(ctypes.c_int)]
        f = getattr(lib, "wrmf_oracle_als_explicit_" + suf)
        f.restype = ctypes.c_double
        f.argtypes = [ctypes.c_int, ctypes.c_int, _i32p, _i32p, _f64p, fp, fp, fp, ctypes.c_int,
                      ctypes.c_double, ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_int,
                      ctypes.POINTER(ctypes.c_int)]
        f = getattr(lib, "wrmf_oracle_gramian_" + suf)
        f.restype = None
        f.argtypes = [fp, ctypes.c_int, ctypes.c_int64, ctypes.c_double, fp]
    lib.wrmf_oracle_max_threads.restype = ctypes.c_int
    return lib


_LIBS = {}


def lib(native=False):
    if native not in _LIBS:
        _LIBS[native] = _load(native)
    return _LIBS[native]


def _ptr(a, ty):
    return a.ctypes.data_as(ty)


def _check(a, dtype, name):
    contiguous = a.flags["F_CONTIGUOUS"] if a.ndim == 2 else a.flags["C_CONTIGUOUS"]
    if a.dtype != dtype or not contiguous:
        raise ValueError("%s must be %s and column-major/contiguous" % (name, dtype))


def als_implicit(col_ptrs, row_indices, values, X, Y, XtX, lam, solver, cg_steps=3, n_threads=1,
                 native=False, with_biases=False, is_x_bias_last_row=False, global_bias=0.0, base_out=None):
    """One implicit half-iteration (als_implicit<T>).  X: (k, n_rows) F-order, Y: (k, n_cols)
    F-order, modified in place.  dtype float32 or float64 selects T.  Returns loss/nnz.
    with_biases: the user/item-bias branch (Cholesky / NNLS), XtX is then (k-1) x (k-1)."""
    dt = X.dtype
    for a, n in ((X, "X"), (Y, "Y"), (XtX, "XtX")):
        _check(a, dt, n)
    k, n_rows = X.shape
    n_cols = Y.shape[1]
    fp = _f32p if dt == np.float32 else _f64p
    if global_bias:
        # a global bias (wrmf_implicit.hpp:108-112,146-157,228-229,262-270); conjugate gradient without user/item biases is
        # cg_solver_implicit_global_bias (:35-57,203); with them the reference cannot run (:189,197)
        if int(solver) == 1 and with_biases:
            raise NotImplementedError("CG + biases with implicit feedback cannot run in the reference (wrmf_implicit.hpp:189,197)")
        f = getattr(lib(native), "wrmf_oracle_als_implicit_gbias_" + ("f32" if dt == np.float32 else "f64"))
        f.restype = ctypes.c_double
        st = ctypes.c_int(0)
        loss = f(n_rows, n_cols, _ptr(col_ptrs, _i32p), _ptr(row_indices, _i32p), _ptr(values, _f64p), _ptr(X, fp),
                 _ptr(Y, fp), _ptr(XtX, fp), k, ctypes.c_double(lam), int(n_threads), ctypes.c_uint(solver),
                 int(bool(with_biases)), int(bool(is_x_bias_last_row)), ctypes.c_double(global_bias),
                 None if base_out is None else _ptr(base_out, fp), ctypes.byref(st), ctypes.c_uint(int(cg_steps)))
        if st.value:
            raise RuntimeError("oracle: %d singular systems" % st.value)
        return loss
    if with_biases:
        if int(solver) == 1:
            raise NotImplementedError("CG + biases with implicit feedback cannot run in the reference (wrmf_implicit.hpp:189,197)")
        assert XtX.shape == (k - 1, k - 1)
        f = getattr(lib(native), "wrmf_oracle_als_implicit_bias_" + ("f32" if dt == np.float32 else "f64"))
        f.restype = ctypes.c_double
        st = ctypes.c_int(0)
        loss = f(n_rows, n_cols, _ptr(col_ptrs, _i32p), _ptr(row_indices, _i32p), _ptr(values, _f64p), _ptr(X, fp),
                 _ptr(Y, fp), _ptr(XtX, fp), k, ctypes.c_double(lam), int(n_threads), ctypes.c_uint(solver),
                 int(bool(is_x_bias_last_row)), ctypes.byref(st))
        if st.value:
            raise RuntimeError("oracle: %d singular systems" % st.value)
        return loss
    assert Y.shape[0] == k and XtX.shape == (k, k) and len(col_ptrs) == n_cols + 1
    f = getattr(lib(native), "wrmf_oracle_als_implicit_" + ("f32" if dt == np.float32 else "f64"))
    st = ctypes.c_int(0)
    loss = f(n_rows, n_cols, _ptr(col_ptrs, _i32p), _ptr(row_indices, _i32p), _ptr(values, _f64p),
             _ptr(X, fp), _ptr(Y, fp), _ptr(XtX, fp), k, float(lam), int(n_threads), int(solver),
             int(cg_steps), ctypes.byref(st))
    if st.value:
        raise RuntimeError("oracle: %d singular systems" % st.value)
    return loss


def als_explicit(col_ptrs, row_indices, values, X, Y, cnt_X, lam, solver, cg_steps=3,
                 dynamic_lambda=True, n_threads=1, native=False, with_biases=False, is_x_bias_last_row=False):
    """One explicit half-iteration (als_explicit<T>); with_biases = the user/item-bias branch
    (wrmf_explicit.hpp:41-64,86-91,113-127), where rank counts the row of ones and the bias row."""
    dt = X.dtype
    for a, n in ((X, "X"), (Y, "Y")):
        _check(a, dt, n)
    k, n_rows = X.shape
    n_cols = Y.shape[1]
    fp = _f32p if dt == np.float32 else _f64p
    sfx = "f32" if dt == np.float32 else "f64"
    st = ctypes.c_int(0)
    cnt = np.ascontiguousarray(cnt_X, dtype=dt) if cnt_X is not None else np.zeros(n_rows, dt)
    if with_biases:
        f = getattr(lib(native), "wrmf_oracle_als_explicit_bias_" + sfx)
        f.restype = ctypes.c_double
        loss = f(n_rows, n_cols, _ptr(col_ptrs, _i32p), _ptr(row_indices, _i32p), _ptr(values, _f64p),
                 _ptr(X, fp), _ptr(Y, fp), _ptr(cnt, fp), k, ctypes.c_double(lam), int(n_threads), ctypes.c_uint(solver),
                 ctypes.c_uint(cg_steps), int(bool(dynamic_lambda)), int(bool(is_x_bias_last_row)), ctypes.byref(st))
        if st.value:
            raise RuntimeError("oracle: %d singular systems" % st.value)
        return loss
    f = getattr(lib(native), "wrmf_oracle_als_explicit_" + sfx)
    loss = f(n_rows, n_cols, _ptr(col_ptrs, _i32p), _ptr(row_indices, _i32p), _ptr(values, _f64p),
             _ptr(X, fp), _ptr(Y, fp), _ptr(cnt, fp), k, float(lam), int(n_threads), int(solver),
             int(cg_steps), int(bool(dynamic_lambda)), ctypes.byref(st))
    if st.value:
        raise RuntimeError("oracle: %d singular systems" % st.value)
    return loss


def init_biases_explicit(csc, csr, user_bias, item_bias, lam, dynamic_lambda=True, non_negative=False,
                         calculate_global_bias=False, native=False):
    """initialize_biases_explicit (wrmf_utils.hpp:32-84).  csc = (p, i, x) of users x items by item column,
    csr = the same matrix by user column; x arrays are float64 and modified in place when the global bias is
    calculated; user_bias / item_bias (float32 or float64) are filled in place.  Returns the global bias."""
    dt = user_bias.dtype
    fp = _f32p if dt == np.float32 else _f64p
    f = getattr(lib(native), "wrmf_oracle_init_biases_explicit_" + ("f32" if dt == np.float32 else "f64"))
    f.restype = ctypes.c_double
    (p1, i1, x1), (p2, i2, x2) = csc, csr
    return f(len(p1) - 1, _ptr(p1, _i32p), _ptr(i1, _i32p), _ptr(x1, _f64p), len(p2) - 1, _ptr(p2, _i32p),
             _ptr(i2, _i32p), _ptr(x2, _f64p), _ptr(user_bias, fp), _ptr(item_bias, fp), ctypes.c_double(lam),
             int(bool(dynamic_lambda)), int(bool(non_negative)), int(bool(calculate_global_bias)))


def init_biases_implicit(csc, csr, user_bias, item_bias, lam, non_negative=False, native=False,
                         calculate_global_bias=False):
    """initialize_biases_implicit (wrmf_utils.hpp:86-165); arguments as init_biases_explicit.  Returns the global bias."""
    dt = user_bias.dtype
    fp = _f32p if dt == np.float32 else _f64p
    f = getattr(lib(native), "wrmf_oracle_init_biases_implicit_" + ("f32" if dt == np.float32 else "f64"))
    f.restype = ctypes.c_double
    (p1, i1, x1), (p2, i2, x2) = csc, csr
    return f(len(p1) - 1, _ptr(p1, _i32p), _ptr(i1, _i32p), _ptr(np.ascontiguousarray(x1, dtype=np.float64), _f64p),
             len(p2) - 1, _ptr(p2, _i32p), _ptr(i2, _i32p), _ptr(np.ascontiguousarray(x2, dtype=np.float64), _f64p),
             _ptr(user_bias, fp), _ptr(item_bias, fp), ctypes.c_double(lam), int(bool(non_negative)),
             int(bool(calculate_global_bias)))


def gramian(X, lam, native=False):
    """XtX = tcrossprod(X) + fl(lambda) I   (R/model_WRMF.R:474-486)."""
    dt = X.dtype
    _check(X, dt, "X")
    k, n = X.shape
    out = np.zeros((k, k), dtype=dt, order="F")
    fp = _f32p if dt == np.float32 else _f64p
    getattr(lib(native), "wrmf_oracle_gramian_" + ("f32" if dt == np.float32 else "f64"))(
        _ptr(X, fp), k, n, float(lam), _ptr(out, fp))
    return out


# ----------------------------------------------------------------------------------------------
# independent dense-numpy statement (float64) of the per-row systems
# ----------------------------------------------------------------------------------------------

def np_half_iteration_implicit(col_ptrs, row_indices, values, X, lam):
    """Exact (float64) solution of every row's implicit normal equations and the reference loss."""
    X = np.asarray(X, dtype=np.float64)
    k = X.shape[0]
    n_cols = len(col_ptrs) - 1
    XtX = X @ X.T + np.float64(np.float32(lam)) * np.eye(k)
    Y = np.zeros((k, n_cols), order="F")
    loss = 0.0
    for i in range(n_cols):
        p1, p2 = col_ptrs[i], col_ptrs[i + 1]
        if p1 == p2:
            continue
        Xn = X[:, row_indices[p1:p2]]
        c = values[p1:p2].astype(np.float64)
        lhs = XtX + (Xn * (c - 1.0)) @ Xn.T
        y = np.linalg.solve(lhs, Xn @ c)
        Y[:, i] = y
        loss += float(((1.0 - y @ Xn) ** 2) @ c + lam * (y @ y))
    if lam > 0:
        loss += lam * float((X * X).sum())
    return Y, loss / float(col_ptrs[-1])


def np_half_iteration_explicit(col_ptrs, row_indices, values, X, lam, dynamic_lambda, cnt_X=None):
    X = np.asarray(X, dtype=np.float64)
    k = X.shape[0]
    n_cols = len(col_ptrs) - 1
    Y = np.zeros((k, n_cols), order="F")
    loss = 0.0
    for i in range(n_cols):
        p1, p2 = col_ptrs[i], col_ptrs[i + 1]
        if p1 == p2:
            continue
        Xn = X[:, row_indices[p1:p2]]
        r = values[p1:p2].astype(np.float64)
        lam_use = lam * ((p2 - p1) if dynamic_lambda else 1.0)
        y = np.linalg.solve(Xn @ Xn.T + lam_use * np.eye(k), Xn @ r)
        Y[:, i] = y
        loss += float(((r - y @ Xn) ** 2).sum() + lam_use * (y @ y))
    if lam > 0:
        if dynamic_lambda:
            loss += lam * float(((X * X) @ np.asarray(cnt_X, dtype=np.float64)).sum())
        else:
            loss += lam * float((X * X).sum())
    return Y, loss / float(col_ptrs[-1])


# ----------------------------------------------------------------------------------------------
# R6 driver semantics on top of the C++ oracle
# ----------------------------------------------------------------------------------------------

def csc_transpose(n_rows, n_cols, p, i, x):
    """CSC of the transpose (== CSR of the input reinterpreted, MatrixExtra::t_shallow(as.csr))."""
    order = np.argsort(i, kind="stable")
    ti = np.repeat(np.arange(n_cols, dtype=np.int32), np.diff(p))[order]
    tp = np.zeros(n_rows + 1, dtype=np.int32)
    np.cumsum(np.bincount(i, minlength=n_rows), out=tp[1:])
    return tp, ti.astype(np.int32), np.ascontiguousarray(x[order])


class OracleWRMF:
    """WRMF$fit_transform / $transform semantics (R/model_WRMF.R:173-360, 365-385, 412-452) for
    the no-bias implicit and explicit models, running on the C++ oracle.  `init_U` (rank x n_user)
    and optionally `init_components` replace R's RNG (large_rand_matrix, src/utils.cpp:131-143)."""

    def __init__(self, rank, lam=0.0, feedback="implicit", solver="conjugate_gradient", cg_steps=3,
                 dynamic_lambda=True, dtype=np.float64, n_threads=1, with_user_item_bias=False, with_global_bias=False):
        self.with_bias, self.with_global_bias = bool(with_user_item_bias), bool(with_global_bias)
        self.global_bias = 0.0
        rank = int(rank) + (2 if self.with_bias else 0)                            # :160
        self.rank, self.lam, self.feedback = int(rank), float(lam), feedback
        self.solver_code = {"cholesky": 0, "conjugate_gradient": 1, "nnls": 2}[solver]   # :99-100
        self.non_negative = solver == "nnls"                                      # :88
        if self.non_negative:                                                     # :90-93 nnls drops the global bias
            self.with_global_bias = False
        self.cg_steps, self.dynamic_lambda = int(cg_steps), bool(dynamic_lambda)
        self.dtype, self.n_threads = np.dtype(dtype), n_threads
        self.components = None
        self.losses = []

    def _solve(self, p, i, x, X, Y, cnt_X=None, XtX=None, avoid_cg=False, is_bias_last_row=False):
        solver = 0 if (avoid_cg and self.solver_code == 1) else self.solver_code   # :112
        if self.feedback == "implicit":
            if XtX is None:
                XX = X
                if self.with_bias:                                                  # :468-473
                    XX = np.asfortranarray(X[:-1, :] if is_bias_last_row else X[1:, :])
                XtX = gramian(XX, self.lam)                                         # :474-486
            return als_implicit(p, i, x, X, Y, XtX, self.lam, solver, self.cg_steps, self.n_threads,
                                with_biases=self.with_bias, is_x_bias_last_row=is_bias_last_row,
                                global_bias=self.global_bias)
        return als_explicit(p, i, x, X, Y, cnt_X, self.lam, solver, self.cg_steps,
                            self.dynamic_lambda, self.n_threads, with_biases=self.with_bias,
                            is_x_bias_last_row=is_bias_last_row)

    def fit_transform(self, n_user, n_item, p_ui, i_ui, x_ui, init_U, n_iter=10,
                      convergence_tol=None, init_components=None):
        if convergence_tol is None:
            convergence_tol = 0.005 if self.feedback == "implicit" else 0.001      # :173
        dt = self.dtype
        self.c_iu = csc_transpose(n_user, n_item, p_ui, i_ui, x_ui)               # :190
        U = np.asfortranarray(init_U, dtype=dt).copy(order="F")
        if init_components is not None:
            comp = np.asfortranarray(init_components, dtype=dt).copy(order="F")
        elif self.solver_code == 1:
            comp = np.zeros((self.rank, n_item), dtype=dt, order="F")               # :219-231
        else:
            raise ValueError("non-CG solvers need init_components (R draws them from its RNG)")
        if self.with_bias:                                                          # :208-245
            U[0, :] = 1.0
            comp[self.rank - 1, :] = 1.0
        if self.non_negative:                                                       # :252-255
            comp, U = np.abs(comp), np.abs(U)
        x_ui = np.array(x_ui, dtype=np.float64)           # deep copy: the global bias is removed in place (:263-264)
        self.c_iu = (self.c_iu[0], self.c_iu[1], np.array(self.c_iu[2], dtype=np.float64))
        if self.with_bias:                                                          # :259-277
            user_bias, item_bias = np.zeros(n_user, dtype=dt), np.zeros(n_item, dtype=dt)
            if self.feedback == "explicit":
                gb = init_biases_explicit((p_ui, i_ui, x_ui), self.c_iu, user_bias, item_bias, self.lam,
                                          self.dynamic_lambda, self.non_negative, self.with_global_bias)
            else:
                gb = init_biases_implicit((p_ui, i_ui, x_ui), self.c_iu, user_bias, item_bias, self.lam,
                                          self.non_negative, calculate_global_bias=self.with_global_bias)
            comp[0, :] = item_bias
            U[self.rank - 1, :] = user_bias
            if self.with_global_bias:
                self.global_bias = gb
        elif self.with_global_bias and self.feedback == "explicit":                 # :278-282
            self.global_bias = float(np.mean(x_ui))
            x_ui -= self.global_bias
            self.c_iu[2][:] -= self.global_bias
        elif self.with_global_bias:                                                 # :285-287
            sm = float(np.sum(x_ui))
            self.global_bias = sm / (sm + float(n_user) * float(n_item) - float(len(x_ui)))
        cnt_u = np.diff(p_ui).astype(dt)     # :311 -- nnz per item (named cnt_u in the reference)
        cnt_i = np.diff(self.c_iu[0]).astype(dt)
        self.cnt_u = cnt_u
        loss_prev = np.inf
        self.losses = []
        for it in range(n_iter):
            li = self._solve(p_ui, i_ui, x_ui, U, comp, cnt_X=cnt_i, is_bias_last_row=True)    # :321 items
            lu = self._solve(*self.c_iu, comp, U, cnt_X=cnt_u, is_bias_last_row=False)         # :327 users
            self.losses.append((li, lu))
            if loss_prev / lu - 1 < convergence_tol:                               # :332
                break
            loss_prev = lu
        self.components, self.U = comp, U
        XX = comp[1:, :] if self.with_bias else comp                                # :345-347
        self.XtX = gramian(np.asfortranarray(XX), self.lam) if self.feedback == "implicit" else None   # :347-353
        return self._transform(*self.c_iu)

    def _transform(self, p, i, x):                                                # :412-452
        res = np.zeros((self.rank, len(p) - 1), dtype=self.dtype, order="F")
        if self.with_bias:
            res[0, :] = 1.0                                                         # :427-429
        self._solve(p, i, x, self.components, res, cnt_X=self.cnt_u, XtX=self.XtX, avoid_cg=True,
                    is_bias_last_row=False)
        return np.ascontiguousarray(res.T)                                          # :444

    def transform(self, p_iu, i_iu, x_iu):
        """x given as CSC of x^T (items x users), i.e. CSR of the users x items matrix (:365-385)."""
        if self.global_bias != 0.0 and self.feedback == "explicit":               # :381-382
            x_iu = np.asarray(x_iu, dtype=np.float64) - self.global_bias
        return self._transform(p_iu, i_iu, x_iu)


# ----------------------------------------------------------------------------------------------
# $predict: top-k of the dense product
# ----------------------------------------------------------------------------------------------

NA_INTEGER = -2147483648


def top_product(x, y, k, nr_p=None, nr_j=None, exclude=(), glob_mean=0.0):
    """Restatement of top_product() (src/matrix_top_product.cpp:20-102) with its exact heap semantics:
    x: nr x rank, y: rank x nc (float64, like find_top_product's dbl() casts, R/utils.R:35-36);
    not_recommend as sorted CSR (nr_p, nr_j) consumed with a moving pointer (:52,70-75); `exclude` holds
    1-based item indices (:78); min-heap of (score, index) pairs, replacement only if top.first < val
    (:80-85); output filled from the end (:88-95) -> best first, equal scores with the larger index first.
    Returns (res 1-based with NA_INTEGER, scores with NaN)."""
    import heapq
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    nr, nc = x.shape[0], y.shape[1]
    res = np.full((nr, k), NA_INTEGER, dtype=np.int32)
    scores = np.full((nr, k), np.nan)
    excl = set(int(e) for e in exclude)
    filt = nr_p is not None and len(nr_p) > 0 and int(nr_p[-1]) > 0
    for j in range(nr):
        yvec = x[j] @ y
        cols = nr_j[nr_p[j]:nr_p[j + 1]] if filt else ()
        u = 0
        q = []
        for i in range(nc):
            val = float(yvec[i])
            skip = False
            if filt and len(cols) > 0 and u < len(cols):
                if i == cols[u]:
                    skip = True
                    u += 1
            if (i + 1) in excl:
                skip = True
            if len(q) < k:
                if not skip:
                    heapq.heappush(q, (val, i))
            elif q[0][0] < val and not skip:
                heapq.heapreplace(q, (val, i))
        qs = len(q)
        for t in range(qs):
            v, i = heapq.heappop(q)
            res[j, qs - t - 1] = i + 1
            scores[j, qs - t - 1] = v
    if glob_mean != 0.0:
        scores += glob_mean
    return res, scores
