"""Host-buffer wrappers with the reference's wrapper signatures.

`als_implicit()` / `als_explicit()` mirror the R functions of the same name
(R/model_WRMF.R:456-496, :498-515): same argument meaning, `Y` modified in place, loss returned,
the k x k Gramian `tcrossprod(X) + fl(lambda) I` computed here when `XtX` is not supplied
(:474-486) -- but every numeric step runs in librsparse_wrmf_hip.so through the stateless C-ABI
entry points that replace `.Call(_rsparse_als_*)` (R/RcppExports.R:88-102).

Matrices are numpy arrays in the reference's layout: X (rank x n) and Y (rank x m) column-major
(Fortran order), CSC slots p / i / x exactly as a dgCMatrix holds them.
"""
import ctypes

import numpy as np

from . import _lib


def _f_contig(a, dtype, name):
    if not isinstance(a, np.ndarray) or a.dtype != dtype or not a.flags["F_CONTIGUOUS"]:
        raise ValueError("%s must be a column-major numpy array of dtype %s" % (name, np.dtype(dtype)))
    return a


def _csc_slots(x):
    """Accepts a scipy.sparse CSC matrix or a (n_rows, n_cols, p, i, x) tuple."""
    if isinstance(x, tuple):
        n_rows, n_cols, p, i, v = x
    else:
        x = x.tocsc()
        n_rows, n_cols = x.shape
        p, i, v = x.indptr, x.indices, x.data
    p = np.ascontiguousarray(p, dtype=np.int32)
    i = np.ascontiguousarray(i, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.float64)   # dgCMatrix@x is always double (src/utils.cpp:71)
    return int(n_rows), int(n_cols), p, i, v


def _vp(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def gramian(X, lambda_, precision):
    """XtX = tcrossprod(X) + fl(diag(lambda))  (R/model_WRMF.R:474-486) on the device."""
    lib = _lib.load()
    dt = np.float32 if precision == "float" else np.float64
    _f_contig(X, dt, "X")
    k, n = X.shape
    out = np.zeros((k, k), dtype=dt, order="F")
    fn = lib.rsparse_hip_gramian_float if dt == np.float32 else lib.rsparse_hip_gramian_double
    _lib.check(fn(_vp(X), k, n, float(lambda_), _vp(out)))
    return out


def als_implicit(x, X, Y, lambda_, n_threads, solver_code, cg_steps, precision, with_user_item_bias,
                 is_bias_last_row, initialize_bias_base=True, global_bias=0.0, XtX=None,
                 global_bias_base=None):
    """R/model_WRMF.R:456-496.  Returns loss / nnz; Y is modified in place."""
    lib = _lib.load()
    dt = np.float32 if precision == "float" else np.float64
    n_rows, n_cols, p, i, v = _csc_slots(x)
    _f_contig(X, dt, "X")
    _f_contig(Y, dt, "Y")
    rank = X.shape[0]
    if X.shape[1] != n_rows or Y.shape != (rank, n_cols):
        raise ValueError("X must be rank x nrow(x) and Y rank x ncol(x)")
    bias = bool(with_user_item_bias)
    fn = lib.rsparse_hip_als_implicit_float if precision == "float" else lib.rsparse_hip_als_implicit_double
    if int(solver_code) == 1 and bias:
        # the C ABI reports this as UNSUPPORTED (include/rsparse_wrmf_hip.h); ask it before computing a Gramian
        _lib.check(fn(n_rows, n_cols, _vp(p), _vp(i), _vp(v), _vp(X), _vp(Y), None, rank, float(lambda_),
                      int(n_threads), int(solver_code), int(cg_steps), int(bias), int(bool(is_bias_last_row)),
                      float(global_bias), None, 0, int(bool(initialize_bias_base)), None))
    if XtX is None:
        # R/model_WRMF.R:474-486: with biases the x_bias row is discarded before tcrossprod
        XX = X
        if bias:
            XX = np.asfortranarray(X[:-1, :] if is_bias_last_row else X[1:, :])
        XtX = gramian(XX, lambda_, precision)
    _f_contig(XtX, dt, "XtX")
    if global_bias_base is not None:
        # self$global_bias_base: numeric(rank - 1) / float(rank - 1) in the R driver (R/model_WRMF.R:291-296) although the
        # vector has `rank` entries; any length is accepted and its length is passed on (the library never touches more)
        if not isinstance(global_bias_base, np.ndarray) or global_bias_base.dtype != dt or global_bias_base.ndim != 1:
            raise ValueError("global_bias_base must be a numpy vector of dtype %s" % np.dtype(dt))
    loss = ctypes.c_double(0.0)
    _lib.check(fn(n_rows, n_cols, _vp(p), _vp(i), _vp(v), _vp(X), _vp(Y), _vp(XtX), rank, float(lambda_),
                  int(n_threads), int(solver_code), int(cg_steps), int(bias), int(bool(is_bias_last_row)),
                  float(global_bias), _vp(global_bias_base), 0 if global_bias_base is None else int(global_bias_base.size),
                  int(bool(initialize_bias_base)), ctypes.addressof(loss)))
    return loss.value


def initialize_biases(m_csc, m_csr, user_bias, item_bias, lambda_, dynamic_lambda, non_negative,
                      calculate_global_bias, is_explicit_feedback=False):
    """R/RcppExports.R:104-110 (`initialize_biases_double` / `_float`), src/wrmf_init.cpp:5-34.  m_csc: users x items
    (columns = items), m_csr: the same matrix transposed (columns = users), both as CSC slots; user_bias / item_bias
    are overwritten; with explicit feedback and calculate_global_bias the global mean leaves BOTH matrices' values in
    place, as in the reference (wrmf_utils.hpp:41-52) -- pass (n_rows, n_cols, p, i, x) tuples with float64 x to see it.
    Returns the global bias."""
    lib = _lib.load()
    dt = user_bias.dtype
    if dt not in (np.float32, np.float64) or item_bias.dtype != dt:
        raise ValueError("user_bias and item_bias must both be float32 or both float64")
    n_users, n_items, p1, i1, v1 = _csc_slots(m_csc)
    n2r, n2c, p2, i2, v2 = _csc_slots(m_csr)
    if (n2r, n2c) != (n_items, n_users) or v1.size != v2.size:
        raise ValueError("m_csr must be the transpose of m_csc")
    if user_bias.size != n_users or item_bias.size != n_items:
        raise ValueError("user_bias / item_bias lengths must be nrow / ncol of m_csc")
    gb = ctypes.c_double(0.0)
    fn = lib.rsparse_hip_initialize_biases_float if dt == np.float32 else lib.rsparse_hip_initialize_biases_double
    _lib.check(fn(n_users, n_items, _vp(p1), _vp(i1), _vp(v1), _vp(p2), _vp(i2), _vp(v2), _vp(user_bias), _vp(item_bias),
                  float(lambda_), int(bool(dynamic_lambda)), int(bool(non_negative)), int(bool(calculate_global_bias)),
                  int(bool(is_explicit_feedback)), ctypes.byref(gb)))
    return gb.value


def als_explicit(x, X, Y, cnt_X, lambda_, n_threads, solver_code, cg_steps, dynamic_lambda, precision,
                 with_user_item_bias, is_bias_last_row):
    """R/model_WRMF.R:498-515.  Returns loss / nnz; Y is modified in place."""
    lib = _lib.load()
    dt = np.float32 if precision == "float" else np.float64
    n_rows, n_cols, p, i, v = _csc_slots(x)
    _f_contig(X, dt, "X")
    _f_contig(Y, dt, "Y")
    rank = X.shape[0]
    if X.shape[1] != n_rows or Y.shape != (rank, n_cols):
        raise ValueError("X must be rank x nrow(x) and Y rank x ncol(x)")
    cnt = None if cnt_X is None else np.ascontiguousarray(cnt_X, dtype=dt)
    loss = ctypes.c_double(0.0)
    fn = lib.rsparse_hip_als_explicit_float if precision == "float" else lib.rsparse_hip_als_explicit_double
    _lib.check(fn(n_rows, n_cols, _vp(p), _vp(i), _vp(v), _vp(X), _vp(Y), _vp(cnt), rank, float(lambda_),
                  int(n_threads), int(solver_code), int(cg_steps), int(bool(dynamic_lambda)),
                  int(bool(with_user_item_bias)), int(bool(is_bias_last_row)), ctypes.addressof(loss)))
    return loss.value
