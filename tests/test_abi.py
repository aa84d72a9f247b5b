"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/rsparse_wrmf_hip.h declares, and rejects bad / unsupported calls with status codes before
touching a device (no compute without a GPU)."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

from rsparse_amd import _lib

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "rsparse_wrmf_hip.h").read_text()


def declared_symbols():
    return sorted(set(re.findall(r"\b(rsparse_hip_[a-z0-9_]+)\s*\(", HEADER)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names      # python binding covers the header one-to-one
    assert lib.rsparse_hip_abi_version() == 6


def test_header_cites_reference_interfaces():
    for needle in ("src/wrmf_implicit.cpp", "src/wrmf_explicit.cpp", "src/RcppExports.cpp",
                   "R/model_WRMF.R:474-486", "inst/include/mapped_csc.hpp"):
        assert needle in HEADER


def _tiny():
    p = np.array([0, 2, 3], dtype=np.int32)
    i = np.array([0, 1, 1], dtype=np.int32)
    x = np.array([1.0, 2.0, 3.0])
    X = np.asfortranarray(np.ones((4, 2), dtype=np.float32))
    Y = np.asfortranarray(np.zeros((4, 2), dtype=np.float32))
    G = np.asfortranarray(np.eye(4, dtype=np.float32))
    return p, i, x, X, Y, G


def _call_implicit(lib, p, i, x, X, Y, G, rank=4, solver=1, with_biases=0, global_bias=0.0):
    loss = ctypes.c_double(-1)
    vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.rsparse_hip_als_implicit_float(2, 2, vp(p), vp(i), vp(x), vp(X), vp(Y), vp(G), rank, 0.1, 1, solver, 3,
                                            with_biases, 0, global_bias, None, 0, 0, ctypes.addressof(loss))
    return rc, lib.rsparse_hip_last_error().decode()


def test_status_codes_without_device():
    lib = _lib.load()
    p, i, x, X, Y, G = _tiny()
    rc, msg = _call_implicit(lib, p, i, x, X, Y, G, rank=0)
    assert rc == _lib.ERR_INVALID and "rank" in msg
    rc, msg = _call_implicit(lib, p, i, x, None, Y, G)
    assert rc == _lib.ERR_INVALID
    rc, msg = _call_implicit(lib, p, i, x, X, Y, G, rank=257)     # ranks up to RSPARSE_HIP_MAX_RANK = 256 are on the device
    assert rc == _lib.ERR_UNSUPPORTED
    rc, msg = _call_implicit(lib, p, i, x, X, Y, G, solver=2)          # nnls is a device solver: past the argument
    assert rc not in (_lib.ERR_UNSUPPORTED, _lib.ERR_INVALID)         # checks (no device here -> runtime error)
    rc, msg = _call_implicit(lib, p, i, x, X, Y, G, with_biases=1)
    assert rc == _lib.ERR_UNSUPPORTED and "bias" in msg
    rc, msg = _call_implicit(lib, p, i, x, X, Y, G, global_bias=0.3)   # cg_solver_implicit_global_bias is on the device
    assert rc not in (_lib.ERR_UNSUPPORTED, _lib.ERR_INVALID)
    rc, msg = _call_implicit(lib, p, i, x, X, Y, G, solver=7)
    assert rc == _lib.ERR_INVALID
    with pytest.raises(_lib.UnsupportedOnDevice):
        _lib.check(_lib.ERR_UNSUPPORTED)
    assert issubclass(_lib.UnsupportedOnDevice, NotImplementedError)


def test_wrapper_argument_checks():
    from rsparse_amd import als
    p, i, x, X, Y, G = _tiny()
    with pytest.raises(ValueError):
        als.als_implicit((2, 2, p, i, x), np.ascontiguousarray(np.ones((2, 4), np.float32)).T[:, :1], Y, 0.1, 1, 1, 3,
                         "float", False, False, XtX=G)
    with pytest.raises(ValueError):   # wrong precision dtype
        als.als_implicit((2, 2, p, i, x), X, Y, 0.1, 1, 1, 3, "double", False, False, XtX=G)
    with pytest.raises(_lib.UnsupportedOnDevice):
        als.als_implicit((2, 2, p, i, x), X, Y, 0.1, 1, 1, 3, "float", True, False, XtX=G)


def test_wrmf_constructor_mirrors_reference_validation():
    from rsparse_amd import WRMF
    with pytest.raises(ValueError):
        WRMF(solver="lbfgs")
    with pytest.raises(ValueError):
        WRMF(feedback="both")
    with pytest.raises(TypeError):
        WRMF(cg_steps=3.0)
    with pytest.raises(TypeError):
        WRMF(init=[[1.0]])
    assert WRMF(solver="nnls")._non_negative                       # R/model_WRMF.R:88
    with pytest.raises(NotImplementedError):
        WRMF(with_user_item_bias=True)
    m = WRMF(rank=8, lambda_=0.1, solver="cholesky", precision="float")
    assert m.components is None and m.global_bias == 0.0
    with pytest.raises(RuntimeError):
        m.transform(np.zeros((1, 1)))
