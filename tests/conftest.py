import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a visible MI355X and the built extension: skip them (instead of failing) on a CPU box."""
    import torch
    so = ROOT / "rsparse_amd" / "lib" / "librsparse_wrmf_hip.so"
    if torch.cuda.device_count() > 0 and so.exists():
        return
    skip = pytest.mark.skip(reason="needs a GPU and rsparse_amd/lib/librsparse_wrmf_hip.so")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_movielens():
    d = np.load(GOLDEN / "movielens100k_csc.npz")
    return int(d["Dim"][0]), int(d["Dim"][1]), d["p"].astype(np.int32), d["i"].astype(np.int32), d["x"].astype(np.float64)


def csc_take_rows(n_rows_keep, p, i, x):
    """dgCMatrix[1:n, ] -- keep rows < n_rows_keep (tests/testthat/test-wrmf.R:6 `train`)."""
    keep = i < n_rows_keep
    cs = np.concatenate([[0], np.cumsum(keep)]).astype(np.int64)
    newp = cs[p].astype(np.int32)
    return newp, i[keep].astype(np.int32), x[keep]


def csc_drop_rows(n_first, p, i, x):
    """dgCMatrix[(n+1):nrow, ] -- rows >= n_first, re-based to 0 (`cv` in test-wrmf.R:7)."""
    keep = i >= n_first
    cs = np.concatenate([[0], np.cumsum(keep)]).astype(np.int64)
    newp = cs[p].astype(np.int32)
    return newp, (i[keep] - n_first).astype(np.int32), x[keep]


@pytest.fixture(scope="session")
def movielens():
    return load_movielens()


@pytest.fixture(scope="session")
def ml_train(movielens):
    n_user, n_item, p, i, x = movielens
    tp, ti, tx = csc_take_rows(900, p, i, x)
    return 900, n_item, tp, ti, tx


def rel_fro(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
