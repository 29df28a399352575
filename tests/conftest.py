import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def orc():
    o = entry.load_oracle()
    o.lib()
    return o


def small_graph(rng, n, e, hub=None, empty=(), n_src=None):
    """Random CSR (rows = destinations) with optional hub row degree and forced-empty rows."""
    n_src = n if n_src is None else n_src
    w = rng.random(n) + 0.05
    w[list(empty)] = 0.0
    deg = rng.multinomial(e, w / w.sum())
    if hub is not None:
        r, d = hub
        deg[r] = d
    for r in empty:
        deg[r] = 0
    row_ptr = np.zeros(n + 1, np.int32)
    row_ptr[1:] = np.cumsum(deg)
    col = np.concatenate([np.sort(rng.integers(0, n_src, size=int(d))) for d in deg]).astype(np.int32) \
        if row_ptr[-1] > 0 else np.zeros(0, np.int32)
    return row_ptr, col


def grad_close(got, want, tol=1e-3, frac=0.005):
    """Gradient comparison that tolerates LeakyReLU-kink hits: LReLU'(s) is discontinuous at s = 0
    (1 vs 0.01, E:774/855), so when some |s| is ~1e-8 two correct fp32 evaluation orders may land on
    different sides and a handful of gradient entries move by O(1e-3) of the tensor's scale.  Accept
    if all but `frac` of the entries agree to `tol` of max|want| and the relative L2 error is small."""
    got = np.asarray(got, np.float64).ravel(); want = np.asarray(want, np.float64).ravel()
    scale = max(1e-12, np.abs(want).max())
    err = np.abs(got - want)
    l2 = np.linalg.norm(err) / max(1e-12, np.linalg.norm(want))
    return bool((err > tol * scale).mean() <= frac and l2 < max(10 * tol, 5e-3)), (float(err.max() / scale), float(l2))
