import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)

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


@pytest.fixture(autouse=True)
def _parity_test_name(request):
    """Achieved errors are recorded per test (tests/parity.py) and written out at the end of the session."""
    import parity
    parity.set_test(request.node.nodeid)
    yield


def pytest_sessionfinish(session, exitstatus):
    import parity
    parity.flush()


def run_snippets_parallel(jobs, workers=5, timeout=900):
    """jobs: {key: (python code, env overrides)} -> {key: CompletedProcess}.  The switch / fuzz matrices run one subprocess per
    setting (the switches are read once per process); `workers` of them at a time share the GPU (the box allows 6 processes on its
    card) — the matrices are bound by the CPU oracle and the fp64 reference, not by the GPU."""
    import concurrent.futures as cf
    import subprocess
    import sys as _sys

    def one(item):
        key, (code, env) = item
        # small cases: the oracle's OpenMP loops are microseconds of work each — with one thread per host core (hundreds) on a box
        # that grants this job a fraction of them, a setting took ~38 s of thread start-up instead of ~2 s
        env = dict({"OMP_NUM_THREADS": "4", "OPENBLAS_NUM_THREADS": "4", "MKL_NUM_THREADS": "4"}, **env)
        try:
            return key, subprocess.run([_sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=timeout)
        except subprocess.TimeoutExpired as e:
            return key, subprocess.CompletedProcess(e.cmd, 124, stdout=(e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or ""),
                                                    stderr="TIMEOUT after %d s" % timeout)
    with cf.ThreadPoolExecutor(max_workers=workers) as ex:
        return dict(ex.map(one, list(jobs.items())))
