import os
import sys

import numpy as np
import pytest

# torch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  If
# libmse_hip.so is loaded first it binds the system runtime and a later `import torch` brings a SECOND
# runtime into the process, which then sees no GPU.  Importing torch first makes both share one runtime.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

SEED_BASE, SEED_QUERY, SEED_CENTRES = 0x5EED0001, 0x5EED0002, 0x5EED0003


_SEGV_C = r"""
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_segv(int sig) {
    static const char msg[] = "\n=== MSE_TEST_SEGV_TRACE: native frames of the crashing thread ===\n";
    void* frames[64];
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, backtrace(frames, 64), 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
void mse_test_install_segv_trace(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_segv;
    sa.sa_flags = SA_NODEFER | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
}
"""


def _install_segv_trace():
    """MSE_TEST_SEGV_TRACE=1: a crash inside a native library (ours, the HIP runtime, RCCL) prints the crashing thread's native
    frames -- library names and offsets -- before the process dies (Python's faulthandler only knows the Python threads)."""
    import ctypes
    import subprocess
    import tempfile
    d = tempfile.mkdtemp(prefix="mse_segv_")
    src, lib = os.path.join(d, "segv.c"), os.path.join(d, "libsegv.so")
    open(src, "w").write(_SEGV_C)
    subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", lib, src])
    ctypes.CDLL(lib).mse_test_install_segv_trace()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    if os.environ.get("MSE_TEST_SEGV_TRACE"):
        _install_segv_trace()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): built on demand from oracle/mse_oracle.c."""
    from oracle import orc as o
    o.build()
    return o


@pytest.fixture(scope="session")
def mse():
    """The product package; its compute calls go through libmse_hip.so only."""
    import mse as m
    return m


@pytest.fixture(scope="session")
def gpu(mse):
    from mse import ffi
    n = ffi.lib().mse_device_count()
    if n <= 0:
        pytest.fail("this test is marked gpu but no HIP device is visible")
    return n


def make_pq(orc, d=1152, dpc=18, n_centroids=256, seed=7, sample=None):
    """Small synthetic OPQ codec: random orthonormal transform + centroids drawn from data-like noise."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((d, d)).astype(np.float64)
    qm, _ = np.linalg.qr(a)
    transform = qm.astype(np.float32)
    centroids = (rng.standard_normal((n_centroids, d)) / np.sqrt(d)).astype(np.float32)
    return centroids, transform, dpc, d
