import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The torch-CPU oracles (test infrastructure) are what the GPU suite waits for.  A GPU box has 256 logical CPUs and torch then
    # starts as many intra-op threads: on the oracles' small convolutions that is SLOWER than a few dozen (bench.py's cpu_baseline
    # sweep finds its best rate at 16 - 32 threads on every box of the pool).  RDPN6D_TEST_THREADS overrides; boxes with fewer cores keep
    # their own default.
    try:
        import torch

        want = int(os.environ.get("RDPN6D_TEST_THREADS", "32"))
        if want > 0 and torch.get_num_threads() > want:
            torch.set_num_threads(want)
    except Exception:  # noqa: BLE001  (a box without torch still collects the pure-C oracle tests)
        pass


@pytest.fixture(scope="session")
def oracle_lib():
    """The C oracle (test infrastructure); built on demand with plain gcc."""
    import ctypes

    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("fps_oracle.c", "ransac_oracle.c", "pnp_oracle.c")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True, capture_output=True)
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
