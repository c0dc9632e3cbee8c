import os
import sys

import pytest

# the oracle's OpenMP loops are small: on a many-core GPU host the default (one thread per core) is slower, not faster
os.environ.setdefault("OMP_NUM_THREADS", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def emu():
    import emu_lib
    return emu_lib.load()


@pytest.fixture(scope="session")
def gpu_ctx():
    import eth_lc_plonky2_amd as m
    ctx = m.Context(0)  # raises (no CPU fallback) when the HIP extension or the device is missing
    yield ctx
    ctx.close()
