import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def hip_ctx():
    """A midas_snps context on GPU 0.  No CPU fallback: the test fails if the HIP path cannot run."""
    from midas_amd import abi
    ctx = abi.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def thr_default():
    from midas_amd import abi
    return abi.Thresholds.from_args(abi.DEFAULT_ARGS)
