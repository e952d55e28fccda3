import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def emu():
    """CPU SIMT emulation of the HIP sources (tests only; index-math checks without a GPU)."""
    from tests.emu.emu_backend import load_emu
    return load_emu()


@pytest.fixture(scope="session")
def hip():
    """The product backend: libvisiondk_hip.so on cuda:0.  Fails loudly if it is missing."""
    import torch
    from visiondk_amd import _lib
    assert torch.cuda.is_available(), "gpu test without a GPU"
    torch.cuda.set_device(0)
    return _lib.load()


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    """Every kernel parity test runs twice: on the CPU SIMT emulation (-m "not gpu") and, through the
    same C ABI, on the real gfx950 library (-m gpu)."""
    return request.getfixturevalue(request.param)


@pytest.fixture
def dev(be):
    return "cuda" if be.device_only else "cpu"
