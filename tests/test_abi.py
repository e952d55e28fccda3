"""The C ABI: every function include/visiondk.h declares is (a) in the ctypes signature table and (b) exported by the built
shared objects (gfx950 product library — loads without a GPU — and the test-only emulation)."""
import ctypes
import re
import subprocess
from pathlib import Path

from visiondk_amd import _abi

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    txt = (ROOT / "include" / "visiondk.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vdk_[a-z0-9_]+)\s*\(", txt)) - {"vdk_grad_ready_fn"})


def test_header_matches_signature_table():
    dec = _declared()
    assert len(dec) >= 30
    assert dec == sorted(_abi.SIGNATURES.keys())


def test_product_library_exports_every_symbol():
    from visiondk_amd import build
    lib = build.build(verbose=False)
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (vdk_[a-z0-9_]+)", out))
    for name in _declared():
        assert name in exported, name
    cdll = ctypes.CDLL(str(lib))           # loads without a GPU; no compute call is made here
    assert _abi.bind(cdll) == list(_abi.SIGNATURES.keys())
    assert cdll.vdk_is_device_build() == 1 and cdll.vdk_abi_version() == 1


def test_product_loader_refuses_cpu_tensors_and_emulation(emu):
    import pytest
    import torch
    from visiondk_amd import _lib
    with pytest.raises(RuntimeError):
        _lib.Backend(emu.lib, device_only=True)              # a non-device build can never be the product backend
    be = _lib.Backend(ctypes.CDLL(str(_lib.LIB_PATH)), device_only=True)
    with pytest.raises(RuntimeError):
        be.ptr(torch.zeros(4))                               # no CPU fallback
