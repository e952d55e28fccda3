"""TEST INFRASTRUCTURE ONLY: load the CPU SIMT emulation of the HIP sources as a `Backend` that accepts CPU
tensors.  The product (`visiondk_amd._lib.load`) can never return this object."""
from __future__ import annotations

import ctypes as C

from visiondk_amd._lib import Backend

from .build_emu import build

_emu = None


def load_emu() -> Backend:
    global _emu
    if _emu is None:
        _emu = Backend(C.CDLL(str(build())), device_only=False, name="emu")
    return _emu
