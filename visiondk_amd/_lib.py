"""Loader for libvisiondk_hip.so (the gfx950 HIP library behind include/visiondk.h).

There is NO CPU fallback: if the shared object is missing or was not built for the device, importing the
product path raises.  (tests/emu/ builds a CPU SIMT emulation of the same sources for index-math checks in
the GPU-less container; it is constructed explicitly by the tests through `Backend(..., device_only=False)`
and is never reachable from here.)
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import torch

from . import _abi

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "libvisiondk_hip.so"


class Backend:
    """A loaded C-ABI library + the rule for which tensors it may touch and which stream it launches on."""

    def __init__(self, lib: C.CDLL, device_only: bool = True, name: str = "hip"):
        self.lib = lib
        self.device_only = device_only
        self.name = name
        _abi.bind(lib)
        if device_only and lib.vdk_is_device_build() != 1:
            raise RuntimeError("refusing to use a non-device build of libvisiondk as the product backend")

    def __deepcopy__(self, memo):  # ModelEMA deep-copies the model (models/ema.py:22); the library handle is shared
        return self

    # ---- plumbing -------------------------------------------------------------------------
    def ptr(self, t: torch.Tensor | None) -> int | None:
        if t is None:
            return None
        if self.device_only and not t.is_cuda:
            raise RuntimeError("visiondk_amd: HIP backend got a CPU tensor (there is no CPU fallback)")
        if not t.is_contiguous():
            raise RuntimeError("visiondk_amd: tensor must be contiguous")
        return t.data_ptr()

    def stream(self) -> int | None:
        if self.device_only:
            return torch.cuda.current_stream().cuda_stream
        return None

    def check(self, rc: int, what: str) -> None:
        _abi.check(self.lib, rc, what)

    def empty(self, shape, dtype, like: torch.Tensor | None = None, device=None) -> torch.Tensor:
        if device is None:
            device = like.device if like is not None else ("cuda" if self.device_only else "cpu")
        return torch.empty(shape, dtype=dtype, device=device)


_default: Backend | None = None


def load() -> Backend:
    """The product backend.  Fails loudly when the HIP library is absent."""
    global _default
    if _default is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m visiondk_amd.build` "
                "(visiondk_amd has no CPU fallback)")
        import os
        alt = os.environ.get("VDK_HIP_LIB")      # tuning only: another build of the SAME library (kernel A/B on one box, tools/ab_build.py)
        _default = Backend(C.CDLL(alt if alt else str(LIB_PATH)), device_only=True, name="hip")
    return _default
