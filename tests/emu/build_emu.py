"""TEST INFRASTRUCTURE ONLY: compile the HIP sources with the host clang++ against the SIMT emulator
(tests/emu/hip_emu.h) into tests/emu/libvisiondk_emu.so.  Never loaded by visiondk_amd itself."""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "visiondk_amd" / "csrc"
LIB = HERE / "libvisiondk_emu.so"
OBJ = HERE / "build"
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-mfma", "-mavx2", "-w",
         "-include", str(HERE / "hip_emu.h"), f"-I{HERE}", f"-I{CSRC}", f"-I{ROOT / 'include'}"]


def build(force: bool = False) -> Path:
    """(pytest -n: several workers call this at once -- one builds under an exclusive file lock, the others wait and find everything up to date; the library is linked to
    a temporary name and renamed, so nobody maps a half-written file)"""
    import fcntl
    OBJ.mkdir(exist_ok=True)
    with open(OBJ / ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return _build_locked(force)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def _build_locked(force: bool) -> Path:
    srcs = sorted(CSRC.glob("*.hip")) + [HERE / "hip_emu.cpp"]
    hdrs = sorted(CSRC.glob("*.h")) + [HERE / "hip_emu.h"] + sorted((ROOT / "include").glob("*.h"))
    OBJ.mkdir(exist_ok=True)
    hd = hashlib.sha256(b"".join(p.read_bytes() for p in hdrs) + " ".join(FLAGS).encode()).hexdigest()

    def one(src: Path) -> tuple[Path, bool]:
        obj = OBJ / (src.stem + ".o")
        tag = OBJ / (src.stem + ".tag")
        body = src.read_bytes()
        for inc in re.findall(rb'#include "([^"]+\.hip)"', body):
            body += (CSRC / inc.decode()).read_bytes()
        d = hashlib.sha256(body + hd.encode()).hexdigest()
        if obj.exists() and tag.exists() and tag.read_text() == d and not force:
            return obj, False
        r = subprocess.run([CXX, *FLAGS, "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError(f"emu compile failed: {src.name}")
        tag.write_text(d)
        return obj, True

    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(one, srcs))
    if LIB.exists() and not any(ch for _, ch in res) and not force:
        return LIB
    tmp = LIB.with_suffix(f".so.tmp{os.getpid()}")
    r = subprocess.run([CXX, "-shared", "-fPIC", "-o", str(tmp), *[str(o) for o, _ in res], "-lpthread"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("emu link failed")
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
