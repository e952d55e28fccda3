"""Build libvisiondk_hip.so (gfx950) in-tree with hipcc.

    python -m visiondk_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container as well as on the
MI355X box.  The shared object lands next to this file (git-ignored, but it travels with gpurun).
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
ROOT = PKG.parent
LIB = PKG / "libvisiondk_hip.so"
OBJ = PKG / "build"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
    f"-I{ROOT / 'include'}", f"-I{CSRC}",
]


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _digest(paths: list[Path]) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    srcs = sources()
    hdrs = sorted(CSRC.glob("*.h")) + sorted((ROOT / "include").glob("*.h"))
    OBJ.mkdir(exist_ok=True)
    stamp = OBJ / "stamp.txt"
    dig = _digest(srcs + hdrs)
    if LIB.exists() and not force and stamp.exists() and stamp.read_text() == dig:
        return LIB
    hdr_dig = _digest(hdrs)

    def compile_one(src: Path) -> Path:
        obj = OBJ / (src.stem + ".o")
        tag = OBJ / (src.stem + ".tag")
        body = src.read_bytes()
        for inc in re.findall(rb'#include "([^"]+\.hip)"', body):      # a translation unit that instantiates another source file (gemm_w4_f16.hip) follows its edits
            body += (CSRC / inc.decode()).read_bytes()
        d = hashlib.sha256(body + hdr_dig.encode()).hexdigest()
        if obj.exists() and tag.exists() and tag.read_text() == d and not force:
            return obj
        cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print("[visiondk build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError(f"hipcc failed on {src.name}")
        if r.stderr.strip() and verbose:
            sys.stderr.write(r.stderr)
        tag.write_text(d)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(srcs)))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    if verbose:
        print("[visiondk build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("link failed")
    stamp.write_text(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
