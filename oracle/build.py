"""TEST INFRASTRUCTURE: compile oracle/*.c (the CPU restatement of the reference's hot path) into
oracle/liboracle.so with gcc.  Used by tests/, bench.py's cpu_baseline leg and smoke() only."""
from __future__ import annotations

import hashlib
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB = HERE / "liboracle.so"
FLAGS = ["-O3", "-mavx2", "-mfma", "-fopenmp", "-fPIC", "-shared", "-ffp-contract=off", "-fno-math-errno",
         "-fno-trapping-math", "-std=c11"]


def build(force: bool = False) -> Path:
    srcs = sorted(HERE.glob("*.c"))
    dig = hashlib.sha256(b"".join(p.read_bytes() for p in srcs) + " ".join(FLAGS).encode()).hexdigest()
    stamp = HERE / ".liboracle.stamp"
    if LIB.exists() and stamp.exists() and stamp.read_text() == dig and not force:
        return LIB
    r = subprocess.run(["gcc", *FLAGS, "-o", str(LIB), *map(str, srcs), "-lm"], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("oracle build failed")
    stamp.write_text(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
