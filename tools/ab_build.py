"""Build a VARIANT of libvisiondk_hip.so for kernel A/B runs on one box: one source recompiled with extra -D flags, every other object taken from the normal build.
    python tools/ab_build.py <name> <source.hip> -DFOO=1 [-DBAR ...]      ->  visiondk_amd/build/ab/lib<name>.so
Use:  VDK_HIP_LIB=visiondk_amd/build/ab/lib<name>.so python bench.py ...   (visiondk_amd/_lib.py; tuning only)."""
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import build as vb  # noqa: E402


def main():
    name, srcs, defs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]      # several sources: comma-separated (gemm_w4.hip,gemm_w4_f16.hip)
    vb.build(verbose=False)
    out = vb.OBJ / "ab"; out.mkdir(exist_ok=True)
    var = {}
    for src in srcs:
        srcp = vb.CSRC / src
        obj = out / f"{name}_{srcp.stem}.o"
        subprocess.run([vb.HIPCC, *vb.FLAGS, *defs, "-c", str(srcp), "-o", str(obj)], check=True)
        var[srcp.stem] = str(obj)
    objs = [var.get(p.stem, str(vb.OBJ / (p.stem + ".o"))) for p in vb.sources()]
    lib = out / f"lib{name}.so"
    subprocess.run([vb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *objs], check=True)
    print(lib)


if __name__ == "__main__":
    main()
