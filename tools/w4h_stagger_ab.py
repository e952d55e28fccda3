"""Timing of the 256x128 / two-workgroups-per-CU GEMM on the long-epilogue forms (set VDK_GEMM_W4H_STAGGER in the environment: percent of the estimated half period)."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402
be = _lib.load()
T = 50432
def timed(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out = []
for name, N, K, ep in [("fc1 gelu", 3072, 768, "gelu"), ("dfc2 dgelu", 3072, 768, "dgelu"), ("proj res", 768, 768, "res"), ("fc2 res", 768, 3072, "res"), ("qkv bias", 2304, 768, "bias")]:
    torch.manual_seed(0)
    a = torch.randn(T, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); bias = torch.randn(N, device="cuda")
    odt = torch.bfloat16
    if ep == "bias": kw = {"bias": bias}
    elif ep == "res": kw = {"bias": bias, "residual": torch.randn(T, N, device="cuda")}; odt = torch.float32
    elif ep == "gelu": kw = {"bias": bias, "act": ops.ACT_GELU, "aux": torch.empty(T, N, device="cuda", dtype=torch.bfloat16)}
    else:
        rows = be.lib.vdk_gemm_c_colsum_rows(T, N, K)
        kw = {"act": ops.ACT_DGELU, "aux": torch.randn(T, N, device="cuda").bfloat16(), "c_colsum": torch.empty(rows, N, device="cuda")}
    o = torch.empty(T, N, dtype=odt, device="cuda")
    be.lib.vdk_gemm_force_kernel(6)
    t = min(timed(lambda: ops.gemm_nt(a, b, out=o, backend=be, **kw)) for _ in range(3))
    out.append(f"{name} {t:.1f}")
print("stagger", os.environ.get("VDK_GEMM_W4H_STAGGER", "default"), " | ".join(out))
