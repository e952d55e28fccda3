"""Round 6: the ViT-B/16 Linear shapes on the persistent four-wave kernel (force 0 = the dispatcher's choice) and on the two-workgroup 256x128 kernel (force 6), for
process-level A/B of library variants (VDK_HIP_LIB):   python tools/r6_w4h_ab.py"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402

be = _lib.load()
T, dt = 50432, torch.float16
out = {}
for name, M, N, K, ep in [("fc1", T, 3072, 768, "gelud"), ("dfc2", T, 3072, 768, "mulaux"), ("proj", T, 768, 768, "res"), ("fc2", T, 768, 3072, "res"), ("qkv", T, 2304, 768, "bias")]:
    torch.manual_seed(0)
    a = torch.randn(M, K, device="cuda").to(dt); b = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    bias = torch.randn(N, device="cuda")
    kw = {}; odt = dt
    if ep == "bias":
        kw = {"bias": bias}
    elif ep == "res":
        kw = {"bias": bias, "residual": torch.randn(M, N, device="cuda")}; odt = torch.float32
    elif ep == "gelud":
        kw = {"bias": bias, "act": ops.ACT_GELU_SAVE_GRAD, "aux": torch.empty(M, N, device="cuda", dtype=dt)}
    elif ep == "mulaux":
        rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
        kw = {"act": ops.ACT_MUL_AUX, "aux": torch.randn(M, N, device="cuda").to(dt), "c_colsum": torch.empty(rows, N, device="cuda")}
    o = torch.empty(M, N, dtype=odt, device="cuda")
    for fk in (0, 6):
        be.lib.vdk_gemm_force_kernel(fk)
        fn = lambda: ops.gemm_nt(a, b, out=o, backend=be, **kw)
        for _ in range(5):
            fn()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        out[f"{name}_k{fk}"] = round(sorted(ts)[2], 1)
    be.lib.vdk_gemm_force_kernel(0)
print(json.dumps(out))
