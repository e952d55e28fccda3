"""LayerNorm backward (bf16 dy + fp32 residual gradient -> fp32 dx + bf16 copy + dgamma / dbeta partials) at the row shapes of the ViT-B/16 and swin_base steps, against its
byte floor (x f32 + dy bf16 + dres f32 in, dx f32 + dxb bf16 out = 16 B per element).   python tools/bench_ln_bwd.py"""
import json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402

be = _lib.load()
for T, C in [(401408, 128), (100352, 256), (25088, 512), (6272, 1024), (25216, 768), (50432, 768), (73856, 1024)]:
    x = torch.randn(T, C, device="cuda"); g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    _, mean, rstd = ops.layernorm_fwd(x, g, b, backend=be)
    dy = torch.randn(T, C, device="cuda").bfloat16(); dres = torch.randn(T, C, device="cuda")
    f = lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dres=dres, backend=be)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    us = sorted(ts)[2]
    # (the op wrapper allocates its outputs per call: torch's caching allocator, no device work)
    print(json.dumps({"T": T, "C": C, "us": round(us, 1), "GB_per_s": round(16.0 * T * C / us / 1e3, 1), "floor_us_at_8TBs": round(16.0 * T * C / 8e6, 1)}), flush=True)
