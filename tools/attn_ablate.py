"""timing-only ablation of the fused attention backward (VDK_ATTN_DBG mask: results are WRONG with any bit set)"""
import json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from visiondk_amd import _lib, ops
be = _lib.load()
B, N, H = 256, 197, 12
D = H * 64
qkv = torch.randn(B, N, 3 * D, device="cuda").bfloat16(); dout = torch.randn(B, N, D, device="cuda").bfloat16()
o, lse = ops.attention_fwd(qkv, H, backend=be)
out = {}
for name, m in (("full", 0), ("no_rmw", 1), ("no_dq", 2), ("no_dkdv_mfma", 4), ("no_dq_no_dkdv", 6), ("no_step", 8), ("no_dq_out", 16), ("no_dkdv_out", 32), ("no_out", 48), ("no_D", 64), ("no_zero", 128),
                ("only_loads", 8 | 48 | 64 | 128), ("only_step", 48 | 64 | 128)):
    os.environ["VDK_ATTN_DBG"] = str(m)
    for _ in range(2): ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
    e1.record(); torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) / 10 * 1e3, 1)
print(json.dumps(out))
