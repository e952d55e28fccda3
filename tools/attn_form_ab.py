"""Same-box A/B of the attention backward forms at ViT-B/16 geometry (B*H = 3072 items of N = 197): VDK_ATTN_BWD_FORM = 3 (two kernels, default) against 4 (one pass).
    python tools/attn_form_ab.py [B N H]   ->  one JSON line {form3_us, form4_us, dq_rel, dkdv_equal}"""
import json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from visiondk_amd import _lib, ops  # noqa: E402

B, N, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 197, 12)
be = _lib.load()
D = H * 64
torch.manual_seed(0)
qkv = torch.randn(B, N, 3 * D, device="cuda").bfloat16(); dout = torch.randn(B, N, D, device="cuda").bfloat16()
o, lse = ops.attention_fwd(qkv, H, backend=be)
out, res = {}, {}
for form in ("3", "4", "3", "4"):
    os.environ["VDK_ATTN_BWD_FORM"] = form
    fn = lambda: ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
    res[form] = fn()
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    out.setdefault(f"form{form}_us", []).append(e0.elapsed_time(e1) / 20 * 1e3)
a, b = res["3"].float(), res["4"].float()
out["dq_rel"] = ((a[..., :D] - b[..., :D]).norm() / a[..., :D].norm()).item()
out["dkdv_equal"] = bool(torch.equal(res["3"][..., D:], res["4"][..., D:]))
print(json.dumps(out))
