"""Same-box A/B of the attention backward forms at ViT-B/16 geometry (B*H = 3072 items of N = 197): VDK_ATTN_BWD_FORM = 3 (two kernels) against 5 (one pass, dS exchange; K^T
fragments in registers / re-read from LDS) and 4 (one pass, partial dQ tiles + reducer wave; bf16 only), both operand formats.
    python tools/attn_form_ab.py [B N H]   ->  one JSON line {dtype: {form: [us, us], ...}, checks}"""
import json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from visiondk_amd import _lib, ops  # noqa: E402

B, N, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 197, 12)
be = _lib.load()
D = H * 64
out = {"B": B, "N": N, "H": H}
for dt, name in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3 * D, device="cuda").to(dt); dout = torch.randn(B, N, D, device="cuda").to(dt)
    o, lse = ops.attention_fwd(qkv, H, backend=be)
    res, tim = {}, {}
    forms = [("3", "regs"), ("5", "regs"), ("5", "lds")] + ([("4", "regs")] if dt == torch.bfloat16 else [])
    for rep in range(2):
        for form, kt in forms:
            os.environ["VDK_ATTN_BWD_FORM"] = form
            os.environ["VDK_ATTN5_KT"] = kt
            key = form if form != "5" else f"5{kt}"
            fn = lambda: ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
            res[key] = fn()
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            tim.setdefault(f"form{key}_us", []).append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
    a = res["3"].float()
    chk = {}
    for key, r in res.items():
        if key == "3":
            continue
        b = r.float()
        chk[key] = {"dq_rel": ((a[..., :D] - b[..., :D]).norm() / a[..., :D].norm()).item(), "dkdv_equal": bool(torch.equal(res["3"][..., D:], r[..., D:]))}
    out[name] = {"us": tim, "vs_form3": chk}
os.environ.pop("VDK_ATTN_BWD_FORM", None)
print(json.dumps(out))
