"""A/B timing of the attention kernels at the bench's geometry (ViT-B/16, batch 256: B*H = 3072 items of N = 197 tokens, head_dim 64).
    python tools/bench_attention.py [B N H]
Prints per-call times of forward / backward for the short-sequence kernels (csrc/attention_small.hip) and the flash-style ones (csrc/attention.hip),
with the HBM floor of each (every operand / result once at 8 TB/s and at the 6.3 TB/s a copy kernel reaches)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from visiondk_amd import _lib, ops  # noqa: E402


def main():
    B, N, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 197, 12)
    be = _lib.load()
    D = H * 64
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3 * D, device="cuda").bfloat16()
    dout = torch.randn(B, N, D, device="cuda").bfloat16()
    out = {"B": B, "N": N, "H": H}
    for name, legacy in (("short_sequence", 0), ("flash_style", 1)):
        be.lib.vdk_attention_force_legacy(legacy)
        o, lse = ops.attention_fwd(qkv, H, backend=be)
        d = ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
        torch.cuda.synchronize()
        res = {}
        for what, fn in (("fwd", lambda: ops.attention_fwd(qkv, H, backend=be)), ("bwd", lambda: ops.attention_bwd(qkv, o, dout, lse, H, backend=be))):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[what + "_us"] = e0.elapsed_time(e1) / 20 * 1e3
        out[name] = res
    be.lib.vdk_attention_force_legacy(-1)
    tok = B * N * D * 2
    out["hbm_floor_us"] = {"fwd_bytes": 4 * tok, "bwd_bytes": 8 * tok, "fwd@8TB/s": 4 * tok / 8e6, "fwd@6.3TB/s": 4 * tok / 6.3e6, "bwd@8TB/s": 8 * tok / 8e6, "bwd@6.3TB/s": 8 * tok / 6.3e6}
    fl = 4.0 * B * H * N * N * 64
    out["tflops"] = {k: {"fwd": fl / (v["fwd_us"] * 1e-6) / 1e12, "bwd": 2.5 * fl / (v["bwd_us"] * 1e-6) / 1e12} for k, v in out.items() if isinstance(v, dict) and "fwd_us" in v}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
