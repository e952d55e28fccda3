"""Round 6: quick timings of the ViT-B/16 Linear shapes with the dispatcher's defaults (process-level A/B: VDK_HIP_LIB, VDK_GEMM_* environment).
    python tools/r6_gemm_quick.py [fp16|bf16] [shape substrings ...]"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402


def main():
    be = _lib.load()
    dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
    only = sys.argv[2:]
    T = 50432
    shapes = [("qkv", T, 2304, 768, "bias"), ("proj", T, 768, 768, "res"), ("fc1", T, 3072, 768, "gelud" if dt == torch.float16 else "gelu"), ("fc2", T, 768, 3072, "res"),
              ("dfc2", T, 3072, 768, "mulaux" if dt == torch.float16 else "dgelu"), ("dfc1", T, 768, 3072, "plain"), ("dproj", T, 768, 768, "plain"), ("dqkv", T, 768, 2304, "plain")]
    out = {}
    for name, M, N, K, ep in shapes:
        if only and not any(o == name for o in only):
            continue
        torch.manual_seed(0)
        a = torch.randn(M, K, device="cuda").to(dt); b = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        bias = torch.randn(N, device="cuda")
        kw = {}; odt = dt
        if ep == "bias":
            kw = {"bias": bias}
        elif ep == "res":
            kw = {"bias": bias, "residual": torch.randn(M, N, device="cuda")}; odt = torch.float32
        elif ep == "gelu":
            kw = {"bias": bias, "act": ops.ACT_GELU, "aux": torch.empty(M, N, device="cuda", dtype=dt)}
        elif ep == "gelud":
            kw = {"bias": bias, "act": ops.ACT_GELU_SAVE_GRAD, "aux": torch.empty(M, N, device="cuda", dtype=dt)}
        elif ep in ("mulaux", "dgelu"):
            rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
            kw = {"act": ops.ACT_MUL_AUX if ep == "mulaux" else ops.ACT_DGELU, "aux": torch.randn(M, N, device="cuda").to(dt), "c_colsum": torch.empty(rows, N, device="cuda")}
        o = torch.empty(M, N, dtype=odt, device="cuda")
        fn = lambda: ops.gemm_nt(a, b, out=o, backend=be, **kw)
        for _ in range(5):
            fn()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        out[name] = round(sorted(ts)[3], 1)
        out[name + "_sum"] = float(o.float().abs().sum().item())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
