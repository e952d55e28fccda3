"""Round 6 A/B: start-phase stagger of the persistent four-wave GEMM (VDK_GEMM_W4_STAGGER, gemm_w4.hip) on the ViT-B/16 Linear shapes whose epilogues are long
(fp32 residual, GELU, dGELU).  Interleaved rounds in one process, random operands.
    python tools/r6_stagger_ab.py [out.json] [rounds] [fp16|bf16]
"""
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    be = _lib.load()
    outp = sys.argv[1] if len(sys.argv) > 1 else None
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dt = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else torch.float16
    T = 50432
    shapes = [("proj bias+res f32", T, 768, 768, "res"), ("fc2 bias+res f32", T, 768, 3072, "res"), ("fc1 gelu+saved derivative", T, 3072, 768, "gelud"),
              ("fc1 gelu+aux", T, 3072, 768, "gelu"), ("dfc2 x saved derivative+ocs", T, 3072, 768, "mulaux"), ("dfc2 dgelu+ocs", T, 3072, 768, "dgelu"),
              ("qkv bias", T, 2304, 768, "bias"), ("dproj plain", T, 768, 768, "plain"), ("dqkv plain", T, 768, 2304, "plain"), ("dfc1 plain", T, 768, 3072, "plain")]
    # (label, force_kernel, VDK_GEMM_W4_SPLIT, VDK_GEMM_W4_STAGGER)
    variants = [("default", 0, "1", "0"), ("w4", 5, "1", "0"), ("w4 nosplit", 5, "0", "0"), ("w4 nosplit st50", 5, "0", "50"), ("w4 nosplit st90", 5, "0", "90"),
                ("w4 nosplit st60all", 5, "0", "60,all"), ("w4 nosplit st100all", 5, "0", "100,all"), ("w4 split st100all", 5, "1", "100,all"), ("w4h", 6, "1", "0")]
    res = {"rows": T, "operand": "fp16" if dt == torch.float16 else "bf16", "shapes": []}
    for name, M, N, K, ep in shapes:
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
        elif ep == "mulaux":
            rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
            kw = {"act": ops.ACT_MUL_AUX, "aux": torch.randn(M, N, device="cuda").to(dt), "c_colsum": torch.empty(rows, N, device="cuda")}
        elif ep == "dgelu":
            rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
            kw = {"act": ops.ACT_DGELU, "aux": torch.randn(M, N, device="cuda").to(dt), "c_colsum": torch.empty(rows, N, device="cuda")}
        o = torch.empty(M, N, dtype=odt, device="cuda")

        def run(v):
            be.lib.vdk_gemm_force_kernel(v[1])
            os.environ["VDK_GEMM_W4_SPLIT"] = v[2]
            os.environ["VDK_GEMM_W4_STAGGER"] = v[3]
            ops.gemm_nt(a, b, out=o, backend=be, **kw)

        ref = None
        t = {v[0]: [] for v in variants}
        bad = {}
        for v in variants:
            o.fill_(float("nan"))
            run(v); torch.cuda.synchronize()
            if ref is None:
                ref = o.clone()
            else:
                bad[v[0]] = int((o != ref).sum().item())
            timed(lambda: run(v), 3)
        for _ in range(rounds):
            for v in variants:
                t[v[0]].append(timed(lambda: run(v), 10))
        rec = {"name": name, "M": M, "N": N, "K": K, "epilogue": ep, "n_differ_vs_default": bad}
        for v in variants:
            ts = sorted(t[v[0]])
            rec[v[0]] = round(ts[len(ts) // 2] * 1e6, 1)
        res["shapes"].append(rec)
        print(json.dumps(rec), flush=True)
    be.lib.vdk_gemm_force_kernel(0)
    os.environ.pop("VDK_GEMM_W4_SPLIT", None); os.environ.pop("VDK_GEMM_W4_STAGGER", None)
    if outp:
        Path(outp).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
