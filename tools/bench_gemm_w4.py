"""A/B of the two 256x256 GEMM structures (2 = eight waves, 5 = four waves / one per SIMD) on the Linear shapes of the ViT-B/16 step
(T = 256 x 197 = 50 432 rows, real epilogues, the wgrad TN forms with the engine's split counts) and on square problems.

Interleaved rounds in one process (cdna_hip_programming.md rule 24), random operands (rule 25); also: outputs of the two kernels compared
(same k order per element -> expected bit-equal), and a repeat screen of the four-wave kernel (20 launches against the first).
    python tools/bench_gemm_w4.py [out.json] [rounds] [epilogues|-] [bf16|fp16]
Round 4: a VENDOR column (same box, same tensors): `torch.matmul` = hipBLASLt / rocBLAS computing the bare matrix product of the same operands (no bias, no activation, no
residual, 16-bit output; for the TN shapes a.T @ b) -- the external yard-stick for "how close to what this part does on these shapes" (SURVEY section 7 allows the vendor
library as a comparison, never in the product), and `rel_err_vs_vendor` for the plain / TN shapes.  Fourth argument: the operand format (fp16: kernels 5 and 6 only).
"""
import json
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


def tn_splitk(M, N, K):
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    s = max(1, 256 // tiles)
    s = min(s, (K // 64) // 4, 64)
    return max(s, 1)


def main():
    be = _lib.load()
    outp = sys.argv[1] if len(sys.argv) > 1 else None
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    T = 50432
    res = {"rows": T, "shapes": []}
    only = sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] != "-" else None
    dt = torch.float16 if (len(sys.argv) > 4 and sys.argv[4] == "fp16") else torch.bfloat16
    kerns = (5, 6) if dt == torch.float16 else (2, 5, 6)
    res["operand"] = "fp16" if dt == torch.float16 else "bf16"
    shapes = [("qkv bias", T, 2304, 768, "bias", False), ("proj bias+res f32", T, 768, 768, "res", False), ("fc1 bias+gelu+aux", T, 3072, 768, "gelu", False),
              ("fc1 bias+gelu+saved derivative", T, 3072, 768, "gelud", False), ("fc2 bias+res f32", T, 768, 3072, "res", False), ("dfc2 dgelu+ocs", T, 3072, 768, "dgelu", False),
              ("dfc2 x saved derivative+ocs", T, 3072, 768, "mulaux", False), ("dfc1 plain", T, 768, 3072, "plain", False),
              ("dproj plain", T, 768, 768, "plain", False), ("dqkv plain", T, 768, 2304, "plain", False),
              ("wgrad qkv", 2304, 768, T, "tn", True), ("wgrad proj", 768, 768, T, "tn", True), ("wgrad fc1", 3072, 768, T, "tn", True), ("wgrad fc2", 768, 3072, T, "tn", True),
              ("4096^3", 4096, 4096, 4096, "plain", False), ("8192^3", 8192, 8192, 8192, "plain", False), ("65536x768x3072", 65536, 768, 3072, "plain", False)]
    for name, M, N, K, ep, trans in shapes:
        if only and ep not in only:
            continue
        torch.manual_seed(0)
        if trans:
            a = torch.randn(K, M, device="cuda").to(dt); b = torch.randn(K, N, device="cuda").to(dt)
        else:
            a = torch.randn(M, K, device="cuda").to(dt); b = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
        bias = torch.randn(N, device="cuda")
        kw = {}
        odt = dt
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
        elif ep == "tn":
            kw = {"trans": True, "splitk": tn_splitk(M, N, K)}; odt = torch.float32
        o = torch.empty(M, N, dtype=odt, device="cuda")

        def run(kern):
            be.lib.vdk_gemm_force_kernel(kern)
            ops.gemm_nt(a, b, out=o, backend=be, **kw)

        if 2 in kerns:
            run(2); assert be.lib.vdk_gemm_last_kernel() == 2; r2 = o.clone()
        o.fill_(float("nan"))
        run(6); assert be.lib.vdk_gemm_last_kernel() == 6; r6 = o.clone()
        o.fill_(float("nan"))
        run(5); assert be.lib.vdk_gemm_last_kernel() == 5; r5 = o.clone()
        torch.cuda.synchronize()
        if 2 not in kerns:
            r2 = r5
        maxdiff = (r2.float() - r5.float()).abs().max().item()
        nbad = int((r2 != r5).sum().item()) + int((r5 != r6).sum().item())
        # repeat screen: the four-wave kernel against its own first result
        unstable = 0
        for _ in range(10):
            run(5)
            unstable += int((o != r5).sum().item() > 0)
            run(6)
            unstable += int((o != r6).sum().item() > 0)
        ref_err = None
        if M * N <= 4096 * 4096 and ep in ("plain", "tn"):
            ref = (a.float().T @ b.float()) if trans else (a.float() @ b.float().T)
            ref_err = ((r5.float() - ref).norm() / ref.norm()).item()
        # the vendor library on the same tensors: the bare product
        vo = torch.empty(M, N, dtype=dt, device="cuda")
        at = a.t() if trans else a
        bt = b if trans else b.t()
        vendor = lambda: torch.matmul(at, bt, out=vo)
        vendor(); torch.cuda.synchronize()
        vend_err = None
        if ep in ("plain", "tn"):
            vend_err = ((r5.float() - vo.float()).norm() / vo.float().norm()).item()
        t = {k: [] for k in kerns}
        t["v"] = []
        iters = 10
        for kern in kerns:
            timed(lambda: run(kern), 3)
        timed(vendor, 3)
        for _ in range(rounds):
            for kern in kerns:
                t[kern].append(timed(lambda: run(kern), iters))
            t["v"].append(timed(vendor, iters))
        fl = 2.0 * M * N * K
        rec = {"name": name, "M": M, "N": N, "K": K, "epilogue": ep, "maxdiff_w8_w4": maxdiff, "n_differ": nbad, "unstable_repeats": unstable, "rel_err_vs_torch": ref_err,
               "rel_err_vs_vendor": vend_err}
        for kern, key in ((2, "w8"), (5, "w4"), (6, "w4h"), ("v", "vendor")):
            if kern not in t:
                continue
            ts = sorted(t[kern])
            rec[key + "_us_median"] = ts[len(ts) // 2] * 1e6
            rec[key + "_us_min"] = ts[0] * 1e6
            rec[key + "_tflops_median"] = fl / ts[len(ts) // 2] / 1e12
        best = min(rec["w4_us_median"], rec["w4h_us_median"])
        rec["vendor_bare_product_over_best"] = rec["vendor_us_median"] / best       # > 1: this library's kernel WITH its epilogue beats the vendor's bare product
        if "w8_us_median" in rec:
            rec["speedup"] = rec["w8_us_median"] / rec["w4_us_median"]
            rec["speedup_h"] = rec["w8_us_median"] / rec["w4h_us_median"]
        res["shapes"].append(rec)
        print(json.dumps(rec), flush=True)
    be.lib.vdk_gemm_force_kernel(0)
    if outp:
        Path(outp).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
