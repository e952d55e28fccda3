"""Every Linear of the swin_base_patch4_window7_224 step at batch 128 (rows 401 408 / 100 352 / 25 088 / 6 272, channels 128 / 256 / 512 / 1024), with its real epilogue: the
dispatcher's own choice against each forced structure (1 = 128x128, 5 = 256x256 four-wave, 6 = 256x128 two-workgroup), and the TN weight-gradient forms over split counts.
The dispatcher's rules were fitted at the ViT-B/16 shapes (K 768 / 3072); this table shows where they misplace the narrower Swin problems.
    python tools/bench_gemm_swin.py [out.json] [rounds]"""
import json, sys
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
    return e0.elapsed_time(e1) / iters * 1e3


def tn_rule(M, N, K):
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    s = max(1, 256 // tiles)
    s = min(s, (K // 64) // 4, 64)
    return max(s, 1)


def main():
    be = _lib.load()
    outp = sys.argv[1] if len(sys.argv) > 1 else None
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dt = torch.bfloat16
    res = []
    stages = [(401408, 128, 2), (100352, 256, 2), (25088, 512, 18), (6272, 1024, 2)]
    only = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None          # e.g. "tn": only the weight-gradient forms
    if len(sys.argv) > 4:          # other row counts / widths, "rows:channels:blocks,..." (e.g. ConvNeXt-B at batch 512: 1605632:128:3,401408:256:3,100352:512:27,25088:1024:3)
        stages = [tuple(int(v) for v in t.split(":")) for t in sys.argv[4].split(",")]
    tot_auto = tot_best = 0.0
    for T, C, nblk in stages:
        shapes = [("qkv", T, 3 * C, C, "bias"), ("proj", T, C, C, "res"), ("fc1", T, 4 * C, C, "gelu"), ("fc2", T, C, 4 * C, "res"), ("dfc2", T, 4 * C, C, "dgelu"),
                  ("dfc1", T, C, 4 * C, "plain"), ("dproj", T, C, C, "plain"), ("dqkv", T, C, 3 * C, "plain"),
                  ("wg_qkv", 3 * C, C, T, "tn"), ("wg_proj", C, C, T, "tn"), ("wg_fc1", 4 * C, C, T, "tn"), ("wg_fc2", C, 4 * C, T, "tn")]
        for name, M, N, K, ep in shapes:
            if only and ep != only:
                continue
            torch.manual_seed(0)
            trans = ep == "tn"
            if trans:
                a = torch.randn(K, M, device="cuda").to(dt); b = torch.randn(K, N, device="cuda").to(dt)
            else:
                a = torch.randn(M, K, device="cuda").to(dt); b = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
            bias = torch.randn(N, device="cuda")
            kw, odt = {}, dt
            if ep == "bias":
                kw = {"bias": bias}
            elif ep == "res":
                kw = {"bias": bias, "residual": torch.randn(M, N, device="cuda")}; odt = torch.float32
            elif ep == "gelu":
                kw = {"bias": bias, "act": ops.ACT_GELU, "aux": torch.empty(M, N, device="cuda", dtype=dt)}
            elif ep == "dgelu":
                rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
                kw = {"act": ops.ACT_DGELU, "aux": torch.randn(M, N, device="cuda").to(dt), "c_colsum": torch.empty(rows, N, device="cuda")}
            o = torch.empty(M, N, dtype=torch.float32 if trans else odt, device="cuda")
            rec = {"stage_rows": T, "C": C, "blocks": nblk, "name": name, "M": M, "N": N, "K": K, "epilogue": ep, "us": {}}
            variants = {}
            if trans:
                r = tn_rule(M, N, K)
                cand = {max(1, r // 2), r, min(64, r * 2)}
                if ((M + 255) // 256) * ((N + 127) // 128) <= 4:          # tiny outputs: one or two tiles -- more splits than the rule's cap of 64 (one workgroup per split and tile)
                    cand |= {128, 256, 512}
                for s in sorted(cand):
                    variants[f"tn_split{s}" + ("(rule)" if s == r else "")] = (0, {"trans": True, "splitk": s})
                    variants[f"tn_k6_split{s}"] = (6, {"trans": True, "splitk": s})
            else:
                for kern, lab in ((0, "auto"), (1, "k1_128x128"), (5, "k5_w4"), (6, "k6_w4h")):
                    variants[lab] = (kern, kw)
            for lab, (kern, kws) in variants.items():
                def run():
                    be.lib.vdk_gemm_force_kernel(kern)
                    ops.gemm_nt(a, b, out=o, backend=be, **kws)
                try:
                    run(); torch.cuda.synchronize()
                except Exception as e:          # a forced structure that cannot serve the shape
                    rec["us"][lab] = None
                    continue
                if kern and be.lib.vdk_gemm_last_kernel() != kern:
                    rec["us"][lab] = None
                    continue
                if lab == "auto":
                    rec["auto_kernel"] = be.lib.vdk_gemm_last_kernel()
                timed(run, 3)
                ts = sorted(timed(run, 10) for _ in range(rounds))
                rec["us"][lab] = round(ts[len(ts) // 2], 2)
            be.lib.vdk_gemm_force_kernel(0)
            ok = {k: v for k, v in rec["us"].items() if v is not None}
            base = ok.get("auto", next((v for k, v in ok.items() if "(rule)" in k), None))
            best = min(ok, key=ok.get)
            rec["best"] = best; rec["gain_us_per_step"] = round((base - ok[best]) * nblk, 1)
            rec["tflops_auto"] = round(2.0 * M * N * K / base / 1e6, 1)
            tot_auto += base * nblk; tot_best += ok[best] * nblk
            res.append(rec)
            print(json.dumps(rec), flush=True)
            del a, b, o, kw
            torch.cuda.empty_cache()
    summary = {"ms_per_step_dispatcher": tot_auto / 1e3, "ms_per_step_best_of_forced": tot_best / 1e3}
    print(json.dumps(summary))
    if outp:
        Path(outp).write_text(json.dumps({"shapes": res, "summary": summary}, indent=1))


if __name__ == "__main__":
    main()
