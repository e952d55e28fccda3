"""What a collective in flight costs a persistent GEMM, on ONE GPU: a stand-in kernel (vdk_debug_occupy_cus: R workgroups resident for a few ms on a second stream, as an
RCCL all-reduce holds one CU per channel) runs while the ViT-B/16 Linear GEMMs are timed on the main stream, with the persistent walk over all CUs (reserve 0) and with R CUs left out
of it (vdk_gemm_reserve_cus).  Also: what the reserve costs when nothing else runs.
    python tools/w4_contention.py [out.json]
"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402


def main():
    be = _lib.load()
    outp = sys.argv[1] if len(sys.argv) > 1 else None
    T = 50432
    side = torch.cuda.Stream()
    res = []
    for name, M, N, K, ep in (("qkv bias", T, 2304, 768, "bias"), ("fc1 bias+gelu", T, 3072, 768, "gelu"), ("fc2 bias+res", T, 768, 3072, "res"), ("dfc1 plain", T, 768, 3072, "plain")):
        torch.manual_seed(0)
        a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        bias = torch.randn(N, device="cuda")
        kw, odt = {}, torch.bfloat16
        if ep == "bias":
            kw = {"bias": bias}
        elif ep == "res":
            kw = {"bias": bias, "residual": torch.randn(M, N, device="cuda")}; odt = torch.float32
        elif ep == "gelu":
            kw = {"bias": bias, "act": ops.ACT_GELU, "aux": torch.empty(M, N, device="cuda", dtype=torch.bfloat16)}
        o = torch.empty(M, N, dtype=odt, device="cuda")
        iters = 12

        def run(reserve, occupy):
            be.check(be.lib.vdk_gemm_reserve_cus(reserve), "reserve")
            for _ in range(3):
                ops.gemm_nt(a, b, out=o, backend=be, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if occupy:
                be.check(be.lib.vdk_debug_occupy_cus(occupy, 30000, side.cuda_stream), "occupy")      # 30 ms: far longer than the timed GEMMs ("a collective is always in flight")
            torch.cuda._sleep(2_000_000)          # let it become resident first (same pause without it: a pause changes the clocks of what follows)
            e0.record()
            for _ in range(iters):
                ops.gemm_nt(a, b, out=o, backend=be, **kw)
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1) / iters * 1e3
            torch.cuda.synchronize()
            return t

        rec = {"name": name, "M": M, "N": N, "K": K, "kernel": None}
        for reserve in (0, 16, 32):
            for occ in (0, 16, 32):
                rec[f"us_reserve{reserve}_occupied{occ}"] = min(run(reserve, occ) for _ in range(2))
        rec["kernel"] = be.lib.vdk_gemm_last_kernel()
        res.append(rec)
        print(json.dumps(rec), flush=True)
    be.lib.vdk_gemm_reserve_cus(0)
    if outp:
        Path(outp).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
