"""fp8 vs bf16 GEMM at the Linear shapes of BASELINE.json configs[4] (SigLIP ViT-L/14 336: 576 tokens x 128 images per GPU = 73 728 rows, dim 1024, mlp 4096).
HIP-event timing on the launch stream; prints one JSON object.  Peaks (MI355X_MICROARCH.md): bf16 2.5 PFLOP/s, fp8 5.0 PFLOP/s dense."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))

from visiondk_amd import _lib, ops


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    be = _lib.load()
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 73728
    out = {"rows": T, "shapes": []}
    for name, N, K, kw in [("qkv", 3072, 1024, {}), ("proj", 1024, 1024, {}), ("fc1+gelu", 4096, 1024, {"act": ops.ACT_GELU}), ("fc2", 1024, 4096, {})]:
        a = torch.randn(T, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        bias = torch.randn(N, device="cuda")
        aux = torch.empty(T, N, dtype=torch.bfloat16, device="cuda") if "act" in kw else None
        a8 = ops.quant_fp8(a, None, ops.FP8_E4M3, None, backend=be); b8 = ops.quant_fp8(b, None, ops.FP8_E4M3, None, backend=be)
        one = torch.ones(1, device="cuda")
        t16 = timeit(lambda: ops.gemm_nt(a, b, bias=bias, aux=aux, backend=be, **kw))
        t8 = timeit(lambda: ops.gemm_fp8_nt(a8, b8, one, one, bias=bias, aux=aux, backend=be, **kw))
        tq = timeit(lambda: ops.quant_fp8(a, one, ops.FP8_E4M3, one, backend=be))
        fl = 2.0 * T * N * K
        out["shapes"].append({"name": name, "M": T, "N": N, "K": K, "bf16_us": t16 * 1e6, "fp8_us": t8 * 1e6, "quant_a_us": tq * 1e6,
                              "bf16_tflops": fl / t16 / 1e12, "fp8_tflops": fl / t8 / 1e12, "bf16_frac_of_2.5PF": fl / t16 / 2.5e15, "fp8_frac_of_5PF": fl / t8 / 5e15,
                              "quant_GBps": (a.numel() * 3) / tq / 1e9, "speedup_incl_quant": t16 / (t8 + tq)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
