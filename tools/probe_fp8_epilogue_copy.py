"""probe: cost of the fp8 by-product in the GELU epilogue of the fp8 GEMM (ViT-L fc1 shape), with and without the amax atomics, against the separate quantisation pass"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import ops, _lib
from visiondk_amd.ops import ACT_GELU, FP8_E4M3
be = _lib.load()
M, N, K = 73728, 4096, 1024
dev = "cuda"
a = ops.quant_fp8(torch.randn(M, K, device=dev).bfloat16(), None, FP8_E4M3, None, backend=be)
b = ops.quant_fp8((torch.randn(N, K, device=dev) * 0.05).bfloat16(), None, FP8_E4M3, None, backend=be)
bias = torch.randn(N, device=dev) * 0.1
u = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
sc = torch.tensor([30.0], device=dev); am = torch.zeros(1, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = ops.gemm_fp8_nt(a, b, bias=bias, act=ACT_GELU, aux=u, backend=be)
out = {}
out["gemm_plain_us"] = t(lambda: ops.gemm_fp8_nt(a, b, bias=bias, act=ACT_GELU, aux=u, backend=be))
out["gemm_q8_amax_us"] = t(lambda: ops.gemm_fp8_nt(a, b, bias=bias, act=ACT_GELU, aux=u, backend=be, q8={"fmt": 0, "scale": sc, "amax": am}))
out["gemm_q8_noamax_us"] = t(lambda: ops.gemm_fp8_nt(a, b, bias=bias, act=ACT_GELU, aux=u, backend=be, q8={"fmt": 0, "scale": sc, "amax": None}))
out["quant_pass_us"] = t(lambda: ops.quant_fp8(g, sc, FP8_E4M3, am, backend=be))
print(json.dumps(out))
