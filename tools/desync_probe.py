"""Does de-synchronising workgroups help the short-K GEMM?  One full GEMM vs the same work as two half-M GEMMs on two streams."""
import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from visiondk_amd import ops
M, N, K = 50432, 2304, 768
a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def full():
    ops.gemm_nt(a, b, out=out)
h = 25088   # 98 row tiles
def halves():
    with torch.cuda.stream(s1): ops.gemm_nt(a[:h], b, out=out[:h])
    with torch.cuda.stream(s2): ops.gemm_nt(a[h:], b, out=out[h:])
def seq_halves():
    ops.gemm_nt(a[:h], b, out=out[:h]); ops.gemm_nt(a[h:], b, out=out[h:])
import time
for name, fn in [("full", full), ("two halves, 2 streams", halves), ("two halves, 1 stream", seq_halves)]:
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name:28s} {dt*1e6:8.1f} us  {2*M*N*K/dt/1e12:7.1f} TF")
