import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from visiondk_amd import ops
for (M, N, K) in [(4096, 4096, 4096), (50432, 3072, 768), (50432, 768, 3072)]:
    a = (torch.randn(M, K, device="cuda") * 0.05).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.gemm_nt(a, b, out=out)
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"skip={os.environ.get('VDK_GEMM_SKIP','0')} {M}x{N}x{K} {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:7.1f} TF/s")
