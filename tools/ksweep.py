import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from visiondk_amd import ops
from tools.microbench import timeit
M = 50432
for N in (2304, 768):
    for K in (64, 256, 768, 1536, 3072):
        a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ms = timeit(lambda: ops.gemm_nt(a, b, out=out), iters=20, warmup=3)
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        rounds = -(-tiles // 256)
        print(f"N={N} K={K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF  tiles={tiles} rounds={rounds}  us/round={ms*1e3/rounds:6.1f}  us/round/ktile={ms*1e3/rounds/(K/64):5.2f}")
print("--- with residual + bias, f32 out (proj/fc2 form) ---")
for K in (64, 768, 3072):
    N = 768
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    res = torch.randn(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ms = timeit(lambda: ops.gemm_nt(a, b, out=out, bias=bias, residual=res), iters=20, warmup=3)
    print(f"N={N} K={K} +bias+residual f32: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF  us/round={ms*1e3/3:6.1f}")
print("--- GELU + aux (fc1 form) ---")
for K in (64, 768):
    N = 3072
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda"); aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm_nt(a, b, out=out, bias=bias, act=ops.ACT_GELU, aux=aux), iters=20, warmup=3)
    print(f"N={N} K={K} gelu+aux: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF  us/round={ms*1e3/10:6.1f}")
