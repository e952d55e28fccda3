"""Per-kernel timings on the MI355X at the hot path's real shapes (ViT-B/16 bs256, CBIR 1M x 10k).
Usage: python tools/microbench.py [gemm] [attn] [ln] [cbir] [opt]   -> prints one line per measurement."""
import sys
import time

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))

from visiondk_amd import cbir, ops


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def main():
    what = set(sys.argv[1:]) or {"gemm", "attn", "ln", "cbir", "opt"}
    dev = "cuda"
    print("device", torch.cuda.get_device_name(0), "count", torch.cuda.device_count(), flush=True)
    T = 256 * 197
    if "gemm" in what:
        for name, (M, N, K, sk) in {
            "qkv_fwd": (T, 2304, 768, 1), "proj_fwd": (T, 768, 768, 1), "fc1_fwd": (T, 3072, 768, 1),
            "fc2_fwd": (T, 768, 3072, 1), "fc1_wgrad": (3072, 768, T, 16), "qkv_wgrad": (2304, 768, T, 16),
            "proj_wgrad": (768, 768, T, 32), "sq4096": (4096, 4096, 4096, 1), "sq8192": (8192, 8192, 8192, 1),
        }.items():
            a = torch.randn(M, K, device=dev).bfloat16()
            b = torch.randn(N, K, device=dev).bfloat16()
            out = torch.empty(M, N, device=dev, dtype=torch.float32 if sk > 1 else torch.bfloat16)
            ms = timeit(lambda: ops.gemm_nt(a, b, out=out, splitk=sk))
            print(f"gemm {name:12s} M={M} N={N} K={K} splitk={sk}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)
            if name == "qkv_fwd":
                ref = (a[:512].float() @ b.float().T)
                err = ((out[:512].float() - ref).norm() / ref.norm()).item()
                print(f"   rel err vs torch fp32 (first 512 rows): {err:.2e}")
            del a, b, out
        x = torch.randn(T, 3072, device=dev).bfloat16()
        ms = timeit(lambda: ops.transpose_pad(x))
        print(f"transpose [{T},3072] bf16: {ms:.3f} ms  {2*x.numel()*2/ms/1e6:.0f} GB/s")
    if "gemmab" in what:
        from visiondk_amd import _lib
        be = _lib.load()
        shapes = {"qkv_fwd": (T, 2304, 768, 1), "proj_fwd": (T, 768, 768, 1), "fc1_fwd": (T, 3072, 768, 1), "fc2_fwd": (T, 768, 3072, 1),
                  "qkv_dgrad": (T, 768, 2304, 1), "fc1_wgrad": (3072, 768, T, 8), "qkv_wgrad": (2304, 768, T, 10), "proj_wgrad": (768, 768, T, 32),
                  "sq4096": (4096, 4096, 4096, 1), "sq8192": (8192, 8192, 8192, 1)}
        for name, (M, N, K, sk) in shapes.items():
            a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
            out = torch.empty(M, N, device=dev, dtype=torch.float32 if sk > 1 else torch.bfloat16)
            res = {}
            outs = {}
            for which in (1, 2):
                be.lib.vdk_gemm_force_kernel(which)
                ms = timeit(lambda: ops.gemm_nt(a, b, out=out, splitk=sk), iters=10, warmup=3)
                res[which] = 2 * M * N * K / ms / 1e9
                outs[which] = out.float().clone() if M * N < 2e8 else None
            be.lib.vdk_gemm_force_kernel(0)
            diff = ""
            if outs[1] is not None:
                diff = f" maxdiff(k1,k2)={(outs[1]-outs[2]).abs().max().item():.3e}"
            print(f"gemmAB {name:11s} M={M} N={N} K={K} sk={sk}: 128^2 reg-staged {res[1]:7.1f} TF | 256^2 lds-dma {res[2]:7.1f} TF{diff}", flush=True)
            del a, b, out, outs
    if "attn" in what:
        B, N, H = 256, 197, 12
        qkv = torch.randn(B, N, 3 * H * 64, device=dev).bfloat16()
        o, lse = ops.attention_fwd(qkv, H)
        ms = timeit(lambda: ops.attention_fwd(qkv, H))
        fl = 4 * B * H * N * N * 64
        print(f"attn fwd B={B} N={N} H={H}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s")
        do = torch.randn_like(o)
        ms = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, H))
        print(f"attn bwd: {ms:.3f} ms  {2.5*fl/ms/1e9:.1f} TFLOP/s (5 GEMM convention)")
    if "ln" in what:
        x = torch.randn(T, 768, device=dev)
        g = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
        y, mean, rstd = ops.layernorm_fwd(x, g, b)
        ms = timeit(lambda: ops.layernorm_fwd(x, g, b))
        print(f"ln fwd [{T},768]: {ms:.3f} ms  {(x.numel()*4 + y.numel()*2)/ms/1e6:.0f} GB/s")
        dy = torch.randn(T, 768, device=dev).bfloat16()
        ms = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dres=x))
        print(f"ln bwd: {ms:.3f} ms  {(x.numel()*4*3 + dy.numel()*2*2)/ms/1e6:.0f} GB/s")
    if "opt" in what:
        n = 86_600_000
        p = torch.randn(n, device=dev); g = torch.randn(n, device=dev); m = torch.zeros(n, device=dev)
        ema = p.clone(); pb = torch.empty(n, device=dev, dtype=torch.bfloat16)
        def step():
            nsq = ops.sumsq(g)
            ops.sgd_step(p, g, m, lr=1e-3, momentum=0.9, weight_decay=5e-4, ema=ema, p_bf16=pb, normsq=nsq, ema_decay=0.99)
        ms = timeit(step)
        print(f"clip+sgd+ema+cast {n/1e6:.1f}M params: {ms:.3f} ms  {n*34/ms/1e6:.0f} GB/s")
    if "cbir" in what:
        torch.manual_seed(0)
        for (nq, n, d) in [(10000, 1_000_000, 128)]:
            g = cbir.l2_normalize(torch.randn(n, d, device=dev))
            q = cbir.l2_normalize(torch.randn(nq, d, device=dev))
            idx = cbir.FlatIPIndex(d)
            idx.add(g)
            s, i = idx.search(q, 100)
            torch.cuda.synchronize()
            t0 = time.time()
            ms = timeit(lambda: idx.search(q, 100), iters=3, warmup=1)
            print(f"cbir search Q={nq} N={n} D={d} k=100: {ms:.2f} ms  {nq*n/ms/1e6:.1f} Gpairs/s  "
                  f"fp32-MFMA {2*nq*n*d/ms/1e9:.1f} TFLOP/s of 157.3")
            # spot-check exactness on a few queries against torch fp64
            ref = (q[:8].double() @ g.double().T)
            rs, ri = ref.topk(100, dim=1)
            print("   top-100 index agreement with fp64 topk on 8 queries:", (ri == i[:8]).float().mean().item())


if __name__ == "__main__":
    main()
