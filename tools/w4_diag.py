"""Diagnostic: each fused epilogue form of the four-wave kernel against torch on the GPU, small and large shapes."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402
be = _lib.load()

def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()

for (M, N, K) in [(512, 512, 256), (2048, 2304, 768), (50432, 3072, 768)]:
    torch.manual_seed(3)
    a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda"); u = torch.randn(M, N, device="cuda").bfloat16()
    ref = a.float() @ b.float().T
    be.lib.vdk_gemm_force_kernel(5)
    for rep in range(2):
        aux = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        out = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU, aux=aux, backend=be)
        k = be.lib.vdk_gemm_last_kernel()
        pre = ref + bias
        bad_aux = (aux.float() - pre.bfloat16().float()).abs() > 0.05 * pre.abs() + 0.05
        g = torch.nn.functional.gelu(pre)
        bad_out = ~((out.float() - g).abs() <= 0.05 * g.abs() + 0.05)
        print(f"{M}x{N}x{K} kern {k} GELU rep{rep}: aux rel {rel(aux.float(), pre):.3e} bad {int(bad_aux.sum())}  out rel {rel(out.float(), g):.3e} bad {int(bad_out.sum())} nan {int(out.float().isnan().sum())}")
        if bad_out.any():
            idx = bad_out.nonzero()
            print("   first bad out (row, col):", idx[:6].tolist(), " rows%32:", sorted(set((idx[:2000, 0] % 32).tolist()))[:40], " cols%128:", sorted(set((idx[:2000, 1] % 128).tolist()))[:40])
        if bad_aux.any():
            idx = bad_aux.nonzero()
            print("   first bad aux (row, col):", idx[:6].tolist(), " rows%32:", sorted(set((idx[:2000, 0] % 32).tolist()))[:40], " cols%128:", sorted(set((idx[:2000, 1] % 128).tolist()))[:40])
    rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
    for act, nm in ((ops.ACT_NONE, "plain"), (ops.ACT_DGELU, "dgelu")):
        for cs in (False, True):
            part = torch.full((rows, N), float("nan"), device="cuda") if cs else None
            out = ops.gemm_nt(a, b, act=act, aux=u if act else None, c_colsum=part, backend=be)
            if act:
                uu = u.float().requires_grad_(True); torch.nn.functional.gelu(uu).sum().backward(); r = ref * uu.grad
            else:
                r = ref
            bad = ~((out.float() - r).abs() <= 0.05 * r.abs() + 0.05)
            msg = f"{M}x{N}x{K} kern {be.lib.vdk_gemm_last_kernel()} {nm} ocs={cs}: rel {rel(out.float(), r):.3e} bad {int(bad.sum())}"
            if cs:
                msg += f"  colsum err {(part.sum(0) - out.float().sum(0)).abs().max().item():.3e}"
            print(msg)
            if bad.any():
                idx = bad.nonzero()
                print("   first bad (row, col):", idx[:6].tolist(), " rows%32:", sorted(set((idx[:2000, 0] % 32).tolist()))[:40], " cols%128:", sorted(set((idx[:2000, 1] % 128).tolist()))[:40])
be.lib.vdk_gemm_force_kernel(0)
