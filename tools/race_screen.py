"""Race screen for the LDS-DMA GEMM schedule: many repetitions at several shapes against a fp32 torch reference; any
mismatch between repetitions (bitwise) or vs the reference is reported."""
import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops
be = _lib.load()
torch.manual_seed(0)
bad = 0
for (M, N, K, trans) in [(4096, 4096, 4096, False), (50432, 768, 768, False), (2048, 2304, 3072, False), (768, 3072, 50432, True), (2304, 768, 12544, True), (300, 264, 192, False)]:
    if trans:
        a = torch.randn(K, M, device="cuda").bfloat16(); b = torch.randn(K, N, device="cuda").bfloat16()
        ref = (a[:, :256].float().T @ b.float())
    else:
        a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
        ref = (a[:256].float() @ b.float().T)
    be.lib.vdk_gemm_force_kernel(2)
    first = None
    for rep in range(30):
        out = ops.gemm_nt(a, b, out_dtype=torch.float32, trans=trans, splitk=(7 if trans else 1))
        if first is None:
            first = out.clone()
            rel = ((out[:256] - ref).norm() / ref.norm()).item()
            print(f"M={M} N={N} K={K} trans={trans}: rel err vs fp32 (first 256 rows) {rel:.2e}", flush=True)
            if rel > 1e-4: bad += 1
        elif not torch.equal(out, first):
            d = (out - first).abs().max().item()
            print(f"  !! repetition {rep} differs from repetition 0 (max abs {d})"); bad += 1; break
    be.lib.vdk_gemm_force_kernel(0)
print("RACE SCREEN", "FAILED" if bad else "clean")
