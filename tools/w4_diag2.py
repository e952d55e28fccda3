import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402
be = _lib.load()
M, N, K = 512, 512, 256
torch.manual_seed(3)
a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
ref = a.float() @ b.float().T
be.lib.vdk_gemm_force_kernel(5)
rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
part = torch.full((rows, N), float("nan"), device="cuda")
out = ops.gemm_nt(a, b, c_colsum=part, backend=be)
bad = ~((out.float() - ref).abs() <= 0.05 * ref.abs() + 0.05)
idx = bad.nonzero().cpu()
from collections import Counter
print("bad", len(idx))
print("rt (row%128//32):", Counter(((idx[:, 0] % 128) // 32).tolist()))
print("row%32:", sorted(Counter((idx[:, 0] % 32).tolist()).items()))
print("col%128:", sorted(Counter((idx[:, 1] % 128).tolist()).items()))
print("wave row (row%256//128), wave col (col%256//128):", Counter(zip(((idx[:, 0] % 256) // 128).tolist(), ((idx[:, 1] % 256) // 128).tolist())))
print("tile:", Counter(zip((idx[:, 0] // 256).tolist(), (idx[:, 1] // 256).tolist())))
v = out[bad][:16].float().tolist(); r = ref[bad][:16].tolist()
print("values", v); print("ref   ", r)
# is the bad value some other element of the output / ref?
be.lib.vdk_gemm_force_kernel(0)
