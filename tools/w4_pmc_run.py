"""One GEMM shape launched a few times with a forced kernel structure (for rocprofv3 --pmc passes).  usage: w4_pmc_run.py KERN M N K [trans] [iters]
KERN = -1: the vendor library's kernel for the same bare product (torch.matmul -> hipBLASLt / rocBLAS; the comparison VERDICT r4 asked for, never on the product path)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402
be = _lib.load()
kern, M, N, K = (int(x) for x in sys.argv[1:5])
trans = len(sys.argv) > 5 and sys.argv[5] == "1"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
splitk = int(sys.argv[7]) if len(sys.argv) > 7 else 1
torch.manual_seed(0)
if trans:
    a = torch.randn(K, M, device="cuda").bfloat16(); b = torch.randn(K, N, device="cuda").bfloat16()
else:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
o = torch.empty(M, N, dtype=torch.float32 if trans else torch.bfloat16, device="cuda")
if kern < 0:
    for _ in range(iters):
        o = torch.matmul(a.t(), b) if trans else torch.matmul(a, b.t())
else:
    be.lib.vdk_gemm_force_kernel(kern)
    for _ in range(iters):
        ops.gemm_nt(a, b, out=o, trans=trans, backend=be, **({"splitk": splitk} if splitk > 1 else {}))
torch.cuda.synchronize()
