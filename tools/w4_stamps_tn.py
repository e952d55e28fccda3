"""Cycle stamps of the four-wave kernel on a weight-gradient (TN, split-K) problem.  usage: w4_stamps_tn.py M N K splitk"""
import ctypes, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402
be = _lib.load()
M, N, K, S = (int(x) for x in sys.argv[1:5])
torch.manual_seed(0)
a = torch.randn(K, M, device="cuda").bfloat16(); b = torch.randn(K, N, device="cuda").bfloat16()
o = torch.empty(M, N, dtype=torch.float32, device="cuda")
be.lib.vdk_gemm_force_kernel(5)
for _ in range(3):
    ops.gemm_nt(a, b, out=o, trans=True, splitk=S, backend=be)
tiles = ((M + 255) // 256) * ((N + 255) // 256)
W = tiles * S
st = torch.zeros(W * 8 * 8, dtype=torch.int64, device="cuda")
be.lib.vdk_gemm_debug_stamps(ctypes.c_void_p(st.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm_nt(a, b, out=o, trans=True, splitk=S, backend=be); e1.record(); torch.cuda.synchronize()
be.lib.vdk_gemm_debug_stamps(ctypes.c_void_p(0))
s = st.cpu().view(W, 8, 8)[:, 0, :].double()
kq = 128
kps = ((K + S - 1) // S + kq - 1) // kq * kq
nk = kps // 64
main = s[:, 1] - s[:, 0]; epi = s[:, 2] - s[:, 1]
print(f"TN {M}x{N}x{K} splitk {S}: {W} workgroups, {nk} k-tiles each; kernel+reduce {e0.elapsed_time(e1) * 1e3:.1f} us")
print(f"  prologue+main loop: mean {main.mean():.0f} cycles = {main.mean() / nk:.0f} per k-tile (min {main.min() / nk:.0f}, max {main.max() / nk:.0f}); epilogue mean {epi.mean():.0f} (max {epi.max():.0f}); until deep wait {(s[:, 4] - s[:, 1]).mean():.0f}")
