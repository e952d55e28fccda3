"""Cycle stamps of the four-wave GEMM (vdk_gemm_debug_stamps): per output tile, main-loop cycles per k-tile and epilogue cycles.  usage: w4_stamps.py M N K [epilogue]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402
be = _lib.load()
M, N, K = (int(x) for x in sys.argv[1:4])
ep = sys.argv[4] if len(sys.argv) > 4 else "plain"
torch.manual_seed(0)
import os
DT = torch.float16 if os.environ.get("DT", "bf16") == "fp16" else torch.bfloat16
a = torch.randn(M, K, device="cuda").to(DT); b = (torch.randn(N, K, device="cuda") * 0.05).to(DT)
bias = torch.randn(N, device="cuda")
kw = {}
odt = DT
if ep == "bias": kw = {"bias": bias}
elif ep == "res": kw = {"bias": bias, "residual": torch.randn(M, N, device="cuda")}; odt = torch.float32
elif ep == "gelu": kw = {"bias": bias, "act": ops.ACT_GELU, "aux": torch.empty(M, N, device="cuda", dtype=DT)}
elif ep == "gelud": kw = {"bias": bias, "act": ops.ACT_GELU_SAVE_GRAD, "aux": torch.empty(M, N, device="cuda", dtype=DT)}
elif ep == "mulaux":
    rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
    kw = {"act": ops.ACT_MUL_AUX, "aux": torch.randn(M, N, device="cuda").to(DT), "c_colsum": torch.empty(rows, N, device="cuda")}
elif ep == "dgelu":
    rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
    kw = {"act": ops.ACT_DGELU, "aux": torch.randn(M, N, device="cuda").to(DT), "c_colsum": torch.empty(rows, N, device="cuda")}
o = torch.empty(M, N, dtype=odt, device="cuda")
be.lib.vdk_gemm_force_kernel(5)
for _ in range(3):
    ops.gemm_nt(a, b, out=o, backend=be, **kw)
G = 256
st = torch.zeros(G * 8 * 8, dtype=torch.int64, device="cuda")
import ctypes
be.lib.vdk_gemm_debug_stamps(ctypes.c_void_p(st.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm_nt(a, b, out=o, backend=be, **kw); e1.record(); torch.cuda.synchronize()
be.lib.vdk_gemm_debug_stamps(ctypes.c_void_p(0))
s = st.cpu().view(G, 8, 8)
nk = K // 64
used = s[:, :, 0] > 0
main = (s[:, :, 1] - s[:, :, 0])[used].double(); epi = (s[:, :, 2] - s[:, :, 1])[used].double()
gap = (s[:, 1:, 0] - s[:, :-1, 2])[used[:, 1:]].double()
span = (s[:, :, 2][used].max() - s[:, :, 0][used].min()).item()
print(f"{M}x{N}x{K} {ep}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, stamped span {span} cycles, tiles stamped {int(used.sum())}")
print(f"  main loop: mean {main.mean():.0f} cycles = {main.mean() / nk:.0f} per k-tile (min {main.min() / nk:.0f}, max {main.max() / nk:.0f}); first tile of a workgroup {(s[:, 0, 1] - s[:, 0, 0]).double().mean() / nk:.0f}, later tiles {(s[:, 1:, 1] - s[:, 1:, 0])[used[:, 1:]].double().mean() / nk:.0f}")
print(f"  epilogue: mean {epi.mean():.0f} cycles (min {epi.min():.0f}, max {epi.max():.0f});  tile-to-tile gap mean {gap.mean():.0f}")
deep = (s[:, :, 4] - s[:, :, 1])[used].double()
print(f"  epilogue parts: until the deep wait passed {deep.mean():.0f} (min {deep.min():.0f} max {deep.max():.0f}); row blocks " + " ".join(f"{(s[:, :, 5 + i] - s[:, :, 4 + i])[used].double().mean():.0f}" for i in range(3)) + f" last {(s[:, :, 2] - s[:, :, 7])[used].double().mean():.0f}")
per_wg = used.sum(1)
print(f"  tiles per workgroup: min {int(per_wg.min())} max {int(per_wg.max())}")
