"""Cycle stamps of the 256x128 / two-workgroups-per-CU GEMM (vdk_gemm_debug_stamps): per workgroup, prologue, main loop per k-tile, epilogue; and how the
workgroups overlap in time.  usage: w4h_stamps.py M N K [epilogue]"""
import ctypes
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402
be = _lib.load()
M, N, K = (int(x) for x in sys.argv[1:4])
ep = sys.argv[4] if len(sys.argv) > 4 else "plain"
torch.manual_seed(0)
a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
bias = torch.randn(N, device="cuda")
kw = {}
odt = torch.bfloat16
if ep == "bias": kw = {"bias": bias}
elif ep == "res": kw = {"bias": bias, "residual": torch.randn(M, N, device="cuda")}; odt = torch.float32
elif ep == "gelu": kw = {"bias": bias, "act": ops.ACT_GELU, "aux": torch.empty(M, N, device="cuda", dtype=torch.bfloat16)}
elif ep == "dgelu":
    rows = be.lib.vdk_gemm_c_colsum_rows(M, N, K)
    kw = {"act": ops.ACT_DGELU, "aux": torch.randn(M, N, device="cuda").bfloat16(), "c_colsum": torch.empty(rows, N, device="cuda")}
o = torch.empty(M, N, dtype=odt, device="cuda")
be.lib.vdk_gemm_force_kernel(6)
for _ in range(3):
    ops.gemm_nt(a, b, out=o, backend=be, **kw)
G = ((M + 255) // 256) * ((N + 127) // 128)
st = torch.zeros(G * 8, dtype=torch.int64, device="cuda")
be.lib.vdk_gemm_debug_stamps(ctypes.c_void_p(st.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm_nt(a, b, out=o, backend=be, **kw); e1.record(); torch.cuda.synchronize()
be.lib.vdk_gemm_debug_stamps(ctypes.c_void_p(0))
s = st.cpu().view(G, 8).double()
nk = K // 64
nk = K // 64
si = st.cpu().view(G, 8)
hw = si[:, 7] & 0xffffffff; xcc = (si[:, 7] >> 32) & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)      # cu_id, sh_id, se_id, xcc
wave_slot = hw & 0xf
rt0 = si[:, 5].double(); rt1 = si[:, 6].double()
base = rt0.min(); rt0 = (rt0 - base) * 10.0; rt1 = (rt1 - base) * 10.0      # ns
dur = s[:, 2] - s[:, 0]
print(f"{M}x{N}x{K} {ep}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {G} workgroups on {len(set(cu.tolist()))} CUs; last end {rt1.max() / 1e3:.1f} us after the first start")
print(f"  cycles: prologue (incl. stagger) {(s[:, 3] - s[:, 0]).mean():.0f}, main loop {(s[:, 1] - s[:, 3]).mean():.0f} = {(s[:, 1] - s[:, 3]).mean() / nk:.0f} per k-tile, epilogue {(s[:, 2] - s[:, 1]).mean():.0f}, whole {dur.mean():.0f}; clock {(dur / (rt1 - rt0) ).mean():.2f} GHz")
print(f"  wave slots seen by wave 0: {sorted(set(wave_slot.tolist()))}")
# per CU: how far apart do its two residents start / how much of a workgroup's main loop lies inside the other resident's epilogue
import collections
by = collections.defaultdict(list)
for i in range(G):
    by[int(cu[i])].append(i)
ph = []
for c, ids in by.items():
    ids = sorted(ids, key=lambda i: float(rt0[i]))
    for a_, b_ in zip(ids[:-1], ids[1:]):
        ph.append(float(rt0[b_] - rt0[a_]))
ph = torch.tensor(ph)
print(f"  per CU, gap between consecutive workgroup starts: mean {ph.mean():.0f} ns, median {ph.median():.0f}, 10% {ph.quantile(0.1):.0f}, 90% {ph.quantile(0.9):.0f}  (mean workgroup duration {(rt1 - rt0).mean():.0f} ns)")
c0 = sorted(by.keys())[0]
print("  one CU's residents (start, end in us):", [(round(float(rt0[i]) / 1e3, 1), round(float(rt1[i]) / 1e3, 1)) for i in sorted(by[c0], key=lambda i: float(rt0[i]))][:14])
