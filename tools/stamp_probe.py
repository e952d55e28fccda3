import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops
be = _lib.load()
for (M, N, K, kw) in [(50432, 2304, 768, {}), (2048, 2304, 768, {}), (1024, 1024, 768, {}), (2048, 768, 768, dict(res=True)), (50432, 768, 768, dict(res=True))]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    res = torch.randn(M, N, device="cuda") if kw.get("res") else None
    bias = torch.randn(N, device="cuda") if kw.get("res") else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if res is not None else torch.bfloat16)
    nwg = ((M + 255) // 256) * ((N + 255) // 256)
    buf = torch.zeros(nwg * 4, dtype=torch.int64, device="cuda")
    for _ in range(2): ops.gemm_nt(a, b, out=out, residual=res, bias=bias)
    be.lib.vdk_gemm_force_kernel(2); be.lib.vdk_gemm_debug_stamps(buf.data_ptr())
    ops.gemm_nt(a, b, out=out, residual=res, bias=bias)
    torch.cuda.synchronize()
    be.lib.vdk_gemm_debug_stamps(None); be.lib.vdk_gemm_force_kernel(0)
    t = buf.view(nwg, 4).cpu().double()
    pro = (t[:, 1] - t[:, 0]); main = (t[:, 2] - t[:, 1]); epi = (t[:, 3] - t[:, 2]); tot = t[:, 3] - t[:, 0]
    span = (t[:, 3].max() - t[:, 0].min())
    print(f"M={M} N={N} K={K} nwg={nwg} res={res is not None}: cycles/workgroup prologue {pro.mean():8.0f}  main {main.mean():8.0f}  epilogue {epi.mean():8.0f}  total {tot.mean():8.0f} | "
          f"kernel span {span:10.0f} cycles = {span/ (-(-nwg//256)):8.0f} per round; sum/round overhead {(span/(-(-nwg//256)) - tot.mean()):7.0f}")
