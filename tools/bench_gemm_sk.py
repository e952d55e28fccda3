"""Whole-tile rounds vs stream-K on the Linear shapes of the ViT-B/16 step (T = 256 x 197 = 50 432 rows): per-launch HIP-event time and TFLOP/s, both launch forms."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    be = _lib.load()
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 50432
    ws = ops.streamk_workspace("cuda", backend=be)
    out = {"rows": T, "shapes": []}
    shapes = [("qkv bias", 2304, 768, "bias"), ("proj bias+res f32", 768, 768, "res"), ("fc1 bias+gelu+aux", 3072, 768, "gelu"), ("fc2 bias+res f32", 768, 3072, "res"),
              ("dfc2 dgelu", 3072, 768, "dgelu"), ("dfc1 plain", 768, 3072, "plain"), ("dproj plain", 768, 768, "plain"), ("dqkv plain", 768, 2304, "plain")]
    for name, N, K, ep in shapes:
        a = torch.randn(T, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        bias = torch.randn(N, device="cuda"); res = torch.randn(T, N, device="cuda"); aux = torch.randn(T, N, device="cuda").bfloat16()
        kw = {"plain": {}, "bias": {"bias": bias}, "res": {"bias": bias, "residual": res, "out_dtype": torch.float32}, "gelu": {"bias": bias, "act": ops.ACT_GELU, "aux": aux},
              "dgelu": {"act": ops.ACT_DGELU, "aux": aux}}[ep]
        o = torch.empty(T, N, dtype=kw.get("out_dtype", torch.bfloat16), device="cuda")
        kw = {k: v for k, v in kw.items() if k != "out_dtype"}
        be.lib.vdk_gemm_force_kernel(4)
        t0 = timeit(lambda: ops.gemm_nt(a, b, out=o, backend=be, **kw)); r0 = o.clone()
        be.lib.vdk_gemm_force_kernel(0)
        t1 = timeit(lambda: ops.gemm_nt(a, b, out=o, streamk_ws=ws, backend=be, **kw))
        be.lib.vdk_gemm_force_kernel(3)
        t2 = timeit(lambda: ops.gemm_nt(a, b, out=o, streamk_ws=ws, backend=be, **kw)); r2 = o.clone()
        be.lib.vdk_gemm_force_kernel(0)
        fl = 2.0 * T * N * K
        out["shapes"].append({"name": name, "N": N, "K": K, "tiles": ((T + 255) // 256) * (N // 256), "rounds_us": t0 * 1e6, "auto_us": t1 * 1e6, "streamk_us": t2 * 1e6,
                              "rounds_tflops": fl / t0 / 1e12, "auto_tflops": fl / t1 / 1e12, "streamk_tflops": fl / t2 / 1e12,
                              "rel_diff": ((r0.float() - r2.float()).norm() / r0.float().norm()).item()})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
