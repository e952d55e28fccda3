"""Round 6: time the LDS-resident attention backward (csrc/attention_small.hip) at the bench's geometry, with the qkv.bias column-sum by-product as the engine calls it.
    VDK_HIP_LIB=<lib> python tools/r6_attn_ab.py [fp16|bf16]      (process-level A/B: run with the old and the new library alternately)"""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from visiondk_amd import _lib, ops  # noqa: E402


def main():
    dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
    B, N, H = 256, 197, 12
    be = _lib.load()
    D = H * 64
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3 * D, device="cuda").to(dt)
    dout = torch.randn(B, N, D, device="cuda").to(dt)
    o, lse = ops.attention_fwd(qkv, H, backend=be)
    dqkv = torch.empty_like(qkv)
    dvec = torch.empty(B, H, N, device="cuda")
    cs = torch.zeros(B, 3 * D, device="cuda")
    prod = C.c_int32(0)
    f = be.lib.vdk_attention_bwd_cs      # (in-library entry of the engine: not in the public table, so the argument types are declared here)
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                  C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]

    def bwd_cs():
        be.check(be.lib.vdk_attention_bwd_cs(be.ptr(qkv), 3 * D, be.ptr(o), be.ptr(dout), D, be.ptr(lse), be.ptr(dqkv), 3 * D, be.ptr(dvec), B, N, H, 64, 0.125, ops._dt(dt),
                                              be.ptr(cs), C.byref(prod), be.stream()), "bwd_cs")

    def bwd():
        ops.attention_bwd(qkv, o, dout, lse, H, backend=be)

    res = {"dtype": str(dt)}
    for name, fn in (("bwd_cs_us", bwd_cs), ("bwd_us", bwd), ("fwd_us", lambda: ops.attention_fwd(qkv, H, backend=be))):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        res[name] = round(sorted(ts)[2], 1)
    # numbers to compare across libraries: checksums of the outputs
    bwd_cs(); torch.cuda.synchronize()
    res["dqkv_abs_sum"] = dqkv.float().abs().sum().item()
    res["cs_abs_sum"] = cs.abs().sum().item()
    res["colsum_err"] = (cs - dqkv.float().sum(1)).abs().max().item()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
