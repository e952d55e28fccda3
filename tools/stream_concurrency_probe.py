"""Do kernels of two HIP streams run concurrently on this box?  A train of GEMMs on stream A, one small long-running kernel (vdk_debug_occupy_cus: 1 / 32 workgroups) on stream B.
Concurrent: wall = max(A, B).  Serialised: wall = A + B.  Variants: which stream is the legacy default stream, who starts first, whether B waits on an event of A (the data-parallel
exchange's pattern).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import _lib, ops

be = _lib.load()
dev = "cuda:0"
torch.cuda.set_device(0)
a = torch.randn(8192, 1024, device=dev).bfloat16(); b = torch.randn(4096, 1024, device=dev).bfloat16()
out = torch.empty(8192, 4096, dtype=torch.bfloat16, device=dev)
NG = 40


def gemms():
    for _ in range(NG):
        ops.gemm_nt(a, b, out=out, backend=be)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


res = {}
gemms(); torch.cuda.synchronize()
res["gemms_alone_ms"] = min(timed(gemms) for _ in range(3))
US = 5000
sb = torch.cuda.Stream()
sa = torch.cuda.Stream()


def occupy(stream, wgs):
    be.check(be.lib.vdk_debug_occupy_cus(wgs, US, stream.cuda_stream), "occupy")


res["occupy_alone_ms"] = timed(lambda: occupy(sb, 1))
for wgs in (1, 32):
    def default_first():
        gemms(); occupy(sb, wgs)
    def occupy_first():
        occupy(sb, wgs); gemms()
    def both_side_streams():
        with torch.cuda.stream(sa):
            gemms()
        occupy(sb, wgs)
    def event_pattern():      # B waits on an event recorded on A after the first few GEMMs (the bucket pattern), A keeps going
        for i in range(NG):
            ops.gemm_nt(a, b, out=out, backend=be)
            if i == 4:
                ev = torch.cuda.Event(); ev.record(); sb.wait_event(ev); occupy(sb, wgs)
    def event_pattern_side():
        with torch.cuda.stream(sa):
            for i in range(NG):
                ops.gemm_nt(a, b, out=out, backend=be)
                if i == 4:
                    ev = torch.cuda.Event(); ev.record(sa); sb.wait_event(ev); occupy(sb, wgs)
    for name, fn in (("default_stream_gemms_then_occupy", default_first), ("occupy_then_default_stream_gemms", occupy_first), ("two_side_streams", both_side_streams),
                     ("event_pattern_default_stream", event_pattern), ("event_pattern_side_stream", event_pattern_side)):
        res[f"{name}_{wgs}wg_ms"] = min(timed(fn) for _ in range(3))
res["env"] = {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG", "AMD_SERIALIZE_KERNEL", "HSA_ENABLE_IPC_MODE_LEGACY", "AMD_LOG_LEVEL")}
print(json.dumps(res))
