"""One-GPU measurement of communication / compute overlap for the data-parallel train step: ViT-B/16, batch 256, fp16 operands, the bench's 24 MB bucket
rule, collectives through the C-ABI route (csrc/comm.hip) with the timing trace on.  A one-rank all-reduce moves nothing, so behind each one a stand-in kernel holds 32 CUs on
the collectives' stream for the time an 8-GPU ring all-reduce of that bucket takes at the given bus bandwidth (2 * 7/8 * bytes / busbw).  Prints one JSON line.
usage: python tools/overlap_probe.py [batch] [busbw_GBps]"""
import json, os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from visiondk_amd import _lib, comm, vit

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
busbw = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
us_per_mb = 2.0 * 7.0 / 8.0 / busbw * 1e3            # MB / (GB/s) = ms * 1e-3 ... 1 MB at 1 GB/s = 1000 us
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
be = _lib.load()


def run(kw, steps=8):
    model = vit.VisionTransformer(vit.spec_from_timm_name("vit_base_patch16_224", 1000), device="cuda:0", backend=be, seed=3, operand="fp16")
    c = comm.GradAllReduce(always_communicate=True, **kw) if kw is not None else None
    step = vit.FusedTrainStep(model, lr=0.006, label_smoothing=0.05, ema=True, comm=c)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(batch, 3, 224, 224, generator=g).cuda(); y = torch.randint(0, 1000, (batch,), generator=g).cuda()
    for _ in range(3):
        step.step(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step.step(x, y)
    e1.record(); torch.cuda.synchronize()
    tr = c.trace_read() if (c is not None and kw.get("trace")) else None
    if c is not None:
        c.close()
    return e0.elapsed_time(e1) / steps, tr


plain, _ = run(None)
c10d, _ = run({"route": "c10d"})
abi, _ = run({"route": "abi"})
standin, tr = run({"route": "abi", "trace": True, "standin_us_per_mb": us_per_mb})
ar, marks = tr["allreduce_ms"], tr["marks_ms"]
total = sum(e - s for s, e in ar)
variants = {}
if len(sys.argv) > 3 and sys.argv[3] == "variants":      # where does the exchange's cost come from: the CUs the collective holds, or its mere presence on another stream?
    for name, kw in (("32cus_notrace", {"standin_cus": 32}), ("8cus_notrace", {"standin_cus": 8}), ("1cu_notrace", {"standin_cus": 1}),
                     ("32cus_reserve64", {"standin_cus": 32, "reserve_cus": 64}), ("32cus_reserve0", {"standin_cus": 32, "reserve_cus": 0})):
        ms_, _ = run(dict({"route": "abi", "standin_us_per_mb": us_per_mb}, **kw))
        variants[name] = ms_
print(json.dumps({"variants_ms": variants, "workload": f"ViT-B/16 bs {batch} fp16 fused step, 1 GPU, RCCL world size 1", "ms_plain": plain, "ms_c10d_route_empty_collectives": c10d,
                  "ms_abi_route_empty_collectives": abi, "standin": {"model": f"8-GPU ring all-reduce at {busbw} GB/s bus bandwidth, 32 CUs held", "us_per_mb": us_per_mb,
                  "ms_step": standin, "collectives_per_step": len(ar), "bucket_mb": [n * 4 / 1e6 for n in tr["numel"]], "collective_ms_total": total,
                  "allreduce_intervals_ms": [[round(s_, 3), round(e_, 3)] for s_, e_ in ar], "backward_end_ms": marks[1], "all_collectives_landed_ms": marks[2],
                  "ended_before_backward_end": sum(1 for s_, e_ in ar if e_ <= marks[1]), "exposed_tail_ms": marks[2] - marks[1],
                  "step_cost_of_the_exchange_ms": standin - plain, "hidden_fraction": 1.0 - (standin - plain) / total}}))
dist.destroy_process_group()
