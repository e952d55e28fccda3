"""What the data-parallel plumbing itself costs per step (host callbacks from vdk_vit_backward, 11 bucket launches, event waits), measured on ONE GPU: the ViT-B/16 bench
step with GradAllReduce(always_communicate=True) in a one-rank RCCL group (every collective is issued; they are identities) against the communication-free step.
    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/bench_comm_overhead.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from visiondk_amd import comm as vcomm, vit

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group("nccl", rank=int(os.environ.get("RANK", 0)), world_size=int(os.environ.get("WORLD_SIZE", 1)), device_id=dev)
spec = vit.spec_from_timm_name("vit_base_patch16_224", 1000)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
out = {}
for name, comm in (("no_comm", None), ("one_rank_rccl", vcomm.GradAllReduce(always_communicate=True)), ("no_comm_again", None)):
    model = vit.VisionTransformer(spec, device=dev, seed=2)
    step = vit.FusedTrainStep(model, lr=0.006, momentum=0.937, weight_decay=5e-4, label_smoothing=0.05, max_norm=10.0, ema=True, comm=comm)
    for _ in range(4): step.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step.step(x, y)
    torch.cuda.synchronize()
    out[name + "_ms_per_step"] = (time.perf_counter() - t0) / 20 * 1e3
    if comm is not None: out["collectives_per_step"] = comm.collectives / 24
    del step, model
print(json.dumps(out))
dist.destroy_process_group()
