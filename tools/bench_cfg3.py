"""cfg3: ConvNeXt-B + neck (feat_dim 512) + ArcFace(C = 1 000 000, m = .35, s = 32), bs 512 on one GPU; one step = compute_loss(face=True)
+ Trainer.update (fwd, fused head+CE, bwd, clip 10, SGD, EMA).  usage: python tools/bench_cfg3.py [batch] [steps] [num_class]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import face

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ncls = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
precision = sys.argv[5] if len(sys.argv) > 5 else "bf16"      # "fp32": the fp32-class arithmetic mode (the reference's no-autocast face / CBIR loop)
planes = int(sys.argv[4]) if len(sys.argv) > 4 else 1      # 1: the head's cosines from single 16-bit operands, 3: split planes (fp32-class)
# operand format of backbone, neck and head: "fp16" (default) = the mode tests/test_parity_fullsize_gpu.py holds to north_star's tolerance against the reference's fp32 loop
# (embeddings <= 1e-3, gradients <= 5e-3 at C = 10^6), with the GradScaler protocol inside the step; "bf16" = rounds 1-4's mode (embeddings 5.8e-3)
operand = sys.argv[6] if len(sys.argv) > 6 else "fp16"
if precision == "fp32":
    operand = "bf16"
dev = torch.device("cuda:0")
cfg = {"task": "cbir", "image_size": 224, "backbone": {"timm-convnext_base": {"pretrained": False, "image_size": 224, "feat_dim": 512, "operand": operand}},
       "head": {"arcface": {"feat_dim": 512, "num_class": ncls, "margin_arc": 0.35, "margin_am": 0.0, "scale": 32}}}
torch.manual_seed(0)
model = face.get_model(cfg, None, 0).model.train()
step = face.FaceTrainStep(model, lr=0.01, momentum=0.9, weight_decay=5e-4, max_norm=10.0, ema=True, layer_wise=True, cos_planes=planes, precision=precision)   # cbir.yaml:111-113
g = torch.Generator(device="cpu"); g.manual_seed(0)
x = torch.randn(B, 3, 224, 224, generator=g).to(dev)
y = torch.randint(0, ncls, (B,), generator=g).to(dev)
losses = []
for _ in range(2):
    losses.append(step.step(x, y).mean().item())
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    rows = step.step(x, y)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
losses.append(rows.mean().item())
flop = (92.1e9 + 0.15e9 + 3.07e9 * ncls / 1e6) * B
print(json.dumps({"workload": f"cfg3 ConvNeXt-B + neck512 + ArcFace(C={ncls}) bs={B}, fwd+bwd+clip+SGD+EMA, " + (f"{operand} operands / fp32 master" + (" + GradScaler protocol" if operand == "fp16" else "") if precision == "bf16" else "fp32-class arithmetic (fp32 activations, fp32 MFMA)"), "ms_per_step": dt * 1e3,
                  "operand": operand, "loss_scale": step.loss_scale(), "skipped_steps": step.skipped_steps(), "steps_run": steps + 2, "head_cos_planes": planes, "images_per_sec": B / dt, "tflops": flop / dt / 1e12, "losses": losses, "max_mem_gib": torch.cuda.max_memory_allocated() / 2**30}))
