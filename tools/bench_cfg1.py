"""cfg1 (BASELINE.json configs[0]): ResNet-18, 5 labels, BCE loss, bs 32, 224 x 224 — the reference's CPU plumbing case, here on the MI355X
through the drop-in surface (get_model -> VisionWrapper -> native ResNet engine), beside the oracle on the host cores.  A third argument names another member of the family
(resnet50: Bottleneck blocks; VERDICT r4 asked for its re-timing); a fourth the operand format: bf16 (default) or fp16 = the reference's autocast dtype with the GradScaler
protocol around the step, the mode whose logits meet north_star's 1e-3 in eval mode (tests/test_resnet.py; eager launches: the hipGraph replay has no scaler variant).
usage: python tools/bench_cfg1.py [batch] [steps] [resnet18|resnet34|resnet50|...] [bf16|fp16]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import face, resnet
from oracle.resnet_ref import ResNetRef
from oracle.cbir import usable_cores

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
name = sys.argv[3] if len(sys.argv) > 3 else "resnet18"
operand = sys.argv[4] if len(sys.argv) > 4 else "bf16"
dev = torch.device("cuda:0")
cfg = {"task": "classification", "name": "timm-" + name, "image_size": 224, "num_classes": 5, "pretrained": False, "kwargs": {"operand": operand} if operand != "bf16" else {}}
wrap = face.get_model(cfg, None, 0)
model = wrap.model
use_graph = os.environ.get("VDK_CFG1_EAGER", "0") != "1" and operand == "bf16"      # default: the step replayed from a hipGraph (launch-bound at this size)
step = resnet.ResNetTrainStep(model, lr=0.01, momentum=0.937, weight_decay=5e-4, loss="bce", max_norm=10.0, ema=True, graph=use_graph)
g = torch.Generator(device="cpu"); g.manual_seed(0)
x = torch.randn(B, 3, 224, 224, generator=g); t = (torch.rand(B, 5, generator=g) > 0.5).float()
xd, td = x.to(dev), t.to(dev)
# parity of the first loss against the oracle with the same weights (fp32 CPU)
spec = dict(resnet.TIMM_RESNETS[name])
ref = ResNetRef(5, **{k: spec[k] for k in ("depths", "mid", "widths") if k in spec})
ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
ref.train()
loss_ref = torch.nn.functional.binary_cross_entropy_with_logits(ref(x), t).item()
loss0 = step.step(xd, td).sum().item() / (B * 5)
for _ in range(3):
    step.step(xd, td)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    rows = step.step(xd, td)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
# the oracle timed on the host cores (fwd + bwd + SGD), the reference's own way of running this config
torch.set_num_threads(usable_cores())
opt = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.937, weight_decay=5e-4)
def cpu_step():
    opt.zero_grad(); torch.nn.functional.binary_cross_entropy_with_logits(ref(x), t).backward(); torch.nn.utils.clip_grad_norm_(ref.parameters(), 10.0); opt.step()
cpu_step(); c0 = time.time(); cpu_step(); cpu_step(); cdt = (time.time() - c0) / 2
print(json.dumps({"workload": f"{'cfg1 ResNet-18' if name == 'resnet18' else name}, 5 labels, BCE, bs={B}, 224x224, fwd+bwd+clip+SGD+EMA", "operand": operand, "launch": "hipGraph replay" if use_graph else "eager", "images_per_sec": B / dt, "ms_per_step": dt * 1e3,
                  "first_loss": loss0, "first_loss_oracle_fp32": loss_ref, "last_loss": rows.sum().item() / (B * 5),
                  "cpu_baseline": {"images_per_sec": B / cdt, "cores": usable_cores(), "kind": "port", "sample": "oracle/resnet_ref.py fwd+bwd+clip+SGD, 2 steps"}}))
