"""swin_base_patch4_window7_224 (the default backbone of pet.yaml / cbir.yaml) classifier step on one MI355X through the native engine (csrc/swin_engine.hip: one C-ABI
call forward, one backward) and vit.FusedTrainStep (CE + clip_grad_norm_ + SGD + EMA in device kernels); `autograd` as third argument times round 3's form beside it
(autograd nodes over the same kernels + torch SGD).
usage: python tools/bench_swin.py [batch] [steps] [native|autograd] [bf16|fp16]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import swin, vit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
form = sys.argv[3] if len(sys.argv) > 3 else "native"
operand = sys.argv[4] if len(sys.argv) > 4 else "bf16"      # fp16 = the reference's autocast dtype (loss scaling inside the fused step)
dev = torch.device("cuda:0")
model = swin.create_model("swin_base_patch4_window7_224", num_classes=37, device=dev, seed=0, native=form == "native", operand=operand)
g = torch.Generator(device="cpu"); g.manual_seed(0)
x = torch.randn(B, 3, 224, 224, generator=g).to(dev); y = torch.randint(0, 37, (B,), generator=g).to(dev)

if form == "native":
    fused = vit.FusedTrainStep(model, lr=0.006, momentum=0.937, weight_decay=5e-4, label_smoothing=0.05, max_norm=10.0, ema=True)

    def step():
        fused.step(x, y)
        return fused
    value = lambda s: s.loss_value()
else:
    opt = torch.optim.SGD(model.parameters(), lr=0.006, momentum=0.937, weight_decay=5e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), y, label_smoothing=0.05)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        return loss
    value = lambda l: l.item()

losses = [value(step()) for _ in range(2)]
torch.cuda.synchronize(); t0 = time.time()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
losses.append(value(l))
flop = 3 * 15.47e9 * B
print(json.dumps({"workload": f"swin_base_patch4_window7_224, 37 classes, bs {B}: fwd + CE + bwd + clip + SGD" + (" + EMA (native engine + fused step)" if form == "native" else
                                                                                                                " (autograd nodes over the HIP kernels + torch SGD)"),
                  "operand": operand, "drop_path_rate": getattr(getattr(model, "engine", None), "drop_path_rate", 0.0), "ms_per_step": dt * 1e3, "images_per_sec": B / dt, "model_tflops": flop / dt / 1e12, "losses": losses, "max_mem_gib": torch.cuda.max_memory_allocated() / 2**30}))
