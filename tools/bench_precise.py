"""Throughput and parity of the PRECISE (fp32-MFMA) forward on ViT-B/16: images/s, and the distance of both forwards from a CPU fp32 torch reference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import vit
from oracle.vit_ref import VisionTransformerRef
dev = torch.device("cuda:0")
model = vit.create_model("vit_base_patch16_224", num_classes=1000, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(B, 3, 224, 224, device=dev)
for name, fn in (("bf16 training-path forward", lambda: model.engine.forward(x)), ("fp32-MFMA precise forward", lambda: model.engine.forward_precise(x))):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print(f"{name}: {B / dt:.0f} img/s ({dt * 1e3:.1f} ms / {B}), {35.13e9 * B / dt / 1e12:.0f} TFLOP/s")
ref = VisionTransformerRef()
ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
with torch.no_grad():
    exp = ref(x[:4].cpu())
rel = lambda a: ((a.double().cpu() - exp.double()).norm() / exp.double().norm()).item()
print("rel. distance from the CPU fp32 reference (4 images): precise", rel(model.forward_precise(x[:4])), " bf16 path", rel(model(x[:4]).detach()))
