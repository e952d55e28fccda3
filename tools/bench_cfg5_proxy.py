"""cfg5 (stretch) is SigLIP-class ViT-L/14 @336 with Mixup + SAM on fp8 MFMA, global batch 1024.  What the engine runs today of that recipe:
ViT-L/14 @336 (timm `vit_large_patch14_224` built at img_size 336: 576 patch tokens + cls) in bf16 with the same training protocol -- a Mixup pair every step
and Trainer.update_sam (two forward-backward passes) -- so the number below is a PROXY with the differences named: cls token + linear head instead of a MAP
head, bf16 instead of fp8.  `--l16` runs ViT-L/16 @224 instead.  usage: python tools/bench_cfg5_proxy.py [batch] [steps] [--l16]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import vit

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 5
dev = torch.device("cuda:0")
l16 = "--l16" in sys.argv
name, img, ntok = ("vit_large_patch16_224", 224, 197) if l16 else ("vit_large_patch14_224", 336, 577)
model = vit.create_model(name, num_classes=1000, device=dev, img_size=img)
out = {"workload": f"cfg5 proxy: {name} @{img} ({ntok} tokens) bf16, bs {B}, Mixup pair + SAM (2 fwd/bwd per step), CE label smoothing 0.05, SGD + EMA"}
x = torch.randn(B, 3, img, img, device=dev); ya = torch.randint(0, 1000, (B,), device=dev); yb = torch.randint(0, 1000, (B,), device=dev)
flop_img = 3 * 2 * (302.3e6 * ntok + 24 * 2 * ntok * ntok * 1024)      # fwd+bwd, 2*MACs: block Linears (302 M weights x tokens) + attention
for sam in (False, True):
    step = vit.FusedTrainStep(model, lr=0.006, momentum=0.937, weight_decay=5e-4, label_smoothing=0.05, max_norm=10.0, ema=True, sam=sam)
    for _ in range(2): step.step(x, ya, yb, 0.4)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(steps): step.step(x, ya, yb, 0.4)
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps
    key = "sam" if sam else "plain"
    out[f"{key}_images_per_sec"] = B / dt; out[f"{key}_ms_per_step"] = dt * 1e3
    out[f"{key}_model_tflops"] = flop_img * B * (2 if sam else 1) / dt / 1e12
out["max_mem_gib"] = torch.cuda.max_memory_allocated() / 2**30
print(json.dumps(out))
