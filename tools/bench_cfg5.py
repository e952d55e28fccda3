"""BASELINE.json configs[4]: SigLIP ViT-L/14 336x336 ICT + Mixup/SAM, bs 1024 over 8 GPUs = 128 per GPU.  What runs here: the REAL architecture (timm
class_token=False + global_pool='map': 576 patch tokens, AttentionPoolLatent head, visiondk_amd/vit.py VisionTransformerMap) with the reference's protocol (a Mixup
pair every step, Trainer.update_sam = two forward-backward passes, clip-free SAM step, EMA) on ONE GPU at the per-GPU batch.  Operand precision: bf16 in the engine
and, in the *_fp8 entries, the engine's fp8 mode (forward and input-gradient GEMMs of the block Linears on e4m3 / e5m2 operands with delayed per-tensor scaling, csrc/gemm_fp8.hip;
weight gradients stay bf16).  VDK_FP8_FUSED_QUANT=0 restores the separate quantisation passes (A/B).
usage: python tools/bench_cfg5.py [batch] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import ops, vit

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 4
dev = torch.device("cuda:0")
model = vit.create_model("vit_large_patch14_siglip_336", num_classes=1000, device=dev)
ntok = model.engine.tokens
out = {"workload": f"cfg5: vit_large_patch14_siglip_336 ({ntok} tokens, MAP head) bf16 operands, per-GPU batch {B}, Mixup pair + SAM (2 fwd/bwd per step), CE ls 0.05, SGD + EMA",
       "dtype": "*_bf16: bf16 operands; *_fp8: fp8 operands in the forward / input-gradient GEMMs of the block Linears (fp8 copies of the LayerNorm / GELU / dGELU outputs written by the producing kernels, attention outputs quantised by a separate pass), bf16 weight gradients"}
x = torch.randn(B, 3, 336, 336, device=dev); ya = torch.randint(0, 1000, (B,), device=dev)
perm = torch.randperm(B, device=dev); yb = ya[perm].contiguous()
flop_img = 3 * 2 * (302.3e6 * ntok + 24 * 2 * ntok * ntok * 1024 + 2 * 1024 * 1024 * ntok)      # fwd+bwd, 2*MACs: block Linears + attention + the kv Linear of the MAP head
for sam, fp8 in ((False, 0), (True, 0), (False, 1), (True, 1)):
    model.engine.enable_fp8(fp8)
    step = vit.MapTrainStep(model, lr=0.006, momentum=0.937, weight_decay=5e-4, label_smoothing=0.05, max_norm=10.0, ema=True, sam=sam)
    def one():
        xm = ops.mixup(x, perm, 0.4)
        step.step(xm, ya, yb, 0.4)
    for _ in range(2): one()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(steps): one()
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps
    key = ("sam" if sam else "plain") + ("_fp8" if fp8 else "_bf16")
    out[f"{key}_images_per_sec"] = B / dt; out[f"{key}_ms_per_step"] = dt * 1e3
    out[f"{key}_model_tflops"] = flop_img * B * (2 if sam else 1) / dt / 1e12
    out[f"{key}_loss"] = step.loss_value()
    del step
out["max_mem_gib"] = torch.cuda.max_memory_allocated() / 2**30
print(json.dumps(out))
