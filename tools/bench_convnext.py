"""cfg3 backbone leg: ConvNeXt-B forward + backward on the native engine (images/sec), optionally with the TimmWrapper neck + ArcFace.
--classifier: the classification task on the same backbone (pet.yaml's `timm-convnext_base`, 35 classes): get_model -> VisionWrapper -> ClassifierTrainStep
(CE label smoothing 0.05 + clip + SGD + EMA), the full step.
usage: python tools/bench_convnext.py [batch] [steps] [--face | --classifier]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import convnext, face

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 5
dev = torch.device("cuda:0")
x = torch.randn(B, 3, 224, 224, device=dev)
if "--face" in sys.argv:
    cfg = {"task": "cbir", "image_size": 224, "backbone": {"timm-convnext_base": {"pretrained": False, "image_size": 224, "feat_dim": 512}},
           "head": {"arcface": {"feat_dim": 512, "num_class": int(os.environ.get("NUM_CLASS", 1000000)), "margin_arc": 0.35, "margin_am": 0.0, "scale": 32}}}
    model = face.get_model(cfg, None, 0).model.train()
    y = torch.randint(0, 1000, (B,), device=dev)
    def step():
        loss = torch.nn.functional.cross_entropy(model(x, y), y)
        loss.backward()
elif "--classifier" in sys.argv:
    from visiondk_amd import resnet
    cfg = {"task": "classification", "name": "timm-convnext_base.clip_laion2b_augreg_ft_in1k", "image_size": 224, "num_classes": 35, "pretrained": False, "kwargs": {}}
    model = face.get_model(cfg, None, 0).model
    ts = resnet.ClassifierTrainStep(model, lr=0.006, momentum=0.937, weight_decay=5e-4, loss="ce", label_smoothing=0.05, max_norm=10.0, ema=True)
    y = torch.randint(0, 35, (B,), device=dev)
    def step():
        ts.step(x, y)
else:
    model = convnext.create_model("convnext_base", device=dev)
    eng = model.engine
    dout = None
    def step():
        global dout
        out = eng.forward(x)
        if dout is None:
            dout = torch.randn_like(out)
        eng.backward(dout)
step(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
flop = 92.1e9 * B
print(f"B={B} {dt*1e3:.1f} ms/step  {B/dt:.0f} img/s  backbone fwd+bwd {flop/dt/1e12:.0f} TFLOP/s (92.1 GFLOP/img)  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
