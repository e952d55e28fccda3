"""TEST INFRASTRUCTURE: one forward + backward of a HIP engine against four evaluations of the oracle (the fourth: the fp32 module under torch.autocast("cpu", bfloat16)) (used by tests/test_parity_bf16.py and by bench.py's
`parity` leg, never by the product): bf16 operands with float32 accumulation (`o32`), the same arithmetic with float64 accumulation (`o64`), and plain
fp32 (what the reference runs on CPU, engine/procedure/train.py:118 with autocast off).  The distance between o32 and o64 -- two valid evaluations of the same
bf16-operand arithmetic -- is the floor below which no bf16-operand engine can be pinned (see tests/test_parity_bf16.py)."""
from __future__ import annotations

import copy

import torch

from . import bf16ops
from .vit_ref import VisionTransformerRef


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


def vit_pair(be, dev, img, patch, dim, depth, heads, mlp, classes, seed=0, operand="bf16", pre_norm=False, eps=1e-6):
    torch.manual_seed(seed)
    ref = VisionTransformerRef(img, patch, 3, classes, dim, depth, heads, mlp, eps=eps, pre_norm=pre_norm)       # reference initialisation (classify_model.py:70-81)
    with torch.no_grad():                                                            # every bias / norm / cls path carries signal
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
        ref.cls_token.add_(torch.randn_like(ref.cls_token) * 0.02)
    from visiondk_amd import vit
    model = vit.VisionTransformer(vit.VitSpec(img_size=img, patch_size=patch, num_classes=classes, dim=dim, depth=depth, heads=heads, mlp_dim=mlp, ln_eps=eps, pre_norm=pre_norm), device=dev, backend=be, seed=1,
                                  operand=operand)
    model.load_state_dict(ref.state_dict())
    return ref, model


def vit_fwd_bwd_vs_oracle(ref, model, x, y, dev, smoothing=0.05, loss_scale=None):
    """one forward + backward of the engine against evaluations of the oracle in the ENGINE's operand format (model.engine.operand: bf16 | fp16): 16-bit operands with
    float32 accumulation (o32), the same with float64 accumulation (o64), plain fp32 (the reference's CPU path), and the fp32 module under torch.autocast("cpu", that
    dtype).  fp16: every backward -- the engine's and the oracle's -- runs on loss * loss_scale and its gradients are divided by it afterwards, as GradScaler does
    (train.py:205-208); default 1024.  Returns the measured errors and the o32-vs-o64 floor."""
    operand = getattr(model.engine, "operand", "bf16")
    omode = "fp16_operands" if operand == "fp16" else "bf16_operands"
    S = float(loss_scale) if loss_scale is not None else (1024.0 if operand == "fp16" else 1.0)
    logits = model(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=smoothing)
    (loss * S).backward()
    eng = {n: p.grad.detach().double().cpu() / S for n, p in model.named_parameters()}
    evals = {}
    for name, mode, net, xx in (("o32", omode, ref, x), ("o64", omode, copy.deepcopy(ref).double(), x.double()), ("fp32", "fp32", ref, x)):
        for p in net.parameters():
            p.grad = None
        sc = S if mode != "fp32" else 1.0
        with bf16ops.precision(mode):
            lr = net(xx)
            l2 = torch.nn.functional.cross_entropy(lr, y, label_smoothing=smoothing)
            (l2 * sc).backward()
        evals[name] = (lr.detach().double(), l2.item(), {n: p.grad.detach().double() / sc for n, p in net.named_parameters()})

    # An arm that is not the builder's own arithmetic: the same fp32 module under torch.autocast("cpu", dtype) -- PyTorch's own choice of which ops read
    # 16-bit operands (what engine/procedure/train.py:118 switches on for the reference, CPU op lists instead of CUDA's).  Its distance from fp32 is an independent
    # measurement of what 16-bit operands cost this network; its distance from o32 says how well oracle/bf16ops.py restates autocast.
    for p in ref.parameters():
        p.grad = None
    try:
        with bf16ops.precision("fp32"), torch.autocast("cpu", dtype=torch.float16 if operand == "fp16" else torch.bfloat16):
            lr = ref(x)
            l2 = torch.nn.functional.cross_entropy(lr.float(), y, label_smoothing=smoothing)
        (l2 * S).backward()
        evals["autocast"] = (lr.detach().double(), l2.item(), {n: p.grad.detach().double() / S for n, p in ref.named_parameters()})
    except RuntimeError:      # (an op without a CPU kernel for the autocast dtype: the arm is simply absent)
        evals["autocast"] = None
    for p in ref.parameters():
        p.grad = None

    def dist(a, b):     # (logits, loss, worst gradient, its name) of evaluation a against evaluation b
        worst, wn = 0.0, None
        for n in a[2]:
            r = _rel(a[2][n], b[2][n])
            if r > worst:
                worst, wn = r, n
        return {"logits": _rel(a[0], b[0]), "loss": abs(a[1] - b[1]) / abs(b[1]), "worst_grad": worst, "worst_grad_name": wn}

    e = (logits.detach().double().cpu(), loss.item(), eng)
    out = {"operand": operand, "loss_scale": S, "vs_o32": dist(e, evals["o32"]), "vs_o64": dist(e, evals["o64"]), "vs_fp32": dist(e, evals["fp32"]),
           "floor_o32_vs_o64": dist(evals["o32"], evals["o64"]), "oracle16_vs_fp32": dist(evals["o32"], evals["fp32"])}
    if evals["autocast"] is not None:
        out.update({"vs_torch_autocast": dist(e, evals["autocast"]), "torch_autocast_vs_fp32": dist(evals["autocast"], evals["fp32"]),
                    "torch_autocast_vs_o32": dist(evals["autocast"], evals["o32"])})
    for p in model.parameters():
        p.grad = None
    return out
