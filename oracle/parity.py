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


def vit_pair(be, dev, img, patch, dim, depth, heads, mlp, classes, seed=0):
    torch.manual_seed(seed)
    ref = VisionTransformerRef(img, patch, 3, classes, dim, depth, heads, mlp)       # reference initialisation (classify_model.py:70-81)
    with torch.no_grad():                                                            # every bias / norm / cls path carries signal
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
        ref.cls_token.add_(torch.randn_like(ref.cls_token) * 0.02)
    from visiondk_amd import vit
    model = vit.VisionTransformer(vit.VitSpec(img_size=img, patch_size=patch, num_classes=classes, dim=dim, depth=depth, heads=heads, mlp_dim=mlp), device=dev, backend=be, seed=1)
    model.load_state_dict(ref.state_dict())
    return ref, model


def vit_fwd_bwd_vs_oracle(ref, model, x, y, dev, smoothing=0.05):
    """one forward + backward of the engine against three evaluations of the oracle: bf16 operands with float32 accumulation (o32), the same with float64
    accumulation (o64), and plain fp32 (the reference's CPU path).  Returns the measured errors and the o32-vs-o64 floor."""
    logits = model(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev), label_smoothing=smoothing)
    loss.backward()
    eng = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()}
    evals = {}
    for name, mode, net, xx in (("o32", "bf16_operands", ref, x), ("o64", "bf16_operands", copy.deepcopy(ref).double(), x.double()), ("fp32", "fp32", ref, x)):
        for p in net.parameters():
            p.grad = None
        with bf16ops.precision(mode):
            lr = net(xx)
            l2 = torch.nn.functional.cross_entropy(lr, y, label_smoothing=smoothing)
            l2.backward()
        evals[name] = (lr.detach().double(), l2.item(), {n: p.grad.detach().double() for n, p in net.named_parameters()})

    # An arm that is not the builder's own arithmetic: the same fp32 module under torch.autocast("cpu", dtype=bfloat16) -- PyTorch's own choice of which ops read
    # bf16 operands (what engine/procedure/train.py:118 switches on for the reference, CPU op lists instead of CUDA's).  Its distance from fp32 is an independent
    # measurement of what bf16 operands cost this network; its distance from o32 says how well oracle/bf16ops.py restates autocast.
    for p in ref.parameters():
        p.grad = None
    with bf16ops.precision("fp32"), torch.autocast("cpu", dtype=torch.bfloat16):
        lr = ref(x)
        l2 = torch.nn.functional.cross_entropy(lr.float(), y, label_smoothing=smoothing)
    l2.backward()
    evals["autocast"] = (lr.detach().double(), l2.item(), {n: p.grad.detach().double() for n, p in ref.named_parameters()})
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
    out = {"vs_o32": dist(e, evals["o32"]), "vs_o64": dist(e, evals["o64"]), "vs_fp32": dist(e, evals["fp32"]), "floor_o32_vs_o64": dist(evals["o32"], evals["o64"]),
           "vs_torch_autocast": dist(e, evals["autocast"]), "torch_autocast_vs_fp32": dist(evals["autocast"], evals["fp32"]),
           "torch_autocast_vs_o32": dist(evals["autocast"], evals["o32"])}
    for p in model.parameters():
        p.grad = None
    return out


