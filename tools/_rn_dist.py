import sys, copy, torch
sys.path.insert(0, '.')
from oracle.resnet_ref import ResNetRef
from visiondk_amd import resnet, _lib
hip = _lib.load()
def rel(a, b): return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()
torch.manual_seed(0)
ref = ResNetRef(1000, 3, (64, 128, 256, 512), (2, 2, 2, 2))
sd0 = copy.deepcopy(ref.state_dict())
torch.manual_seed(1)
x = torch.randn(16, 3, 224, 224); y = torch.randint(0, 1000, (16,))
def run_ref(net, xx, train):
    net.load_state_dict({k: v.to(next(net.parameters()).dtype) if v.is_floating_point() else v for k, v in sd0.items()})
    net.train(train)
    for p in net.parameters(): p.grad = None
    l = net(xx); torch.nn.functional.cross_entropy(l, y, label_smoothing=0.05).backward()
    return l.detach(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}
ref64 = copy.deepcopy(ref).double()
for train in (True, False):
    l32, g32 = run_ref(ref, x, train)
    l64, g64 = run_ref(ref64, x.double(), train)
    spread = sorted((rel(g32[n], g64[n]), n) for n in g32)
    print("mode", "train" if train else "eval", "| fp32 oracle vs fp64 oracle: logits %.2e grads median %.2e max %.2e (%s)" % (rel(l32, l64), spread[len(spread)//2][0], spread[-1][0], spread[-1][1]))
    for operand in ("fp16", "bf16"):
        model = resnet.create_model("resnet18", num_classes=1000, device="cuda:0", backend=hip, operand=operand)
        model.load_state_dict(sd0, strict=True)
        model.train(train)
        S = 1024.0 if operand == "fp16" else 1.0
        lo = model(x.cuda()); (torch.nn.functional.cross_entropy(lo, y.cuda(), label_smoothing=0.05) * S).backward()
        g = {n: p.grad / S for n, p in model.named_parameters()}
        e32 = sorted((rel(g[n], g32[n]), n) for n in g32); e64 = sorted((rel(g[n], g64[n]), n) for n in g32)
        print("   %s: logits vs fp32 %.2e vs fp64 %.2e | grads vs fp32: median %.2e max %.2e (%s) | vs fp64: median %.2e max %.2e" % (operand, rel(lo.detach(), l32), rel(lo.detach(), l64),
              e32[len(e32)//2][0], e32[-1][0], e32[-1][1], e64[len(e64)//2][0], e64[-1][0]))
        del model
