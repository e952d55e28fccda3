"""Native ConvNeXt engine (csrc/convnext_engine.hip) vs the pinned oracle (oracle/convnext_ref.py): forward map and every
parameter gradient, on the CPU SIMT emulation and (-m gpu) on the MI355X through the same C ABI."""
import pytest
import torch

from oracle.convnext_ref import ConvNeXtRef
from visiondk_amd import convnext


def _pair(be, dev, depths, dims, img, seed=0):
    spec = convnext.ConvNeXtSpec(img_size=img, depths=depths, dims=dims)
    model = convnext.ConvNeXt(spec, device=dev, backend=be, seed=seed)
    ref = ConvNeXtRef(3, depths, dims)
    torch.manual_seed(seed)
    with torch.no_grad():   # non-trivial values everywhere (timm's init leaves biases at 0 and gamma at 1e-6: gradients would hide bugs)
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.copy_(torch.rand_like(p) * 0.5 + 0.25)
            elif p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
            else:
                p.copy_(torch.randn_like(p) * (0.5 / (p[0].numel() ** 0.5)))
    missing, unexpected = model.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected
    return model, ref


def test_state_dict_names_match_timm_layout(be, dev):
    model, ref = _pair(be, dev, (1, 1, 2, 1), (8, 16, 24, 32), 32)
    assert list(model.state_dict().keys()) == list(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert model.state_dict()[k].shape == v.shape, k


@pytest.mark.parametrize("B,img,depths,dims", [(2, 32, (1, 1, 2, 1), (8, 16, 24, 32)), (4, 64, (2, 1, 1, 1), (16, 32, 64, 72))])
def test_forward_backward_vs_oracle(be, dev, B, img, depths, dims):
    model, ref = _pair(be, dev, depths, dims, img)
    torch.manual_seed(3)
    x = torch.randn(B, 3, img, img)
    y = model(x.to(dev))
    yr = ref(x)
    assert y.shape == yr.shape
    rel = ((y.detach().cpu() - yr.detach()).norm() / yr.detach().norm()).item()
    assert rel < 2e-2, rel            # bf16 GEMM operands vs the fp32 oracle
    dy = torch.randn_like(yr)
    y.backward(dy.to(dev))
    yr.backward(dy)
    worst = []
    for (n, p), (nr, pr) in zip(model.named_parameters(), ref.named_parameters()):
        assert n == nr
        g, gr = p.grad.detach().cpu(), pr.grad
        r = ((g - gr).norm() / (gr.norm() + 1e-12)).item()
        worst.append((r, n))
        assert r < 6e-2, (n, r)
    # the bulk is far better than the bound
    worst.sort()
    assert worst[len(worst) // 2][0] < 2e-2, worst[len(worst) // 2]
