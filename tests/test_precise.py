"""PRECISE inference path (fp32-MFMA contractions, csrc/gemm_f32.hip + vdk_*_forward_f32): logits / tokens / embeddings against the pinned fp32
oracles within north_star's 1e-3 relative tolerance (measured ~1e-6), and the consequence that matters for path B: the cosine top-k lists built from
these embeddings equal the ones built from the oracle's embeddings."""
import numpy as np
import torch

from oracle import cbir as ocbir
from oracle.convnext_ref import ConvNeXtRef, TimmWrapperCNNRef
from oracle.vit_ref import TimmWrapperRef, VisionTransformerRef
from visiondk_amd import convnext, face, vit

TOL = 1e-4   # north_star: 1e-3


def _rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm()).item()


def _perturb(ref, scale=3.0):
    torch.manual_seed(11)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.copy_(torch.rand_like(p) * 0.5 + 0.25)
            elif p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
            elif "attn" in n or "mlp" in n:
                p.mul_(scale)


def test_vit_logits_precise(be, dev):
    spec = vit.VitSpec(img_size=32, patch_size=8, num_classes=10, dim=128, depth=3, heads=2, mlp_dim=512)
    model = vit.VisionTransformer(spec, device=dev, backend=be, seed=0)
    ref = VisionTransformerRef(32, 8, 3, 10, 128, 3, 2, 512)
    _perturb(ref)
    model.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(5, 3, 32, 32)
    with torch.no_grad():
        exp = ref(x)
    got = model.forward_precise(x.to(dev))
    assert got.shape == exp.shape and _rel(got, exp) < TOL
    assert _rel(model(x.to(dev)).detach(), exp) > 10 * _rel(got, exp)      # the bf16 training path is (much) further away: the mode matters


def test_convnext_map_precise(be, dev):
    depths, dims = (1, 1, 2, 1), (8, 16, 24, 32)
    model = convnext.ConvNeXt(convnext.ConvNeXtSpec(img_size=64, depths=depths, dims=dims), device=dev, backend=be, seed=0)
    ref = ConvNeXtRef(3, depths, dims)
    _perturb(ref)
    model.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(3, 3, 64, 64)
    with torch.no_grad():
        exp = ref(x)
    got = model.forward_precise(x.to(dev))
    assert got.shape == exp.shape and _rel(got, exp) < TOL


def test_embeddings_precise_and_topk_equal(be, dev, monkeypatch):
    """TimmWrapper (backbone + neck, eval) for both backbone families; the gallery / query top-k from our embeddings equals the oracle's"""
    monkeypatch.setitem(convnext.TIMM_CONVNEXTS, "convnext_test", dict(depths=(1, 1, 2, 1), dims=(8, 16, 24, 32)))
    monkeypatch.setitem(vit.TIMM_VITS, "vit_test_patch16", dict(dim=128, depth=3, heads=2, mlp_dim=512))
    for name, img, ref in (("convnext_test", 32, TimmWrapperCNNRef(32, 32, 3, (1, 1, 2, 1), (8, 16, 24, 32))),
                           ("vit_test_patch16", 32, TimmWrapperRef(32, 32, 16, 128, 3, 2, 512))):
        _perturb(ref)
        with torch.no_grad():
            for m in ref.output_layer:
                if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                    m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
        tw = face.TimmWrapper(name, feat_dim=32, image_size=img, backend=be, device=dev)
        tw.load_state_dict({k: v.to(dev) for k, v in ref.state_dict().items()}, strict=True)
        tw.eval(); ref.eval()
        torch.manual_seed(4)
        x = torch.randn(16, 3, img, img)
        with torch.no_grad():
            exp = torch.nn.functional.normalize(ref(x)).numpy()
        got = face.FeatureExtractor(tw, precise=True).extract_cbir([x[:7], x[7:]], dev)
        assert _rel(got, exp) < TOL, (name, _rel(got, exp))
        # 12 gallery rows, 4 queries, top-5: identical index lists from our embeddings and from the oracle's
        _, i_got = ocbir.flat_ip_search(got[12:], got[:12], 5)
        _, i_exp = ocbir.flat_ip_search(exp[12:], exp[:12], 5)
        np.testing.assert_array_equal(i_got, i_exp)
