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


def test_gemm_f32_weight_gradient_form_and_pieces(be, dev):
    """vdk_gemm_f32_nt with k-major A (dW = dY^T X as the operands lie) incl. the contraction split over batch1 (k_total), and the elementwise pieces of the fp32 training
    path against torch: GELU / GELU', row scale, column sums, depth-to-space"""
    import ctypes as C
    from visiondk_amd import _abi, ops
    torch.manual_seed(5)
    rows, out, inn = 1030, 72, 40
    dY = torch.randn(rows, out).to(dev); X = torch.randn(rows, inn).to(dev)
    ref = dY.double().T @ X.double()
    dW = ops.gemm_f32(dY, X, a_kmajor=True, b_kmajor=True, backend=be)
    assert ((dW.double() - ref).norm() / ref.norm()).item() < 2e-6
    # split over 5 slabs of 208 rows (the last one short), summed by the caller
    S, kc = 5, 208
    slabs = torch.full((S, out, inn), float("nan"), dtype=torch.float32, device=dev)
    d = _abi.GemmF32Desc()
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = dY.data_ptr(), out, X.data_ptr(), inn, slabs.data_ptr(), inn
    d.M, d.N, d.K, d.alpha, d.a_kmajor, d.b_kmajor = out, inn, kc, 1.0, 1, 1
    d.batch1, d.batch2, d.sa1, d.sb1, d.sc1, d.k_total = S, 1, kc * out, kc * inn, out * inn, rows
    be.check(be.lib.vdk_gemm_f32_nt(C.byref(d), be.stream()), "vdk_gemm_f32_nt")
    assert ((slabs.double().sum(0) - ref).norm() / ref.norm()).item() < 2e-6
    part = dY[4 * kc:].double().T @ X[4 * kc:].double()
    assert ((slabs[4].double() - part).norm() / part.norm()).item() < 2e-6
    # elementwise pieces
    u = (torch.randn(64, 36) * 2).to(dev)
    g = torch.empty_like(u)
    be.check(be.lib.vdk_gelu_f32(be.ptr(u), be.ptr(g), u.numel(), be.stream()), "vdk_gelu_f32")
    torch.testing.assert_close(g.cpu(), torch.nn.functional.gelu(u.cpu()), rtol=2e-6, atol=2e-6)      # (1 + erf cancels in the negative tail: absolute, not relative)
    uu = u.cpu().clone().requires_grad_(True)
    torch.nn.functional.gelu(uu).sum().backward()
    dd = torch.randn(64, 36).to(dev); want = dd.cpu() * uu.grad
    be.check(be.lib.vdk_dgelu_f32(be.ptr(dd), be.ptr(u), u.numel(), be.stream()), "vdk_dgelu_f32")
    torch.testing.assert_close(dd.cpu(), want, rtol=3e-6, atol=3e-6)
    w = torch.randn(24, 96).to(dev); sc = torch.randn(24).to(dev); o = torch.empty_like(w)
    be.check(be.lib.vdk_rowscale_f32(be.ptr(w), be.ptr(sc), be.ptr(o), 24, 96, be.stream()), "vdk_rowscale_f32")
    assert torch.equal(o.cpu(), (w.cpu() * sc.cpu()[:, None]))
    xs = torch.randn(5000, 72).to(dev); cs = torch.empty(72, device=dev)
    need = C.c_size_t(0)
    be.check(be.lib.vdk_colsum_f32_workspace_bytes(5000, 72, C.byref(need)), "ws")
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    be.check(be.lib.vdk_colsum_f32(be.ptr(xs), 72, 5000, 72, be.ptr(cs), be.ptr(ws), ws.numel(), be.stream()), "vdk_colsum_f32")
    torch.testing.assert_close(cs.cpu().double(), xs.cpu().double().sum(0), rtol=1e-5, atol=1e-4)
    a = torch.randn(2 * 6 * 6 * 8).to(dev); mid = torch.empty_like(a); back = torch.empty_like(a)
    be.check(be.lib.vdk_space_to_depth2_f32(be.ptr(a), be.ptr(mid), 2, 6, 6, 8, be.stream()), "s2d")
    be.check(be.lib.vdk_depth_to_space2_f32(be.ptr(mid), be.ptr(back), 2, 6, 6, 8, be.stream()), "d2s")
    assert torch.equal(a.cpu(), back.cpu())
