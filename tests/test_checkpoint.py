"""SURVEY §8(f).1 — checkpoint format compatibility: files written the way the reference writes them (vision_engine.py:387-403 for the classifier,
train.py:266-278 for face / CBIR) from the ORACLE models load into the HIP-backed models through the reference's loader methods, and a
checkpoint exported from the HIP-backed model loads back into the oracle (timm key names both ways)."""
import copy

import torch

from oracle.convnext_ref import TimmWrapperCNNRef
from oracle.vit_ref import VisionTransformerRef
from visiondk_amd import convnext, face, vit


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm()).item()


def test_classifier_checkpoint_both_ways(be, dev, tmp_path):
    vit.TIMM_VITS.setdefault("vit_test_patch16", dict(dim=128, depth=3, heads=2, mlp_dim=512))
    torch.manual_seed(0)
    ref = VisionTransformerRef(32, 16, 3, 10, 128, 3, 2, 512)
    ema = copy.deepcopy(ref)
    with torch.no_grad():
        for p in ema.parameters():
            p.mul_(0.5)
    # the reference's classifier checkpoint: 'model' = state_dict, 'ema' = the pickled EMA module (vision_engine.py:387-392)
    path = tmp_path / "best.pt"
    torch.save({"epoch": 3, "model": ref.state_dict(), "ema": ema, "updates": 7}, path)
    cfg = {"task": "classification", "name": "timm-vit_test_patch16", "image_size": 32, "num_classes": 10, "pretrained": True, "kwargs": {}}
    wrap = face.get_model(cfg, None, 0, backend=be, device=dev)
    x = torch.randn(3, 3, 32, 32)
    wrap.load_weight(str(path), ema=False, device="cpu")
    assert _rel(wrap.model(x.to(dev)).detach(), ref(x).detach()) < 3e-2
    wrap.load_weight(str(path), ema=True, device="cpu")
    assert _rel(wrap.model(x.to(dev)).detach(), ema(x).detach()) < 3e-2
    # export from the HIP-backed model -> loads into the oracle strictly and bit-exactly (fp32 master weights)
    out = tmp_path / "export.pt"
    torch.save({"model": {k: v.cpu() for k, v in wrap.model.state_dict().items()}}, out)
    back = VisionTransformerRef(32, 16, 3, 10, 128, 3, 2, 512)
    back.load_state_dict(torch.load(out, weights_only=False)["model"], strict=True)
    for k, v in ema.state_dict().items():
        assert torch.equal(back.state_dict()[k], v), k


def test_face_backbone_checkpoint(be, dev, tmp_path, monkeypatch):
    depths, dims, img = (1, 1, 2, 1), (8, 16, 24, 32), 64
    monkeypatch.setitem(convnext.TIMM_CONVNEXTS, "convnext_test", dict(depths=depths, dims=dims))
    torch.manual_seed(1)
    ref = TimmWrapperCNNRef(64, img, 3, depths, dims)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.fill_(0.3)
        ref.output_layer[0].running_var.uniform_(0.5, 1.5); ref.output_layer[3].running_mean.normal_(0, 0.1)
    ema_sd = {k: (v * 0.9 if v.dtype.is_floating_point else v) for k, v in ref.state_dict().items()}
    path = tmp_path / "Epoch_1.pt"   # train.py:266-278: 'state_dict' = backbone, 'ema' = backbone state_dict
    torch.save({"epoch": 0, "state_dict": ref.state_dict(), "ema": ema_sd, "updates": 3}, path)
    cfg = {"backbone": {"timm-convnext_test": {"pretrained": False, "image_size": img, "feat_dim": 64}}}
    loader = face.FaceModelLoader(cfg, backend=be, device=dev)
    model = loader.load_weight(str(path))
    x = torch.randn(4, 3, img, img)
    ref.eval(); model.eval()
    with torch.no_grad():
        assert _rel(model(x.to(dev)), ref(x)) < 3e-2
    model = loader.load_weight(str(path), ema=True)
    for k, v in ema_sd.items():
        assert torch.equal(model.state_dict()[k].cpu(), v), k
    assert torch.equal(loader.load_weight_default(str(path)).state_dict()["output_layer.2.weight"].cpu(), ref.state_dict()["output_layer.2.weight"])
