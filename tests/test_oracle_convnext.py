"""Pin the timm-ConvNeXt restatement (oracle/convnext_ref.py) against the independent `transformers.ConvNextModel` through a
weight map (timm itself is not installable here)."""
import torch

from oracle.convnext_ref import ConvNeXtRef


def test_convnext_ref_matches_transformers():
    from transformers import ConvNextConfig, ConvNextModel
    torch.manual_seed(0)
    depths, dims, img = (2, 1, 2, 1), (16, 32, 48, 64), 64
    ref = ConvNeXtRef(3, depths, dims).eval()
    with torch.no_grad():  # non-trivial biases / norms / layer scales
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.copy_(torch.rand_like(p) + 0.5)
            elif p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    cfg = ConvNextConfig(num_channels=3, patch_size=4, num_stages=4, hidden_sizes=list(dims), depths=list(depths), hidden_act="gelu",
                         layer_norm_eps=1e-6, layer_scale_init_value=1e-6, drop_path_rate=0.0, image_size=img)
    hf = ConvNextModel(cfg).eval()
    sd, hsd = ref.state_dict(), hf.state_dict()
    used = set()

    def put(name, value):
        assert name in hsd, (name, [k for k in hsd if k.split(".")[0] == name.split(".")[0]][:8])
        assert hsd[name].shape == value.shape, (name, hsd[name].shape, value.shape)
        hsd[name] = value.clone(); used.add(name)

    for kind in ("weight", "bias"):
        put(f"embeddings.patch_embeddings.{kind}", sd[f"stem.0.{kind}"])
        put(f"embeddings.layernorm.{kind}", sd[f"stem.1.{kind}"])
        put(f"layernorm.{kind}", sd[f"head.norm.{kind}"])
    for i in range(4):
        if i > 0:
            for kind in ("weight", "bias"):
                put(f"encoder.stages.{i}.downsampling_layer.0.{kind}", sd[f"stages.{i}.downsample.0.{kind}"])
                put(f"encoder.stages.{i}.downsampling_layer.1.{kind}", sd[f"stages.{i}.downsample.1.{kind}"])
        for j in range(depths[i]):
            t, h = f"stages.{i}.blocks.{j}", f"encoder.stages.{i}.layers.{j}"
            put(f"{h}.layer_scale_parameter", sd[f"{t}.gamma"])
            for kind in ("weight", "bias"):
                put(f"{h}.dwconv.{kind}", sd[f"{t}.conv_dw.{kind}"])
                put(f"{h}.layernorm.{kind}", sd[f"{t}.norm.{kind}"])
                put(f"{h}.pwconv1.{kind}", sd[f"{t}.mlp.fc1.{kind}"])
                put(f"{h}.pwconv2.{kind}", sd[f"{t}.mlp.fc2.{kind}"])
    assert used == set(hsd.keys()), set(hsd.keys()) - used
    hf.load_state_dict(hsd)
    x = torch.randn(2, 3, img, img)
    with torch.no_grad():
        out = hf(pixel_values=x)
        trunk = ref.forward_trunk(x)
        rel = ((trunk - out.last_hidden_state).norm() / out.last_hidden_state.norm()).item()
        assert rel < 1e-5, rel
        # head.norm: HF applies the same LayerNorm after pooling; LN is per pixel so compare on the pooled map
        pooled = ref.head.norm(trunk.mean((-2, -1), keepdim=True)).flatten(1)
        rel2 = ((pooled - out.pooler_output).norm() / out.pooler_output.norm()).item()
        assert rel2 < 1e-5, rel2


def test_convnext_classifier_head_matches_transformers():
    """timm's classifier head (pool -> head.norm -> head.fc) against transformers.ConvNextForImageClassification (pooler LayerNorm -> classifier)."""
    from transformers import ConvNextConfig, ConvNextForImageClassification
    torch.manual_seed(1)
    depths, dims, img, ncls = (1, 1, 2, 1), (16, 32, 48, 64), 64, 7
    ref = ConvNeXtRef(3, depths, dims, num_classes=ncls).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.copy_(torch.rand_like(p) + 0.5)
            elif p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    cfg = ConvNextConfig(num_channels=3, patch_size=4, num_stages=4, hidden_sizes=list(dims), depths=list(depths), hidden_act="gelu",
                         layer_norm_eps=1e-6, layer_scale_init_value=1e-6, drop_path_rate=0.0, image_size=img, num_labels=ncls)
    hf = ConvNextForImageClassification(cfg).eval()
    sd, hsd = ref.state_dict(), hf.state_dict()
    m = {"classifier.weight": "head.fc.weight", "classifier.bias": "head.fc.bias"}
    for kind in ("weight", "bias"):
        m[f"convnext.embeddings.patch_embeddings.{kind}"] = f"stem.0.{kind}"
        m[f"convnext.embeddings.layernorm.{kind}"] = f"stem.1.{kind}"
        m[f"convnext.layernorm.{kind}"] = f"head.norm.{kind}"
    for i in range(4):
        for kind in ("weight", "bias"):
            if i > 0:
                m[f"convnext.encoder.stages.{i}.downsampling_layer.0.{kind}"] = f"stages.{i}.downsample.0.{kind}"
                m[f"convnext.encoder.stages.{i}.downsampling_layer.1.{kind}"] = f"stages.{i}.downsample.1.{kind}"
            for j in range(depths[i]):
                t, h = f"stages.{i}.blocks.{j}", f"convnext.encoder.stages.{i}.layers.{j}"
                m[f"{h}.layer_scale_parameter"] = f"{t}.gamma"
                for a, b in (("dwconv", "conv_dw"), ("layernorm", "norm"), ("pwconv1", "mlp.fc1"), ("pwconv2", "mlp.fc2")):
                    m[f"{h}.{a}.{kind}"] = f"{t}.{b}.{kind}"
    assert set(m) == set(hsd), set(hsd) ^ set(m)
    hf.load_state_dict({k: sd[v].clone() for k, v in m.items()})
    x = torch.randn(3, 3, img, img)
    with torch.no_grad():
        a, b = ref(x), hf(pixel_values=x).logits
    assert a.shape == (3, ncls) and ((a - b).norm() / b.norm()).item() < 1e-5
