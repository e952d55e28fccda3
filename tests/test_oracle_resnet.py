"""Pin the timm-ResNet restatement (oracle/resnet_ref.py) against the independent `transformers.ResNetModel` through a weight map."""
import torch

from oracle.resnet_ref import ResNetRef


def test_resnet_ref_matches_transformers():
    from transformers import ResNetConfig, ResNetModel
    torch.manual_seed(0)
    widths, depths = (8, 16, 24, 32), (2, 2, 2, 2)
    ref = ResNetRef(5, 3, widths, depths).eval()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            elif isinstance(m, torch.nn.Conv2d):
                m.weight.normal_(0, 0.2)
    cfg = ResNetConfig(num_channels=3, embedding_size=widths[0], hidden_sizes=list(widths), depths=list(depths), layer_type="basic", hidden_act="relu",
                       downsample_in_first_stage=False)
    hf = ResNetModel(cfg).eval()
    sd, hsd = ref.state_dict(), hf.state_dict()
    used = set()

    def put(h, t):
        for suffix in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            hk, tk = f"{h}.{suffix}", f"{t}.{suffix}"
            if tk in sd:
                assert hk in hsd and hsd[hk].shape == sd[tk].shape, (hk, tk)
                hsd[hk] = sd[tk].clone(); used.add(hk)

    put("embedder.embedder.convolution", "conv1"); put("embedder.embedder.normalization", "bn1")
    for i in range(4):
        for j in range(depths[i]):
            t, h = f"layer{i + 1}.{j}", f"encoder.stages.{i}.layers.{j}"
            put(f"{h}.layer.0.convolution", f"{t}.conv1"); put(f"{h}.layer.0.normalization", f"{t}.bn1")
            put(f"{h}.layer.1.convolution", f"{t}.conv2"); put(f"{h}.layer.1.normalization", f"{t}.bn2")
            put(f"{h}.shortcut.convolution", f"{t}.downsample.0"); put(f"{h}.shortcut.normalization", f"{t}.downsample.1")
    assert used == set(hsd.keys()), sorted(set(hsd.keys()) - used)[:8]
    hf.load_state_dict(hsd)
    x = torch.randn(2, 3, 64, 64)
    with torch.no_grad():
        out = hf(pixel_values=x)
        feats = ref.forward_features(x)
    rel = ((feats - out.last_hidden_state).norm() / out.last_hidden_state.norm()).item()
    assert rel < 1e-5, rel
    rel = ((feats.mean((-2, -1)) - out.pooler_output.flatten(1)).norm() / out.pooler_output.norm()).item()
    assert rel < 1e-5, rel


def test_bottleneck_resnet_ref_matches_transformers():
    """resnet50-family blocks (1x1 -> 3x3 with the stride -> 1x1, expansion 4) against transformers' layer_type="bottleneck" """
    from transformers import ResNetConfig, ResNetModel
    torch.manual_seed(1)
    widths, mid, depths = (32, 64, 96, 128), (8, 16, 24, 32), (2, 1, 2, 1)
    ref = ResNetRef(5, 3, widths, depths, mid=mid, stem_width=16).eval()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            elif isinstance(m, torch.nn.Conv2d):
                m.weight.normal_(0, 0.2)
    cfg = ResNetConfig(num_channels=3, embedding_size=16, hidden_sizes=list(widths), depths=list(depths), layer_type="bottleneck", hidden_act="relu",
                       downsample_in_first_stage=False, downsample_in_bottleneck=False)
    hf = ResNetModel(cfg).eval()
    sd, hsd = ref.state_dict(), hf.state_dict()
    used = set()

    def put(h, t):
        for suffix in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            hk, tk = f"{h}.{suffix}", f"{t}.{suffix}"
            if tk in sd:
                assert hk in hsd and hsd[hk].shape == sd[tk].shape, (hk, tk, hsd.get(hk, torch.zeros(0)).shape, sd[tk].shape)
                hsd[hk] = sd[tk].clone(); used.add(hk)

    put("embedder.embedder.convolution", "conv1"); put("embedder.embedder.normalization", "bn1")
    for i in range(4):
        for j in range(depths[i]):
            t, h = f"layer{i + 1}.{j}", f"encoder.stages.{i}.layers.{j}"
            for n in range(3):
                put(f"{h}.layer.{n}.convolution", f"{t}.conv{n + 1}"); put(f"{h}.layer.{n}.normalization", f"{t}.bn{n + 1}")
            put(f"{h}.shortcut.convolution", f"{t}.downsample.0"); put(f"{h}.shortcut.normalization", f"{t}.downsample.1")
    assert used == set(hsd.keys()), sorted(set(hsd.keys()) - used)[:8]
    hf.load_state_dict(hsd)
    x = torch.randn(2, 3, 64, 64)
    with torch.no_grad():
        out = hf(pixel_values=x)
        feats = ref.forward_features(x)
    assert ((feats - out.last_hidden_state).norm() / out.last_hidden_state.norm()).item() < 1e-5
