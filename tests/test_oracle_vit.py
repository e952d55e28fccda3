"""Pin the timm-ViT restatement (oracle/vit_ref.py) against the independent `transformers.ViTModel` through the
weight map of SURVEY.md §10 (timm itself is not installable here)."""
import torch

from oracle.vit_ref import VisionTransformerRef


def test_vit_ref_matches_transformers():
    from transformers import ViTConfig, ViTModel
    torch.manual_seed(0)
    D, depth, heads, img, ps = 64, 2, 2, 32, 8
    ref = VisionTransformerRef(img, ps, 3, 10, D, depth, heads, mlp_dim=4 * D).eval()
    with torch.no_grad():  # make biases / norms non-trivial
        for p in ref.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
        ref.cls_token.add_(torch.randn_like(ref.cls_token) * 0.1)
    cfg = ViTConfig(hidden_size=D, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=4 * D, image_size=img,
                    patch_size=ps, num_channels=3, layer_norm_eps=1e-6, hidden_act="gelu", hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0, qkv_bias=True)
    hf = ViTModel(cfg, add_pooling_layer=False).eval()
    sd = ref.state_dict()
    hsd = hf.state_dict()
    keys = list(hsd.keys())

    def put(name_options, value):
        for n in name_options:
            if n in hsd:
                assert hsd[n].shape == value.shape, (n, hsd[n].shape, value.shape)
                hsd[n] = value.clone()
                return
        raise KeyError(f"none of {name_options} in HF state_dict; keys sample: {keys[:12]}")

    put(["embeddings.cls_token"], sd["cls_token"])
    put(["embeddings.position_embeddings"], sd["pos_embed"])
    put(["embeddings.patch_embeddings.projection.weight"], sd["patch_embed.proj.weight"])
    put(["embeddings.patch_embeddings.projection.bias"], sd["patch_embed.proj.bias"])
    for i in range(depth):
        w, b = sd[f"blocks.{i}.attn.qkv.weight"], sd[f"blocks.{i}.attn.qkv.bias"]
        for j, nm in enumerate(["query", "key", "value"]):  # timm fused rows: [0:D]=q, [D:2D]=k, [2D:3D]=v
            short = {"query": "q_proj", "key": "k_proj", "value": "v_proj"}[nm]
            put([f"encoder.layer.{i}.attention.attention.{nm}.weight", f"layers.{i}.attention.{short}.weight"], w[j * D:(j + 1) * D])
            put([f"encoder.layer.{i}.attention.attention.{nm}.bias", f"layers.{i}.attention.{short}.bias"], b[j * D:(j + 1) * D])
        for kind in ("weight", "bias"):
            put([f"encoder.layer.{i}.attention.output.dense.{kind}", f"layers.{i}.attention.o_proj.{kind}"], sd[f"blocks.{i}.attn.proj.{kind}"])
            put([f"encoder.layer.{i}.layernorm_before.{kind}", f"layers.{i}.layernorm_before.{kind}"], sd[f"blocks.{i}.norm1.{kind}"])
            put([f"encoder.layer.{i}.layernorm_after.{kind}", f"layers.{i}.layernorm_after.{kind}"], sd[f"blocks.{i}.norm2.{kind}"])
            put([f"encoder.layer.{i}.intermediate.dense.{kind}", f"layers.{i}.mlp.fc1.{kind}"], sd[f"blocks.{i}.mlp.fc1.{kind}"])
            put([f"encoder.layer.{i}.output.dense.{kind}", f"layers.{i}.mlp.fc2.{kind}"], sd[f"blocks.{i}.mlp.fc2.{kind}"])
    for kind in ("weight", "bias"):
        put([f"layernorm.{kind}"], sd[f"norm.{kind}"])
    hf.load_state_dict(hsd)
    x = torch.randn(3, 3, img, img)
    with torch.no_grad():
        feats = ref.forward_features(x)
        hfeats = hf(pixel_values=x).last_hidden_state
    rel = ((feats - hfeats).norm() / hfeats.norm()).item()
    assert rel < 1e-5, rel


def test_state_dict_keys_are_timm_names():
    ref = VisionTransformerRef(32, 8, 3, 10, 64, 2, 1)
    keys = list(ref.state_dict().keys())
    for k in ["cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias", "blocks.0.norm1.weight",
              "blocks.0.attn.qkv.weight", "blocks.0.attn.proj.bias", "blocks.1.mlp.fc1.weight", "blocks.1.mlp.fc2.bias",
              "norm.weight", "head.weight", "head.bias"]:
        assert k in keys
    assert len(keys) == 4 + 12 * 2 + 4
    assert ref.state_dict()["blocks.0.attn.qkv.weight"].shape == (192, 64)


def test_siglip_map_pool_ref_matches_transformers():
    """oracle/vit_ref.SiglipVisionTransformerRef (timm class_token=False + global_pool='map' AttentionPoolLatent) against transformers.SiglipVisionModel:
    last_hidden_state and pooler_output (nn.MultiheadAttention with the probe as the only query) through the timm <- HF weight map of timm's own converter
    (in_proj rows [0:D] -> attn_pool.q, [D:3D] -> attn_pool.kv, probe -> latent)."""
    from transformers import SiglipVisionConfig, SiglipVisionModel
    from oracle.vit_ref import SiglipVisionTransformerRef
    torch.manual_seed(0)
    D, depth, heads, img, ps = 128, 2, 2, 32, 8
    ref = SiglipVisionTransformerRef(img, ps, 3, 0, D, depth, heads, mlp_dim=2 * D).eval()
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    cfg = SiglipVisionConfig(hidden_size=D, intermediate_size=2 * D, num_hidden_layers=depth, num_attention_heads=heads, image_size=img, patch_size=ps,
                             hidden_act="gelu", layer_norm_eps=1e-6, attention_dropout=0.0)
    hf = SiglipVisionModel(cfg).eval()
    sd, hsd = ref.state_dict(), hf.state_dict()
    pre = "vision_model." if any(k.startswith("vision_model.") for k in hsd) else ""

    def put(name, value):
        assert hsd[pre + name].shape == value.shape, (name, hsd[pre + name].shape, value.shape)
        hsd[pre + name] = value.clone()

    put("embeddings.patch_embedding.weight", sd["patch_embed.proj.weight"]); put("embeddings.patch_embedding.bias", sd["patch_embed.proj.bias"])
    put("embeddings.position_embedding.weight", sd["pos_embed"][0])
    for i in range(depth):
        w, b = sd[f"blocks.{i}.attn.qkv.weight"], sd[f"blocks.{i}.attn.qkv.bias"]
        for j, nm in enumerate(["q_proj", "k_proj", "v_proj"]):
            put(f"encoder.layers.{i}.self_attn.{nm}.weight", w[j * D:(j + 1) * D]); put(f"encoder.layers.{i}.self_attn.{nm}.bias", b[j * D:(j + 1) * D])
        for kind in ("weight", "bias"):
            put(f"encoder.layers.{i}.self_attn.out_proj.{kind}", sd[f"blocks.{i}.attn.proj.{kind}"])
            put(f"encoder.layers.{i}.layer_norm1.{kind}", sd[f"blocks.{i}.norm1.{kind}"]); put(f"encoder.layers.{i}.layer_norm2.{kind}", sd[f"blocks.{i}.norm2.{kind}"])
            put(f"encoder.layers.{i}.mlp.fc1.{kind}", sd[f"blocks.{i}.mlp.fc1.{kind}"]); put(f"encoder.layers.{i}.mlp.fc2.{kind}", sd[f"blocks.{i}.mlp.fc2.{kind}"])
    for kind in ("weight", "bias"):
        put(f"post_layernorm.{kind}", sd[f"norm.{kind}"])
        put(f"head.layernorm.{kind}", sd[f"attn_pool.norm.{kind}"])
        put(f"head.attention.out_proj.{kind}", sd[f"attn_pool.proj.{kind}"])
        put(f"head.mlp.fc1.{kind}", sd[f"attn_pool.mlp.fc1.{kind}"]); put(f"head.mlp.fc2.{kind}", sd[f"attn_pool.mlp.fc2.{kind}"])
    put("head.probe", sd["attn_pool.latent"])
    put("head.attention.in_proj_weight", torch.cat([sd["attn_pool.q.weight"], sd["attn_pool.kv.weight"]], 0))
    put("head.attention.in_proj_bias", torch.cat([sd["attn_pool.q.bias"], sd["attn_pool.kv.bias"]], 0))
    hf.load_state_dict(hsd)
    x = torch.randn(3, 3, img, img)
    with torch.no_grad():
        feats = ref.forward_features(x)
        pooled = ref.attn_pool(feats)
        out = hf(pixel_values=x)
    assert ((feats - out.last_hidden_state).norm() / out.last_hidden_state.norm()).item() < 1e-5
    assert ((pooled - out.pooler_output).norm() / out.pooler_output.norm()).item() < 1e-5
    assert list(sd.keys())[:3] == ["pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias"] and "attn_pool.latent" in sd and "cls_token" not in sd


def test_clip_pre_norm_ref_matches_transformers():
    """timm's CLIP ViTs (`vit_*_clip_*`: pre_norm=True, patch embedding without a bias, LayerNorm eps 1e-5) as restated by VisionTransformerRef(pre_norm=True), pinned against
    the independent `transformers.CLIPVisionModel` (class embedding + position embedding -> pre_layrnorm -> pre-norm encoder layers; hidden_act gelu: the laion / datacomp
    weights timm's plain clip ids carry; the quick-GELU ones are separate timm ids).  last_hidden_state there is in front of post_layernorm: compared with the tokens before
    `norm`."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    torch.manual_seed(0)
    D, depth, heads, img, ps = 64, 2, 2, 32, 8
    ref = VisionTransformerRef(img, ps, 3, 10, D, depth, heads, mlp_dim=4 * D, eps=1e-5, pre_norm=True).eval()
    assert ref.patch_embed.proj.bias is None and "norm_pre.weight" in ref.state_dict() and "patch_embed.proj.bias" not in ref.state_dict()
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
        ref.cls_token.add_(torch.randn_like(ref.cls_token) * 0.1)
    cfg = CLIPVisionConfig(hidden_size=D, intermediate_size=4 * D, num_hidden_layers=depth, num_attention_heads=heads, num_channels=3, image_size=img, patch_size=ps,
                           hidden_act="gelu", layer_norm_eps=1e-5, attention_dropout=0.0)
    hf = CLIPVisionModel(cfg).eval()
    sd, hsd = ref.state_dict(), hf.state_dict()

    def put(name, value):
        assert hsd[name].shape == value.shape, (name, hsd[name].shape, value.shape)
        hsd[name] = value.clone()

    pre = "vision_model." if "vision_model.embeddings.class_embedding" in hsd else ""      # (the prefix depends on the transformers version)
    put(pre + "embeddings.class_embedding", sd["cls_token"].view(D))
    put(pre + "embeddings.position_embedding.weight", sd["pos_embed"][0])
    put(pre + "embeddings.patch_embedding.weight", sd["patch_embed.proj.weight"])
    ln_pre = "pre_layrnorm" if pre + "pre_layrnorm.weight" in hsd else "pre_layernorm"      # (the attribute's spelling in transformers)
    for kind in ("weight", "bias"):
        put(pre + f"{ln_pre}.{kind}", sd[f"norm_pre.{kind}"])
        put(pre + f"post_layernorm.{kind}", sd[f"norm.{kind}"])
    for i in range(depth):
        w, b = sd[f"blocks.{i}.attn.qkv.weight"], sd[f"blocks.{i}.attn.qkv.bias"]
        for j, nm in enumerate(["q_proj", "k_proj", "v_proj"]):
            put(pre + f"encoder.layers.{i}.self_attn.{nm}.weight", w[j * D:(j + 1) * D])
            put(pre + f"encoder.layers.{i}.self_attn.{nm}.bias", b[j * D:(j + 1) * D])
        for kind in ("weight", "bias"):
            put(pre + f"encoder.layers.{i}.self_attn.out_proj.{kind}", sd[f"blocks.{i}.attn.proj.{kind}"])
            put(pre + f"encoder.layers.{i}.layer_norm1.{kind}", sd[f"blocks.{i}.norm1.{kind}"])
            put(pre + f"encoder.layers.{i}.layer_norm2.{kind}", sd[f"blocks.{i}.norm2.{kind}"])
            put(pre + f"encoder.layers.{i}.mlp.fc1.{kind}", sd[f"blocks.{i}.mlp.fc1.{kind}"])
            put(pre + f"encoder.layers.{i}.mlp.fc2.{kind}", sd[f"blocks.{i}.mlp.fc2.{kind}"])
    hf.load_state_dict(hsd)
    x = torch.randn(3, 3, img, img)
    with torch.no_grad():
        tokens = ref.blocks(ref.norm_pre(torch.cat([ref.cls_token.expand(3, -1, -1), ref.patch_embed(x)], 1) + ref.pos_embed))      # in front of `norm`
        out = hf(pixel_values=x)
        rel = ((tokens - out.last_hidden_state).norm() / out.last_hidden_state.norm()).item()
        assert rel < 1e-5, rel
        pooled = ref.forward_features(x)[:, 0]                                               # norm(tokens)[:, 0] = transformers' pooler_output (post_layernorm of the class token)
        relp = ((pooled - out.pooler_output).norm() / out.pooler_output.norm()).item()
        assert relp < 1e-5, relp
