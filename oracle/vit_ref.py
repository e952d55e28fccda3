"""TEST INFRASTRUCTURE: plain-torch fp32 restatement of the model the reference's classifier path runs.

The reference builds its classifier with `timm.create_model('vit_base_patch16_224', pretrained, num_classes=C)`
(/root/reference models/classifier/classify_model.py:23,49-54) and then re-initialises it
(`reset_parameters`, :70-81).  timm==0.9.16 (requirements.txt:10) is an un-vendored dependency that is absent
from /root/reference and not installable here, so its published architecture is restated below; every choice
that matters numerically is listed, and tests/test_oracle_vit.py pins the restatement against the independent
`transformers.ViTModel` implementation through a weight map.  PARITY PINNING: the
reference itself has no tests or golden vectors for this path.

timm 0.9.16 VisionTransformer (vit_base_patch16_224 defaults) as restated:
  * patch_embed.proj = Conv2d(in_chans, D, kernel=patch, stride=patch, bias=True); flatten(2).transpose(1,2)
  * x = cat([cls_token.expand(B,-1,-1), x], dim=1); x = x + pos_embed      (cls token FIRST, then pos add)
  * depth x Block:  x = x + attn(norm1(x));  x = x + mlp(norm2(x))          (pre-norm, no layer-scale, no drop)
      - norm1/norm2 = LayerNorm(D, eps=1e-6)
      - attn: qkv = Linear(D, 3D, bias=True) -> reshape(B,N,3,H,hd).permute(2,0,3,1,4); q,k,v = unbind(0)
              attn = softmax((q * hd**-0.5) @ k^T, dim=-1); x = (attn @ v).transpose(1,2).reshape(B,N,D); proj = Linear(D,D)
      - mlp: fc1 = Linear(D, 4D) -> nn.GELU() (exact erf) -> fc2 = Linear(4D, D)
  * norm = LayerNorm(D, eps=1e-6) on all tokens; global_pool='token' -> x[:, 0]; fc_norm = Identity; head = Linear(D, C)
  * pre_norm=True (the CLIP ViTs, `vit_*_clip_*`): patch_embed.proj has NO bias (timm: `bias=not pre_norm`), `norm_pre = LayerNorm(D, eps)` is applied to the
    embedded tokens (after the class token and pos_embed) in front of the blocks, norm eps 1e-5; pinned against transformers.CLIPVisionModel (tests/test_oracle_vit.py)
state_dict key names equal timm's, so reference checkpoints (`ckpt['model']`, vision_engine.py:387-403) load.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import bf16ops as ops   # fp32 mode = plain torch (bit-identical to F.linear / softmax / F.gelu); bf16_operands mode: see oracle/bf16ops.py


class PatchEmbed(nn.Module):
    def __init__(self, patch_size: int, in_chans: int, embed_dim: int, bias: bool = True):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)

    def forward(self, x):
        if ops.mode() == "fp32":
            return self.proj(x).flatten(2).transpose(1, 2)
        # the same stride == kernel convolution as a GEMM over (c, ky, kx) patches, bf16 operands; the bias gradient is the fp32 column sum
        # (the engine derives it from the fp32 pos_embed gradient rows, csrc/vit_engine.hip)
        p = self.proj.kernel_size[0]
        B, Cc, H, W = x.shape
        pt = x.reshape(B, Cc, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // p) * (W // p), Cc * p * p)
        return ops.linear(pt, self.proj.weight.reshape(self.proj.out_channels, -1), self.proj.bias, bias_grad_unrounded=self.proj.bias is not None)


class Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = ops.q(ops.linear(x, self.qkv.weight, self.qkv.bias)).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        x = ops.q(ops.attention(q, k, v, self.scale)).transpose(1, 2).reshape(B, N, C)
        return ops.linear(x, self.proj.weight, self.proj.bias)


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return ops.linear(ops.gelu(ops.linear(x, self.fc1.weight, self.fc1.bias)), self.fc2.weight, self.fc2.bias)


class Block(nn.Module):
    def __init__(self, dim: int, num_heads: int, mlp_dim: int, eps: float):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, mlp_dim)

    def forward(self, x):
        x = x + self.attn(ops.q(self.norm1(x)))
        return x + self.mlp(ops.q(self.norm2(x)))


class VisionTransformerRef(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_dim=None, eps=1e-6, pre_norm=False):
        super().__init__()
        mlp_dim = mlp_dim or 4 * embed_dim
        n = (img_size // patch_size) ** 2 + 1
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, bias=not pre_norm)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n, embed_dim) * .02)
        self.norm_pre = nn.LayerNorm(embed_dim, eps=eps) if pre_norm else nn.Identity()
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_dim, eps) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=eps)
        self.head = nn.Linear(embed_dim, num_classes)
        self.reset_parameters()

    def reset_parameters(self):
        """timm init (trunc_normal .02 pos_embed, normal 1e-6 cls_token) followed by the reference's override
        classify_model.py:70-81: N(0, 0.02) for every Conv2d/Linear weight, zeros for Linear bias."""
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, mean=0, std=0.02)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward_features(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1)
        x = self.norm_pre(x + self.pos_embed)
        x = self.blocks(x)
        return self.norm(x)

    def forward(self, x):
        if ops.mode() == "fp32":
            return self.head(self.forward_features(x)[:, 0])
        # LayerNorm is per token: norm(x)[:, 0] == norm(x[:, 0]); the engine normalises the class-token rows only
        x = self.patch_embed(x)
        x = self.norm_pre(torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed)
        x = self.blocks(x)
        return ops.linear(ops.q(self.norm(x[:, 0])), self.head.weight, self.head.bias)


def train_step_reference(model: nn.Module, x, y, *, lr, momentum, weight_decay, label_smoothing, max_norm=10.0,
                         momentum_bufs=None, ema=None, updates=0):
    """One step of Trainer.compute_loss + Trainer.update on CPU (engine/procedure/train.py:196,203-215): CE(label
    smoothing) -> backward -> clip_grad_norm_(10) -> SGD(momentum, wd) -> EMA (models/ema.py:28-37).
    On CPU the reference disables autocast and the GradScaler (train.py:118, vision_engine.py:232): plain fp32."""
    params = [p for p in model.parameters()]
    for p in params:
        p.grad = None
    logits = model(x)
    loss = F.cross_entropy(logits, y, label_smoothing=label_smoothing)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    total_norm = torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm)
    if momentum_bufs is None:
        momentum_bufs = {}
    with torch.no_grad():
        for n, p in model.named_parameters():
            g = p.grad + weight_decay * p
            if n not in momentum_bufs:
                momentum_bufs[n] = g.clone()
            else:
                momentum_bufs[n].mul_(momentum).add_(g)
            p.add_(momentum_bufs[n], alpha=-lr)
        if ema is not None:
            d = 0.9999 * (1 - math.exp(-(updates + 1) / 2000))
            for n, p in model.named_parameters():
                ema[n].mul_(d).add_(p.detach(), alpha=1 - d)
    return logits.detach(), loss.detach(), grads, total_norm, momentum_bufs


def train_step_reference_sam(model: nn.Module, x, y, *, lr, momentum, weight_decay, label_smoothing, rho=0.05, momentum_bufs=None):
    """Trainer.update_sam (engine/procedure/train.py:150-175) with engine/optimizer.py's SAM(adaptive=True) over SGD: first
    fwd/bwd -> e(w) = w^2 * g * rho / (|| |w| * g || + 1e-12) -> w += e(w) -> second fwd/bwd -> w restored -> SGD step (no clip)."""
    params = list(model.parameters())
    for p in params:
        p.grad = None
    loss = F.cross_entropy(model(x), y, label_smoothing=label_smoothing)
    loss.backward()
    with torch.no_grad():
        gnorm = torch.norm(torch.stack([(p.abs() * p.grad).norm(p=2) for p in params]), p=2)
        scale = rho / (gnorm + 1e-12)
        old = [p.detach().clone() for p in params]
        for p in params:
            p.add_(p.pow(2) * p.grad * scale)
    for p in params:
        p.grad = None
    F.cross_entropy(model(x), y, label_smoothing=label_smoothing).backward()
    if momentum_bufs is None:
        momentum_bufs = {}
    with torch.no_grad():
        for (n, p), o in zip(model.named_parameters(), old):
            p.copy_(o)
            g = p.grad + weight_decay * p
            if n not in momentum_bufs:
                momentum_bufs[n] = g.clone()
            else:
                momentum_bufs[n].mul_(momentum).add_(g)
            p.add_(momentum_bufs[n], alpha=-lr)
    return loss.detach(), momentum_bufs


class AttentionPoolLatentRef(nn.Module):
    """timm layers/attention_pool.py AttentionPoolLatent as the SigLIP ViTs build it (global_pool='map': latent_len 1, qkv_bias, no qk_norm, pool_type 'token',
    norm_layer = LayerNorm(eps 1e-6), mlp_ratio 4, exact-erf GELU):
        q = q(latent) -> [B, H, 1, hd];  k, v = kv(x) -> [B, H, N, hd];  x = softmax(q k^T / sqrt(hd)) v -> [B, 1, D];  x = proj(x);  x = x + mlp(norm(x));  x[:, 0]
    tests/test_oracle_vit.py pins it against transformers' SiglipMultiheadAttentionPoolingHead (same arithmetic through nn.MultiheadAttention) via a weight map."""

    def __init__(self, dim: int, num_heads: int, mlp_dim: int = None, eps: float = 1e-6):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.latent = nn.Parameter(torch.zeros(1, 1, dim))
        self.q = nn.Linear(dim, dim)
        self.kv = nn.Linear(dim, 2 * dim)
        self.proj = nn.Linear(dim, dim)
        self.norm = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, mlp_dim or 4 * dim)
        nn.init.trunc_normal_(self.latent, std=dim ** -0.5)

    def forward(self, x):
        B, N, C = x.shape
        q = self.q(self.latent.expand(B, -1, -1)).reshape(B, 1, self.num_heads, self.head_dim).transpose(1, 2)
        kv = self.kv(x).reshape(B, N, 2, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        k, v = kv.unbind(0)
        a = torch.softmax((q * self.scale) @ k.transpose(-2, -1), dim=-1)
        x = (a @ v).transpose(1, 2).reshape(B, 1, C)
        x = self.proj(x)
        x = x + self.mlp(self.norm(x))
        return x[:, 0]


class SiglipVisionTransformerRef(nn.Module):
    """timm VisionTransformer(class_token=False, global_pool='map') -- the vit_*_siglip_* family (BASELINE.json configs[4]): no cls token, pos_embed over the
    patches only, the usual pre-norm blocks, norm, AttentionPoolLatent, head.  state_dict names equal timm's (pos_embed, patch_embed.*, blocks.*, norm.*, attn_pool.*, head.*)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12, mlp_dim=None, eps=1e-6):
        super().__init__()
        mlp_dim = mlp_dim or 4 * embed_dim
        n = (img_size // patch_size) ** 2
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim)
        self.pos_embed = nn.Parameter(torch.randn(1, n, embed_dim) * .02)
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_dim, eps) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=eps)
        self.attn_pool = AttentionPoolLatentRef(embed_dim, num_heads, mlp_dim, eps)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.normal_(m.weight, mean=0, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward_features(self, x):
        return self.norm(self.blocks(self.patch_embed(x) + self.pos_embed))

    def forward(self, x):
        return self.head(self.attn_pool(self.forward_features(x)))


class TimmWrapperRef(nn.Module):
    """Restatement of the reference's TimmWrapper for a transformer backbone (models/faceX/backbone/timm_wrapper.py:16-21,39-54):
    `timm.create_model(name, num_classes=0, global_pool='')` -> forward = final-normed tokens [B, N, C]; neck =
    LayerNorm(C) -> Flatten -> Linear(N*C, feat_dim) -> BatchNorm1d(feat_dim)."""

    def __init__(self, feat_dim, img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_dim=None):
        super().__init__()
        self.model = VisionTransformerRef(img_size, patch_size, 3, 1, embed_dim, depth, num_heads, mlp_dim)
        del self.model.head
        n = (img_size // patch_size) ** 2 + 1
        self.output_layer = nn.Sequential(nn.LayerNorm(embed_dim), nn.Flatten(1), nn.Linear(n * embed_dim, feat_dim), nn.BatchNorm1d(feat_dim))

    def forward(self, x):
        return self.output_layer(self.model.forward_features(x))
