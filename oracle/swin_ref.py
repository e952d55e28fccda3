"""TEST INFRASTRUCTURE: plain-torch fp32 restatement of timm's Swin Transformer (v1), the default backbone of both shipped configs of the reference
(`timm-swin_base_patch4_window7_224`: /root/reference configs/classification/pet.yaml:25, configs/faceX/cbir.yaml:26; built by
`timm.create_model(...)` in models/classifier/classify_model.py:49-54 and models/faceX/backbone/timm_wrapper.py:16-21).

timm==0.9.16 (requirements.txt:10) is an un-vendored dependency, absent from /root/reference and not installable here: its published architecture is
restated below with timm's state_dict names, and tests/test_oracle_swin.py pins the restatement against the independent `transformers.SwinModel`
through a weight map.  PARITY PINNING: the reference has no tests or golden vectors for this path.

timm 0.9.16 SwinTransformer (swin_*_patch4_window7_224) as restated:
  * patch_embed: Conv2d(3, C, 4, stride 4) -> NHWC -> LayerNorm(C)                                   (patch_norm=True, eps 1e-5 everywhere)
  * 4 stages (`layers.i`): stage i > 0 STARTS with PatchMerging (`layers.i.downsample`): [B, H, W, C] -> [B, H/2, W/2, 4C] with the channel blocks ordered
    (w parity, h parity) = x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]; LayerNorm(4C); Linear(4C, 2C, bias=False)
  * SwinTransformerBlock j of a stage at resolution R: window 7 (the whole map when R <= 7), shift 3 for odd j (0 when R <= 7)
        x = x + attn(norm1(x))   -- cyclic shift by -shift, 7x7 window partition, WindowAttention, reverse, shift back
        x = x + mlp(norm2(x))    -- Linear(C, 4C), exact-erf GELU, Linear(4C, C)
  * WindowAttention: qkv = Linear(C, 3C) -> [B_, N, 3, heads, hd]; attn = (q * hd^-0.5) k^T + relative_position_bias_table[relative_position_index]
        (+ mask of the shifted windows: 0 where two tokens come from the same region, -100 otherwise); softmax; attn @ v; proj = Linear(C, C)
  * norm = LayerNorm(C_last) on the last map; head: global average pool -> Linear (`head.fc`); num_classes = 0 / global_pool '' -> the NHWC map
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def window_partition(x: torch.Tensor, ws: int) -> torch.Tensor:
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)


def window_reverse(w: torch.Tensor, ws: int, H: int, W: int) -> torch.Tensor:
    C = w.shape[-1]
    x = w.view(-1, H // ws, W // ws, ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, H, W, C)


def relative_position_index(ws: int) -> torch.Tensor:
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)     # [2, N]
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)                                                                                       # [N, N]


def shifted_window_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """[nW, N, N]: 0 where the two tokens of a shifted window come from the same image region, -100 otherwise"""
    img = torch.zeros(1, H, W, 1)
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    m = mw[:, None, :] - mw[:, :, None]
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


class WindowAttention(nn.Module):
    def __init__(self, dim: int, heads: int, ws: int):
        super().__init__()
        self.heads, self.scale, self.ws = heads, (dim // heads) ** -0.5, ws
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        self.register_buffer("relative_position_index", relative_position_index(ws), persistent=False)
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    def bias(self) -> torch.Tensor:
        N = self.ws * self.ws
        return self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1).permute(2, 0, 1)   # [heads, N, N]

    def forward(self, x, mask=None):
        B_, N, C = x.shape
        qkv = self.qkv(x).reshape(B_, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = (q * self.scale) @ k.transpose(-2, -1) + self.bias().unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = (attn.view(-1, nW, self.heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, self.heads, N, N)
        attn = attn.softmax(-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B_, N, C))


class Mlp(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class SwinBlock(nn.Module):
    def __init__(self, dim: int, res: int, heads: int, ws: int, shift: int):
        super().__init__()
        if res <= ws:
            ws, shift = res, 0
        self.res, self.ws, self.shift = res, ws, shift
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, heads, ws)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim)
        self.register_buffer("attn_mask", shifted_window_mask(res, res, ws, shift) if shift > 0 else None, persistent=False)
        # timm DropPath (stochastic depth) on both branches: per-sample factors mask / keep_prob, handed in by the test so that oracle and engine drop the same samples
        # (timm: `x = x + self.drop_path1(attn branch); x = x + self.drop_path2(mlp branch)`; drop_path(x) = x * bernoulli(keep_prob) / keep_prob per sample, training only)
        self.drop = None                       # None, or (f1 [B], f2 [B])

    def forward(self, x):                      # [B, H, W, C]
        B, H, W, C = x.shape
        h = self.norm1(x)
        if self.shift:
            h = torch.roll(h, (-self.shift, -self.shift), (1, 2))
        w = window_partition(h, self.ws).view(-1, self.ws * self.ws, C)
        w = self.attn(w, self.attn_mask)
        h = window_reverse(w.view(-1, self.ws, self.ws, C), self.ws, H, W)
        if self.shift:
            h = torch.roll(h, (self.shift, self.shift), (1, 2))
        if self.drop is None:
            x = x + h
            return x + self.mlp(self.norm2(x))
        f1, f2 = self.drop
        x = x + h * f1.view(B, 1, 1, 1)
        return x + self.mlp(self.norm2(x)) * f2.view(B, 1, 1, 1)


class PatchMerging(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.norm = nn.LayerNorm(4 * dim)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)

    def forward(self, x):
        B, H, W, C = x.shape
        x = x.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 4, 2, 5).flatten(3)
        return self.reduction(self.norm(x))


class SwinStage(nn.Module):
    def __init__(self, dim_in: int, dim: int, res: int, depth: int, heads: int, ws: int, downsample: bool):
        super().__init__()
        self.downsample = PatchMerging(dim_in) if downsample else nn.Identity()
        self.blocks = nn.Sequential(*[SwinBlock(dim, res, heads, ws, 0 if j % 2 == 0 else ws // 2) for j in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class _PatchEmbed(nn.Module):
    def __init__(self, in_chans: int, dim: int):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, 4, 4)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return self.norm(self.proj(x).permute(0, 2, 3, 1))


class _Head(nn.Module):
    def __init__(self, dim: int, num_classes: int):
        super().__init__()
        self.fc = nn.Linear(dim, num_classes) if num_classes > 0 else nn.Identity()

    def forward(self, x):                      # [B, H, W, C] -> global average pool -> fc
        return self.fc(x.mean((1, 2)))


class SwinTransformerRef(nn.Module):
    """swin_base_patch4_window7_224: embed_dim 128, depths (2, 2, 18, 2), heads (4, 8, 16, 32)"""

    def __init__(self, img_size=224, in_chans=3, num_classes=1000, embed_dim=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32), window=7):
        super().__init__()
        self.num_classes = num_classes
        self.patch_embed = _PatchEmbed(in_chans, embed_dim)
        res = img_size // 4
        layers, dim_in = [], embed_dim
        for i, (d, h) in enumerate(zip(depths, heads)):
            dim = embed_dim * 2 ** i
            if i > 0:
                res //= 2
            layers.append(SwinStage(dim_in, dim, res, d, h, window, downsample=i > 0))
            dim_in = dim
        self.layers = nn.Sequential(*layers)
        self.norm = nn.LayerNorm(dim_in)
        self.head = _Head(dim_in, num_classes)
        self.num_features = dim_in

    def set_drop_path(self, factors):
        """factors [2 * blocks, B] (row 2 k: block k's attention branch, 2 k + 1: its MLP branch; entries 0 or 1 / keep_prob), or None to switch stochastic depth off"""
        k = 0
        for stage in self.layers:
            for blk in stage.blocks:
                blk.drop = None if factors is None else (factors[2 * k], factors[2 * k + 1])
                k += 1

    def forward_features(self, x):
        return self.norm(self.layers(self.patch_embed(x)))        # [B, H/32, W/32, C]

    def forward(self, x):
        x = self.forward_features(x)
        return self.head(x) if self.num_classes > 0 else x
