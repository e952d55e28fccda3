"""TEST INFRASTRUCTURE: plain-torch fp32 restatement of the CNN backbone the reference's face / CBIR path runs.

The reference builds it with `timm.create_model('convnext_base', pretrained, num_classes=0, global_pool='')`
(/root/reference models/faceX/backbone/timm_wrapper.py:16-21; configs/faceX/cbir.yaml:4-8) and wraps it with the CNN neck
`BatchNorm2d(C) -> Flatten -> Linear(C*H*W, feat_dim) -> BatchNorm1d(feat_dim)` (timm_wrapper.py:23-37).  timm==0.9.16
(requirements.txt:10) is an un-vendored dependency, absent from /root/reference and not installable here, so its published
architecture is restated below; tests/test_oracle_convnext.py pins the restatement against the independent
`transformers.ConvNextModel` through a weight map.  PARITY PINNING: the reference itself has no tests
or golden vectors for this path.

timm 0.9.16 ConvNeXt (`convnext_base`: depths (3,3,27,3), dims (128,256,512,1024)) as restated:
  * stem = Sequential(Conv2d(in_chans, dims[0], 4, stride 4, bias), LayerNorm2d(dims[0], eps 1e-6))
  * stage i: downsample = Identity (i == 0) or Sequential(LayerNorm2d(dims[i-1], eps 1e-6), Conv2d(dims[i-1], dims[i], 2, stride 2, bias));
    blocks = depths[i] x ConvNeXtBlock
  * ConvNeXtBlock(x): shortcut = x; x = conv_dw(x) [7x7, padding 3, groups=C, bias]; NCHW->NHWC; x = norm(x) [LayerNorm(C, eps 1e-6)];
    x = mlp.fc2(GELU_erf(mlp.fc1(x))) [Linear(C,4C), Linear(4C,C)]; NHWC->NCHW; x = x * gamma[C] (init 1e-6); x = x + shortcut
  * norm_pre = Identity; head = NormMlpClassifierHead: with num_classes=0 and global_pool='' the pool and fc are Identity but
    head.norm = LayerNorm2d(dims[-1], eps 1e-6) IS applied -> output [B, dims[-1], H/32, W/32]
LayerNorm2d = LayerNorm over the channel dim of an NCHW tensor.  state_dict key names equal timm's
(stem.0/1, stages.{i}.downsample.0/1, stages.{i}.blocks.{j}.{gamma,conv_dw,norm,mlp.fc1,mlp.fc2}, head.norm).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class LayerNorm2d(nn.LayerNorm):
    def forward(self, x):
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(4 * dim, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class ConvNeXtBlock(nn.Module):
    def __init__(self, dim, eps=1e-6, ls_init_value=1e-6):
        super().__init__()
        self.gamma = nn.Parameter(ls_init_value * torch.ones(dim))
        self.conv_dw = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim, bias=True)
        self.norm = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim)

    def forward(self, x):
        shortcut = x
        x = self.conv_dw(x).permute(0, 2, 3, 1)
        x = self.mlp(self.norm(x)).permute(0, 3, 1, 2)
        x = x * self.gamma.reshape(1, -1, 1, 1)
        return x + shortcut


class ConvNeXtStage(nn.Module):
    def __init__(self, in_chs, out_chs, depth, first, eps):
        super().__init__()
        self.downsample = nn.Identity() if first else nn.Sequential(LayerNorm2d(in_chs, eps=eps), nn.Conv2d(in_chs, out_chs, 2, stride=2, bias=True))
        self.blocks = nn.Sequential(*[ConvNeXtBlock(out_chs, eps) for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class _Head(nn.Module):
    """timm NormMlpClassifierHead (hidden_size=None): global_pool -> norm (LayerNorm2d) -> flatten -> fc.  num_classes=0, global_pool='' (TimmWrapper):
    pool, flatten and fc are Identity, the norm runs per pixel.  num_classes>0 (VisionWrapper, models/classifier/classify_model.py:49-54): global average
    pool to [B, C, 1, 1] FIRST, then the norm on the pooled vector, then Linear(C, num_classes)."""

    def __init__(self, dim, eps, num_classes=0):
        super().__init__()
        self.norm = LayerNorm2d(dim, eps=eps)
        self.num_classes = num_classes
        if num_classes > 0:
            self.fc = nn.Linear(dim, num_classes)

    def forward(self, x):
        if self.num_classes > 0:
            return self.fc(self.norm(x.mean((-2, -1), keepdim=True)).flatten(1))
        return self.norm(x)


class ConvNeXtRef(nn.Module):
    """timm ConvNeXt with num_classes=0, global_pool='' (what TimmWrapper builds): [B, Cin, H, W] -> [B, dims[-1], H/32, W/32]."""

    def __init__(self, in_chans=3, depths=(3, 3, 27, 3), dims=(128, 256, 512, 1024), eps=1e-6, num_classes=0):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(in_chans, dims[0], 4, stride=4, bias=True), LayerNorm2d(dims[0], eps=eps))
        self.stages = nn.Sequential(*[ConvNeXtStage(dims[max(i - 1, 0)], dims[i], depths[i], i == 0, eps) for i in range(4)])
        self.head = _Head(dims[-1], eps, num_classes)
        self.reset_parameters()

    def reset_parameters(self):
        """timm init (trunc_normal .02 on Conv/Linear weights, zero bias) — the reference's FaceTrainingWrapper.reset_parameters is
        defined but never called, so timm's own init is what training starts from."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)

    def forward_trunk(self, x):
        return self.stages(self.stem(x))          # before head.norm: what transformers.ConvNextModel calls last_hidden_state

    def forward(self, x):
        return self.head(self.forward_trunk(x))


class TimmWrapperCNNRef(nn.Module):
    """The reference's TimmWrapper for a CNN backbone (models/faceX/backbone/timm_wrapper.py:23-37,49-54): backbone map [B,C,H,W]
    -> BatchNorm2d(C) -> Flatten -> Linear(C*H*W, feat_dim) -> BatchNorm1d(feat_dim)."""

    def __init__(self, feat_dim, image_size=224, in_chans=3, depths=(3, 3, 27, 3), dims=(128, 256, 512, 1024)):
        super().__init__()
        self.model = ConvNeXtRef(in_chans, depths, dims)
        hw = image_size // 32
        self.output_layer = nn.Sequential(nn.BatchNorm2d(dims[-1]), nn.Flatten(1), nn.Linear(dims[-1] * hw * hw, feat_dim), nn.BatchNorm1d(feat_dim))

    def forward(self, x):
        return self.output_layer(self.model(x))
