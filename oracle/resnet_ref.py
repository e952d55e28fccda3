"""TEST INFRASTRUCTURE: plain-torch fp32 restatement of timm's ResNet (BasicBlock family) — BASELINE.json configs[0] is "ResNet-18 ICT on
toy-multi-cls.csv" (`timm-resnet18` through models/classifier/classify_model.py:49-54).  timm==0.9.16 is un-vendored and not installable here; the
architecture below is the published one (it equals torchvision's) and tests/test_oracle_resnet.py pins it against the independent
`transformers.ResNetModel` through a weight map.  PARITY PINNING: the reference has no tests or golden vectors for this path.

timm 0.9.16 `resnet18` as restated (state_dict keys equal timm's):
  * conv1 = Conv2d(in_chans, 64, 7, stride 2, padding 3, bias=False); bn1 = BatchNorm2d(64) (eps 1e-5, momentum 0.1); act1 = ReLU
  * maxpool = MaxPool2d(3, stride 2, padding 1)
  * layer1..4, 2 BasicBlocks each, widths (64, 128, 256, 512), first block of layers 2-4 has stride 2 and
    downsample = Sequential(Conv2d(cin, cout, 1, stride 2, bias=False), BatchNorm2d(cout))
  * BasicBlock(x): shortcut = x (or downsample(x)); x = act1(bn1(conv1(x))) [3x3, stride, padding 1, no bias]; x = bn2(conv2(x)) [3x3, stride 1];
    x = act2(x + shortcut)
  * global_pool = adaptive average pool -> flatten; fc = Linear(512, num_classes)
The reference's re-init (classify_model.py:70-81) then overwrites Conv2d/Linear weights with N(0, 0.02), Linear bias 0, BatchNorm (1, 0).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        x = F.relu(self.bn1(self.conv1(x)))
        x = self.bn2(self.conv2(x))
        return F.relu(x + shortcut)


class Bottleneck(nn.Module):
    """timm / torchvision v1.5 Bottleneck (resnet50 / 101 / 152, wide_resnet*_2: `timm-wide_resnet101_2` is one of pet.yaml's listed models): 1x1 (cin -> mid),
    3x3 with the block's stride (mid -> mid), 1x1 (mid -> cout = 4 * planes); mid = planes * base_width / 64; downsample as in BasicBlock."""

    def __init__(self, cin, mid, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, mid, 1, bias=False); self.bn1 = nn.BatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride=stride, padding=1, bias=False); self.bn2 = nn.BatchNorm2d(mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=False); self.bn3 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        return F.relu(x + shortcut)


class ResNetRef(nn.Module):
    def __init__(self, num_classes=1000, in_chans=3, widths=(64, 128, 256, 512), depths=(2, 2, 2, 2), mid=None, stem_width=None):
        """mid = None: BasicBlocks with `widths` channels.  mid = (m1..m4): Bottlenecks, `widths` are the block OUTPUT channels (4 * planes)."""
        super().__init__()
        stem_width = stem_width or (widths[0] if mid is None else 64)
        self.conv1 = nn.Conv2d(in_chans, stem_width, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(stem_width)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = stem_width
        for i, (w, d) in enumerate(zip(widths, depths)):
            blocks = []
            for j in range(d):
                st = 2 if (j == 0 and i > 0) else 1
                blocks.append(BasicBlock(cin, w, st) if mid is None else Bottleneck(cin, mid[i], w, st))
                cin = w
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.fc = nn.Linear(widths[-1], num_classes)
        self.reset_parameters()

    def reset_parameters(self):
        """classify_model.py:70-81"""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.normal_(m.weight, mean=0, std=0.02)
                if getattr(m, "bias", None) is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight); nn.init.zeros_(m.bias)

    def forward_features(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
        return x

    def forward(self, x):
        return self.fc(self.forward_features(x).mean((-2, -1)))


def _q(t, dtype=torch.bfloat16):
    """16-bit round trip (bf16, or fp16 for the engine's operand="fp16" mode) with a straight-through gradient: where the engine stores an activation in 16 bits"""
    return t + (t.to(dtype).to(t.dtype) - t).detach()


def forward_16bit_storage(ref, x, dtype=torch.bfloat16):
    """The oracle's forward with the engine's storage precision made explicit: conv operands (image, activations) are bf16, conv outputs / BatchNorm /
    shortcut sums are fp32, the BatchNorm'd shortcut stays fp32.  With ReLU + small-batch BatchNorm a plain fp32 run differs from ANY bf16 run by tens of
    percent in the gradients (a pre-activation that rounds across zero flips its mask; torch's own CPU autocast shows 25-40 % here), so the comparison
    has to put the rounding points in the same places."""
    F = torch.nn.functional
    q = lambda t: _q(t, dtype)
    a = ref.maxpool(q(F.relu(ref.bn1(ref.conv1(q(x))))))
    for i in range(1, 5):
        for blk in getattr(ref, f"layer{i}"):
            idn = a if blk.downsample is None else blk.downsample(a)
            a1 = q(F.relu(blk.bn1(blk.conv1(a))))
            if hasattr(blk, "conv3"):                       # Bottleneck
                a2 = q(F.relu(blk.bn2(blk.conv2(a1))))
                a = q(F.relu(blk.bn3(blk.conv3(a2)) + idn))
            else:
                a = q(F.relu(blk.bn2(blk.conv2(a1)) + idn))
    return ref.fc(q(a.mean((-2, -1))))


def forward_bf16_storage(ref, x):
    return forward_16bit_storage(ref, x, torch.bfloat16)
