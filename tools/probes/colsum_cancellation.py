"""Why the fc2 bias gradient of ConvNeXt's last blocks was the worst gradient of rounds 2-4 (4.2e-2 in bf16), on the CPU in a few seconds.
The neck is BatchNorm2d -> Flatten -> Linear -> BatchNorm1d behind head.norm (LayerNorm2d): the rows of a BatchNorm input gradient sum to zero per channel, LayerNorm2d only perturbs
that, so the column sums of the gradient entering the backbone's last block nearly cancel (|sum| ~ 0.03 x the column's l2 norm).  Summing a 16-bit COPY of that gradient amplifies
its rounding by ~30x: fp16 5.6e-3, bf16 4.1e-2 -- what tests/test_parity_fullsize_gpu.py measured on the MI355X (5.9e-3 / 4.2e-2) before the fp16 mode summed the fp32 stream.
    python tools/probes/colsum_cancellation.py"""
import torch

torch.manual_seed(0)
B, C, H, F = 8, 1024, 7, 512
x = torch.randn(B, C, H, H, requires_grad=True)          # the map in front of head.norm
ln_w = torch.ones(C) + 0.1 * torch.randn(C); ln_b = 0.1 * torch.randn(C)
bn2 = torch.nn.BatchNorm2d(C).train(); lin = torch.nn.Linear(C * H * H, F); bn1 = torch.nn.BatchNorm1d(F).train()
y = torch.nn.functional.layer_norm(x.permute(0, 2, 3, 1), (C,), ln_w, ln_b, 1e-6).permute(0, 3, 1, 2)
emb = bn1(lin(bn2(y).flatten(1)))
emb.backward(torch.randn(B, F) * 1e-3)
g = x.grad.permute(0, 2, 3, 1).reshape(-1, C)            # rows = (image, position), columns = channels: its column sums are the last block's fc2 bias gradient / gamma
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
kappa = g.sum(0).abs() / g.pow(2).sum(0).sqrt()
print("per-column |sum| / l2 norm: median %.3f" % kappa.median().item())
S = 1024.0                                               # a loss scale keeps fp16 in its normal range, as GradScaler does
print("column sums of the fp16 copy vs of the fp32 tensor: %.2e" % rel((g * S).half().float().sum(0) / S, g.sum(0)))
print("column sums of the bf16 copy vs of the fp32 tensor: %.2e" % rel(g.bfloat16().float().sum(0), g.sum(0)))
