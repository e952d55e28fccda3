"""visiondk_amd.evaluate.valuate vs the formulas of the reference's classification `valuate` (engine/procedure/evaluation.py:52-185) restated with
plain torch on the same logits."""
import torch

from visiondk_amd import evaluate


class _FakeModel(torch.nn.Module):
    def __init__(self, table, be):
        super().__init__()
        self.table = table
        class _E: pass
        self.engine = _E(); self.engine.be = be

    def forward(self, x):
        return self.table[x[:, 0].long()]


def _loader(n, bs, labels, dev):
    ids = torch.arange(n, dtype=torch.float32)[:, None]
    return [(ids[i:i + bs], labels[i:i + bs]) for i in range(0, n, bs)]


def test_single_label_top1_topk(be, dev):
    torch.manual_seed(0)
    n, C = 57, 12
    logits = torch.randn(n, C)
    labels = torch.randint(0, C, (n,))
    model = _FakeModel(logits.to(dev), be)
    top1, top5, loss = evaluate.valuate(model, _loader(n, 16, labels, dev), dev, lossfn=lambda y, t: torch.nn.functional.cross_entropy(y, t),
                                        top_k=5, class_indices=[str(i) for i in range(C)])
    pred = logits.argsort(1, descending=True)[:, :5]
    correct = (labels[:, None] == pred).float()
    assert abs(top1 - correct[:, 0].mean().item()) < 1e-6 and abs(top5 - correct.max(1).values.mean().item()) < 1e-6
    exp_loss = sum(torch.nn.functional.cross_entropy(logits[i:i + 16], labels[i:i + 16]).item() for i in range(0, n, 16)) / 4
    assert abs(loss - exp_loss) < 1e-5


def test_multi_label_precision_recall_f1(be, dev):
    torch.manual_seed(1)
    n, C = 40, 5
    logits = torch.randn(n, C)
    labels = (torch.rand(n, C) > 0.6).float()
    labels[:, 4] = 0                                   # a class without positives: recall 0 by convention
    model = _FakeModel(logits.to(dev), be)
    thr = [0.5, 0.3, 0.7, 0.5, 0.5]
    p, r, f = evaluate.valuate(model, _loader(n, 8, labels, dev), dev, thresh=thr, class_indices=list("abcde"))
    pred = logits.sigmoid() >= torch.tensor(thr)
    tgt = labels.bool()
    tp = (pred & tgt).sum(0).double(); fp = (pred & ~tgt).sum(0).double(); fn = (~pred & tgt).sum(0).double()
    z = lambda a, b: torch.where(b > 0, a / b.clamp_min(1), torch.zeros_like(a))
    assert abs(p - z(tp, tp + fp).mean().item()) < 1e-9
    assert abs(r - z(tp, tp + fn).mean().item()) < 1e-9
    assert abs(f - z(2 * tp, 2 * tp + fp + fn).mean().item()) < 1e-9
