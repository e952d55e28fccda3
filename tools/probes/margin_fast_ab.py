import os, sys, json, torch
sys.path.insert(0, '/root/repo')
from visiondk_amd import _lib, heads
be = _lib.load()
B, D, C = 512, 512, 1_000_000
h = heads.ArcFace(D, C, margin_arc=0.35, margin_am=0.0, scale=32, backend=be, device="cuda")
feats = torch.randn(B, D, device="cuda"); labels = torch.randint(0, C, (B,), device="cuda")
res = {}
outs = {}
for rnd in range(3):
    for g in ("1", "0"):
        os.environ["VDK_MARGIN_GENERIC"] = g
        for _ in range(2): o = h.margin_ce(feats, labels, label_smoothing=0.0, cos_planes=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): o = h.margin_ce(feats, labels, label_smoothing=0.0, cos_planes=1)
        e1.record(); torch.cuda.synchronize()
        res.setdefault("generic_ms" if g == "1" else "arc_ms", []).append(round(e0.elapsed_time(e1) / 5, 3)); outs[g] = o
res["bit_equal"] = all(torch.equal(a, b) for a, b in zip(outs["0"], outs["1"]))
print(json.dumps(res))
