"""ms per search of the headline CBIR problem under the current environment knobs (same-box A/B of schedules):  python tools/cbir_quick.py [default|small|guaranteed] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import cbir
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(0)
gal = cbir.l2_normalize(torch.randn(1_000_000, 128, generator=g).to(dev))
g.manual_seed(1)
qry = cbir.l2_normalize(torch.randn(10_000, 128, generator=g).to(dev))
kw = {} if mode == "default" else {"small_lists": mode == "small"}
if os.environ.get("CBIR_CAP"): kw["cap"] = int(os.environ["CBIR_CAP"])
index = cbir.FlatIPIndex(128, device=dev, **kw)
index.add(gal)
for _ in range(4):
    s, i = index.search(qry, 100)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    s, i = index.search(qry, 100)
e1.record(); torch.cuda.synchronize()
print(mode, {k: v for k, v in os.environ.items() if k.startswith("VDK_CBIR") or k.startswith("CBIR_")}, "ms/search %.3f" % (e0.elapsed_time(e1) / iters), "fallbacks", index.fallbacks,
      "checksum", int(i.sum().item()), float(s.double().sum().item()))
