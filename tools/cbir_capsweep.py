"""search time of the headline CBIR problem against the candidate-list capacity (= stage size) of the guaranteed schedule"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import cbir
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(0)
gal = cbir.l2_normalize(torch.randn(1_000_000, 128, generator=g).to(dev)); g.manual_seed(1)
qry = cbir.l2_normalize(torch.randn(10_000, 128, generator=g).to(dev))
out = {}
ref = None
for cap in (32768, 65536, 98304, 131072, 196608, 262144):
    index = cbir.FlatIPIndex(128, device=dev, cap=cap); index.add(gal)
    s, i = index.search(qry, 100); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): s, i = index.search(qry, 100)
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = i.clone()
    out[cap] = {"ms": round(e0.elapsed_time(e1) / 4, 3), "equal": bool(torch.equal(i, ref))}
    del index; torch.cuda.empty_cache()
print(json.dumps(out))
