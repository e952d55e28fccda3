"""N searches of the headline CBIR problem (10 k x 1 M x 128, k = 100, default method) and nothing else: the command the PMC passes of tools/pmc_cbir.py wrap.
usage: python tools/cbir_pmc_run.py [searches] [d] [default|small|guaranteed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondk_amd import cbir
n_search = int(sys.argv[1]) if len(sys.argv) > 1 else 6
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(0)
gal = cbir.l2_normalize(torch.randn(1_000_000, d, generator=g).to(dev))
g.manual_seed(1)
qry = cbir.l2_normalize(torch.randn(10_000, d, generator=g).to(dev))
mode = sys.argv[3] if len(sys.argv) > 3 else "default"
kw = {} if mode == "default" else {"small_lists": mode == "small"}
index = cbir.FlatIPIndex(d, device=dev, **kw)
index.add(gal)
for _ in range(n_search):
    s, i = index.search(qry, 100)
torch.cuda.synchronize()
print("searches", n_search, "fallbacks", index.fallbacks, "checksum", int(i.sum().item()))
