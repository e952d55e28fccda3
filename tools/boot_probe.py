"""Fixed-vs-per-tile cost of the CBIR scan kernel: the bootstrap pass scans G = 4k tiles, so sweeping k sweeps the work."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiondk_amd import cbir
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(0)
gal = cbir.l2_normalize(torch.randn(1_000_000, 128, generator=g).to(dev))
qry = cbir.l2_normalize(torch.randn(10000, 128, generator=g).to(dev))
for k in (30, 60, 120, 240):
    index = cbir.FlatIPIndex(128, device=dev); index.add(gal)
    index.search(qry, k); index.search(qry, k)
    torch.cuda.synchronize()
