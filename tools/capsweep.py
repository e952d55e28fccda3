import sys, os, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visiondk_amd import cbir
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(0)
gal = cbir.l2_normalize(torch.randn(1_000_000, 128, generator=g).to(dev))
g.manual_seed(1)
qry = cbir.l2_normalize(torch.randn(10000, 128, generator=g).to(dev))
ref = None
for cap in (65536, 131072, 262144, 524288, 1048576 + 100):
    index = cbir.FlatIPIndex(128, device=dev, cap=cap); index.add(gal)
    s, i = index.search(qry, 100); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): s, i = index.search(qry, 100)
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = (s.clone(), i.clone())
    print(cap, e0.elapsed_time(e1) / 3, "ms", bool(torch.equal(i, ref[1]) and torch.equal(s, ref[0])), flush=True)
    del index; torch.cuda.empty_cache()
