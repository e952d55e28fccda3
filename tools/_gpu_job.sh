python -m pytest tests/test_cbir.py -m gpu -q 2>&1 | grep -E "passed|failed" | tail -1
python - <<'PY'
import os,sys,torch
sys.path.insert(0,'.')
from visiondk_amd import cbir
dev=torch.device("cuda:0")
g=torch.Generator(device="cpu"); g.manual_seed(0)
gal=cbir.l2_normalize(torch.randn(1_000_000,128,generator=g).to(dev)); g.manual_seed(1)
qry=cbir.l2_normalize(torch.randn(10_000,128,generator=g).to(dev))
ref=None
for name,kw in (("guaranteed",{}),("small_lists",{"small_lists":True}),("guaranteed",{}),("small_lists",{"small_lists":True})):
    index=cbir.FlatIPIndex(128,device=dev,**kw); index.add(gal)
    for _ in range(4): s,i=index.search(qry,100)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): s,i=index.search(qry,100)
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref=i.clone()
    print(name,"ms",e0.elapsed_time(e1)/8,"equal",bool(torch.equal(i,ref)),"fallbacks",index.fallbacks,"ws GB",index._ws.numel()/1e9)
    del index; torch.cuda.empty_cache()
PY
