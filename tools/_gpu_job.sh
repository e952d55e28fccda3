for m in 4 8 16 32; do
VDK_CBIR_BOOT_MULT=$m python - <<'PY'
import os,sys,torch
sys.path.insert(0,'.')
from visiondk_amd import cbir
dev=torch.device("cuda:0")
g=torch.Generator(device="cpu"); g.manual_seed(0)
gal=cbir.l2_normalize(torch.randn(1_000_000,128,generator=g).to(dev)); g.manual_seed(1)
qry=cbir.l2_normalize(torch.randn(10_000,128,generator=g).to(dev))
index=cbir.FlatIPIndex(128,device=dev); index.add(gal)
for _ in range(4): s,i=index.search(qry,100)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8): s,i=index.search(qry,100)
e1.record(); torch.cuda.synchronize()
print("boot mult",os.environ["VDK_CBIR_BOOT_MULT"],"ms",e0.elapsed_time(e1)/8,"checksum",int(i.sum().item()))
PY
done
