import os, sys, json, torch
sys.path.insert(0, '/root/repo')
from visiondk_amd import _lib, ops
be = _lib.load()
B,N,H=256,197,12; D=H*64
dt=torch.float16
torch.manual_seed(0)
qkv=torch.randn(B,N,3*D,device="cuda").to(dt); dout=torch.randn(B,N,D,device="cuda").to(dt)
o,lse=ops.attention_fwd(qkv,H,backend=be)
os.environ["VDK_ATTN_BWD_FORM"]="5"
res={}
for dbg in (0,1,2,3,4,8,7,15,0):
    os.environ["VDK_ATTN5_DBG"]=str(dbg)
    fn=lambda: ops.attention_bwd(qkv,o,dout,lse,H,backend=be)
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    res.setdefault(dbg,[]).append(round(e0.elapsed_time(e1)/20*1e3,1))
print(json.dumps(res))
