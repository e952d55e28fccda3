import os, sys, torch
sys.path.insert(0, '/root/repo')
from visiondk_amd import _lib, ops
be = _lib.load()
def rel(a,b): return ((a.double()-b.double()).norm()/b.double().norm()).item()
dt=torch.float16
for (B,N,H) in ((1,224,1),(4,197,12)):
    torch.manual_seed(5)
    D=H*64
    qkv=(torch.randn(B,N,3*D)*1.4).to(dt); qkv[0,N//2,:D]*=3.0; qkv=qkv.cuda()
    dout=torch.randn(B,N,D).to(dt).cuda()
    o,lse=ops.attention_fwd(qkv,H,backend=be)
    qr=qkv.float().requires_grad_(True)
    x=qr.reshape(B,N,3,H,64).permute(2,0,3,1,4)
    att=torch.softmax((x[0]@x[1].transpose(-1,-2))*0.125,-1)@x[2]
    att.transpose(1,2).reshape(B,N,D).backward(dout.float())
    outs={}
    for form,grid in (("3",None),("3","1"),("5",None),("5","1"),("5","3"),("5",None)):
        os.environ["VDK_ATTN_BWD_FORM"]=form
        if grid: os.environ["VDK_ATTN_GRID"]=grid
        else: os.environ.pop("VDK_ATTN_GRID",None)
        r=ops.attention_bwd(qkv,o,dout,lse,H,backend=be)
        torch.cuda.synchronize()
        key=(form,grid)
        if key in outs: print(N,"repeat",key,"equal",torch.equal(outs[key],r))
        outs[key]=r
        print(N,key,"vs fp32 ref: dq %.3e dk %.3e dv %.3e"%tuple(rel(r[...,i*D:(i+1)*D].float(), qr.grad[...,i*D:(i+1)*D]) for i in range(3)))
    print(N,"form3 grid-indep",torch.equal(outs[("3",None)],outs[("3","1")]),"form5 grid-indep",torch.equal(outs[("5",None)],outs[("5","1")]),torch.equal(outs[("5",None)],outs[("5","3")]))
    print(N,"dk equal 3 vs 5",torch.equal(outs[("3",None)][...,D:2*D],outs[("5",None)][...,D:2*D]))
