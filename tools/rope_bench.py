import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from ullsam_amd import ops
B,S,KVH,G,K,hd=4,1081,8,4,4096,128
dev="cuda"
x=torch.randn(B*S,K,device=dev).bfloat16(); w=(torch.randn(KVH*(G+2)*hd,K,device=dev)*K**-0.5).bfloat16()
pos=torch.arange(S,dtype=torch.int32,device=dev)[None].repeat(B,1).contiguous()
t=torch.arange(2048).float()[:,None]*(1.0/(1e6**(torch.arange(0,hd,2).float()/hd)))[None]
emb=torch.cat([t,t],-1); cos,sin=emb.cos().to(dev).contiguous(),emb.sin().to(dev).contiguous()
kc=torch.zeros(B,KVH,S,hd,device=dev,dtype=torch.bfloat16); vc=torch.zeros_like(kc)
def fused(): return ops.gemm_qkv_rope(x,w,None,kc,vc,pos,cos,sin,B,S,KVH,G,0)
def unfused():
    qkv=ops.gemm(x,w); return ops.rope_split(qkv,kc,vc,pos,cos,sin,B,S,KVH,G,hd,0)
from ullsam_amd import _lib
lib=_lib.load()
def fused7():
    lib.ullsam_set_gemm_variant(7)
    try: return fused()
    finally: lib.ullsam_set_gemm_variant(0)
ref=fused()[0].float() if isinstance(fused(),tuple) else fused().float()
got=fused7(); got=(got[0] if isinstance(got,tuple) else got).float()
print("variant 7 vs auto: max |dq|", (got-ref).abs().max().item())
for name,fn in (("unfused",unfused),("fused",fused),("fused, four-wave kernel",fused7),("unfused",unfused),("fused",fused),("fused, four-wave kernel",fused7)):
    fn(); ts=[]
    for r in range(7):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/8)
    print(name, round(sorted(ts)[3]*1e3,1),"us")
