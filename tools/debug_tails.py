import sys, math, ctypes as C
sys.path.insert(0,'/root/repo')
import torch
from mmmm_amd import kernels as K, hip
dev=torch.device('cuda:0')
lib=hip.lib()
def ops(M,N,Kd,K2,seed):
    g=torch.Generator(device='cpu').manual_seed(seed)
    a=torch.randn(M,Kd,generator=g).to(dev).bfloat16(); w=(torch.randn(N,Kd,generator=g)/math.sqrt(Kd)).to(dev).bfloat16(); w1=(torch.randn(N,Kd,generator=g)/math.sqrt(Kd)).to(dev).bfloat16()
    a2=torch.randn(M,K2,generator=g).to(dev).bfloat16() if K2 else None
    b2=(torch.randn(N,K2,generator=g)*0.05).to(dev).bfloat16() if K2 else None
    b21=(torch.randn(N,K2,generator=g)*0.05).to(dev).bfloat16() if K2 else None
    return a,w,w1,a2,b2,b21
for (M,N,Kd,K2,split,plain) in [(4128,4096,4096,64,2064,True),(4128,4096,4096,64,2064,False),(4128,12288,4096,64,2064,True),(4128,4096,11008,64,2064,False),(4176,4096,4096,64,4104,True),(16392,15360,1792,64,None,True)]:
    a,w,w1,a2,b2,b21=ops(M,N,Kd,K2,1)
    counts=torch.tensor([split,M],dtype=torch.int32,device=dev) if split is not None else None
    kw=dict(w1=w1 if split is not None else None,a2=a2,b2=b2,b2_1=b21 if (K2 and split is not None) else None,counts=counts)
    if not plain: kw.update(drop_p=0.05, drop_seed=1234)
    two=K.gemm(a,w,**kw)
    lib.vm_gemm_tails_mode_(0); one=K.gemm(a,w,**kw); lib.vm_gemm_tails_mode_(1)
    torch.cuda.synchronize()
    d=(two.float()-one.float()).abs()
    rows=(d.max(dim=1).values>0).nonzero().flatten()
    print(M,N,Kd,K2,split,plain,'maxdiff',d.max().item(),'nrows differing',rows.numel(), rows[:8].tolist(), rows[-8:].tolist())
    import time
    for mode in (0,1,3,4):
        lib.vm_gemm_tails_mode_(mode)
        for _ in range(3): K.gemm(a,w,**kw)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(20): K.gemm(a,w,**kw)
        torch.cuda.synchronize(); print('  tails mode',mode,(time.perf_counter()-t0)/20*1e6,'us')
    lib.vm_gemm_tails_mode_(1)
