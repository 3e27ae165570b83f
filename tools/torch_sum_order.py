import torch, numpy as np
torch.manual_seed(1)
def f32(x): return np.float32(x)
def seq(p):  # p: [fl, N] float32 products; sequential
    acc=p[0].copy()
    for k in range(1,p.shape[0]): acc=(acc+p[k]).astype(np.float32)
    return acc
def chunked(p, step):  # cascade: level0 accumulates 'step' rows then is flushed into level1 ...
    fl=p.shape[0]
    acc1=None; acc0=None
    for k in range(fl):
        acc0 = p[k].copy() if acc0 is None else (acc0+p[k]).astype(np.float32)
        if (k+1)%step==0:
            acc1 = acc0 if acc1 is None else (acc1+acc0).astype(np.float32); acc0=None
    if acc0 is not None:
        acc1 = acc0 if acc1 is None else (acc1+acc0).astype(np.float32)
    return acc1
def four_acc(p):  # 4 interleaved accumulators (k mod 4), then ((a0+a1)+(a2+a3))
    fl=p.shape[0]
    a=[None]*4
    for k in range(fl):
        j=k%4
        a[j]=p[k].copy() if a[j] is None else (a[j]+p[k]).astype(np.float32)
    a=[x if x is not None else np.zeros_like(p[0]) for x in a]
    return (((a[0]+a[1]).astype(np.float32))+((a[2]+a[3]).astype(np.float32))).astype(np.float32)
for (H,W) in [(7,5),(96,119),(384,474)]:
  for fl in [3,5,7,9,13,15,17,19,25,31,33,41,65]:
    w=torch.rand(1,1,fl,H,W)*100+1
    cf=torch.randn(1,1,fl,1,1)*0.3
    out=(w*cf).sum(dim=-3,keepdim=True).numpy().reshape(-1)
    p=(w*cf).numpy().reshape(fl,-1)
    res={}
    res['seq']=(seq(p)==out).mean()
    for st in (4,8,16,32): res['chunk%d'%st]=(chunked(p,st)==out).mean()
    res['4acc']=(four_acc(p)==out).mean()
    print((H,W),fl,{k:round(float(v),3) for k,v in res.items()})
