import torch, numpy as np, torch.nn.functional as F
torch.manual_seed(0)
def fma32(a,b,c):  # float32 fma via float64 (product exact in f64; one extra rounding, rare double-rounding)
    return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(np.float32)
K=np.array([0.25-0.4/2,0.25,0.4,0.25,0.25-0.4/2],dtype=np.float64).astype(np.float32)
Kt=torch.tensor([0.25 - 0.4 / 2.0, 0.25, 0.4, 0.25, 0.25 - 0.4 / 2.0], dtype=torch.float32)
print(K, Kt.numpy(), (K==Kt.numpy()).all())
for (H,W,N) in [(13,10,56),(7,5,56),(25,19,56),(97,76,8),(385,304,8),(2160,3840,2)]:
    x=(torch.rand(N,1,H,W)*100+50)
    ya=F.conv2d(x,Kt.view(1,1,5,1),stride=(2,1),padding=(2,0)).numpy()
    xp=np.pad(x.numpy(),((0,0),(0,0),(2,2),(0,0)))
    Ho=ya.shape[2]
    rows=[xp[:,:,k:k+2*Ho:2,:][:,:,:Ho] for k in range(5)]
    # sequential fma k0..k4 from 0
    acc=rows[0]*K[0]
    for k in range(1,5): acc=fma32(rows[k],np.float32(K[k])*np.ones_like(acc),acc)
    # reverse order
    accr=rows[4]*K[4]
    for k in range(3,-1,-1): accr=fma32(rows[k],np.float32(K[k])*np.ones_like(accr),accr)
    # non-fma sequential
    accn=rows[0]*K[0]
    for k in range(1,5): accn=(accn+rows[k]*K[k]).astype(np.float32)
    print((H,W), 'seq-fma eq', (acc==ya).mean(), 'rev-fma eq', (accr==ya).mean(), 'seq-nofma eq', (accn==ya).mean())
    # horizontal
    y=F.conv2d(x,Kt.view(1,1,1,5),stride=(1,2),padding=(0,2)).numpy()
    xp=np.pad(x.numpy(),((0,0),(0,0),(0,0),(2,2)))
    Wo=y.shape[3]
    cols=[xp[:,:,:,k:k+2*Wo:2][:,:,:,:Wo] for k in range(5)]
    acc=cols[0]*K[0]
    for k in range(1,5): acc=fma32(cols[k],np.float32(K[k])*np.ones_like(acc),acc)
    accn=cols[0]*K[0]
    for k in range(1,5): accn=(accn+cols[k]*K[k]).astype(np.float32)
    print('   horiz: seq-fma eq', (acc==y).mean(), 'seq-nofma', (accn==y).mean())
print(torch.__config__.show()[:600]); print(torch.get_num_threads())
