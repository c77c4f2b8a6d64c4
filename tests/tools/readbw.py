import torch
dev=torch.device("cuda:0")
x=torch.randn(256*16384,128,device=dev); x2=torch.randn_like(x)
def t(fn,n=6):
    fn(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1)/n*1e-3
by=x.numel()*4
ts=t(lambda:(x.sum(),x2.sum()))/2
print(f"torch.sum read-only: {ts*1e6:.1f} us {by/ts/1e9:.0f} GB/s")
tm=t(lambda:(x.max(),x2.max()))/2
print(f"torch.max read-only: {tm*1e6:.1f} us {by/tm/1e9:.0f} GB/s")
y=torch.empty_like(x)
tc=t(lambda:torch.add(x,1.0,out=y))
print(f"torch add (r+w): {tc*1e6:.1f} us {2*by/tc/1e9:.0f} GB/s")
tz=t(lambda:y.fill_(1.0))
print(f"torch fill (write-only): {tz*1e6:.1f} us {by/tz/1e9:.0f} GB/s")
