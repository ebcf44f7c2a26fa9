import torch,time
n=6_000_000_000
a=torch.empty(n,dtype=torch.complex64,device='cuda')
f=torch.view_as_real(a).reshape(-1)
for name,fn in (("fill",lambda: f.fill_(1.0)),("zero",lambda: f.zero_())):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/3
    print(name, ms,"ms", n*8/ms/1e6,"GB/s")
b=torch.empty(n//2*2,dtype=torch.float32,device='cuda')
src=f[:n]; dst=f[n:2*n]
dst.copy_(src); torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): dst.copy_(src)
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/3
print("copy", ms, "ms", 2*n*4/ms/1e6, "GB/s (r+w)")
