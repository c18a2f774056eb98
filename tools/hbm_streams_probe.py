import torch, time
dev=torch.device("cuda:0")
n=1<<30
a=torch.empty(n, dtype=torch.uint8, device=dev); b=torch.empty(n, dtype=torch.uint8, device=dev)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it
ms=t(lambda: a.zero_()); print(f"memset 1 GiB: {ms:.3f} ms  {n/ms/1e9:.2f} TB/s write")
ms=t(lambda: b.copy_(a)); print(f"copy 1 GiB: {ms:.3f} ms  {2*n/ms/1e9:.2f} TB/s total (read+write)")
af=a.view(torch.float32)
ms=t(lambda: af.sum()); print(f"read-reduce 1 GiB: {ms:.3f} ms  {n/ms/1e9:.2f} TB/s read")
x=torch.empty(n//2, dtype=torch.bfloat16, device=dev)
y=torch.empty(n//2, dtype=torch.bfloat16, device=dev); z=torch.empty(n//2, dtype=torch.bfloat16, device=dev)
def one_in_two_out():
    torch.add(x, 1.0, out=y); 
ms=t(lambda: torch.add(x,1.0,out=y)); print(f"1 read + 1 write (0.5+0.5 GiB): {ms:.3f} ms {n/ms/1e9:.2f} TB/s")
