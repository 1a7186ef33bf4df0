import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
for mb in (268, 1073):
    x=torch.empty(mb*1000*1000//2, dtype=torch.bfloat16, device='cuda'); y=torch.empty_like(x)
    tf=t(lambda: x.zero_()); tc=t(lambda: y.copy_(x)); tr=t(lambda: x.sum())
    print(f"{mb} MB: fill {tf*1e6:.0f} us = {mb/tf/1e6:.2f} TB/s | copy {tc*1e6:.0f} us = {2*mb/tc/1e6:.2f} TB/s (r+w) | read-reduce {tr*1e6:.0f} us = {mb/tr/1e6:.2f} TB/s")
