import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
for mb in (88, 351, 1404):
    y=torch.empty(mb*1024*1024//4, device='cuda')
    ms=t(lambda: y.fill_(1.0))
    x=torch.empty_like(y)
    ms2=t(lambda: y.copy_(x))
    print(f"{mb} MB fill {ms*1e3:.1f} us -> {mb*1.048576/ms:.0f} GB/s ; copy {ms2*1e3:.1f} us -> {2*mb*1.048576/ms2:.0f} GB/s (r+w)")
