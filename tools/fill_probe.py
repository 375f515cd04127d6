import torch, time
dev="cuda:0"
for mb in (75, 300, 1200):
    x=torch.empty(mb*1024*1024//4, device=dev)
    for _ in range(5): x.fill_(1.0)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): x.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/50
    print(f"fill {mb} MB: {ms*1e3:.1f} us  {mb*1.048576/ms:.0f} GB/s")
    y=torch.empty_like(x)
    e0.record()
    for _ in range(50): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/50
    print(f"copy {mb} MB: {ms*1e3:.1f} us  {2*mb*1.048576/ms:.0f} GB/s (r+w)")
