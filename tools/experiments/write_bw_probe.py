import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n*1e-3
for mb in (64, 420, 840, 2048):
    x=torch.empty(mb*1024*1024//2, dtype=torch.bfloat16, device='cuda'); y=torch.empty_like(x)
    tf=t(lambda: x.fill_(1.0)); tc=t(lambda: y.copy_(x)); tr=t(lambda: x.float().sum()) if mb<=840 else 0
    tz=t(lambda: x.zero_())
    print(f"{mb} MB: fill {mb/1024/tf/1e3*1.0737:.2f} TB/s, zero_ {mb/1024/tz/1e3*1.0737:.2f} TB/s, copy (r+w) {2*mb/1024/tc/1e3*1.0737:.2f} TB/s")
