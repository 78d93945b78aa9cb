import torch, time
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (26, 52, 128, 512, 2048):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.float16, device=dev); b = torch.empty(n, dtype=torch.float16, device=dev)
    a.fill_(1.0)
    tw = timeit(lambda: a.fill_(2.0))
    tc = timeit(lambda: b.copy_(a))
    tr = timeit(lambda: a.sum())
    print(f"{mb:5d} MB  fill {mb/1024/tw/1e3*1.048576:6.2f} TB/s ({tw*1e6:7.1f} us)   copy {2*mb/1024/tc/1e3*1.048576:6.2f} TB/s ({tc*1e6:7.1f} us)   sum(read) {mb/1024/tr/1e3*1.048576:6.2f} TB/s ({tr*1e6:7.1f} us)")
