import torch, time
d = torch.device("cuda:0")
x = torch.ones(6 * 1024**3 // 8, dtype=torch.int64, device=d)
for f, name in ((lambda: x.sum(), "sum int64 6GiB"), (lambda: (x.view(torch.int32)).sum(), "sum int32"), (lambda: x.max(), "max int64")):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(name, round(x.numel() * 8 / dt / 1e12, 3), "TB/s")
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print("copy (read+write)", round(2 * x.numel() * 8 / dt / 1e12, 3), "TB/s")
