import torch
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (268, 537, 1074):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    t = timeit(lambda: a.fill_(1.0)); print(f"fill  {mb} MB: {t:7.1f} us  {mb/t:.2f} TB/s write")
    t = timeit(lambda: b.copy_(a)); print(f"copy  {mb} MB: {t:7.1f} us  {2*mb/t:.2f} TB/s r+w")
    t = timeit(lambda: a.sum()); print(f"sum   {mb} MB: {t:7.1f} us  {mb/t:.2f} TB/s read")
    t = timeit(lambda: torch.relu_(a)); print(f"relu_ {mb} MB: {t:7.1f} us  {2*mb/t:.2f} TB/s r+w")
