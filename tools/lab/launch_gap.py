"""Lab: what does one more (tiny, dependent) kernel cost inside a replayed hipGraph?"""
import torch, time
dev = torch.device("cuda:0")
a = torch.zeros(256, device=dev)
big = torch.zeros(1 << 22, device=dev)   # 16 MB: ~6 us of work
s = torch.cuda.Stream(device=dev)
def graph_of(fn, n):
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n): fn()
    return g
def timed(g, reps=20):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
for name, fn in (("tiny add (256 floats)", lambda: a.add_(1)), ("16 MB add", lambda: big.add_(1))):
    t1, t2 = timed(graph_of(fn, 200)), timed(graph_of(fn, 1000))
    print(f"{name}: 200 kernels {t1:8.1f} us, 1000 kernels {t2:8.1f} us -> {(t2 - t1) / 800:.2f} us per additional kernel", flush=True)
