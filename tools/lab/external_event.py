import torch, time
dev = torch.device("cuda:0")
print(torch.__version__)
try:
    ev = torch.cuda.Event(external=True)
except TypeError as e:
    print("no external events:", e); raise SystemExit
a = torch.zeros(1 << 24, device=dev); b = torch.zeros(1 << 24, device=dev); c = torch.zeros(4, device=dev)
side = torch.cuda.Stream(device=dev)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    for _ in range(2):
        a.add_(1)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    for _ in range(20): a.add_(1)          # part 1
    ev.record(torch.cuda.current_stream())
    for _ in range(200): b.add_(1)         # part 2 (long)
torch.cuda.synchronize()
a.zero_(); b.zero_()
g.replay()
side.wait_event(ev)
with torch.cuda.stream(side):
    c[0] = a[0]          # should see 20 (part 1 done), while part 2 still runs
    c[1] = b[0]          # likely < 200 if the side stream ran early
torch.cuda.synchronize()
print("a seen by side stream:", c[0].item(), " b seen:", c[1].item(), "(final b", b[0].item(), ")")
