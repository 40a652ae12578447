import torch
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 8*256*256
X = torch.randn(M, 128, device='cuda'); G = torch.randn(M, 128, device='cuda'); G9 = torch.randn(M, 9, device='cuda')
ref = G.t() @ X
for S in (64, 128, 256, 512, 1024):
    f = lambda: torch.bmm(G.view(S, M // S, 128).transpose(1, 2), X.view(S, M // S, 128)).sum(0)
    out = f()
    print('splitK', S, t(f), 'ms  err', float((out - ref).abs().max() / ref.abs().max()))
    f9 = lambda: torch.bmm(G9.view(S, M // S, 9).transpose(1, 2), X.view(S, M // S, 128)).sum(0)
    print('   9x128', t(f9))
print('plain', t(lambda: G.t() @ X), t(lambda: G9.t() @ X))
print('bias grad sum(0)', t(lambda: G.sum(0)))
