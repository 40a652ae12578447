import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
for rows, C in [(524288,9),(524288,3),(524288,128)]:
    g=torch.randn(rows,C,device='cuda')
    r=[('sum0',lambda: g.sum(0))]
    for S in (256,1024,4096,16384):
        r.append((f'slab{S}',lambda S=S: g.view(S,rows//S,C).sum(1).sum(0)))
    r.append(('flat', lambda: g.view(rows//64, 64*C).sum(0).view(64,C).sum(0)))
    r.append(('flat512', lambda: g.view(rows//512, 512*C).sum(0).view(512,C).sum(0)))
    print(rows,C,' '.join(f'{n}={t(f)*1e3:.1f}us' for n,f in r))
