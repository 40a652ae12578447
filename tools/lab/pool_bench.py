import torch, torch.nn.functional as F
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for shape in [(8,128,2048,64), (8,256,1024,32), (8,256,512,16)]:
    x = torch.randn(*shape, device='cuda', requires_grad=True)
    g = torch.randn(shape[:3], device='cuda')
    print(shape, 'GB', x.numel()*4/1e9)
    print('  max_pool2d fwd', t(lambda: F.max_pool2d(x, kernel_size=[1, x.size(3)])))
    print('  torch.max  fwd', t(lambda: torch.max(x, dim=3)))
    print('  amax       fwd', t(lambda: torch.amax(x, dim=3)))
    y = F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
    print('  max_pool2d bwd', t(lambda: torch.autograd.grad(y, x, g, retain_graph=True)))
    y = torch.max(x, dim=3)[0]
    print('  torch.max  bwd', t(lambda: torch.autograd.grad(y, x, g, retain_graph=True)))
    y = torch.amax(x, dim=3)
    print('  amax       bwd', t(lambda: torch.autograd.grad(y, x, g, retain_graph=True)))
