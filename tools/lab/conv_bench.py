import torch, torch.nn.functional as F
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, Cin, Cout, P, S) in [(8,4,64,2048,64),(8,64,64,2048,64),(8,64,128,2048,64),(8,131,128,1024,32),(8,128,128,1024,32),(8,128,256,1024,32),(8,259,128,512,16)]:
    x = torch.randn(B, Cin, P, S, device='cuda', requires_grad=True); w = torch.randn(Cout, Cin, 1, 1, device='cuda', requires_grad=True)
    g = torch.randn(B, Cout, P, S, device='cuda')
    L = P * S
    y = F.conv2d(x, w)
    fwd_c = t(lambda: F.conv2d(x, w))
    bwd_c = t(lambda: torch.autograd.grad(y, (x, w), g, retain_graph=True))
    w2 = w.view(Cout, Cin); x3 = x.view(B, Cin, L); g3 = g.view(B, Cout, L)
    fwd_m = t(lambda: torch.matmul(w2, x3))
    dx_m = t(lambda: torch.matmul(w2.t(), g3))
    dw_plain = t(lambda: torch.bmm(g3, x3.transpose(1, 2)).sum(0))
    SL = 16
    dw_split = t(lambda: torch.bmm(g3.reshape(B, Cout, SL, L // SL).permute(0, 2, 1, 3).reshape(B * SL, Cout, L // SL), x3.reshape(B, Cin, SL, L // SL).permute(0, 2, 3, 1).reshape(B * SL, L // SL, Cin)).sum(0))
    dw_split2 = t(lambda: torch.einsum('bol,bil->oi', g3, x3))
    print(f"{(B,Cin,Cout,P,S)} conv2d fwd {fwd_c:.3f} bwd {bwd_c:.3f} | matmul fwd {fwd_m:.3f} dx {dx_m:.3f} dw(bmm+sum) {dw_plain:.3f} dw(split16) {dw_split:.3f} einsum {dw_split2:.3f}")
