import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch, torch.nn.functional as F
from spacap3d_amd import attention as att
from spacap3d_amd.transformer_captioner import tall_linear
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B,H,K,D = 8,8,256,16
P = torch.rand(B,H,K,K, device='cuda', requires_grad=True); V = torch.randn(B,K,H,D, device='cuda').transpose(1,2).requires_grad_(True)
lin = torch.nn.Linear(128,128).cuda()
w = torch.randn(B,K,K,128, device='cuda')
def old():
    return F.relu(tall_linear(att.relation_feature(P, V), lin))
def new():
    return att.relation_layer1(P, V, lin.weight, lin.bias)
for name, f in (("old", old), ("new", new)):
    print(name, "fwd", t(f))
    y = f()
    print(name, "bwd", t(lambda: torch.autograd.grad(y, (P, V, lin.weight, lin.bias), w, retain_graph=True)))
