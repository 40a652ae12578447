"""Lab: the proposal head (Conv1d-BN-ReLU x2 + Conv1d) alone, CPU float64 vs GPU, with and without the Conv1x1 op."""
import copy, os, sys
import torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spacap3d_amd import backend
from spacap3d_amd.detector import _conv
torch.manual_seed(0)
head = nn.Sequential(nn.Conv1d(128, 128, 1, bias=False), nn.BatchNorm1d(128), nn.ReLU(), nn.Conv1d(128, 128, 1, bias=False),
                     nn.BatchNorm1d(128), nn.ReLU(), nn.Conv1d(128, 97, 1)).train()
for P in (64, 256):
    x = torch.relu(torch.randn(2, 128, P))
    dout = torch.randn(2, 97, P)
    ref = copy.deepcopy(head).double()
    xr = x.double().requires_grad_(True)
    (ref(xr) * dout.double()).sum().backward()
    for mode in ("module", "conv1x1"):
        m = copy.deepcopy(head).cuda()
        xg = x.cuda().requires_grad_(True)
        net = xg
        for layer in m:
            if mode == "conv1x1" and isinstance(layer, nn.Conv1d):
                net = _conv(layer, net, True)
            else:
                net = layer(net)
        (net * dout.cuda()).sum().backward()
        rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
        print(f"P={P} {mode:8s} dx {rel(xg.grad, xr.grad):.2e} " + " ".join(f"{n}:{rel(p.grad, q.grad):.1e}" for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters())))
