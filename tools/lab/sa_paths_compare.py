import sys, copy
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from spacap3d_amd import backend, synthetic as S
from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
DEV = "cuda:0"
torch.manual_seed(3)
sa = PointnetSAModuleVotes(npoint=128, radius=0.8, nsample=32, mlp=[128, 128, 128, 256], use_xyz=True, normalize_xyz=True).to(DEV).train()
sb = copy.deepcopy(sa)
xyz = S.scene_batch(2, 1024, use_height=False, seed=1).to(DEV)
feats = F.relu(torch.randn(2, 128, 1024, device=DEV))
fa = feats.clone().requires_grad_(True); fb = feats.clone().requires_grad_(True)
_, oa, _ = sa(xyz, fa)
hip = backend.ops(); saved = hip.sa_mlp_train
hip.sa_mlp_train = None
_, ob, _ = sb(xyz, fb)
hip.sa_mlp_train = saved
w = torch.randn_like(oa)
(oa * w).sum().backward(); (ob * w).sum().backward()
print("out max abs diff", float((oa - ob).abs().max()), "scale", float(ob.abs().max()))
d = (fa.grad - fb.grad).abs(); tol = 1e-3 * fb.grad.abs() + 1e-4 * float(fb.grad.abs().max())
print("feature grad: elements outside", int((d > tol).sum()), "of", d.numel(), "max diff", float(d.max()), "scale", float(fb.grad.abs().max()))
for la, lb in zip(sa.mlp_module.children(), sb.mlp_module.children()):
    ga, gb = la.conv.weight.grad, lb.conv.weight.grad
    dd = (ga - gb).abs(); t2 = 1e-3 * gb.abs() + 1e-4 * float(gb.abs().max())
    print("dW outside", int((dd > t2).sum()), "of", dd.numel(), "max", float(dd.max()), "scale", float(gb.abs().max()))
