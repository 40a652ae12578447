"""Lab: where does the 4e-3 first-layer weight-gradient error at C = 132 come from?  Fused op vs float64, per column
block, and the same for the per-operator (unfused, torch / rocBLAS) path."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spacap3d_amd import backend, synthetic as S, pointnet2_utils as pu
from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
DEV = "cuda:0"
C = int(sys.argv[1]) if len(sys.argv) > 1 else 132
torch.manual_seed(C)
pc = S.scene_batch(2, 40000, use_color=(C == 7), use_normal=True, use_multiview=(C == 132), seed=C).to(DEV)
xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
MODE = sys.argv[2] if len(sys.argv) > 2 else "plain"
if MODE == "center":
    feats = (feats - feats.mean(dim=2, keepdim=True)).contiguous()
elif MODE == "offset":
    feats = (feats + 10.0).contiguous()
elif MODE == "randn":
    feats = torch.randn_like(feats)
print("mode", MODE, "C", C)
sa = PointnetSAModuleVotes(npoint=2048, radius=0.2, nsample=64, mlp=[C, 64, 64, 128], use_xyz=True, normalize_xyz=True).to(DEV).train()
new_xyz, out, inds = sa(xyz, feats)
wsum = torch.randn(out.shape, device=DEV)
(out * wsum).sum().backward()
g_fused = [l.conv.weight.grad.clone().view(l.conv.out_channels, -1) for l in sa.mlp_module.children()]
# unfused path
sa.zero_grad()
hip = backend.ops()
saved = hip.sa_mlp_train
hip.sa_mlp_train = None
try:
    _, out2, _ = sa(xyz, feats, inds)
    (out2 * wsum).sum().backward()
finally:
    hip.sa_mlp_train = saved
g_unf = [l.conv.weight.grad.clone().view(l.conv.out_channels, -1) for l in sa.mlp_module.children()]
with torch.no_grad():
    idx = pu.ball_query(0.2, 64, xyz, new_xyz).long()
    B, P, Sn = idx.shape
    flat = idx.view(B, -1)
x64, f64 = xyz.double(), feats.double()
g_xyz = torch.gather(x64, 1, flat.unsqueeze(-1).expand(-1, -1, 3)).view(B, P, Sn, 3)
rel = (g_xyz - new_xyz.double().unsqueeze(2)) / 0.2
g_f = torch.gather(f64, 2, flat.unsqueeze(1).expand(-1, C, -1)).view(B, C, P, Sn).permute(0, 2, 3, 1)
h = torch.cat([rel, g_f], -1)
ws = [l.conv.weight.detach().double().view(l.conv.out_channels, -1).requires_grad_(True) for l in sa.mlp_module.children()]
for w, l in zip(ws, sa.mlp_module.children()):
    z = h @ w.t()
    mu, var = z.mean((0, 1, 2)), z.var((0, 1, 2), unbiased=False)
    h = torch.relu((z - mu) / torch.sqrt(var + l.bn.bn.eps) * l.bn.bn.weight.double() + l.bn.bn.bias.double())
ref = h.max(2).values.permute(0, 2, 1)
(ref * wsum.double()).sum().backward()
for i, l in enumerate(sa.mlp_module.children()):
    pass
print("out fused vs f64", float((out.double() - ref).abs().max() / ref.abs().max()), " unfused", float((out2.double() - ref).abs().max() / ref.abs().max()))
for i, w in enumerate(ws):
    r = w.grad
    for nm, g in (("fused", g_fused[i]), ("unfused", g_unf[i])):
        e = (g.double() - r)
        msg = f"layer {i} {nm:8s} l2 {float(e.norm() / r.norm()):.2e}"
        if i == 0:
            msg += f"  xyz cols {float(e[:, :3].norm() / r[:, :3].norm()):.2e}  feat cols {float(e[:, 3:].norm() / r[:, 3:].norm()):.2e}"
            msg += f"  |r xyz| {float(r[:, :3].norm()):.3e} |r feat| {float(r[:, 3:].norm()):.3e}"
        print(msg)
# statistics of the pre-activations (float64 restatement): |mean| / std per layer
hh = torch.cat([rel, g_f], -1)
for i, (w, l) in enumerate(zip(ws, sa.mlp_module.children())):
    z = hh @ w.detach().t()
    mu, sd = z.mean((0, 1, 2)), z.std((0, 1, 2))
    print(f"layer {i}: |mean|/std max {float((mu.abs() / sd).max()):.2f} median {float((mu.abs() / sd).median()):.2f}  std min {float(sd.min()):.3e} max {float(sd.max()):.3e}")
    hh = torch.relu((z - mu) / torch.sqrt(z.var((0, 1, 2), unbiased=False) + l.bn.bn.eps) * l.bn.bn.weight.double() + l.bn.bn.bias.double())
