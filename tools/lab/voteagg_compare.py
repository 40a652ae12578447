"""Lab: the vote-aggregation SA module on the model's own votes: fused op vs per-operator path vs float64, gradient w.r.t. xyz."""
import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_engine_gpu as T
import test_sa_mlp_gpu as TS
from spacap3d_amd import backend, pointnet2_utils as PU
DEV = "cuda:0"
data = T._anchored_batch()
model = T._fresh_model(DEV)
with torch.no_grad():
    d = model({k: v.to(DEV) for k, v in data.items()})
vx, vf = d["vote_xyz"].detach().clone(), d["vote_features"].detach().contiguous().clone()
print("vote_features", vf.shape, "norm per point", float(vf.norm(dim=1).mean()))
sa = model.proposal.vote_aggregation
inds = data["proposal_inds"].to(DEV)
dout = torch.randn(2, 128, 64, device=DEV)
res = {}
for mode in ("fused", "unfused"):
    m = copy.deepcopy(sa)
    x = vx.clone().requires_grad_(True)
    f = vf.clone().requires_grad_(True)
    hip = backend.ops()
    saved = hip.sa_mlp_train
    if mode == "unfused":
        hip.sa_mlp_train = None
    try:
        new_xyz, out, _ = m(x, f, inds)
        (out * dout).sum().backward()
    finally:
        hip.sa_mlp_train = saved
    res[mode] = (out.detach(), x.grad.clone(), f.grad.clone(), [l.conv.weight.grad.clone() for l in m.mlp_module.children()])
new_xyz = torch.gather(vx, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3))
idx = PU.ball_query(0.3, 16, vx, new_xyz)
print("groups: distinct neighbours per group (mean)", float(torch.tensor([[len(set(r.tolist())) for r in b] for b in idx.cpu()]).float().mean()))
want, dxyz, dnew, dfeat, dparams = TS._reference(sa, vx, new_xyz, vf, idx, dout, 0.3)
full = dxyz.clone()
full.scatter_add_(1, inds.cpu().long().unsqueeze(-1).expand(-1, -1, 3), dnew)
rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
for mode in ("fused", "unfused"):
    o, gx, gf, gw = res[mode]
    print(f"{mode:8s} out {rel(o, want):.2e}  dxyz {rel(gx, full):.2e}  dfeat {rel(gf, dfeat):.2e}  dW1 {rel(gw[0].view(128, -1), dparams[0]):.2e}  dW3 {rel(gw[2].view(128, -1), dparams[6]):.2e}")
print("fused vs unfused dxyz", rel(res["fused"][1], res["unfused"][1].double().cpu()))
