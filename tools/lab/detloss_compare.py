"""Lab: gradient of every detection-loss term w.r.t. the proposal head rows / centres / votes, fused HIP op vs the
torch composition (CPU checker backend), on the anchored trajectory batch."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_engine_gpu as T
from oracle.attention_ref import OracleBackend
from spacap3d_amd import backend, synthetic as S
from spacap3d_amd.loss_helper import get_scene_cap_loss

data = T._anchored_batch()
PN = ["proposal.proposal.6.weight", "proposal.proposal.6.bias", "proposal.proposal.4.weight", "proposal.proposal.3.weight",
      "proposal.proposal.1.weight", "proposal.proposal.0.weight", "proposal.vote_aggregation.mlp_module.layer2.conv.weight"]
TERMS = ("loss", "cap_loss", "relation_loss", "det_loss", "box_loss", "vote_loss", "objectness_loss", "center_loss", "size_cls_loss", "size_reg_loss", "sem_cls_loss", "heading_cls_loss", "heading_reg_loss")
out = {}
for name, be, dev in (("cpu", OracleBackend(), "cpu"), ("hip", backend.HipBackend(), "cuda:0")):
    with backend.use_backend(be):
        model = T._fresh_model(dev)
        d = model({k: v.to(dev) for k, v in data.items()})
        d = get_scene_cap_loss(d, use_relation=True, mean_size_arr=S.mean_size_arr().numpy())
        wrt = [d["_proposal_net"], d["center"], d["vote_xyz"], d["aggregated_vote_xyz"], d["aggregated_vote_features"]]
        res = {}
        for t in TERMS:
            gs = torch.autograd.grad(d[t], wrt, retain_graph=True, allow_unused=True)
            res[t] = [None if g is None else g.detach().cpu().double() for g in gs]
        loss_params = dict(model.named_parameters())
        pg = torch.autograd.grad(d["loss"], [loss_params[n] for n in PN], retain_graph=True, allow_unused=True)
        res["pgrad"] = [None if g is None else g.detach().cpu().double() for g in pg]
        res["labels"] = (d["objectness_label"].cpu(), d["object_assignment"].cpu(), d["objectness_mask"].cpu())
        res["vals"] = {t: float(d[t]) for t in TERMS}
        out[name] = res
c, h = out["cpu"], out["hip"]
print("labels equal:", [bool(torch.equal(a, b)) for a, b in zip(c["labels"], h["labels"])], "positives", int(c["labels"][0].sum()))
for t in TERMS:
    msg = f"{t:18s} val cpu {c['vals'][t]:.6f} hip {h['vals'][t]:.6f} |"
    for nm, a, b in zip(("net", "center", "vote", "aggxyz", "aggfeat"), c[t], h[t]):
        if a is None and b is None:
            continue
        if a is None or b is None:
            msg += f" {nm}: ONE-SIDED(cpu {None if a is None else float(a.norm()):} hip {None if b is None else float(b.norm()):})"
            continue
        msg += f" {nm}: {float((a - b).norm() / (a.norm() + 1e-30)):.2e} (|g| {float(a.norm()):.2e})"
    print(msg)

for n, a, b in zip(PN, c["pgrad"], h["pgrad"]):
    print(f"{n:60s} {float((a - b).norm() / a.norm()):.2e}")
