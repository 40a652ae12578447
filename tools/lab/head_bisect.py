"""Lab: values and gradients at every layer of the proposal head inside the full model, CPU checker vs HIP."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_engine_gpu as T
from oracle.attention_ref import OracleBackend
from spacap3d_amd import backend, synthetic as S
from spacap3d_amd.loss_helper import get_scene_cap_loss
data = T._anchored_batch()
out = {}
for name, be, dev in (("cpu", OracleBackend(), "cpu"), ("hip", backend.HipBackend(), "cuda:0")):
    with backend.use_backend(be):
        model = T._fresh_model(dev)
        acts = {}
        def hook(i):
            def f(mod, inp, o):
                o.retain_grad(); acts[i] = o
            return f
        hs = [l.register_forward_hook(hook(i)) for i, l in enumerate(model.proposal.proposal)]
        def sa_hook(m, i, o):
            o[1].retain_grad(); acts["sa"] = o[1]
        h0 = model.proposal.vote_aggregation.register_forward_hook(sa_hook)
        d = model({k: v.to(dev) for k, v in data.items()})
        d = get_scene_cap_loss(d, use_relation=True, mean_size_arr=S.mean_size_arr().numpy())
        d["det_loss"].backward()
        out[name] = {k: (v.detach().cpu().double(), None if v.grad is None else v.grad.detach().cpu().double()) for k, v in acts.items()}
        bn = model.proposal.proposal[4]
        out[name]["bn4"] = (bn.running_mean.detach().cpu().double(), bn.running_var.detach().cpu().double())
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
for k in ("sa", 1, 2, 4, 5):
    (vc, gc), (vh, gh) = out["cpu"][k], out["hip"][k]
    msg = f"layer {k}: value {rel(vh, vc):.2e}"
    if gc is not None and gh is not None:
        msg += f"  grad {rel(gh, gc):.2e} |g| {float(gc.norm()):.3e}"
        e = (gh - gc)
        if e.dim() == 3:
            per_ch = e.pow(2).sum((0, 2)).sqrt() / (gc.pow(2).sum((0, 2)).sqrt() + 1e-30)
            top = torch.topk(per_ch, 4)
            msg += "  worst channels " + ", ".join(f"{int(i)}:{float(v):.1e}" for v, i in zip(top.values, top.indices))
    else:
        msg += f"  grad cpu {gc is not None} hip {gh is not None}"
    print(msg)
for k in (1, 4):
    v = out["cpu"][k][0]
    print(f"BN{k} output: per-channel std min {float(v.std((0, 2)).min()):.3e}; |y| < 1e-4 count {int((v.abs() < 1e-4).sum())}, < 1e-3 count {int((v.abs() < 1e-3).sum())} of {v.numel()}")
    mc, mh = out["cpu"][k][0] > 0, out["hip"][k][0] > 0
    print(f"    ReLU masks differing between cpu and hip: {int((mc != mh).sum())}")
sa = out["cpu"]["sa"][0]
print("sa out per-channel std: min", float(sa.std((0, 2)).min()), "median", float(sa.std((0, 2)).median()), "frac zero", float((sa == 0).double().mean()))
