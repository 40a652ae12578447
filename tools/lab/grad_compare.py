"""Lab: every parameter gradient (and the first optimizer update) of one training step, HIP product path vs the CPU checker
backend, same weights / batch / pinned proposal indices.  Prints the parameters sorted by relative l2 error."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_engine_gpu as T
from oracle.attention_ref import OracleBackend
from spacap3d_amd import backend, synthetic as S
from spacap3d_amd.engine import Trainer

data = T._anchored_batch()
res = {}
for name, be, dev in (("cpu", OracleBackend(), "cpu"), ("hip", backend.HipBackend(), "cuda:0")):
    with backend.use_backend(be):
        model = T._fresh_model(dev)
        tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-4, adam_eps=1e-3)
        d = {k: v.to(dev) for k, v in data.items()}
        p0 = {n: p.detach().clone().cpu() for n, p in model.named_parameters()}
        tr._setup(dict(d))
        tr._core(dict(d), with_optimizer=False)
        g = {n: p.grad.detach().clone().cpu() for n, p in model.named_parameters() if p.grad is not None}
        tr._optimizer_step(None)
        p1 = {n: p.detach().clone().cpu() for n, p in model.named_parameters()}
        l1 = float(tr._core(dict(d)))
        res[name] = (g, p0, p1, {k: float(v) for k, v in tr.last_losses.items()})
gc, gh = res["cpu"][0], res["hip"][0]
# the same gradients WITHOUT the Trainer (plain forward / loss / backward, as the golden tests do)
from spacap3d_amd.loss_helper import get_scene_cap_loss
direct = {}
for name, be, dev in (("cpu", OracleBackend(), "cpu"), ("hip", backend.HipBackend(), "cuda:0")):
    with backend.use_backend(be):
        model = T._fresh_model(dev)
        d = model({k: v.to(dev) for k, v in data.items()})
        d = get_scene_cap_loss(d, use_relation=True, mean_size_arr=S.mean_size_arr().numpy())
        d["loss"].backward()
        direct[name] = {n: p.grad.detach().clone().cpu() for n, p in model.named_parameters() if p.grad is not None}


def table(a_all, b_all, title):
    print("==", title)
    rows = []
    for n in a_all:
        a, b = a_all[n].double().flatten(), b_all[n].double().flatten()
        rows.append((float((a - b).norm() / (a.norm() + 1e-30)), float(b.norm() / (a.norm() + 1e-30)),
                     float((a @ b) / (a.norm() * b.norm() + 1e-30)), float(a.norm()), n))
    rows.sort(reverse=True)
    for e, ratio, cos, nrm, n in rows[:14] + rows[len(rows) // 2:len(rows) // 2 + 3]:
        print(f"{e:10.3e} ratio {ratio:8.5f} cos {cos:9.6f} |g| {nrm:10.3e}  {n}")
    by = {}
    for e, ratio, cos, nrm, n in rows:
        if nrm > 1e-4:
            by.setdefault(".".join(n.split(".")[:2]), []).append(e)
    for k, v in sorted(by.items()):
        print(f"   {k:40s} n {len(v):3d} max {max(v):.2e} median {sorted(v)[len(v) // 2]:.2e}")


table(gc, gh, "Trainer: cpu vs hip")
table(direct["cpu"], direct["hip"], "direct: cpu vs hip")
table(gc, direct["cpu"], "cpu: Trainer vs direct")
table(gh, direct["hip"], "hip: Trainer vs direct")
