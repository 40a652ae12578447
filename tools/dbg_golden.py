import os, sys, numpy as np, torch
sys.path.insert(0, 'tests/golden'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_golden as T
from spacap3d_amd import backend
from spacap3d_amd.loss_helper import get_scene_cap_loss
be, device = T._backend("hip")
fx = np.load("tests/golden/train_step_cfg1.npz")
with backend.use_backend(be):
    model = T._build(fx, device).train()
    d = model(T._inputs(fx, device))
    d = get_scene_cap_loss(d, use_relation=True, mean_size_arr=fx["mean_size_arr"])
    d["loss"].backward()
for k in fx.files:
    if k.startswith("out_"):
        name = k[4:]; flat = name.endswith("__flat7")
        if flat: name = name[:-7]
        got = d[name].detach().cpu().numpy()
        if flat: got = got.reshape(got.shape[0], -1)[:, ::7]
        want = fx[k]
        if got.dtype.kind in "iub": print(f"{name:28s} int equal={np.array_equal(got, want)}"); continue
        err = np.abs(got - want).max(); sc = np.abs(want).max()
        print(f"{name:28s} maxabs={err:.3e} scale={sc:.3e} rel_to_scale={err/sc:.2e}")
    if k.startswith("loss_"):
        print(f"{k:28s} got={float(d[k[5:]]):.6f} want={float(fx[k]):.6f}")
params = dict(model.named_parameters())
for k in fx.files:
    if k.startswith("grad_") and k != "grad_absent":
        g = params[k[5:]].grad.detach().cpu().numpy().reshape(-1); g = g[::3] if g.size > 4096 else g
        print(f"{k[:60]:60s} maxabs={np.abs(g-fx[k]).max():.3e} scale={np.abs(fx[k]).max():.3e}")
