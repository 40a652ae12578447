"""Lab: prints the error of every gradient of the cfg1 golden training step on the HIP leg (to set test tolerances)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from spacap3d_amd import backend
from spacap3d_amd.loss_helper import get_scene_cap_loss
import test_golden as T
fx = np.load(os.path.join(ROOT, "tests", "golden", "train_step_cfg1.npz"))
with backend.use_backend(backend.HipBackend()):
    model = T._build(fx, "cuda:0").train()
    d = model(T._inputs(fx, "cuda:0"))
    d = get_scene_cap_loss(d, use_relation=True, mean_size_arr=fx["mean_size_arr"])
    d["loss"].backward()
params = dict(model.named_parameters())
for k in fx.files:
    if k.startswith("grad_") and k != "grad_absent":
        g = params[k[5:]].grad.detach().cpu().numpy().reshape(-1)
        g = g[::3] if g.size > 4096 else g
        w = fx[k]
        print(f"{k:70s} linf {np.abs(g-w).max()/np.abs(w).max():.2e}  l2 {np.linalg.norm(g-w)/np.linalg.norm(w):.2e}")
for k in fx.files:
    if k.startswith("out_"):
        name = k[4:]
        flat = name.endswith("__flat7")
        name = name[:-7] if flat else name
        got = d[name].detach().cpu().numpy()
        got = got.reshape(got.shape[0], -1)[:, ::7] if flat else got
        if got.dtype.kind == "f":
            w = fx[k]
            print(f"{k:70s} linf {np.abs(got-w).max()/np.abs(w).max():.2e}  l2 {np.linalg.norm(got-w)/np.linalg.norm(w):.2e}")
