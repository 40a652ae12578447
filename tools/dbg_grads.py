import os, sys, numpy as np, torch
sys.path.insert(0, 'tests/golden'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_golden as T
from spacap3d_amd import backend
from spacap3d_amd.loss_helper import get_scene_cap_loss
fx = np.load("tests/golden/train_step_cfg1.npz")
res = {}
for kind in ("oracle", "hip"):
    be, device = T._backend(kind)
    with backend.use_backend(be):
        model = T._build(fx, device).train()
        d = model(T._inputs(fx, device))
        d = get_scene_cap_loss(d, use_relation=True, mean_size_arr=fx["mean_size_arr"])
        keep = {}
        for k in ("sa1_features", "sa2_features", "sa3_features", "sa4_features", "fp2_features", "seed_features", "vote_features", "aggregated_vote_features"):
            d[k].retain_grad(); keep[k] = d[k]
        d["loss"].backward()
    res[kind] = ({n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None},
                 {k: v.grad.detach().cpu() for k, v in keep.items()})
ga, gb = res["oracle"][0], res["hip"][0]
for n in ga:
    e = float((ga[n]-gb[n]).abs().max()); s = float(ga[n].abs().max()) + 1e-12
    if 'caption' in n and e/s < 1e-3: continue
    print(f"{n:70s} rel={e/s:.2e} scale={s:.2e}")
print('--- activation grads')
for k in res["oracle"][1]:
    a, b = res["oracle"][1][k], res["hip"][1][k]
    print(f"{k:30s} rel={float((a-b).abs().max())/float(a.abs().max()):.2e}")
