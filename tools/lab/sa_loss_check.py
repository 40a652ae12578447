import torch, sys
import spacap3d_amd
from spacap3d_amd import backend
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
from spacap3d_amd import synthetic as S
dev = torch.device("cuda:0")
def run(fused, steps=12):
    torch.manual_seed(0)
    model = build_default().to(dev)
    tr = Trainer(model, S.mean_size_arr().numpy())
    hip = backend.ops()
    saved = hip.sa_mlp_train
    if not fused: hip.sa_mlp_train = None
    out = []
    try:
        for it in range(steps):
            batch = synthetic_batch(8, 40000, dev, seed=it % 3)
            out.append(float(tr.step(batch)))
    finally:
        hip.sa_mlp_train = saved
    return out
a = run(True); b = run(False); c = run(False)
for i,(x,y,z) in enumerate(zip(a,b,c)): print(i, f"fused {x:.4f}  unfused {y:.4f}  unfused-again {z:.4f}")
