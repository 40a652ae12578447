import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import test_engine_gpu as T
from spacap3d_amd import backend, synthetic as S
from spacap3d_amd.engine import Trainer
data = T._anchored_batch()
def run(device, be, steps=5):
    with backend.use_backend(be):
        model = T._fresh_model(device)
        tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-4, adam_eps=1e-3)
        d = {k: v.to(device) for k, v in data.items()}
        out = []
        for _ in range(steps):
            tr.step(d)
            out.append({k: float(v) for k, v in tr.last_losses.items()})
        return out
for rep in range(int(os.environ.get("REPS", "2"))):
    g = run("cuda:0", backend.HipBackend())
    print([round(x["cap_loss"], 6) for x in g], [round(x["loss"], 5) for x in g], flush=True)
