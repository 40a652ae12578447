"""Lab: which gradients are not finite after a replayed step at 2 scenes per GPU (relation head forked)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(os.environ.get("B", "2"))
model = build_default(input_feature_dim=1, num_proposal=256).to(dev).train()
tr = Trainer(model, S.mean_size_arr().numpy(), lr=float(os.environ.get("LR", "0")))
data = synthetic_batch(B, 40000, dev, seed=1000)
tr.step(data, next_data=data)
def report(tag):
    torch.cuda.synchronize()
    bad = []
    for (n, p), v in zip([(n, p) for n, p in model.named_parameters() if any(p is q for q in tr.bucket.params)], []):
        pass
    names = {id(p): n for n, p in model.named_parameters()}
    for p, off in zip(tr.bucket.params, tr.bucket.offsets):
        g = tr.bucket.flat[off:off + p.numel()]
        if not torch.isfinite(g).all():
            bad.append((names.get(id(p), "?"), int((~torch.isfinite(g)).sum())))
    print(tag, "loss", {k: float(v) for k, v in tr.last_losses.items() if k in ("loss", "relation_loss", "cap_loss")}, "non-finite grads:", bad[:12], len(bad), flush=True)
report("eager")
assert tr.enable_graph(data, warmup=1), tr.graph_error
report("after capture warm-up")
for i in range(3):
    tr.step(data, next_data=data)
    report(f"replay {i}")
