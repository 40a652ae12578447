"""Lab: does torch.profiler lose kernels of a forked branch?  cfg5's model (wide relation head on the relation stream): 12 profiled
eager steps, per step which of the head's kernels the trace holds, and whether the head's weight gradient was produced."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
dev = torch.device("cuda:0")
torch.manual_seed(0)
# earlier tests of the suite leave other Trainers / streams behind: imitate with a cfg2 trainer first
if os.environ.get("WARM"):
    m0 = build_default().to(dev).train(); t0 = Trainer(m0, S.mean_size_arr().numpy()); d0 = synthetic_batch(2, 8192, dev, seed=1)
    for _ in range(3): t0.step(d0, next_data=d0)
model = build_default(vocab_size=3001, num_proposal=512, input_feature_dim=1, d_model=512, h=32).to(dev).train()
tr = Trainer(model, S.mean_size_arr().numpy())
data = synthetic_batch(2, 8192, dev, seed=1)
for _ in range(2):
    tr.step(data, next_data=data)
w = model.caption.relation_proposal[2].weight
for i in range(12):
    before = w.detach().clone()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        tr.step(data, next_data=data)
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    has = lambda s: any(s in n for n in names)
    print(i, len(names), "l1_fwd", has("rel_wide_l1_fwd_kernel"), "l1_bwd", has("rel_wide_l1_bwd_kernel"), "tail_bwd", has("rel_wide_tail_bwd"),
          "weight moved", float((w.detach() - before).abs().max()) > 0, flush=True)
