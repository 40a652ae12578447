"""Lab: a few eager training steps at 2 scenes -- workload for tools/lab/pmc_passes.sh on single small kernels of the step
(bash tools/lab/pmc_passes.sh <tag> tools/lab/det_loss_steps.py det_proposal)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_default().to(dev).train()
tr = Trainer(model, S.mean_size_arr().numpy())
data = synthetic_batch(8, 40000, dev, seed=1000)
for _ in range(4):
    tr.step(data, next_data=data)
torch.cuda.synchronize()
