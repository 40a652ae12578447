"""Lab: what each flush of the deferred weight-gradient queue carries in one eager training step (jobs = Linear weight gradients,
conv = 1x1-convolution weight gradients, sums = slab sums), in backward order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spacap3d_amd import _native, synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_default().to(dev).train()
tr = Trainer(model, S.mean_size_arr().numpy())
data = synthetic_batch(8, 40000, dev, seed=1000)
tr.step(data, next_data=data)
orig = _native.deferred_slab_sums.flush
def flush(self):
    conv = [tuple(j[2]) for j in self.conv_jobs]
    print(f"flush: {len(self.jobs)} linear jobs, {len(conv)} conv jobs {conv}, {len(self.items)} sums "
          f"({sum(p.numel() for p, _ in self.items) * 4 / 1e6:.1f} MB of partials) on stream {torch.cuda.current_stream().cuda_stream:#x}", flush=True)
    return orig(self)
_native.deferred_slab_sums.flush = flush
tr.step(data, next_data=data)
torch.cuda.synchronize()
