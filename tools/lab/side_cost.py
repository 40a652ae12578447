"""Lab: which main-stream kernels pay for the side-stream sampling chain?  Runs N pipelined steps (pyramid of the next batch on the
side stream) and N steps with the pyramid re-attached (no side-stream work) in one process; under
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/lab/side_cost.py
tools/lab/side_cost_diff.py <kernel_trace.csv> then compares every launch position of the step between the two phases."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
dev = torch.device("cuda:0")
torch.manual_seed(0)
N = int(os.environ.get("N", "12"))
model = build_default(input_feature_dim=1, num_proposal=256).to(dev).train()
trainer = Trainer(model, S.mean_size_arr().numpy(), use_relation=True)
data = synthetic_batch(8, 40000, dev, seed=1000)
trainer.step(data, next_data=data)
assert trainer.enable_graph(data), trainer.graph_error
def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
timed(lambda: trainer.step(data, next_data=data), 10)
print("pipelined: %.3f ms" % timed(lambda: trainer.step(data, next_data=data), N), flush=True)
trainer.prefetch(data); torch.cuda.synchronize()
saved = data["_fps_prefetch"]
def reuse():
    data["_fps_prefetch"] = saved
    trainer.step(data, next_data=None)
timed(reuse, 3)
print("no side-stream work: %.3f ms" % timed(reuse, N), flush=True)
