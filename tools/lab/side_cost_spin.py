"""Lab: per-kernel cost of a RESIDENT neighbour (8 sleeping single-wave workgroups on another queue for the whole step) -- run under
rocprofv3 --kernel-trace, analysed by side_cost_diff.py with MARK=spin_kernel."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
probe = ctypes.CDLL(os.path.join(ROOT, "tools", "lab", "libcumask_probe.so"))
probe.probe_spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda:0")
torch.manual_seed(0)
N = int(os.environ.get("N", "12")); US = int(os.environ.get("US", "14000")); G = int(os.environ.get("G", "8")); T = int(os.environ.get("T", "64"))
model = build_default(input_feature_dim=1, num_proposal=256).to(dev).train()
trainer = Trainer(model, S.mean_size_arr().numpy(), use_relation=True)
data = synthetic_batch(8, 40000, dev, seed=1000)
trainer.step(data, next_data=data)
assert trainer.enable_graph(data), trainer.graph_error
trainer.prefetch(data); torch.cuda.synchronize()
saved = data["_fps_prefetch"]
side = torch.cuda.Stream(device=dev)
def step(spin):
    data["_fps_prefetch"] = saved
    if spin:
        torch.cuda.synchronize()
        probe.probe_spin(ctypes.c_void_p(side.cuda_stream), G, T, 0, US)
    trainer.step(data, next_data=None)
    if spin:
        torch.cuda.synchronize()
for _ in range(5): step(False)
torch.cuda.synchronize()
for _ in range(N): step(True)
torch.cuda.synchronize()
for _ in range(N): step(False)
torch.cuda.synchronize()
