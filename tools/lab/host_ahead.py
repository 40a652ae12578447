"""Lab: does the host run ahead of the GPU in the replayed step loop?  Host time per trainer.step() call (no sync) against wall time per
step, pipelined and with the pyramid re-attached; and the host cost of each piece of the pipelined call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_default(input_feature_dim=1, num_proposal=256).to(dev).train()
trainer = Trainer(model, S.mean_size_arr().numpy(), use_relation=True)
data = synthetic_batch(8, 40000, dev, seed=1000)
trainer.step(data, next_data=data)
assert trainer.enable_graph(data), trainer.graph_error
def loop(fn, n=40):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter(); fn(); host.append(time.perf_counter() - a)
    t_host_done = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    host.sort()
    return wall, sum(host) / n * 1e3, host[n // 2] * 1e3, host[-1] * 1e3, t_host_done / n * 1e3
w = loop(lambda: trainer.step(data, next_data=data))
print("pipelined       : wall %.3f ms/step | host per call mean %.3f median %.3f max %.3f | host loop done after %.3f ms/step" % w, flush=True)
trainer.prefetch(data); torch.cuda.synchronize()
saved = data["_fps_prefetch"]
def reuse():
    data["_fps_prefetch"] = saved
    trainer.step(data, next_data=None)
w = loop(reuse)
print("pyramid reused  : wall %.3f ms/step | host per call mean %.3f median %.3f max %.3f | host loop done after %.3f ms/step" % w, flush=True)
w = loop(lambda: trainer.graph.replay())
print("replay only     : wall %.3f ms/step | host per call mean %.3f median %.3f max %.3f | host loop done after %.3f ms/step" % w, flush=True)
g = trainer._prefetch_graph_obj
side = trainer.side_stream
def two():
    with torch.cuda.stream(side): g.replay()
    trainer.graph.replay()
w = loop(two)
print("side graph + main graph replays, no events between: wall %.3f | host mean %.3f median %.3f max %.3f | host done %.3f" % w, flush=True)
def side_only():
    with torch.cuda.stream(side): g.replay()
w = loop(side_only)
print("side graph only : wall %.3f | host mean %.3f median %.3f max %.3f | host done %.3f" % w, flush=True)
