"""Lab: the training step with the geometry pyramid prefetched ONCE and re-attached every step (no sampling work on the side
stream at all) against the normal pipelined step: how much do the side-stream kernels cost the main stream?"""
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
assert trainer.enable_graph(data)
def timed(fn, n=40, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("pipelined step (pyramid of the next batch on the side stream): %.3f ms" % timed(lambda: trainer.step(data, next_data=data)), flush=True)
trainer.prefetch(data); torch.cuda.synchronize()
saved = data["_fps_prefetch"]
def reuse():
    data["_fps_prefetch"] = saved
    trainer.step(data, next_data=None)
print("same step, pyramid computed once and re-attached (no side-stream work): %.3f ms" % timed(reuse), flush=True)
def replay_only():
    trainer.graph.replay()
print("graph replay alone (no copies into the static buffers): %.3f ms" % timed(replay_only), flush=True)

# -- the same side-stream work as ONE graph launch instead of ~60 eager launches per step
from spacap3d_amd.detector import geometry_pyramid, sampling_pyramid
side = torch.cuda.Stream(device=dev)
pc = data["point_clouds"][..., :3].contiguous()
with torch.cuda.stream(side), torch.no_grad():
    for _ in range(2):
        geometry_pyramid(pc)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g, stream=side):
    pyr_static = geometry_pyramid(pc)
torch.cuda.synchronize()
def side_graph():
    cur = torch.cuda.current_stream(dev)
    data["_fps_prefetch"] = saved
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        g.replay()
    trainer.step(data, next_data=None)
print("step + the pyramid as one side-stream graph launch: %.3f ms" % timed(side_graph), flush=True)
g2 = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g2, stream=side):
    pyr2 = sampling_pyramid(pc)
torch.cuda.synchronize()
def side_graph2():
    cur = torch.cuda.current_stream(dev)
    data["_fps_prefetch"] = saved
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        g2.replay()
    trainer.step(data, next_data=None)
print("step + only the sampling chain (no groupings / neighbour searches) as a side-stream graph: %.3f ms" % timed(side_graph2), flush=True)
