"""Lab: stream priorities of the step's replay stream and of the side stream (PRIO=main,side e.g. -1,0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spacap3d_amd import engine, synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
dev = torch.device("cuda:0")
pm, ps = [int(v) for v in os.environ.get("PRIO", "0,0").split(",")]
torch.zeros(1, device=dev)
engine._STREAMS[("cuda", 0)] = {"side": torch.cuda.Stream(device=dev, priority=ps), "capture": torch.cuda.Stream(device=dev, priority=pm), "wgrad": torch.cuda.Stream(device=dev), "relation": torch.cuda.Stream(device=dev),
                                "comm": torch.cuda.Stream(device=dev)}
torch.manual_seed(0)
model = build_default(input_feature_dim=1, num_proposal=256).to(dev).train()
trainer = Trainer(model, S.mean_size_arr().numpy(), use_relation=True)
if os.environ.get("SKEW"): trainer.prefetch_skew_us = int(os.environ["SKEW"])
if os.environ.get("RESERVE"):   # a preset side stream keeps prefetch() from setting the reservation itself
    from spacap3d_amd._native import check, lib
    trainer.side_stream = engine._role_stream(dev, "side")
    check(lib.spacap_sa_reserve_cus(int(os.environ["RESERVE"])), "reserve")
data = synthetic_batch(8, 40000, dev, seed=1000)
trainer.step(data, next_data=data)
assert trainer.enable_graph(data), trainer.graph_error
def timed(fn, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
timed(lambda: trainer.step(data, next_data=data), 20)
ts = sorted(timed(lambda: trainer.step(data, next_data=data)) for _ in range(3))
trainer.prefetch(data); torch.cuda.synchronize()
saved = data["_fps_prefetch"]
def reuse():
    data["_fps_prefetch"] = saved
    trainer.step(data, next_data=None)
timed(reuse, 10)
print(f"PRIO main={pm} side={ps} SKEW={os.environ.get('SKEW')} RESERVE={os.environ.get('RESERVE')}: pipelined {ts[0]:.3f} {ts[1]:.3f} {ts[2]:.3f} | no side work {timed(reuse):.3f}", flush=True)
