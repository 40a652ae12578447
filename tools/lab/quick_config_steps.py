"""Per-step device time of a configuration's pipelined step right after its graph was captured (events on the step's stream,
no host sync between steps): does a sub-record of the bench line measure a transient?
    python tools/lab/quick_config_steps.py cfg3 cfg4 cfg3 cfg4"""
import os
import sys
sys.path.insert(0, "/root/repo")
os.environ.setdefault("WORLD_SIZE", "1")
import gc
import torch
import bench as B
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default

dev = torch.device("cuda", 0)
N = int(os.environ.get("STEPS", "30"))
for name in sys.argv[1:]:
    cfg = B.CFG[name]
    torch.manual_seed(0)
    model = build_default(input_feature_dim=S.num_extra_channels(**cfg["feats"]), num_proposal=cfg["proposals"], **cfg["transformer"]).to(dev).train()
    tr = Trainer(model, S.mean_size_arr().numpy())
    data = synthetic_batch(cfg["batch"], cfg["n_points"], dev, seed=1000, **cfg["feats"])
    tr.step(data, next_data=data)
    tr.enable_graph(data)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    sv = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        tr.step(data, next_data=data)
        ev[i + 1].record()
        sv[i + 1].record(tr.side_stream)     # end of the pyramid launched beside step i
    torch.cuda.synchronize()
    print(name, " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.2f}" for i in range(N)), flush=True)
    # pyramid end relative to the end of the step it ran beside (negative: finished earlier) and relative to that step's start
    print("   side end - step end:", " ".join(f"{ev[i + 1].elapsed_time(sv[i + 1]):+.2f}" for i in range(1, N)), flush=True)
    print("   side end - step start:", " ".join(f"{ev[i].elapsed_time(sv[i + 1]):.2f}" for i in range(1, N)), flush=True)
    del tr, model, data
    gc.collect()
    torch.cuda.empty_cache()
