"""Lab: which Python lines of the step launch the small torch kernels?  One eager training step under torch.profiler
with stacks; prints (aten op, innermost spacap3d_amd frame) -> number of device kernels and their summed time."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from torch.profiler import profile, ProfilerActivity
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_default().to(dev).train()
tr = Trainer(model, S.mean_size_arr().numpy())
data = synthetic_batch(8, 40000, dev, seed=1000)
for _ in range(3):
    tr.step(data, next_data=data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    tr.step(data, next_data=data)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0:
        continue
    # only leaf-ish ops: those that directly own kernels
    if not ev.kernels:
        continue
    frame = "?"
    for fr in ev.stack or []:
        if "spacap3d_amd" in fr or "bench.py" in fr:
            frame = fr.split("spacap3d_amd/")[-1] if "spacap3d_amd/" in fr else fr
            break
    if frame == "?" and ev.stack:
        frame = "| ".join(f.split("/")[-1] for f in ev.stack[:2])
    k = (ev.name, frame[:90])
    agg[k][0] += len(ev.kernels)
    agg[k][1] += sum(kk.duration for kk in ev.kernels)
rows = sorted(((v[1], v[0], k) for k, v in agg.items()), reverse=True)
print("total aten-launched kernels:", sum(r[1] for r in rows), "time us:", round(sum(r[0] for r in rows)))
for t, n, (op, fr) in rows[:90]:
    print(f"{t:8.1f} us {n:4d}  {op:32s} {fr}")
