import os, sys, traceback, collections
sys.path.insert(0, "/root/repo")
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from spacap3d_amd.engine import synthetic_batch
from spacap3d_amd.spacapnet import build_default
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = build_default().to(dev).eval()
data = synthetic_batch(2, 8192, dev, seed=0)
agg = collections.OrderedDict()
class T(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(k in name for k in ("mm", "conv", "linear", "matmul", "bmm", "einsum")):
            fr = [f for f in traceback.extract_stack() if "spacap3d_amd" in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "?"
            ts = [a for a in args if isinstance(a, torch.Tensor)]
            agg.setdefault((name, where), [0, "x".join(str(tuple(t.shape)) for t in ts[:3])])[0] += 1
        return out
with torch.no_grad():
    model(dict(data), is_eval=True)
    with T():
        model(dict(data), is_eval=True)
for (n, w), (c, shp) in agg.items():
    print(c, n, w, shp)
