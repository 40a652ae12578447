"""Lab: which Python lines of spacap3d_amd call the torch (aten) operators that launch kernels in one eager training step.
    python tools/lab/glue_stacks.py        (on the GPU box)"""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer
from spacap3d_amd.spacapnet import build_default

dev = torch.device("cuda", 0)
cfg = B.CFG[os.environ.get("CFG", "cfg2")]
SMALL = os.environ.get("SMALL") == "1"     # 2 scenes x 8 192 points: call sites only
ONLY = os.environ.get("ONLY", "")          # e.g. ONLY=mm,conv: only operators whose name contains one of these
torch.manual_seed(0)
model = build_default(input_feature_dim=S.num_extra_channels(**cfg["feats"]), num_proposal=cfg["proposals"], **cfg["transformer"]).to(dev).train()
tr = Trainer(model, S.mean_size_arr().numpy())
data = B.synthetic_batch(2 if SMALL else cfg["batch"], 8192 if SMALL else cfg["n_points"], dev, seed=1000, **cfg["feats"])
for _ in range(3):
    tr.step(data, next_data=data)
torch.cuda.synchronize()
NOKERNEL = ("view", "reshape", "expand", "transpose", "permute", "select", "slice", "unsqueeze", "squeeze", "detach", "alias", "as_strided",
            "empty", "aten.t.default", "unbind", "split", "_unsafe_view", "size", "stride", "is_", "sym_", "_local_scalar", "lift", "_to_copy_noop",
            "new_empty", "empty_like", "record_stream", "unfold", "chunk", "narrow", "diagonal", "movedim", "flatten", "unflatten")
agg = collections.OrderedDict()


class Tracer(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(k in name for k in NOKERNEL) and (not ONLY or any(k in name for k in ONLY.split(","))):
            ts = [a for a in list(args) + [out] if isinstance(a, torch.Tensor)]
            if any(t.is_cuda for t in ts):
                fr = [f for f in traceback.extract_stack() if "spacap3d_amd" in f.filename]
                where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "(no python frame: autograd)"
                if not fr:
                    node = torch._C._current_autograd_node()
                    where += " in/after " + (node.name() if node is not None else "-")
                shp = "x".join(str(tuple(t.shape)) for t in ts[:2])
                a = agg.setdefault((name, where), [0, shp])
                a[0] += 1
        return out


with Tracer():
    tr.step(data, next_data=data)
torch.cuda.synchronize()
for (name, where), (n, shp) in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{n:3d}x {name:30s} {where:84s} {shp[:60]}")
