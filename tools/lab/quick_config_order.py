import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
os.environ.setdefault("WORLD_SIZE", "1")
import torch, bench as B
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
dev = torch.device("cuda", 0)
which = sys.argv[1]
if which in ("eval", "train+eval", "train"):
    torch.manual_seed(0)
    model = build_default().to(dev).train()
    data = synthetic_batch(8, 40000, dev, seed=1000)
    if "train" in which:
        tr = Trainer(model, S.mean_size_arr().numpy())
        tr.step(data, next_data=data); tr.enable_graph(data)
        for _ in range(5): tr.step(data, next_data=data)
        torch.cuda.synchronize()
    if "eval" in which:
        print("eval", B.eval_record(model, data)["ms_per_forward"])
    del model, data
    torch.cuda.empty_cache()
for name in sys.argv[2:]:
    print(name, B.quick_config(name, 0, dev)["ms_per_step"])
