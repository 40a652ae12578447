"""Lab: how much of an inference forward (eval, greedy decoding of B*K captions) is the decode loop."""
import time, torch, sys
sys.path.insert(0, "/root/repo")
import bench as B
from spacap3d_amd import synthetic as S, tf_layer
from spacap3d_amd.spacapnet import build_default
dev = torch.device("cuda", 0)
cfg = B.CFG["cfg2"]
torch.manual_seed(0)
model = build_default(input_feature_dim=S.num_extra_channels(**cfg["feats"]), num_proposal=cfg["proposals"], **cfg["transformer"]).to(dev).eval()
data = B.synthetic_batch(cfg["batch"], cfg["n_points"], dev, seed=1000, **cfg["feats"])
orig = tf_layer.greedy_decode
times = []
def timed(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(*a, **k)
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    return r
tf_layer.greedy_decode = timed
for g in (False,):
    times.clear()
    with torch.no_grad():
        for _ in range(4):
            d = {k: v for k, v in data.items() if k != "_fps_prefetch"}
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model(d, is_eval=True)
            torch.cuda.synchronize(); tot = (time.perf_counter() - t0) * 1e3
    print("graph" if g else "direct", "decode ms:", [round(t, 2) for t in times], "whole forward ms (last):", round(tot, 2))
