"""Greedy-decoding (eval) timing: KV-cached vs reference-style prefix recomputation.  python tools/eval_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import synthetic_batch
from spacap3d_amd.spacapnet import build_default
torch.manual_seed(0)
dev = "cuda:0"
model = build_default().to(dev).eval()
data = synthetic_batch(8, 40000, dev, seed=0)
with torch.no_grad():
    for use_cache in (True, False, True):
        d = model.backbone_net(dict(data))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d = model(dict(data), is_eval=True) if use_cache else None
        if not use_cache:
            d = model.backbone_net(dict(data))
            xyz, f = d["fp2_xyz"], d["fp2_features"]; d["seed_inds"] = d["fp2_inds"]; d["seed_xyz"] = xyz; d["seed_features"] = f
            xyz, f = model.vgen(xyz, f); f = f.div(torch.norm(f, p=2, dim=1).unsqueeze(1)); d["vote_xyz"] = xyz; d["vote_features"] = f
            d = model.proposal(xyz, f, d)
            d = model.caption.forward_eval(d, use_cache=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"eval forward B=8 (2048 captions x 31 steps) use_cache={use_cache}: {dt*1e3:.1f} ms  caps[0,0,:6]={d['lang_cap'][0,0,:6].tolist()}")
