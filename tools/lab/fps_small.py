"""Small-N furthest point sampling: per-round latency of the register kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import spacap3d_amd  # noqa
from spacap3d_amd import ext
dev = torch.device("cuda:0")
for N, m in ((512, 256), (1024, 256), (1025, 256), (2048, 1024)):
    xyz = torch.rand(8, N, 3, device=dev) * 5
    for _ in range(3):
        ext.furthest_point_sampling(xyz, m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20):
        ext.furthest_point_sampling(xyz, m)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    print(f"N={N} m={m}: {t:.1f} us = {t / (m - 1):.3f} us/round")
