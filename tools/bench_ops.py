"""Per-operator timing of the HIP kernels at the SA1..SA4 / FP shapes (cfg2: B=8, N=40000).
Usage: python tools/bench_ops.py [B] [N]
"""
import sys
import time

import torch

import spacap3d_amd.ext as ext
from spacap3d_amd import synthetic as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
dev = "cuda:0"


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


xyz = S.scene_batch(B, N, use_height=False, seed=0).to(dev)
levels = [(N, 2048, 0.2, 64), (2048, 1024, 0.4, 32), (1024, 512, 0.8, 16), (512, 256, 1.2, 16), (1024, 256, 0.3, 16)]
cur = xyz
for (n, m, r, ns) in levels:
    if cur.shape[1] != n:
        cur = xyz[:, :n].contiguous()
    t = timeit(lambda: ext.furthest_point_sampling(cur, m))
    inds = ext.furthest_point_sampling(cur, m)
    new_xyz = torch.gather(cur, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    print(f"fps        B={B} {n:6d}->{m:5d}: {t:8.3f} ms  {B*n/t*1e-3:9.2f} Mpts/s  "
          f"{B*n*(m-1)/t*1e-6:9.2f} Gupd/s  {t/(m-1)*1e3:6.3f} us/round")
    t = timeit(lambda: ext.ball_query(new_xyz, cur, r, ns))
    print(f"ball_query B={B} {m:5d}x{n:6d} r={r} ns={ns}: {t:8.3f} ms  {B*n/t*1e-3:9.2f} Mpts/s  "
          f"{B*m*n/t*1e-6:9.2f} Gpair/s")
    idx = ext.ball_query(new_xyz, cur, r, ns)
    for C in (3, 128):
        f = torch.randn(B, C, n, device=dev)
        t = timeit(lambda: ext.group_points(f, idx))
        by = B * m * ns * (4 + 8 * C)
        print(f"group      C={C:3d} -> (B,{C},{m},{ns}): {t:8.3f} ms  {by/t*1e-6:8.1f} GB/s")
        go = torch.randn(B, C, m, ns, device=dev)
        t = timeit(lambda: ext.group_points_grad(go, idx, n))
        print(f"group_grad C={C:3d}: {t:8.3f} ms  {by/t*1e-6:8.1f} GB/s")
    cur = new_xyz

unk = xyz[:, :1024].contiguous(); kn = xyz[:, :512].contiguous()
t = timeit(lambda: ext.three_nn(unk, kn))
print(f"three_nn 1024x512: {t:8.3f} ms")
d, i = ext.three_nn(unk, kn)
w = torch.rand(B, 1024, 3, device=dev)
f = torch.randn(B, 256, 512, device=dev)
t = timeit(lambda: ext.three_interpolate(f, i, w))
print(f"three_interpolate (B,256,512)->(B,256,1024): {t:8.3f} ms")
go = torch.randn(B, 256, 1024, device=dev)
t = timeit(lambda: ext.three_interpolate_grad(go, i, w, 512))
print(f"three_interpolate_grad: {t:8.3f} ms")
