"""Exhaustive vs cell-grid ball query at the SA1 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import spacap3d_amd  # noqa
from spacap3d_amd import ext, synthetic as S

dev = torch.device("cuda:0")
xyz = S.scene_batch(8, 40000, seed=0)[..., :3].contiguous().to(dev)
inds = ext.furthest_point_sampling(xyz, 2048).long()
new_xyz = torch.gather(xyz, 1, inds.unsqueeze(-1).expand(-1, -1, 3)).contiguous()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


ext.BALL_QUERY_GRID_MIN_N = 10 ** 9
a = ext.ball_query(new_xyz, xyz, 0.2, 64)
t0 = timeit(lambda: ext.ball_query(new_xyz, xyz, 0.2, 64))
ext.BALL_QUERY_GRID_MIN_N = 1
b = ext.ball_query(new_xyz, xyz, 0.2, 64)
t1 = timeit(lambda: ext.ball_query(new_xyz, xyz, 0.2, 64))
print(f"SA1 ball query: exhaustive {t0:.1f} us, cell grid {t1:.1f} us, equal {bool(torch.equal(a, b))}")
