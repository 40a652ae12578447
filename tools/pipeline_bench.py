"""Throughput of the device input pipeline (spacap3d_amd/dataset.py) at the cfg2 shape: 8 descriptions per batch,
40 000 points each, synthetic scenes of 150 000 vertices / 40 instances in HBM; beside it the numpy restatement of
the reference's per-item path on one host core (what a DataLoader worker does per item)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from spacap3d_amd import synthetic as S
from spacap3d_amd.dataset import DeviceSceneDataset
from oracle import scene_pipeline_ref as R

NYU = [3, 4, 5, 7, 8, 12, 14, 24, 33, 39]


def scene(rng, n, m):
    xyz = np.concatenate([rng.uniform(-4, 4, (n, 2)), rng.uniform(0, 3, (n, 1))], 1)
    ins = rng.integers(0, m + 1, n)
    sem = np.where(ins == 0, 1, np.array(NYU)[(ins - 1) % len(NYU)])
    box = np.concatenate([rng.uniform(-3, 3, (m, 3)), rng.uniform(0.3, 1.5, (m, 3)),
                          np.array(NYU)[np.arange(m) % len(NYU)][:, None], np.arange(1, m + 1)[:, None]], 1)
    vert = np.concatenate([xyz, rng.uniform(0, 255, (n, 3)), rng.normal(size=(n, 3))], 1).astype(np.float32)
    rel = [rng.integers(0, 3, (m, m)) for _ in range(3)]
    return vert, ins, sem, box, rel


def main():
    rng = np.random.default_rng(0)
    msa = S.mean_size_arr().numpy()
    n2c = {k: i % 18 for i, k in enumerate(R.NYU40IDS.tolist())}
    ds = DeviceSceneDataset("cuda:0", msa, n2c, num_points=40000)
    ref = R.SceneStoreRef(msa, n2c, {})
    for s in range(16):
        v, i, se, b, rel = scene(rng, 150000, 40)
        ds.add_scene(f"s{s}", v, i, se, b, *rel)
        ref.add_scene(f"s{s}", v, i, se, b, *rel)
        ds.add_item(f"s{s}", 1 + s % 40)
    g = torch.Generator(device="cuda:0").manual_seed(0)
    idx = list(range(8))
    for _ in range(3):
        ds.batch(idx, ds.draw(idx, generator=g))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for k in range(n):
        idx = [(k * 8 + j) % 16 for j in range(8)]
        d = ds.batch(idx, ds.draw(idx, generator=g))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    draws = ds.draw(idx, generator=g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ds.batch(idx, draws)
    e1.record()
    torch.cuda.synchronize()
    print(f"device pipeline: {dt * 1e3:.2f} ms per batch of 8 incl. host draws ({8 / dt:.0f} scenes/s); "
          f"{e0.elapsed_time(e1) / n:.2f} ms per batch with given draws")
    t0 = time.perf_counter()
    for j in range(4):
        dr = R.draws_from_seed(j, 150000, 40000)
        ref.get_item(f"s{j}", 1, "", dr, 40000)
    dt_ref = (time.perf_counter() - t0) / 4
    print(f"numpy restatement of the reference item path: {dt_ref * 1e3:.1f} ms per item on one core ({1 / dt_ref:.1f} scenes/s/core)")


if __name__ == "__main__":
    main()
