"""Quantifies the FMA-contraction caveat of the native-op parity claim.

The reference's setup.py compiles its .cu files with nvcc -O3 and nvcc's default --fmad=true, which turns the
`a*a + b*b + c*c` distance sums (sampling_gpu.cu:100-104, ball_query_gpu.cu:31-32, interpolate_gpu.cu:31-32) into
mul, fma, fma (three roundings instead of five).  This repo's canonical arithmetic (oracle and HIP kernels) is the
un-contracted source semantics.  This script runs BOTH arithmetics of the CPU oracle (liboracle_omp.so vs
liboracle_fma.so) on the cfg2 synthetic scenes and counts how many distance VALUES and how many index DECISIONS differ.

    python tools/fma_caveat.py [--scenes 8] [--points 40000] > profiles/r02_fma_caveat.json
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ext_cpu import OracleExt  # noqa: E402
from spacap3d_amd import synthetic as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=8)
    ap.add_argument("--points", type=int, default=40000)
    ap.add_argument("--seeds", type=int, nargs="+", default=[1000, 1001, 1002, 1003])
    a = ap.parse_args()
    canon, fma = OracleExt(openmp=True), OracleExt(fma=True)
    out = {"workload": f"{len(a.seeds)} batches x {a.scenes} synthetic scenes x {a.points} points (bench.py seeds)",
           "arithmetic": {"canonical": "((a*a + b*b) + c*c), five roundings (-ffp-contract=off)",
                          "fma": "fma(c,c, fma(b,b, a*a)), three roundings (nvcc --fmad=true default)"},
           "levels": []}
    tot = {}
    for seed in a.seeds:
        xyz = S.scene_batch(a.scenes, a.points, use_height=False, seed=seed)
        cur = xyz
        for lvl, (m, r, ns) in enumerate(((2048, 0.2, 64), (1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16))):
            ia, i_f = canon.furthest_point_sampling(cur, m), fma.furthest_point_sampling(cur, m)
            nx = torch.gather(cur, 1, ia.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
            qa, qf = canon.ball_query(nx, cur, r, ns), fma.ball_query(nx, cur, r, ns)
            t = tot.setdefault(lvl, dict(level=f"SA{lvl + 1}", fps_indices=0, fps_indices_differing=0, fps_scenes_differing=0,
                                         ball_query_rows=0, ball_query_rows_differing=0))
            t["fps_indices"] += ia.numel()
            t["fps_indices_differing"] += int((ia != i_f).sum())
            t["fps_scenes_differing"] += int((ia != i_f).any(1).sum())
            t["ball_query_rows"] += qa.shape[0] * qa.shape[1]
            t["ball_query_rows_differing"] += int((qa != qf).any(-1).sum())
            cur = nx
        u, k = xyz[:, :4096].contiguous(), xyz[:, 4096:8192].contiguous()
        (da, ja), (df, jf) = canon.three_nn(u, k), fma.three_nn(u, k)
        t = tot.setdefault("nn", dict(level="three_nn 4096 x 4096", dist2_values=0, dist2_values_differing=0, idx_differing=0))
        t["dist2_values"] += da.numel()
        t["dist2_values_differing"] += int((da != df).sum())
        t["idx_differing"] += int((ja != jf).sum())
    out["levels"] = list(tot.values())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
