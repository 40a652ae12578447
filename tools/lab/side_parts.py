"""Lab: which part of the side-stream pyramid costs the step what?  The pyramid function is replaced by one that computes only PART
of the chain (the rest is re-attached from a pyramid computed once): PARTS = comma list of fps1,fps2,fps3,fps4,bq1,bq,nn,rows."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spacap3d_amd import engine, detector, synthetic as S
from spacap3d_amd import pointnet2_utils as pu
from spacap3d_amd.pointnet2_modules import PointnetFPModule
from spacap3d_amd.sa_mlp import rows_index
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_default(input_feature_dim=1, num_proposal=256).to(dev).train()
data = synthetic_batch(8, 40000, dev, seed=1000)
with torch.no_grad():
    FULL = detector.geometry_pyramid(data["point_clouds"][..., :3].contiguous())
torch.cuda.synchronize()
def make(parts):
    parts = set(parts)
    def pyr(xyz):
        # FULL layout: inds[0:4], idx[4:8], fp1 (8, 9), fp2 (10, 11), rows_index[12:15], xyzs[15:19]
        xyzs = [xyz] + list(FULL[15:19])
        for l, n in enumerate(detector.SA_NPOINTS):
            if f"fps{l + 1}" in parts:
                pu.furthest_point_sample(xyzs[l], n)
        for l, (r, ns) in enumerate(zip(detector.SA_RADII, detector.SA_NSAMPLES)):
            if (l == 0 and "bq1" in parts) or (l > 0 and "bq" in parts):
                pu.ball_query(r, ns, xyzs[l], xyzs[l + 1])
        if "nn" in parts:
            PointnetFPModule.neighbours(xyzs[3], xyzs[4]); PointnetFPModule.neighbours(xyzs[2], xyzs[3])
        if "rows" in parts:
            for l in (1, 2, 3): rows_index(FULL[4 + l], xyzs[l].shape[1])
        return tuple(t.clone() for t in FULL)
    return pyr
def run(parts):
    engine.geometry_pyramid = make(parts) if parts != ["full"] else detector.geometry_pyramid
    trainer = Trainer(model, S.mean_size_arr().numpy(), use_relation=True)
    data.pop("_fps_prefetch", None)
    trainer.step(data, next_data=data)
    assert trainer.enable_graph(data), trainer.graph_error
    def timed(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): trainer.step(data, next_data=data)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    timed(15)
    ts = sorted(timed(30) for _ in range(3))
    print(f"side chain = {','.join(parts):40s} {ts[0]:.3f} {ts[1]:.3f} {ts[2]:.3f} ms/step", flush=True)
for parts in os.environ.get("PARTS", "full;none;fps1;fps2,fps3,fps4;bq1,bq,nn,rows;fps1,fps2,fps3,fps4;full").split(";"):
    run([p for p in parts.split(",") if p and p != "none"] or ["none"])
