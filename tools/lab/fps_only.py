"""SA1-shaped FPS launches only (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import spacap3d_amd.ext as ext
from spacap3d_amd import synthetic as S
xyz = S.scene_batch(8, 40000, use_height=False, seed=1000).to('cuda')
for _ in range(5):
    ext.furthest_point_sampling(xyz, 2048)
torch.cuda.synchronize()
