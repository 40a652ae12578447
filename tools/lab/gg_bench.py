import torch, sys
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import spacap3d_amd.ext as ext
from spacap3d_amd import synthetic as S
B = 8
xyz = S.scene_batch(B, 40000, use_height=False, seed=0).to('cuda')
inds = ext.furthest_point_sampling(xyz, 2048)
x1 = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
inds2 = ext.furthest_point_sampling(x1, 1024)
x2 = torch.gather(x1, 1, inds2.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
idx = ext.ball_query(x2, x1, 0.4, 32)
go = torch.randn(B, 128, 1024, 32, device='cuda')
for _ in range(10): ext.group_points_grad(go, idx, 2048)
torch.cuda.synchronize()
