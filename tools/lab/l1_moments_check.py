"""Lab: first-layer BatchNorm statistics, summed-z1 form against the moment form, both against float64."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spacap3d_amd import pointnet2_utils as pu, synthetic as S
from spacap3d_amd._native import check, lib
DEV = torch.device("cuda:0")
torch.manual_seed(0)
for B, n, N, Sn in ((2, 6000, 512, 64), (8, 40000, 2048, 64), (2, 1024, 64, 16)):
    pc = S.scene_batch(B, n, use_height=True, seed=4).to(DEV)
    xyz, feat = pc[..., :3].contiguous(), pc[..., 3].contiguous()
    inds = pu.furthest_point_sample(xyz, N)
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = pu.ball_query(0.2, Sn, xyz, new_xyz).contiguous()
    R = B * N * Sn
    W1 = (torch.randn(64, 4, device=DEV) * 0.5).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    nparts = int(lib.spacap_sa_nparts())
    ra, rb = torch.empty(R, 4, device=DEV), torch.empty(R, 4, device=DEV)
    part, mom = torch.empty(nparts * 2 * 64, dtype=torch.float64, device=DEV), torch.empty(nparts * 16, dtype=torch.float64, device=DEV)
    check(lib.spacap_sa_l1_stats_f32(feat.data_ptr(), xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), W1.data_ptr(), 4, 0.2, B, n, N, Sn, 64,
                                     ra.data_ptr(), part.data_ptr(), st), "a")
    check(lib.spacap_sa_l1_moments_f32(feat.data_ptr(), xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), 0.2, B, n, N, Sn, rb.data_ptr(),
                                       mom.data_ptr(), st), "b")
    g, b_ = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    sa_, sb_ = torch.empty(64, 4, device=DEV), torch.empty(64, 4, device=DEV)
    check(lib.spacap_sa_bn_finalize_f32(part.data_ptr(), 64, R, 1e-5, 0.1, g.data_ptr(), b_.data_ptr(), None, None, sa_.data_ptr(), st), "fa")
    check(lib.spacap_sa_l1_moments_finalize_f32(mom.data_ptr(), W1.data_ptr(), 4, 1, 64, R, 1e-5, 0.1, g.data_ptr(), b_.data_ptr(), None, None,
                                                sb_.data_ptr(), st), "fb")
    torch.cuda.synchronize()
    # float64 truth from the fp32 z1 the kernels rebuild (fma chain) and from the exact linear form
    z32 = (W1[:, 0] * ra[:, 0:1] + W1[:, 1] * ra[:, 1:2] + W1[:, 2] * ra[:, 2:3] + W1[:, 3] * ra[:, 3:4])
    for name, z in (("fp32 z1", z32.double()), ("exact z1", ra.double() @ W1.double().t())):
        mean, var = z.mean(0), z.var(0, unbiased=False)
        istd = 1.0 / torch.sqrt(var + 1e-5)
        for lab, s_ in (("summed", sa_), ("moments", sb_)):
            em = ((s_[:, 0].double() - mean).abs() / torch.sqrt(var)).max().item()
            ei = ((s_[:, 1].double() - istd).abs() / istd).max().item()
            print(f"R={R:8d} truth={name:9s} {lab:8s}: |mean err| / std {em:.2e}   rel err of 1/std {ei:.2e}")
