"""The fused point-major SA shared-MLP (spacap3d_amd/sa_mlp.py, csrc/sa_mlp.hip) against the per-operator path
(QueryAndGroup -> Conv2d 1x1 -> BN -> ReLU -> max over samples: the reference's own structure,
lib/pointnet2/pointnet2_modules.py:241-259) evaluated in float64 on the CPU with torch, on identical inputs and
grouping indices.  Tolerance: fp32 re-association (the GEMMs run on the matrix cores with a different summation
order; the first layer is evaluated as gather(W f) instead of W gather(f))."""
import copy

import numpy as np
import pytest
import torch
from spacap3d_amd.layout import point_major_of
import torch.nn.functional as F

from spacap3d_amd import backend
from spacap3d_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(sa, xyz, new_xyz, feats, idx, dout, rdiv):
    """float64 restatement of grouping + SharedMLP (train-mode BN) + max on the CPU; returns out and gradients."""
    xyz = xyz.double().cpu().requires_grad_(True)
    new_xyz = new_xyz.double().cpu().requires_grad_(True)
    feats = feats.double().cpu().requires_grad_(True) if feats is not None else None
    idx = idx.cpu().long()
    B, N, Sn = idx.shape
    bi = torch.arange(B).view(B, 1, 1).expand(-1, N, Sn)
    gx = (xyz[bi, idx] - new_xyz.unsqueeze(2)) / rdiv                     # (B,N,S,3)
    x = gx
    if feats is not None:
        x = torch.cat([gx, feats.transpose(1, 2)[bi, idx]], dim=-1)       # (B,N,S,3+Cf)
    params = []
    for layer in sa.mlp_module.children():
        W = layer.conv.weight.detach().double().cpu().view(layer.conv.out_channels, -1).requires_grad_(True)
        g = layer.bn.bn.weight.detach().double().cpu().requires_grad_(True)
        b = layer.bn.bn.bias.detach().double().cpu().requires_grad_(True)
        z = x @ W.t()
        m = z.mean(dim=(0, 1, 2))
        v = z.var(dim=(0, 1, 2), unbiased=False)
        x = F.relu((z - m) / torch.sqrt(v + layer.bn.bn.eps) * g + b)
        params += [W, g, b]
    out = x.max(dim=2).values.permute(0, 2, 1)                            # (B,C3,N)
    (out * dout.double().cpu()).sum().backward()
    return out.detach(), xyz.grad, new_xyz.grad, (feats.grad if feats is not None else None), [p.grad for p in params]


CASES = [
    # name, Np, N, S, Cf, mlp, radius, normalize, xyz_grad
    ("sa1", 3000, 256, 64, 1, [1, 64, 64, 128], 0.4, True, False),
    ("sa2", 1024, 256, 32, 128, [128, 128, 128, 256], 0.8, True, False),
    ("sa3", 512, 128, 16, 256, [256, 128, 128, 256], 1.2, True, False),
    ("vote_agg", 512, 64, 16, 128, [128, 128, 128, 128], 0.6, True, True),
    ("sa1_unnormalised_ragged", 1500, 101, 48, 1, [1, 64, 64, 128], 0.5, False, False),
]


@pytest.mark.parametrize("name,Np,N,Sn,Cf,mlp,radius,normalize,xyz_grad", CASES)
def test_fused_sa_mlp_matches_float64_reference(name, Np, N, Sn, Cf, mlp, radius, normalize, xyz_grad):
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    from spacap3d_amd import pointnet2_utils as PU
    torch.manual_seed(sum(map(ord, name)))
    B = 2
    sa = PointnetSAModuleVotes(npoint=N, radius=radius, nsample=Sn, mlp=list(mlp), use_xyz=True,
                               normalize_xyz=normalize).to(DEV).train()
    for layer in sa.mlp_module.children():   # non-trivial affine parameters, some negative gammas
        layer.bn.bn.weight.data.uniform_(-1.0, 1.5)
        layer.bn.bn.bias.data.uniform_(-0.3, 0.3)
    xyz = S.scene_batch(B, Np, use_height=False, seed=7).to(DEV)
    feats = torch.randn(B, Cf, Np, device=DEV)
    if Cf > 1:
        feats = F.relu(feats)
    xyz_in = xyz.clone().requires_grad_(xyz_grad)
    feats_in = feats.clone().requires_grad_(Cf > 1)
    rm0 = [l.bn.bn.running_mean.clone() for l in sa.mlp_module.children()]
    new_xyz, out, inds = sa(xyz_in, feats_in)
    assert out.shape == (B, mlp[-1], N) and point_major_of(out).is_contiguous()
    dout = torch.randn_like(out)
    (out * dout).sum().backward()
    idx = PU.ball_query(radius, Sn, xyz, new_xyz.detach())
    rdiv = radius if normalize else 1.0
    # reference: new_xyz is a gather of xyz; its gradient flows back into xyz (only checked when xyz needs grad)
    want, dxyz, dnew, dfeat, dparams = _reference(sa, xyz, new_xyz.detach(), feats, idx, dout, rdiv)

    def close(a, b, what, rtol=2e-4):
        a, b = a.detach().double().cpu().numpy(), b.numpy()
        scale = np.abs(b).max() + 1e-12
        err = np.abs(a - b).max() / scale
        assert err < rtol, f"{name}/{what}: max err {err:.3e} of scale {scale:.3e}"

    close(out, want, "out")
    got = []
    for layer in sa.mlp_module.children():
        got += [layer.conv.weight.grad.view(layer.conv.out_channels, -1), layer.bn.bn.weight.grad, layer.bn.bn.bias.grad]
    for i, (g, w) in enumerate(zip(got, dparams)):
        close(g, w, f"param{i}", rtol=1e-3)
    if Cf > 1:
        close(feats_in.grad, dfeat, "dfeat", rtol=1e-3)
    if xyz_grad:
        # d/dxyz = direct term + the term through new_xyz = xyz[inds]
        full = dxyz.clone()
        full.scatter_add_(1, inds.cpu().long().unsqueeze(-1).expand(-1, -1, 3), dnew)
        close(xyz_in.grad, full, "dxyz", rtol=1e-3)
    # running statistics moved (momentum update happened once per layer)
    for l, r0 in zip(sa.mlp_module.children(), rm0):
        assert not torch.equal(l.bn.bn.running_mean, r0)
        assert int(l.bn.bn.num_batches_tracked) == 1


def test_fused_and_per_operator_paths_agree_including_running_stats(monkeypatch):
    """Same module, same inputs, fused op switched off for the second run.  (The per-operator path's 1x1 convolutions on the
    exact-fp32 kernel: the split-bf16 one differs from the fused op's split-bf16 layer kernels by ~1e-6 in a pre-activation, which
    flips a handful of pooling arg-max / ReLU decisions among the padded groups of this input and with them whole gradient routes
    -- 174 of 262 144 feature-gradient entries; both are within 1e-6 of float64, see the float64 gates above.)"""
    from spacap3d_amd import linear
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    monkeypatch.setattr(linear, "CONV_FWD_EXACT_F32", True)
    torch.manual_seed(3)
    sa = PointnetSAModuleVotes(npoint=128, radius=0.8, nsample=32, mlp=[128, 128, 128, 256], use_xyz=True,
                               normalize_xyz=True).to(DEV).train()
    sb = copy.deepcopy(sa)
    xyz = S.scene_batch(2, 1024, use_height=False, seed=1).to(DEV)
    feats = F.relu(torch.randn(2, 128, 1024, device=DEV))
    fa = feats.clone().requires_grad_(True)
    fb = feats.clone().requires_grad_(True)
    _, oa, _ = sa(xyz, fa)
    hip = backend.ops()
    saved = hip.sa_mlp_train
    try:
        hip.sa_mlp_train = None
        _, ob, _ = sb(xyz, fb)
    finally:
        hip.sa_mlp_train = saved
    w = torch.randn_like(oa)
    (oa * w).sum().backward()
    (ob * w).sum().backward()
    assert torch.allclose(oa, ob, rtol=1e-4, atol=1e-4)
    assert torch.allclose(fa.grad, fb.grad, rtol=1e-3, atol=1e-4 * float(fb.grad.abs().max()))
    for la, lb in zip(sa.mlp_module.children(), sb.mlp_module.children()):
        assert torch.allclose(la.bn.bn.running_mean, lb.bn.bn.running_mean, rtol=1e-4, atol=1e-5)
        assert torch.allclose(la.bn.bn.running_var, lb.bn.bn.running_var, rtol=1e-4, atol=1e-6)
        ga, gb = la.conv.weight.grad, lb.conv.weight.grad
        assert torch.allclose(ga, gb, rtol=1e-3, atol=1e-4 * float(gb.abs().max()))


def test_prebuilt_inverted_index_gives_the_same_feature_gradient():
    """sa_mlp.rows_index(idx, Np) handed to the module (as the geometry pyramid does) must give bit-identical
    outputs and gradients to the module sorting its grouping itself in the backward."""
    from spacap3d_amd import pointnet2_utils as pu
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    from spacap3d_amd.sa_mlp import rows_index
    torch.manual_seed(5)
    sa = PointnetSAModuleVotes(npoint=256, radius=0.6, nsample=32, mlp=[128, 128, 128, 256], use_xyz=True,
                               normalize_xyz=True).to(DEV).train()
    sb = copy.deepcopy(sa)
    xyz = S.scene_batch(3, 1000, use_height=False, seed=2).to(DEV)
    feats = F.relu(torch.randn(3, 128, 1000, device=DEV))
    inds = pu.furthest_point_sample(xyz, 256)
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3))
    idx = pu.ball_query(0.6, 32, xyz, new_xyz)
    fa, fb = feats.clone().requires_grad_(True), feats.clone().requires_grad_(True)
    _, oa, _ = sa(xyz, fa, inds, idx, rows_index(idx, 1000))
    _, ob, _ = sb(xyz, fb, inds, idx)
    w = torch.randn_like(oa)
    (oa * w).sum().backward()
    (ob * w).sum().backward()
    assert torch.equal(oa, ob) and torch.equal(fa.grad, fb.grad)
    for la, lb in zip(sa.mlp_module.children(), sb.mlp_module.children()):
        assert torch.equal(la.conv.weight.grad, lb.conv.weight.grad)


@pytest.mark.parametrize("B,Np,E", [(3, 1000, 8192), (2, 4096, 65535), (1, 1, 100), (8, 2048, 32768), (2, 37, 64),
                                    (2, 5000, 4096), (1, 300, 70000)])
def test_rows_scatter_sums_in_ascending_row_order(B, Np, E):
    """spacap_sa_rows_scatter_f32 (inverted index of a grouping + ordered gather-sum; the one-launch counting kernel for
    Np <= 4096 and E <= 65535, the radix sort otherwise) against numpy's sequential ``np.add.at``: bit-identical fp32 sums,
    i.e. every point's rows are visited in ascending order.  Groupings with long runs of one index (ball query pads with the
    first neighbour) and with every row on one point included."""
    from spacap3d_amd._native import check, lib
    g = torch.Generator().manual_seed(B * 7 + Np + E)
    idx = torch.randint(0, Np, (B, E), generator=g, dtype=torch.int32)
    run = torch.randint(0, Np, (B, (E + 31) // 32), generator=g, dtype=torch.int32).repeat_interleave(32, 1)[:, :E]
    idx = torch.where(torch.rand(B, E, generator=g) < 0.5, run, idx)      # padded groups: runs of one index
    C = 8
    dz = torch.randn(B * E, C, generator=g)
    want = np.zeros((B * Np, C), np.float32)
    keys = (idx.long() + torch.arange(B).view(B, 1) * Np).reshape(-1).numpy()
    np.add.at(want, keys, dz.numpy())
    out = torch.empty(B, Np, C, device=DEV)
    ws = torch.empty(int(lib.spacap_sa_rows_scatter_workspace_bytes(B, Np, E)), dtype=torch.uint8, device=DEV)
    idx_d, dz_d = idx.to(DEV), dz.to(DEV)
    check(lib.spacap_sa_rows_scatter_f32(dz_d.data_ptr(), idx_d.data_ptr(), B, Np, E, C, out.data_ptr(), ws.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "spacap_sa_rows_scatter_f32")
    assert np.array_equal(out.cpu().numpy().reshape(B * Np, C), want)


@pytest.mark.parametrize("B,Np,N,Sn", [(8, 1024, 256, 16), (2, 300, 37, 32), (1, 5000, 64, 64), (3, 40, 40, 16)])
def test_relative_coordinate_gradient_in_one_launch(B, Np, N, Sn):
    """spacap_sa_drel_sums_f32: the gradient of rel = (xyz[idx] - new_xyz) / r (lib/pointnet2/pointnet2_utils.py:350-355) to the
    source points (rows that reference a point, ascending: bit-identical to numpy's sequential add.at) and to the centres
    (minus the sum over each group's rows, in row order), in one launch; either output alone as well."""
    from spacap3d_amd._native import check, lib
    g = torch.Generator().manual_seed(B + Np + N)
    E = N * Sn
    idx = torch.randint(0, Np, (B, E), generator=g, dtype=torch.int32)
    run = torch.randint(0, Np, (B, (E + 15) // 16), generator=g, dtype=torch.int32).repeat_interleave(16, 1)[:, :E]
    idx = torch.where(torch.rand(B, E, generator=g) < 0.5, run, idx)
    drel = torch.randn(B * E, 3, generator=g)
    want_x = np.zeros((B * Np, 3), np.float32)
    np.add.at(want_x, (idx.long() + torch.arange(B).view(B, 1) * Np).reshape(-1).numpy(), drel.numpy())
    want_n = np.zeros((B * N, 3), np.float32)
    d4 = drel.numpy().reshape(B * N, Sn, 3)
    for s_ in range(Sn):
        want_n += d4[:, s_]
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(int(lib.spacap_sa_rows_scatter_workspace_bytes(B, Np, E)), dtype=torch.uint8, device=DEV)
    idx_d, drel_d = idx.to(DEV), drel.to(DEV)
    check(lib.spacap_sa_rows_index_f32(idx_d.data_ptr(), B, Np, E, ws.data_ptr(), st), "rows_index")
    dxyz, dnew = torch.full((B, Np, 3), float("nan"), device=DEV), torch.full((B, N, 3), float("nan"), device=DEV)
    check(lib.spacap_sa_drel_sums_f32(drel_d.data_ptr(), B, Np, N, Sn, ws.data_ptr(), dxyz.data_ptr(), dnew.data_ptr(), st), "drel_sums")
    assert np.array_equal(dxyz.cpu().numpy().reshape(-1, 3), want_x)
    assert np.array_equal(dnew.cpu().numpy().reshape(-1, 3), -want_n)
    only_n = torch.full((B, N, 3), float("nan"), device=DEV)
    check(lib.spacap_sa_drel_sums_f32(drel_d.data_ptr(), B, Np, N, Sn, None, None, only_n.data_ptr(), st), "drel_sums")
    only_x = torch.full((B, Np, 3), float("nan"), device=DEV)
    check(lib.spacap_sa_drel_sums_f32(drel_d.data_ptr(), B, Np, N, Sn, ws.data_ptr(), only_x.data_ptr(), None, st), "drel_sums")
    assert torch.equal(only_n, dnew) and torch.equal(only_x, dxyz)


@pytest.mark.parametrize("C1,Cf,n1,nf", [(128, 256, 1024, 64), (128, 128, 1024, 7), (64, 128, 3, 1), (128, 256, 1, 130)])
def test_first_layer_weight_gradient_assembled_in_one_launch(C1, Cf, n1, nf):
    """spacap_sa_dw1_assemble_f32 = spacap_sum_slabs_f32 on the coordinate partials and on the feature-product partials + the
    concatenation [rel columns | feature columns], bit for bit."""
    from spacap3d_amd._native import check, lib, sum_slabs
    g = torch.Generator().manual_seed(C1 + Cf + n1)
    pw1 = torch.randn(n1, C1, 4, generator=g).to(DEV)
    pf = torch.randn(nf, C1 * Cf, generator=g).to(DEV)
    out = torch.full((C1, 3 + Cf), float("nan"), device=DEV)
    check(lib.spacap_sa_dw1_assemble_f32(pw1.data_ptr(), n1, pf.data_ptr(), nf, C1, Cf, out.data_ptr(), torch.cuda.current_stream().cuda_stream),
          "dw1_assemble")
    want = torch.cat([sum_slabs(pw1)[:, :3], sum_slabs(pf).view(C1, Cf)], 1)
    assert torch.equal(out, want)


@pytest.mark.parametrize("C,npoint,nsample,n", [(1, 512, 64, 6000), (0, 200, 32, 3000), (1, 77, 16, 1000)])
def test_first_layer_rebuilt_instead_of_stored(C, npoint, nsample, n, monkeypatch):
    """SA1-shaped modules (3 relative coordinates + at most one inline feature, 64 -> 64 -> 128): with sa_mlp.RECOMPUTE_Z1 the
    first layer's pre-activation is never written -- the statistics pass leaves 16 bytes per row and the second layer, its weight
    gradient and the fused first-layer backward rebuild z1 with the same arithmetic.  Outputs, BatchNorm buffers and every
    gradient must equal the stored-z1 path bit for bit (row counts that are not multiples of the tiles included)."""
    from spacap3d_amd import pointnet2_utils as pu
    from spacap3d_amd import sa_mlp
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(11)
    sa = PointnetSAModuleVotes(npoint=npoint, radius=0.3, nsample=nsample, mlp=[C, 64, 64, 128], use_xyz=True,
                               normalize_xyz=True).to(DEV).train()
    sb = copy.deepcopy(sa)
    pc = S.scene_batch(2, n, use_height=C == 1, seed=4).to(DEV)
    xyz = pc[..., :3].contiguous()
    feats = pc[..., 3:].transpose(1, 2).contiguous() if C else None
    inds = pu.furthest_point_sample(xyz, npoint)
    monkeypatch.setattr(sa_mlp, "L1_MOMENTS", False)     # (statistics from the summed z1 on both sides: see the next test)
    monkeypatch.setattr(sa_mlp, "FUSE_L2_WGRAD", False)  # (the second layer's weight gradient by its own kernel on both sides)
    res = []
    for mod, flag in ((sa, True), (sb, False)):
        monkeypatch.setattr(sa_mlp, "RECOMPUTE_Z1", flag)
        _, out, _ = mod(xyz, feats, inds)
        w = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
        (out * w).sum().backward()
        res.append(out)
    assert torch.equal(res[0], res[1])
    for (na, pa), (nb, pb) in zip(sa.named_parameters(), sb.named_parameters()):
        assert torch.equal(pa.grad, pb.grad), na
    for (na, ba), (nb, bb) in zip(sa.named_buffers(), sb.named_buffers()):
        assert torch.equal(ba, bb), na


@pytest.mark.parametrize("C,npoint,nsample,n", [(1, 512, 64, 6000), (0, 200, 32, 3000), (1, 77, 16, 1000)])
def test_first_layer_statistics_from_the_moments_of_its_inputs(C, npoint, nsample, n, monkeypatch):
    """sa_mlp.L1_MOMENTS: z1 = W1 . (rel x, rel y, rel z, feature) is linear in the row's inputs, so SA1's first BatchNorm takes
    its batch statistics from the 4 sums and 10 products of those inputs (spacap_sa_l1_moments_f32 + _finalize) instead of
    summing 64 channels per row (lib/pointnet2/pytorch_utils.py:11-36 BatchNorm2d over all grouped rows).  Same rel4 rows bit
    for bit; statistics, running buffers, outputs and gradients equal the summed form up to fp32 rounding."""
    from spacap3d_amd import pointnet2_utils as pu
    from spacap3d_amd import sa_mlp
    from spacap3d_amd._native import check, lib
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(11)
    sa = PointnetSAModuleVotes(npoint=npoint, radius=0.3, nsample=nsample, mlp=[C, 64, 64, 128], use_xyz=True,
                               normalize_xyz=True).to(DEV).train()
    sb = copy.deepcopy(sa)
    pc = S.scene_batch(2, n, use_height=C == 1, seed=4).to(DEV)
    xyz = pc[..., :3].contiguous()
    feats = pc[..., 3:].transpose(1, 2).contiguous() if C else None
    inds = pu.furthest_point_sample(xyz, npoint)
    # the two statistics passes side by side on the module's own inputs
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = pu.ball_query(0.3, nsample, xyz, new_xyz).contiguous()
    B, Np, N, Sn = 2, n, npoint, nsample
    R = B * N * Sn
    W1 = sa.mlp_module.layer0.conv.weight.detach().view(64, -1).contiguous()
    feat = feats.reshape(B, -1).contiguous() if C else None
    st = torch.cuda.current_stream().cuda_stream
    nparts = int(lib.spacap_sa_nparts())
    ra, rb = torch.empty(R, 4, device=DEV), torch.empty(R, 4, device=DEV)
    part, mom = torch.empty(nparts * 2 * 64, dtype=torch.float64, device=DEV), torch.empty(nparts * 16, dtype=torch.float64, device=DEV)
    fp = feat.data_ptr() if C else None
    check(lib.spacap_sa_l1_stats_f32(fp, xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), W1.data_ptr(), W1.shape[1], 0.3, B, Np, N, Sn, 64,
                                     ra.data_ptr(), part.data_ptr(), st), "l1_stats")
    check(lib.spacap_sa_l1_moments_f32(fp, xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), 0.3, B, Np, N, Sn, rb.data_ptr(), mom.data_ptr(), st),
          "l1_moments")
    assert torch.equal(ra, rb)
    g, b_ = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)
    sa_, sb_ = torch.empty(64, 4, device=DEV), torch.empty(64, 4, device=DEV)
    rm = [torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)]
    rv = [torch.ones(64, device=DEV), torch.ones(64, device=DEV)]
    check(lib.spacap_sa_bn_finalize_f32(part.data_ptr(), 64, R, 1e-5, 0.1, g.data_ptr(), b_.data_ptr(), rm[0].data_ptr(), rv[0].data_ptr(),
                                        sa_.data_ptr(), st), "finalize")
    check(lib.spacap_sa_l1_moments_finalize_f32(mom.data_ptr(), W1.data_ptr(), W1.shape[1], int(C == 1), 64, R, 1e-5, 0.1, g.data_ptr(),
                                                b_.data_ptr(), rm[1].data_ptr(), rv[1].data_ptr(), sb_.data_ptr(), st), "moments finalize")
    torch.cuda.synchronize()
    scale = sa_[:, 0].abs().max().clamp_min(1e-3)
    assert ((sa_[:, 0] - sb_[:, 0]).abs().max() / scale).item() < 2e-6                    # mean
    assert ((sa_[:, 1] - sb_[:, 1]).abs() / sa_[:, 1].abs()).max().item() < 2e-5          # 1 / std
    assert torch.equal(sa_[:, 3], sb_[:, 3])
    assert ((rm[0] - rm[1]).abs().max() / rm[0].abs().max().clamp_min(1e-4)).item() < 2e-5
    assert ((rv[0] - rv[1]).abs() / rv[0].abs()).max().item() < 2e-5
    # the module end to end with the switch on / off
    res = []
    for mod, flag in ((sa, True), (sb, False)):
        monkeypatch.setattr(sa_mlp, "L1_MOMENTS", flag)
        _, out, _ = mod(xyz, feats, inds)
        w = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
        (out * w).sum().backward()
        res.append(out)
    assert ((res[0] - res[1]).abs().max() / res[1].abs().max()).item() < 2e-4
    for (na, pa), (nb, pb) in zip(sa.named_parameters(), sb.named_parameters()):
        e = ((pa.grad - pb.grad).abs().max() / pb.grad.abs().max().clamp_min(1e-20)).item()
        assert e < 2e-3, (na, e)
    for (na, ba), (nb, bb) in zip(sa.named_buffers(), sb.named_buffers()):
        if ba.dtype.is_floating_point:
            assert ((ba - bb).abs().max() / bb.abs().max().clamp_min(1e-6)).item() < 1e-4, na
        else:
            assert torch.equal(ba, bb), na


@pytest.mark.parametrize("Cf,mlp,npoint,nsample,n", [(1, [64, 64, 128], 512, 64, 6000), (128, [128, 128, 256], 300, 32, 2048),
                                                     (256, [128, 128, 128], 65, 32, 1024)])
def test_pooled_layer_weight_gradient_from_z2_matches_the_dense_kernel(Cf, mlp, npoint, nsample, n, monkeypatch):
    """sa_mlp.POOL_WGRAD: the pooled layer's weight gradient from z2 alone (csrc/sa_l3bwd.inc: sa_wgrad_pool_kernel; sparse
    term + Gram matrix) against the dense kernel that streams z3 and z2 (lib/pointnet2/pytorch_utils.py:11-36,
    lib/pointnet2/pointnet2_modules.py:256-259; autograd backward).  Everything but dW3 is computed by the same kernels on the
    same inputs: bit-identical; dW3 agrees at fp32 rounding level."""
    from spacap3d_amd import pointnet2_utils as pu
    from spacap3d_amd import sa_mlp
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(5)
    sa = PointnetSAModuleVotes(npoint=npoint, radius=0.4, nsample=nsample, mlp=[Cf] + mlp, use_xyz=True, normalize_xyz=True).to(DEV).train()
    sb = copy.deepcopy(sa)
    xyz = S.scene_batch(2, n, use_height=False, seed=4).to(DEV)[..., :3].contiguous()
    feats = torch.randn(2, Cf, n, generator=torch.Generator().manual_seed(1)).to(DEV)
    inds = pu.furthest_point_sample(xyz, npoint)
    monkeypatch.setattr(sa_mlp, "POOL_WGRAD_MIN_ROWS", 0)
    res, fg = [], []
    for mod, flag in ((sa, True), (sb, False)):
        monkeypatch.setattr(sa_mlp, "POOL_WGRAD", flag)
        f = feats.clone().requires_grad_(Cf > 1)
        _, out, _ = mod(xyz, f, inds)
        w = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
        (out * w).sum().backward()
        res.append(out)
        fg.append(f.grad)
    assert torch.equal(res[0], res[1])
    last = [n_ for n_, _ in sa.named_parameters() if n_.endswith("conv.weight")][-1]
    for (na, pa), (nb, pb) in zip(sa.named_parameters(), sb.named_parameters()):
        if na == last:
            assert not torch.equal(pa.grad, pb.grad), "the weight gradient did not come from the new kernel"
            e = (pa.grad - pb.grad).abs().max() / pb.grad.abs().max().clamp_min(1e-20)
            assert e.item() < 2e-5, (na, e.item())
        else:
            assert torch.equal(pa.grad, pb.grad), na
    if fg[0] is not None:
        assert torch.equal(fg[0], fg[1])


@pytest.mark.parametrize("C,npoint,nsample,n,B", [(1, 512, 64, 6000, 2), (0, 200, 32, 3000, 2), (1, 77, 16, 1000, 3), (1, 2048, 64, 40000, 2)])
def test_second_layer_weight_gradient_from_the_data_gradient_pass(C, npoint, nsample, n, B, monkeypatch):
    """sa_mlp.FUSE_L2_WGRAD (SA1-shaped modules with the rebuilt first layer): dW2 = dz2^T relu(bn(z1)) accumulated by the
    data-gradient kernel from the tile it already holds (csrc/sa_mlp.hip: sa_dgrad_kernel<.., WG>) against the separate
    weight-gradient kernel (lib/pointnet2/pytorch_utils.py:11-36; autograd backward of the second Conv2d).  Everything but dW2
    comes from the same arithmetic: bit-identical; dW2 agrees at fp32 rounding level (row counts that are not multiples of the
    64-row tile included)."""
    from spacap3d_amd import pointnet2_utils as pu
    from spacap3d_amd import sa_mlp
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(13)
    sa = PointnetSAModuleVotes(npoint=npoint, radius=0.3, nsample=nsample, mlp=[C, 64, 64, 128], use_xyz=True,
                               normalize_xyz=True).to(DEV).train()
    sb = copy.deepcopy(sa)
    pc = S.scene_batch(B, n, use_height=C == 1, seed=4).to(DEV)
    xyz = pc[..., :3].contiguous()
    feats = pc[..., 3:].transpose(1, 2).contiguous() if C else None
    inds = pu.furthest_point_sample(xyz, npoint)
    monkeypatch.setattr(sa_mlp, "RECOMPUTE_Z1", True)
    res = []
    for mod, flag in ((sa, True), (sb, False)):
        monkeypatch.setattr(sa_mlp, "FUSE_L2_WGRAD", flag)
        _, out, _ = mod(xyz, feats, inds)
        w = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
        (out * w).sum().backward()
        res.append(out)
    assert torch.equal(res[0], res[1])
    second = [n_ for n_, _ in sa.named_parameters() if n_.endswith("conv.weight")][1]
    for (na, pa), (nb, pb) in zip(sa.named_parameters(), sb.named_parameters()):
        if na == second:
            assert not torch.equal(pa.grad, pb.grad), "the weight gradient did not come from the fused kernel"
            e = (pa.grad - pb.grad).abs().max() / pb.grad.abs().max().clamp_min(1e-20)
            assert e.item() < 2e-5, (na, e.item())
        else:
            assert torch.equal(pa.grad, pb.grad), na


def test_unsupported_mlp_uses_the_per_operator_path():
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    from spacap3d_amd import sa_mlp
    sa = PointnetSAModuleVotes(npoint=32, radius=0.4, nsample=16, mlp=[6, 16, 32], use_xyz=True).to(DEV).train()
    assert not sa_mlp.supported(sa.mlp_module, 16)
    xyz = S.scene_batch(1, 256, use_height=False, seed=1).to(DEV)
    _, out, _ = sa(xyz, torch.randn(1, 6, 256, device=DEV))
    assert out.shape == (1, 32, 32)


@pytest.mark.parametrize("mlp,Cf,Sn", [([1, 64, 64, 128], 1, 64), ([7, 64, 64, 128], 7, 64), ([128, 128, 128, 256], 128, 32),
                                       ([256, 128, 128, 128], 256, 16)])
def test_eval_mode_fused_path_folds_the_running_statistics(mlp, Cf, Sn):
    """model.eval(): the fused kernels with BatchNorm folded to its running statistics (sa_mlp.sa_mlp_eval) against the
    per-operator inference path (QueryAndGroup -> Conv2d -> BatchNorm2d(running stats) -> ReLU -> max) on the same
    module, after two training steps have moved the running statistics away from their initial values."""
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(11)
    sa = PointnetSAModuleVotes(npoint=128, radius=0.5, nsample=Sn, mlp=list(mlp), use_xyz=True, normalize_xyz=True).to(DEV)
    for layer in sa.mlp_module.children():
        layer.bn.bn.weight.data.uniform_(0.5, 1.5)
        layer.bn.bn.bias.data.uniform_(-0.3, 0.3)
    xyz = S.scene_batch(2, 1500, use_height=False, seed=4).to(DEV)
    feats = torch.randn(2, Cf, 1500, device=DEV)
    sa.train()
    for _ in range(2):
        sa(xyz, feats + 0.3)
    sa.eval()
    hip = backend.ops()
    with torch.no_grad():
        _, fused, inds = sa(xyz, feats)
        assert point_major_of(fused) is not None, "the fused inference path did not run"
        saved = hip.sa_mlp_eval
        hip.sa_mlp_eval = None
        try:
            _, plain, inds2 = sa(xyz, feats)
        finally:
            hip.sa_mlp_eval = saved
    assert torch.equal(inds, inds2)
    assert point_major_of(plain) is None
    err = float((fused - plain).abs().max() / plain.abs().max())
    assert err < 2e-5, err


def test_library_self_test_of_the_streaming_kernels():
    """sa_mlp._selftest (run once per process and device before the first fused SA op): the streaming split-bf16 kernels, whose
    prefetched rows land in registers the compiler does not know about, against float64 on the library that is loaded."""
    from spacap3d_amd import sa_mlp
    dev = torch.device(DEV)
    sa_mlp._SELFTESTED.discard((dev.type, dev.index))
    sa_mlp._selftest(dev)
    assert (dev.type, dev.index) in sa_mlp._SELFTESTED
    sa_mlp._selftest(dev)      # second call: nothing to do
