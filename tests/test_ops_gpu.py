"""Parity of the HIP operators (through the C ABI) against the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): FPS / ball_query / three_nn indices and every gathered value bit-exact;
scatter-add gradients within fp32 re-association error (the reference itself uses atomics).
"""
import os
import sys

import numpy as np
import pytest
import torch

from spacap3d_amd import synthetic as S

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(n, seed, B=2):
    return S.scene_batch(B, n, use_height=False, seed=seed)  # (B, n, 3)


FPS_CASES = [  # (N, m) -- every kernel variant: 1-wave, 256-thread, 1024-thread register, bucketed (NB 3..16), generic
    (37, 37), (64, 16), (100, 30), (128, 128), (300, 64), (512, 256), (513, 512), (700, 100), (1000, 1000), (1024, 256), (1024, 512),
    (2048, 1024), (3000, 200), (4096, 512), (8192, 128), (8193, 100), (10000, 256), (12288, 64), (16000, 300),
    (20000, 256), (24576, 64),
    (30000, 128), (40000, 2048), (40960, 64), (50000, 64), (65535, 48), (65536, 48), (70000, 200), (80000, 96), (81920, 40), (81921, 24),
]


@pytest.mark.parametrize("N,m", FPS_CASES)
def test_fps_bit_exact(hip_ext, oracle_ext, N, m):
    xyz = _scene(N, seed=N + m)
    want = oracle_ext.furthest_point_sampling(xyz, m)
    got = hip_ext.furthest_point_sampling(xyz.to(DEV), m).cpu()
    assert got.dtype == torch.int32 and got.shape == (xyz.shape[0], m)
    assert torch.equal(got, want), f"first mismatch at {(got != want).nonzero()[0].tolist()}"


@pytest.mark.parametrize("N", [26000, 28672, 33000, 36864, 41000, 45056, 50000, 61440, 66000, 70000, 77824])
def test_fps_stays_inside_its_workspace(oracle_ext, N):
    """The bucketed kernel lays its planes out with NPAD = NB*4096 for the DISPATCHED template NB (3,4,5,6,8,10,12,16,20),
    which exceeds ceil(N/4096) for these N: spacap_fps_workspace_bytes must size for it.  The workspace handed to the C
    ABI is the exact-size prefix of a larger buffer whose tail is a guard pattern."""
    from spacap3d_amd._native import check, lib
    B, m = 2, 24
    xyz = _scene(N, seed=N, B=B)
    want = oracle_ext.furthest_point_sampling(xyz, m)
    nbytes = int(lib.spacap_fps_workspace_bytes(B, N))
    guard = 4 << 20
    buf = torch.full((nbytes + guard,), 0xA5, dtype=torch.uint8, device=DEV)
    x = xyz.to(DEV)
    out = torch.empty(B, m, dtype=torch.int32, device=DEV)
    check(lib.spacap_fps_f32(x.data_ptr(), B, N, m, buf.data_ptr(), out.data_ptr(),
                             torch.cuda.current_stream().cuda_stream), "fps")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), want)
    assert bool((buf[nbytes:] == 0xA5).all()), "FPS wrote past spacap_fps_workspace_bytes()"


@pytest.mark.parametrize("N,m", [(512, 128), (700, 64), (1024, 256), (2048, 256), (5000, 300), (40000, 400)])
def test_fps_ties_on_a_grid(hip_ext, oracle_ext, N, m):
    """Integer-lattice points: almost every round has many exactly equal maxima, so the result is decided
    by the reference's tree tie-break (sampling_gpu.cu:59-65)."""
    g = torch.Generator().manual_seed(N)
    xyz = (torch.randint(0, 6, (2, N, 3), generator=g).float() * 0.5).contiguous()
    xyz[:, 5] = 0.0            # skipped by the |p|^2 <= 1e-3 rule
    xyz[:, 9] = torch.tensor([0.01, 0.02, 0.0])
    want = oracle_ext.furthest_point_sampling(xyz, m)
    got = hip_ext.furthest_point_sampling(xyz.to(DEV), m).cpu()
    assert torch.equal(got, want)


def test_fps_all_points_skipped(hip_ext, oracle_ext):
    xyz = (torch.rand(1, 300, 3) * 0.01).contiguous()  # every |p|^2 <= 1e-3 -> all indices 0
    want = oracle_ext.furthest_point_sampling(xyz, 10)
    got = hip_ext.furthest_point_sampling(xyz.to(DEV), 10).cpu()
    assert torch.equal(got, want) and int(want.abs().sum()) == 0


def test_fps_skip_threshold_is_a_double_compare(hip_ext, oracle_ext):
    """|p|^2 == float32(1e-3) is > the double literal 1e-3, so that point is NOT skipped."""
    xyz = torch.zeros(1, 64, 3)
    xyz[0, :, 0] = torch.linspace(1.0, 2.0, 64)
    r = np.sqrt(np.float32(1e-3)).astype(np.float32)
    xyz[0, 3] = torch.tensor([float(r), 0.0, 0.0])
    want = oracle_ext.furthest_point_sampling(xyz.contiguous(), 64)
    got = hip_ext.furthest_point_sampling(xyz.contiguous().to(DEV), 64).cpu()
    assert torch.equal(got, want)


BQ_CASES = [  # (N, m, radius, nsample)
    (40000, 2048, 0.2, 64), (2048, 1024, 0.4, 32), (1024, 512, 0.8, 16), (512, 256, 1.2, 16),
    (1024, 256, 0.3, 16), (4096, 64, 0.3, 16), (777, 33, 0.5, 5), (100, 7, 0.05, 8), (5000, 130, 10.0, 128),
]


@pytest.mark.parametrize("N,m,radius,nsample", BQ_CASES)
def test_ball_query_bit_exact(hip_ext, oracle_ext, N, m, radius, nsample):
    xyz = _scene(N, seed=N)
    inds = oracle_ext.furthest_point_sampling(xyz, m).long()
    new_xyz = torch.gather(xyz, 1, inds.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    new_xyz[:, -1] += 50.0  # one centre with an empty ball -> all-zero row
    want = oracle_ext.ball_query(new_xyz, xyz, radius, nsample)
    got = hip_ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), radius, nsample).cpu()
    assert torch.equal(got, want)
    assert int(want[:, -1].abs().sum()) == 0


@pytest.mark.parametrize("C,n,m", [(256, 1024, 512), (256, 512, 256), (5, 70, 9), (64, 300, 1), (130, 33, 200)])
def test_three_interpolate_point_major_gradient_is_the_same_gather(hip_ext, oracle_ext, C, n, m):
    """spacap_three_interpolate_grad_pm_f32 on the transposed gradient gives the transposed result of
    spacap_three_interpolate_grad_f32 / the oracle, bit for bit (same lists, same (n, k) order); and
    pointnet2_utils.three_interpolate_train routes both layouts of the features correctly."""
    from spacap3d_amd import pointnet2_utils as pu
    g = torch.Generator().manual_seed(C + n)
    B = 2
    idx = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32)
    idx[:, : n // 2, 0] = 0                       # one very popular known point (beyond the list capacity when n is large)
    w = torch.rand(B, n, 3, generator=g)
    go = torch.randn(B, C, n, generator=g)
    want = oracle_ext.three_interpolate_grad(go, idx, w, m)
    got_cm = hip_ext.three_interpolate_grad(go.to(DEV), idx.to(DEV), w.to(DEV), m)
    got_pm = hip_ext.three_interpolate_grad_pm(go.transpose(1, 2).contiguous().to(DEV), idx.to(DEV), w.to(DEV), m)
    assert torch.equal(got_cm.cpu(), want)
    assert torch.equal(got_pm.transpose(1, 2).cpu(), want)
    feats = torch.randn(B, C, m, generator=g)
    for layout in ("channel", "point"):
        f = feats.to(DEV).requires_grad_(True)
        if layout == "point":
            from spacap3d_amd.layout import ChannelMajorOf
            pm = f.transpose(1, 2).contiguous()
            out = pu.three_interpolate_train(ChannelMajorOf.wrap(pm), idx.to(DEV), w.to(DEV))
        else:
            out = pu.three_interpolate_train(f, idx.to(DEV), w.to(DEV))
        assert torch.equal(out.detach().cpu(), oracle_ext.three_interpolate(feats, idx, w))
        (out * go.to(DEV)).sum().backward()
        assert torch.equal(f.grad.cpu(), want)


@pytest.mark.parametrize("N,m,radius,nsample", BQ_CASES + [(40001, 300, 0.2, 64), (9000, 200, 0.05, 32), (20000, 64, 3.0, 64)])
def test_ball_query_cell_grid_is_bit_exact(hip_ext, oracle_ext, monkeypatch, N, m, radius, nsample):
    """spacap_ball_query_grid_f32 (cell grid + hit bitmap) must give the exhaustive kernel's / the oracle's rows
    exactly, also when it is forced onto small clouds, and for centres that are not points of the cloud."""
    from spacap3d_amd import ext
    monkeypatch.setattr(ext, "BALL_QUERY_GRID_MIN_N", 1)
    xyz = _scene(N, seed=N + 1)
    inds = oracle_ext.furthest_point_sampling(xyz, m).long()
    new_xyz = torch.gather(xyz, 1, inds.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    g = torch.Generator().manual_seed(N)
    new_xyz[:, : m // 2] += 0.3 * radius * torch.randn(new_xyz.shape[0], m // 2, 3, generator=g)   # off-cloud centres
    new_xyz[:, -1] += 50.0     # empty ball, far outside the grid
    new_xyz[:, -2] = xyz.amin(1) - 0.5 * radius   # just outside the bounding box
    want = oracle_ext.ball_query(new_xyz, xyz, radius, nsample)
    got = hip_ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), radius, nsample).cpu()
    assert torch.equal(got, want)


def test_ball_query_cell_grid_degenerate_clouds(hip_ext, oracle_ext, monkeypatch):
    """Duplicated points, a cloud that is one point, and an extent so large that the cell size is set by the
    64 x 64 x 16 cap instead of the radius."""
    from spacap3d_amd import ext
    monkeypatch.setattr(ext, "BALL_QUERY_GRID_MIN_N", 1)
    g = torch.Generator().manual_seed(3)
    base = torch.rand(2, 50, 3, generator=g)
    dup = base.repeat(1, 40, 1)                                   # every point 40 times
    one = torch.full((2, 300, 3), 0.25)
    wide = torch.rand(2, 5000, 3, generator=g) * torch.tensor([300.0, 200.0, 90.0])
    for xyz, radius, nsample in ((dup, 0.2, 64), (one, 0.1, 16), (wide, 2.0, 32)):
        new_xyz = xyz[:, ::7].contiguous()
        want = oracle_ext.ball_query(new_xyz, xyz.contiguous(), radius, nsample)
        got = hip_ext.ball_query(new_xyz.to(DEV), xyz.contiguous().to(DEV), radius, nsample).cpu()
        assert torch.equal(got, want)


@pytest.mark.parametrize("C,N,P,Sn", [(1, 40000, 2048, 64), (3, 40000, 2048, 64), (128, 2048, 1024, 32),
                                      (7, 300, 17, 5), (256, 1024, 256, 16)])
def test_group_points_and_grad(hip_ext, oracle_ext, C, N, P, Sn):
    g = torch.Generator().manual_seed(C * N)
    B = 2
    pts = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, P, Sn), generator=g, dtype=torch.int32)
    want = oracle_ext.group_points(pts, idx)
    got = hip_ext.group_points(pts.to(DEV), idx.to(DEV)).cpu()
    assert torch.equal(got, want)
    go = torch.randn(B, C, P, Sn, generator=g)
    want_g = oracle_ext.group_points_grad(go, idx, N)
    got_g = hip_ext.group_points_grad(go.to(DEV), idx.to(DEV), N).cpu()
    torch.testing.assert_close(got_g, want_g, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("C,N,m", [(3, 40000, 2048), (3, 2048, 1024), (5, 100, 100), (64, 1000, 10)])
def test_gather_points_and_grad(hip_ext, oracle_ext, C, N, m):
    g = torch.Generator().manual_seed(C + N)
    B = 2
    pts = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, m), generator=g, dtype=torch.int32)
    assert torch.equal(hip_ext.gather_points(pts.to(DEV), idx.to(DEV)).cpu(), oracle_ext.gather_points(pts, idx))
    go = torch.randn(B, C, m, generator=g)
    torch.testing.assert_close(hip_ext.gather_points_grad(go.to(DEV), idx.to(DEV), N).cpu(),
                               oracle_ext.gather_points_grad(go, idx, N), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,m", [(512, 256), (1024, 512), (100, 2), (77, 1), (300, 1000)])
def test_three_nn_bit_exact(hip_ext, oracle_ext, n, m):
    g = torch.Generator().manual_seed(n * m)
    B = 2
    unknown = torch.rand(B, n, 3, generator=g)
    known = torch.rand(B, m, 3, generator=g)
    if m > 4:
        known[:, 3] = known[:, 1]  # exact duplicate: tie decided by the strict `<` cascade
    d_want, i_want = oracle_ext.three_nn(unknown, known)
    d_got, i_got = hip_ext.three_nn(unknown.to(DEV), known.to(DEV))
    assert torch.equal(i_got.cpu(), i_want)
    assert torch.equal(d_got.cpu(), d_want)  # includes +inf slots when m < 3


@pytest.mark.parametrize("n,m", [(512, 256), (1024, 512), (100, 3), (300, 1000)])
def test_three_nn_weights_equal_the_tensor_composition(hip_ext, n, m):
    """ext.three_nn_weights (search + normalised inverse-distance weights in one launch, csrc/interpolate.hip) against the
    reference's separate tensor operations on the search result (pointnet2_modules.py:399-405): identical indices and
    bit-identical weights, a point coinciding with a known point (distance 0) included."""
    from spacap3d_amd import ext
    g = torch.Generator().manual_seed(n + m)
    unknown = torch.rand(2, n, 3, generator=g).to(DEV)
    known = torch.rand(2, m, 3, generator=g).to(DEV)
    unknown[:, 5] = known[:, 1]
    d2, idx = hip_ext.three_nn(unknown, known)
    r = 1.0 / (torch.sqrt(d2) + 1e-8)
    want = r / torch.sum(r, dim=2, keepdim=True)
    i_got, w_got = ext.three_nn_weights(unknown, known)
    assert torch.equal(i_got, idx) and torch.equal(w_got, want)


def test_gather_xyz_is_a_row_gather(hip_ext):
    from spacap3d_amd import ext
    g = torch.Generator().manual_seed(3)
    xyz = torch.randn(3, 500, 3, generator=g).to(DEV)
    idx = torch.randint(0, 500, (3, 77), generator=g, dtype=torch.int32).to(DEV)
    want = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(ext.gather_xyz(xyz, idx), want)
    assert ext.gather_xyz(xyz, idx[:, :0].contiguous()).shape == (3, 0, 3)


@pytest.mark.parametrize("C,m,n", [(256, 256, 512), (256, 512, 1024), (2, 4, 2), (33, 50, 77)])
def test_three_interpolate_and_grad(hip_ext, oracle_ext, C, m, n):
    g = torch.Generator().manual_seed(C + m + n)
    B = 2
    pts = torch.randn(B, C, m, generator=g)
    idx = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32)
    w = torch.rand(B, n, 3, generator=g)
    w = (w / w.sum(-1, keepdim=True)).contiguous()
    assert torch.equal(hip_ext.three_interpolate(pts.to(DEV), idx.to(DEV), w.to(DEV)).cpu(),
                       oracle_ext.three_interpolate(pts, idx, w))
    go = torch.randn(B, C, n, generator=g)
    # gather form in the oracle's (n, k) order: bit-identical (the reference's float atomics are order dependent)
    assert torch.equal(hip_ext.three_interpolate_grad(go.to(DEV), idx.to(DEV), w.to(DEV), m).cpu(),
                       oracle_ext.three_interpolate_grad(go, idx, w, m))


def test_three_interpolate_grad_with_a_hot_known_point(hip_ext, oracle_ext):
    """Half of all unknown points reference known point 5 (more pairs than the per-point list of the gather kernel
    holds: the ordered fallback must give the same sums), point 6 is never referenced (gradient exactly 0)."""
    g = torch.Generator().manual_seed(11)
    B, C, m, n = 2, 40, 32, 700
    idx = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32)
    idx[idx == 6] = 7
    idx[:, ::2, 1] = 5
    w = torch.rand(B, n, 3, generator=g).contiguous()
    go = torch.randn(B, C, n, generator=g)
    got = hip_ext.three_interpolate_grad(go.to(DEV), idx.to(DEV), w.to(DEV), m).cpu()
    assert torch.equal(got, oracle_ext.three_interpolate_grad(go, idx, w, m))
    assert float(got[:, :, 6].abs().max()) == 0.0


def test_reference_known_answer_three_interpolate(hip_ext):
    """The one native-op test the reference ships: lib/pointnet2/pointnet2_test.py:14-26 (fixed idx / weight)."""
    feats = torch.tensor([[[1.0, 2.0, 3.0, 4.0], [-1.0, 0.5, 2.5, 8.0]]], device=DEV)
    idx = torch.tensor([[[0, 1, 2], [1, 2, 3]]], dtype=torch.int32, device=DEV)
    w = torch.tensor([[[1.0, 1.0, 1.0], [2.0, 2.0, 2.0]]], device=DEV)
    out = hip_ext.three_interpolate(feats, idx, w).cpu()
    assert torch.equal(out, torch.tensor([[[6.0, 18.0], [2.0, 22.0]]]))
    g = hip_ext.three_interpolate_grad(torch.ones(1, 2, 2, device=DEV), idx, w, 4).cpu()
    assert torch.equal(g, torch.tensor([[[1.0, 3.0, 3.0, 2.0], [1.0, 3.0, 3.0, 2.0]]]))


def test_rejects_cpu_and_bad_dtypes(hip_ext):
    xyz = torch.rand(1, 64, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        hip_ext.furthest_point_sampling(xyz, 8)
    with pytest.raises(RuntimeError, match="contiguous"):
        hip_ext.furthest_point_sampling(torch.rand(1, 3, 64, device=DEV).transpose(1, 2), 8)
    with pytest.raises(RuntimeError, match="int tensor"):
        hip_ext.gather_points(torch.rand(1, 3, 64, device=DEV), torch.zeros(1, 8, dtype=torch.int64, device=DEV))


@pytest.mark.parametrize("kind", ["uniform_volume", "line", "all_duplicates", "two_clusters", "nan_inf"])
def test_fps_bucketed_kernel_on_adversarial_layouts(hip_ext, oracle_ext, kind):
    """The large-scene kernel skips whole buckets when that provably changes nothing; layouts that stress the
    bounding-box test (degenerate boxes, huge extents, non-finite coordinates) must still be bit-exact."""
    g = torch.Generator().manual_seed(5)
    N, m = 30000, 300
    if kind == "uniform_volume":
        xyz = torch.rand(2, N, 3, generator=g) * 4
    elif kind == "line":
        xyz = torch.zeros(2, N, 3)
        xyz[..., 0] = torch.rand(2, N, generator=g) * 100 + 1
    elif kind == "all_duplicates":
        xyz = torch.ones(2, N, 3) * 2.5
        xyz[:, ::1000] += torch.rand(2, 30, 3, generator=g)
    elif kind == "two_clusters":
        xyz = torch.randn(2, N, 3, generator=g) * 0.05 + 1
        xyz[:, N // 2:] += 1e4
    else:
        xyz = torch.rand(2, N, 3, generator=g) * 4 + 1
        xyz[0, 17, 0] = float("nan")
        xyz[0, 4000, 1] = float("inf")
        xyz[1, 29999, 2] = float("-inf")
    xyz = xyz.contiguous()
    want = oracle_ext.furthest_point_sampling(xyz, m)
    got = hip_ext.furthest_point_sampling(xyz.to(DEV), m).cpu()
    assert torch.equal(got, want), f"first mismatch at {(got != want).nonzero()[0].tolist()}"


def test_group_points_grad_indexed_path_is_deterministic_and_matches_oracle_bitwise(hip_ext, oracle_ext):
    """C >= 16 takes the inverted-index (gather) path: ascending (centre, sample) summation = the oracle's order."""
    g = torch.Generator().manual_seed(3)
    B, C, N, P, Sn = 2, 128, 2048, 1024, 32
    idx = torch.randint(0, N, (B, P, Sn), generator=g, dtype=torch.int32)
    idx[:, :, 5:] = idx[:, :, 4:5]  # ball-query style padding: long duplicate runs
    go = torch.randn(B, C, P, Sn, generator=g)
    want = oracle_ext.group_points_grad(go, idx, N)
    a = hip_ext.group_points_grad(go.to(DEV), idx.to(DEV), N)
    b = hip_ext.group_points_grad(go.to(DEV), idx.to(DEV), N)
    assert torch.equal(a, b)
    assert torch.equal(a.cpu(), want)


@pytest.mark.parametrize("C,P,Sn", [(128, 2048, 64), (256, 1024, 32), (64, 512, 16), (5, 100, 7), (1, 2048, 64)])
def test_group_max_matches_max_pool2d(hip_ext, C, P, Sn):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(C + P)
    x = torch.randn(2, C, P, Sn, generator=g)
    x[:, :, :, 3:9] = x[:, :, :, 2:3]          # exact ties: the first maximum must win
    x[0, 0, 0, 5] = float("nan")               # NaN wins, as in PyTorch's pooling
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, kernel_size=[1, Sn]).squeeze(-1)
    w = torch.randn(2, C, P, generator=g)
    (yr[~yr.isnan()] * w[~yr.isnan()]).sum().backward()
    out, arg = hip_ext.group_max(x.to(DEV))
    assert torch.equal(out.cpu().nan_to_num(123.0), yr.detach().nan_to_num(123.0))
    w0 = w.clone(); w0[yr.isnan()] = 0
    gi = hip_ext.group_max_grad(w0.to(DEV), arg, Sn).cpu()
    assert torch.equal(gi, xr.grad)


@pytest.mark.parametrize("B,C,P,Sn,pool", [(4, 64, 512, 64, False), (4, 128, 256, 64, True), (2, 256, 128, 16, True),
                                           (3, 37, 50, 32, True), (2, 16, 33, 7, False), (8, 128, 2048, 1, False)])
def test_fused_bn_relu_max_matches_torch(hip_ext, B, C, P, Sn, pool):
    """BatchNorm2d(train) -> ReLU [-> max_pool2d([1,S])] : values, gradients (z, gamma, beta) and running statistics."""
    import torch.nn as nn
    import torch.nn.functional as F
    from spacap3d_amd.fused_bn import bn_relu_train
    g = torch.Generator().manual_seed(B * C + P)
    z = torch.randn(B, C, P, Sn, generator=g) * 2 + 0.5
    w = torch.randn(B, C, P, generator=g) if pool else torch.randn(B, C, P, Sn, generator=g)
    bn_r = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn_r.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn_r.weight[0] = -0.7  # a negative scale reverses the order inside the group
        bn_r.bias.copy_(torch.randn(C, generator=g) * 0.3)
    bn_g = nn.BatchNorm2d(C)
    bn_g.load_state_dict(bn_r.state_dict())
    bn_g = bn_g.to(DEV)
    zr = z.clone().requires_grad_(True)
    yr = F.relu(bn_r(zr))
    if pool:
        yr = F.max_pool2d(yr, kernel_size=[1, Sn]).squeeze(-1)
    (yr * w).sum().backward()
    zg = z.to(DEV).requires_grad_(True)
    yg = bn_relu_train(zg, bn_g, pool_S=Sn if pool else None)
    (yg * w.to(DEV)).sum().backward()
    torch.testing.assert_close(yg.detach().cpu(), yr.detach(), rtol=1e-4, atol=1e-5)
    for got, want, name in ((zg.grad, zr.grad, "dz"), (bn_g.weight.grad, bn_r.weight.grad, "dgamma"),
                            (bn_g.bias.grad, bn_r.bias.grad, "dbeta")):
        err = float((got.cpu() - want).abs().max()) / (float(want.abs().max()) + 1e-12)
        assert err < 2e-4, (name, err)
    torch.testing.assert_close(bn_g.running_mean.cpu(), bn_r.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn_g.running_var.cpu(), bn_r.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn_g.num_batches_tracked) == int(bn_r.num_batches_tracked) == 1


@pytest.mark.gpu
def test_single_launch_bn_equals_the_three_pass_form(hip_ext):
    """Small tensors take one launch per direction (csrc/bn_relu.hip: bn_relu_train_small_kernel / bn_relu_bwd_small_kernel):
    outputs, gradients and running statistics must equal the partial / final / apply form (spacap_bn_set_single_launch(0))
    bit for bit -- same fp32 expressions, fp64 sums."""
    from spacap3d_amd._native import lib
    from spacap3d_amd.fused_bn import bn_relu_train

    def run():
        torch.manual_seed(5)
        outs = []
        for (B, C, L) in ((8, 256, 1024), (8, 128, 256), (2, 64, 16384)):
            z = torch.randn(B, C, L, device="cuda:0", requires_grad=True)
            bn = torch.nn.BatchNorm1d(C).cuda().train()
            with torch.no_grad():
                bn.weight.uniform_(-1.0, 1.5)
                bn.bias.normal_()
            y = bn_relu_train(z, bn)
            (y * torch.randn_like(y)).sum().backward()
            outs += [y.detach().cpu(), z.grad.cpu(), bn.weight.grad.cpu(), bn.bias.grad.cpu(), bn.running_mean.cpu(), bn.running_var.cpu()]
        return outs
    A = run()
    try:
        assert lib.spacap_bn_set_single_launch(0) == 0
        Bq = run()
    finally:
        lib.spacap_bn_set_single_launch(1)
    for i, (x, y) in enumerate(zip(A, Bq)):
        assert torch.equal(x, y), (i, (x - y).abs().max().item())
