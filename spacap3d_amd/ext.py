"""``spacap3d_amd.ext`` -- the object that stands where the reference binds ``pointnet2._ext``.

Same nine names, argument order, dtypes, shapes and error behaviour as the pybind module of the
reference (lib/pointnet2/_ext_src/src/bindings.cpp:6-19; checks in include/utils.h:5-25):
inputs must be contiguous float32 / int32 tensors on a GPU, violations raise ``RuntimeError``,
outputs are freshly allocated on the inputs' device, kernels are enqueued on the current stream
without synchronisation.  CPU tensors raise "CPU not supported" exactly as the reference does
(src/sampling.cpp:33-35 etc.): there is no fallback path.

A maintainer of the reference switches over with one line in ``lib/pointnet2/pointnet2_utils.py``:
``import spacap3d_amd.ext as _ext``.
"""
import torch

from . import _native
from ._native import check, lib


def _chk_contig(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def _chk_float(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float tensor")


def _chk_int(t, name):
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")


def _chk_gpu(t, name, like=None):
    if not t.is_cuda:
        if like is None:
            raise RuntimeError("CPU not supported")
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if like is not None and t.device != like.device:
        raise RuntimeError(f"{name} must be on {like.device}")


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def opt_n_threads(work_size: int) -> int:
    """include/cuda_utils.h:15-19 of the reference (decides the FPS tie-break)."""
    return int(lib.spacap_opt_n_threads(int(work_size)))


# ---- sampling.cpp ---------------------------------------------------------------------------
def gather_points(points, idx):
    _chk_contig(points, "points"); _chk_contig(idx, "idx")
    _chk_float(points, "points"); _chk_int(idx, "idx")
    _chk_gpu(points, "points"); _chk_gpu(idx, "idx", points)
    B, C, N = points.shape
    m = idx.shape[1]
    with torch.cuda.device(points.device):
        out = torch.empty(B, C, m, dtype=torch.float32, device=points.device)
        check(lib.spacap_gather_points_f32(points.data_ptr(), idx.data_ptr(), B, C, N, m, out.data_ptr(),
                                           _stream(points)), "gather_points")
    return out


def gather_points_grad(grad_out, idx, n):
    _chk_contig(grad_out, "grad_out"); _chk_contig(idx, "idx")
    _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx")
    _chk_gpu(grad_out, "grad_out"); _chk_gpu(idx, "idx", grad_out)
    B, C, m = grad_out.shape
    with torch.cuda.device(grad_out.device):
        out = torch.empty(B, C, int(n), dtype=torch.float32, device=grad_out.device)
        check(lib.spacap_gather_points_grad_f32(grad_out.data_ptr(), idx.data_ptr(), B, C, int(n), m,
                                                out.data_ptr(), _stream(grad_out)), "gather_points_grad")
    return out


def furthest_point_sampling(points, nsamples):
    _chk_contig(points, "points")
    _chk_float(points, "points")
    _chk_gpu(points, "points")
    B, N, _ = points.shape
    nsamples = int(nsamples)
    with torch.cuda.device(points.device):
        out = torch.empty(B, nsamples, dtype=torch.int32, device=points.device)
        ws = torch.empty(max(int(lib.spacap_fps_workspace_bytes(B, N)), 16), dtype=torch.uint8,
                         device=points.device)
        check(lib.spacap_fps_f32(points.data_ptr(), B, N, nsamples, ws.data_ptr(), out.data_ptr(),
                                 _stream(points)), "furthest_point_sampling")
    return out


# ---- interpolate.cpp -------------------------------------------------------------------------
def three_nn(unknowns, knows):
    _chk_contig(unknowns, "unknowns"); _chk_contig(knows, "knows")
    _chk_float(unknowns, "unknowns"); _chk_float(knows, "knows")
    _chk_gpu(unknowns, "unknowns"); _chk_gpu(knows, "knows", unknowns)
    B, n, _ = unknowns.shape
    m = knows.shape[1]
    with torch.cuda.device(unknowns.device):
        idx = torch.empty(B, n, 3, dtype=torch.int32, device=unknowns.device)
        dist2 = torch.empty(B, n, 3, dtype=torch.float32, device=unknowns.device)
        check(lib.spacap_three_nn_f32(unknowns.data_ptr(), knows.data_ptr(), B, n, m, dist2.data_ptr(),
                                      idx.data_ptr(), _stream(unknowns)), "three_nn")
    return [dist2, idx]


def three_nn_weights(unknowns, knows):
    """(idx, weight) of the three nearest known points with the feature-propagation modules' normalised inverse-distance
    weights (pointnet2_modules.py:399-405) from one launch."""
    _chk_contig(unknowns, "unknowns"); _chk_contig(knows, "knows")
    _chk_float(unknowns, "unknowns"); _chk_float(knows, "knows")
    _chk_gpu(unknowns, "unknowns"); _chk_gpu(knows, "knows", unknowns)
    B, n, _ = unknowns.shape
    m = knows.shape[1]
    with torch.cuda.device(unknowns.device):
        idx = torch.empty(B, n, 3, dtype=torch.int32, device=unknowns.device)
        weight = torch.empty(B, n, 3, dtype=torch.float32, device=unknowns.device)
        check(lib.spacap_three_nn_weights_f32(unknowns.data_ptr(), knows.data_ptr(), B, n, m, weight.data_ptr(),
                                              idx.data_ptr(), _stream(unknowns)), "three_nn_weights")
    return idx, weight


def gather_xyz(xyz, idx):
    """xyz (B,N,3) float32, idx (B,m) int32 -> (B,m,3): the coordinates of the sampled points."""
    _chk_contig(xyz, "xyz"); _chk_contig(idx, "idx")
    _chk_float(xyz, "xyz"); _chk_int(idx, "idx")
    _chk_gpu(xyz, "xyz"); _chk_gpu(idx, "idx", xyz)
    B, N, _ = xyz.shape
    m = idx.shape[1]
    with torch.cuda.device(xyz.device):
        out = torch.empty(B, m, 3, dtype=torch.float32, device=xyz.device)
        check(lib.spacap_gather_xyz_f32(xyz.data_ptr(), idx.data_ptr(), B, N, m, out.data_ptr(), _stream(xyz)), "gather_xyz")
    return out


def three_interpolate(points, idx, weight):
    _chk_contig(points, "points"); _chk_contig(idx, "idx"); _chk_contig(weight, "weight")
    _chk_float(points, "points"); _chk_int(idx, "idx"); _chk_float(weight, "weight")
    _chk_gpu(points, "points"); _chk_gpu(idx, "idx", points); _chk_gpu(weight, "weight", points)
    B, C, m = points.shape
    n = idx.shape[1]
    with torch.cuda.device(points.device):
        out = torch.empty(B, C, n, dtype=torch.float32, device=points.device)
        check(lib.spacap_three_interpolate_f32(points.data_ptr(), idx.data_ptr(), weight.data_ptr(), B, C, m, n,
                                               out.data_ptr(), _stream(points)), "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _chk_contig(grad_out, "grad_out"); _chk_contig(idx, "idx"); _chk_contig(weight, "weight")
    _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx"); _chk_float(weight, "weight")
    _chk_gpu(grad_out, "grad_out"); _chk_gpu(idx, "idx", grad_out); _chk_gpu(weight, "weight", grad_out)
    B, C, n = grad_out.shape
    with torch.cuda.device(grad_out.device):
        out = torch.empty(B, C, int(m), dtype=torch.float32, device=grad_out.device)
        check(lib.spacap_three_interpolate_grad_f32(grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(), B, C,
                                                    n, int(m), out.data_ptr(), _stream(grad_out)),
              "three_interpolate_grad")
    return out


def three_interpolate_grad_pm(grad_pm, idx, weight, m):
    """Point-major form of ``three_interpolate_grad``: grad_pm (B,n,C) -> (B,m,C), same values."""
    _chk_contig(grad_pm, "grad_pm"); _chk_contig(idx, "idx"); _chk_contig(weight, "weight")
    _chk_float(grad_pm, "grad_pm"); _chk_float(weight, "weight")
    _chk_gpu(grad_pm, "grad_pm"); _chk_gpu(idx, "idx", grad_pm); _chk_gpu(weight, "weight", grad_pm)
    B, n, C = grad_pm.shape
    with torch.cuda.device(grad_pm.device):
        out = torch.empty(B, int(m), C, dtype=torch.float32, device=grad_pm.device)
        check(lib.spacap_three_interpolate_grad_pm_f32(grad_pm.data_ptr(), idx.data_ptr(), weight.data_ptr(), B, C, n, int(m),
                                                       out.data_ptr(), _stream(grad_pm)), "three_interpolate_grad_pm")
    return out


# ---- ball_query.cpp --------------------------------------------------------------------------
BALL_QUERY_GRID_MIN_N = 8192   # below this the exhaustive kernel is as fast as building the grid


def ball_query(new_xyz, xyz, radius, nsample):
    _chk_contig(new_xyz, "new_xyz"); _chk_contig(xyz, "xyz")
    _chk_float(new_xyz, "new_xyz"); _chk_float(xyz, "xyz")
    _chk_gpu(new_xyz, "new_xyz"); _chk_gpu(xyz, "xyz", new_xyz)
    B, m, _ = new_xyz.shape
    N = xyz.shape[1]
    with torch.cuda.device(new_xyz.device):
        idx = torch.empty(B, m, int(nsample), dtype=torch.int32, device=new_xyz.device)
        if BALL_QUERY_GRID_MIN_N <= N <= 131000 and B >= 1 and m >= 1 and nsample >= 1 and radius > 0:
            # large clouds (SA1): cell grid, same output (csrc/ball_query.hip)
            nbytes = int(lib.spacap_ball_query_grid_workspace_bytes(B, N))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=new_xyz.device)
            check(lib.spacap_ball_query_grid_f32(new_xyz.data_ptr(), xyz.data_ptr(), B, N, m, float(radius), int(nsample),
                                                 idx.data_ptr(), ws.data_ptr(), nbytes, _stream(new_xyz)), "ball_query_grid")
            return idx
        check(lib.spacap_ball_query_f32(new_xyz.data_ptr(), xyz.data_ptr(), B, N, m, float(radius), int(nsample),
                                        idx.data_ptr(), _stream(new_xyz)), "ball_query")
    return idx


# ---- group_points.cpp ------------------------------------------------------------------------
def group_points(points, idx):
    _chk_contig(points, "points"); _chk_contig(idx, "idx")
    _chk_float(points, "points"); _chk_int(idx, "idx")
    _chk_gpu(points, "points"); _chk_gpu(idx, "idx", points)
    B, C, N = points.shape
    _, P, S = idx.shape
    with torch.cuda.device(points.device):
        out = torch.empty(B, C, P, S, dtype=torch.float32, device=points.device)
        check(lib.spacap_group_points_f32(points.data_ptr(), idx.data_ptr(), B, C, N, P, S, out.data_ptr(),
                                          _stream(points)), "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    _chk_contig(grad_out, "grad_out"); _chk_contig(idx, "idx")
    _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx")
    _chk_gpu(grad_out, "grad_out"); _chk_gpu(idx, "idx", grad_out)
    B, C, P, S = grad_out.shape
    with torch.cuda.device(grad_out.device):
        out = torch.empty(B, C, int(n), dtype=torch.float32, device=grad_out.device)
        nws = int(lib.spacap_group_points_grad_workspace_bytes(B, C, int(n), P, S))
        ws = torch.empty(nws, dtype=torch.uint8, device=grad_out.device) if nws else None
        check(lib.spacap_group_points_grad_f32(grad_out.data_ptr(), idx.data_ptr(), B, C, int(n), P, S,
                                               out.data_ptr(), ws.data_ptr() if nws else None,
                                               _stream(grad_out)), "group_points_grad")
    return out


# ---- max over the samples of a group (F.max_pool2d(x, [1, nsample]) in pointnet2_modules.py:256-259) ----------
def group_max(x):
    """x f32 (B,C,P,S) contiguous -> (values f32 (B,C,P), arg u8 (B,C,P))."""
    _chk_contig(x, "x"); _chk_float(x, "x"); _chk_gpu(x, "x")
    B, C, P, S = x.shape
    with torch.cuda.device(x.device):
        out = torch.empty(B, C, P, dtype=torch.float32, device=x.device)
        arg = torch.empty(B, C, P, dtype=torch.uint8, device=x.device)
        check(lib.spacap_group_max_f32(x.data_ptr(), B * C * P, S, out.data_ptr(), arg.data_ptr(), _stream(x)),
              "group_max")
    return out, arg


def group_max_grad(grad_out, arg, S):
    _chk_contig(grad_out, "grad_out"); _chk_float(grad_out, "grad_out"); _chk_gpu(grad_out, "grad_out")
    _chk_contig(arg, "arg"); _chk_gpu(arg, "arg", grad_out)
    if arg.dtype != torch.uint8:
        raise RuntimeError("arg must be a uint8 tensor")
    B, C, P = grad_out.shape
    with torch.cuda.device(grad_out.device):
        gi = torch.empty(B, C, P, int(S), dtype=torch.float32, device=grad_out.device)
        check(lib.spacap_group_max_grad_f32(grad_out.data_ptr(), arg.data_ptr(), B * C * P, int(S), gi.data_ptr(),
                                            _stream(grad_out)), "group_max_grad")
    return gi


LIB_PATH = _native.LIB_PATH
