"""``OracleExt`` -- the nine ``pointnet2._ext`` entry points on CPU tensors.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Signatures, dtype / contiguity checks
and zero-initialised outputs follow the reference host wrappers
(lib/pointnet2/_ext_src/src/{sampling,ball_query,group_points,interpolate}.cpp);
the arithmetic is ``oracle/pointnet2_oracle.c``.
"""
import ctypes

import torch

from . import build, lib_path

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_vp = ctypes.c_void_p


def _check(t, name, dtype):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        kind = "a float" if dtype == torch.float32 else "an int"
        raise RuntimeError(f"{name} must be {kind} tensor")
    if t.device.type != "cpu":
        raise RuntimeError(f"{name}: the oracle only runs on CPU tensors")


class OracleExt:
    """Drop-in for the object the reference binds as ``pointnet2_utils._ext``."""

    def __init__(self, openmp: bool = False, fma: bool = False):
        build()
        self.openmp = openmp or fma
        self.fma = fma
        self._lib = ctypes.CDLL(lib_path(openmp, fma))
        L = self._lib
        L.oracle_opt_n_threads.restype = _c_int
        L.oracle_opt_n_threads.argtypes = [_c_int]
        L.oracle_num_threads.restype = _c_int
        L.oracle_furthest_point_sampling.argtypes = [_c_int, _c_int, _c_int, _vp, _vp, _vp]
        L.oracle_gather_points.argtypes = [_c_int] * 4 + [_vp] * 3
        L.oracle_gather_points_grad.argtypes = [_c_int] * 4 + [_vp] * 3
        L.oracle_ball_query.argtypes = [_c_int, _c_int, _c_int, _c_float, _c_int, _vp, _vp, _vp]
        L.oracle_group_points.argtypes = [_c_int] * 5 + [_vp] * 3
        L.oracle_group_points_grad.argtypes = [_c_int] * 5 + [_vp] * 3
        L.oracle_three_nn.argtypes = [_c_int] * 3 + [_vp] * 4
        L.oracle_three_interpolate.argtypes = [_c_int] * 4 + [_vp] * 4
        L.oracle_three_interpolate_grad.argtypes = [_c_int] * 4 + [_vp] * 4

    # -- helpers ----------------------------------------------------------
    def num_threads(self) -> int:
        return int(self._lib.oracle_num_threads())

    def opt_n_threads(self, work_size: int) -> int:
        return int(self._lib.oracle_opt_n_threads(int(work_size)))

    # -- sampling.cpp -------------------------------------------------------
    def gather_points(self, points, idx):
        _check(points, "points", torch.float32)
        _check(idx, "idx", torch.int32)
        B, C, N = points.shape
        m = idx.shape[1]
        out = torch.zeros(B, C, m, dtype=torch.float32)
        self._lib.oracle_gather_points(B, C, N, m, points.data_ptr(), idx.data_ptr(), out.data_ptr())
        return out

    def gather_points_grad(self, grad_out, idx, n):
        _check(grad_out, "grad_out", torch.float32)
        _check(idx, "idx", torch.int32)
        B, C, m = grad_out.shape
        out = torch.zeros(B, C, n, dtype=torch.float32)
        self._lib.oracle_gather_points_grad(B, C, int(n), m, grad_out.data_ptr(), idx.data_ptr(), out.data_ptr())
        return out

    def furthest_point_sampling(self, points, nsamples):
        _check(points, "points", torch.float32)
        B, N, _ = points.shape
        out = torch.zeros(B, nsamples, dtype=torch.int32)
        tmp = torch.empty(B, N, dtype=torch.float32)
        self._lib.oracle_furthest_point_sampling(B, N, int(nsamples), points.data_ptr(), tmp.data_ptr(), out.data_ptr())
        return out

    # -- interpolate.cpp ------------------------------------------------------
    def three_nn(self, unknowns, knows):
        _check(unknowns, "unknowns", torch.float32)
        _check(knows, "knows", torch.float32)
        B, n, _ = unknowns.shape
        m = knows.shape[1]
        idx = torch.zeros(B, n, 3, dtype=torch.int32)
        dist2 = torch.zeros(B, n, 3, dtype=torch.float32)
        self._lib.oracle_three_nn(B, n, m, unknowns.data_ptr(), knows.data_ptr(), dist2.data_ptr(), idx.data_ptr())
        return [dist2, idx]

    def three_interpolate(self, points, idx, weight):
        _check(points, "points", torch.float32)
        _check(idx, "idx", torch.int32)
        _check(weight, "weight", torch.float32)
        B, C, m = points.shape
        n = idx.shape[1]
        out = torch.zeros(B, C, n, dtype=torch.float32)
        self._lib.oracle_three_interpolate(B, C, m, n, points.data_ptr(), idx.data_ptr(), weight.data_ptr(), out.data_ptr())
        return out

    def three_interpolate_grad(self, grad_out, idx, weight, m):
        _check(grad_out, "grad_out", torch.float32)
        _check(idx, "idx", torch.int32)
        _check(weight, "weight", torch.float32)
        B, C, n = grad_out.shape
        out = torch.zeros(B, C, int(m), dtype=torch.float32)
        self._lib.oracle_three_interpolate_grad(B, C, n, int(m), grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(), out.data_ptr())
        return out

    # -- ball_query.cpp ---------------------------------------------------------
    def ball_query(self, new_xyz, xyz, radius, nsample):
        _check(new_xyz, "new_xyz", torch.float32)
        _check(xyz, "xyz", torch.float32)
        B, m, _ = new_xyz.shape
        N = xyz.shape[1]
        idx = torch.zeros(B, m, int(nsample), dtype=torch.int32)
        self._lib.oracle_ball_query(B, N, m, float(radius), int(nsample), new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr())
        return idx

    # -- group_points.cpp ---------------------------------------------------------
    def group_points(self, points, idx):
        _check(points, "points", torch.float32)
        _check(idx, "idx", torch.int32)
        B, C, N = points.shape
        _, P, S = idx.shape
        out = torch.zeros(B, C, P, S, dtype=torch.float32)
        self._lib.oracle_group_points(B, C, N, P, S, points.data_ptr(), idx.data_ptr(), out.data_ptr())
        return out

    def group_points_grad(self, grad_out, idx, n):
        _check(grad_out, "grad_out", torch.float32)
        _check(idx, "idx", torch.int32)
        B, C, P, S = grad_out.shape
        out = torch.zeros(B, C, int(n), dtype=torch.float32)
        self._lib.oracle_group_points_grad(B, C, int(n), P, S, grad_out.data_ptr(), idx.data_ptr(), out.data_ptr())
        return out
