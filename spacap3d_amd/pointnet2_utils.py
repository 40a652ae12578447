"""Autograd wrappers around the native operators -- the drop-in counterpart of the reference's
``lib/pointnet2/pointnet2_utils.py`` (same public names, argument order and return values):

    furthest_point_sample(xyz, npoint)            pointnet2_utils.py:51-80
    gather_operation(features, idx)               :83-117
    three_nn(unknown, known) -> (dist, idx)       :120-149   (dist = sqrt of the native dist2, :142)
    three_interpolate(features, idx, weight)      :152-206
    grouping_operation(features, idx)             :209-257
    ball_query(radius, nsample, xyz, new_xyz)     :260-291   (note the argument order)
    QueryAndGroup                                 :294-380

The native calls go through ``backend.ops()`` (the HIP library); the model code above this file is
unchanged relative to the reference's call pattern.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from .layout import ChannelMajorOf, point_major_of
from .backend import ops


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        inds = ops().furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.n = features.size(2)
        ctx.save_for_backward(idx)
        return ops().gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return ops().gather_points_grad(grad_out.contiguous(), idx, ctx.n), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = ops().three_nn(unknown, known)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.m = features.size(2)
        ctx.save_for_backward(idx, weight)
        return ops().three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return ops().three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m), None, None


three_interpolate = ThreeInterpolate.apply


class ThreeInterpolatePM(Function):
    """``three_interpolate`` whose backward gathers on point-major gradients (ext.three_interpolate_grad_pm).
    ``features``: (B,C,m) channel-major, or -- ``point_major=True`` -- the (B,m,C) tensor an SA module produced, in
    which case the gradient goes back in that layout too and no transposed copies are made for it."""

    @staticmethod
    def forward(ctx, features, idx, weight, point_major):
        cm = features.transpose(1, 2).contiguous() if point_major else features.contiguous()
        ctx.m, ctx.point_major = cm.size(2), point_major
        ctx.save_for_backward(idx, weight)
        return ops().three_interpolate(cm, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        g = ops().three_interpolate_grad_pm(grad_out.transpose(1, 2).contiguous(), idx, weight, ctx.m)   # (B,m,C)
        return (g if ctx.point_major else g.transpose(1, 2)), None, None, None


class FPConcat(Function):
    """``torch.cat([three_interpolate(known_feats, idx, weight), unknow_feats], dim=1)`` of PointnetFPModule.forward
    (lib/pointnet2/pointnet2_modules.py:406-412) as ONE launch each way (csrc/interpolate.hip: fp_concat_*): ``known`` is the
    point-major (B, m, K1) output of an SA module (``known_pm``) or a channel-major (B, K1, m) tensor, ``skip_pm`` the point-major
    (B, n, K2) skip features; returns the channel-major (B, K1 + K2, n) input of the module's first 1x1 convolution.  The backward
    splits the gradient into its two halves, both point-major, and gathers the interpolation's gradient on point-major rows."""

    @staticmethod
    def forward(ctx, known, idx, weight, skip_pm, known_pm):
        from ._native import check, lib
        known, idx, weight, skip_pm = known.contiguous(), idx.contiguous(), weight.contiguous(), skip_pm.contiguous()
        B, n, K2 = skip_pm.shape
        m, K1 = (known.shape[1], known.shape[2]) if known_pm else (known.shape[2], known.shape[1])
        dev = known.device
        with torch.cuda.device(dev):
            out = torch.empty(B, K1 + K2, n, dtype=torch.float32, device=dev)
            check(lib.spacap_fp_concat_fwd_f32(known.data_ptr(), 1 if known_pm else 0, idx.data_ptr(), weight.data_ptr(), skip_pm.data_ptr(),
                                               B, K1, K2, m, n, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                  "spacap_fp_concat_fwd_f32")
        ctx.save_for_backward(idx, weight)
        ctx.dims = (B, K1, K2, m, n, bool(known_pm))
        return out

    @staticmethod
    def backward(ctx, g):
        from ._native import check, lib
        idx, weight = ctx.saved_tensors
        B, K1, K2, m, n, known_pm = ctx.dims
        g = g.contiguous()
        dev = g.device
        with torch.cuda.device(dev):
            g1 = torch.empty(B, n, K1, dtype=torch.float32, device=dev)
            g2 = torch.empty(B, n, K2, dtype=torch.float32, device=dev)
            check(lib.spacap_fp_concat_bwd_f32(g.data_ptr(), B, K1, K2, n, g1.data_ptr(), g2.data_ptr(),
                                               torch.cuda.current_stream(dev).cuda_stream), "spacap_fp_concat_bwd_f32")
        dk = ops().three_interpolate_grad_pm(g1, idx, weight, m) if ctx.needs_input_grad[0] else None     # (B, m, K1)
        if dk is not None and not known_pm:
            dk = dk.transpose(1, 2)
        return dk, None, None, (g2 if ctx.needs_input_grad[3] else None), None


def fp_concat_train(known_feats, idx, weight, unknow_feats):
    """The concatenated input of a feature-propagation MLP (see FPConcat), or None when it does not apply (CPU, widths that are
    not multiples of 32, skip features without a point-major twin)."""
    if not (known_feats.is_cuda and known_feats.dtype == torch.float32 and getattr(ops(), "three_interpolate_grad_pm", None) is not None):
        return None
    skip_pm = point_major_of(unknow_feats)
    if skip_pm is None or known_feats.shape[1] % 32:
        return None
    kpm = point_major_of(known_feats)
    return FPConcat.apply(kpm if kpm is not None else known_feats, idx, weight, skip_pm, kpm is not None)


def three_interpolate_train(features, idx, weight):
    """Training-path interpolation: point-major gradient gather when the backend has it (same values)."""
    if getattr(ops(), "three_interpolate_grad_pm", None) is None or not features.is_cuda:
        return three_interpolate(features.contiguous(), idx, weight)
    pm = point_major_of(features)
    if pm is not None:
        return ThreeInterpolatePM.apply(pm, idx, weight, True)
    return ThreeInterpolatePM.apply(features, idx, weight, False)


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.n = features.size(2)
        ctx.save_for_backward(idx)
        return ops().group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return ops().group_points_grad(grad_out.contiguous(), idx, ctx.n), None


grouping_operation = GroupingOperation.apply


class GroupMax(Function):
    """max over the last (sample) dimension of a (B,C,npoint,nsample) tensor -- the
    ``F.max_pool2d(x, kernel_size=[1, nsample])`` + ``squeeze(-1)`` of pointnet2_modules.py:256-271."""

    @staticmethod
    def forward(ctx, x):
        out, arg = ops().group_max(x.contiguous())
        ctx.S = x.size(3)
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (arg,) = ctx.saved_tensors
        return ops().group_max_grad(grad_out.contiguous(), arg, ctx.S)


group_max = GroupMax.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        inds = ops().ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball-query grouping: idx -> grouped xyz (centre-subtracted, optionally / radius) ++ grouped features,
    xyz channels first (pointnet2_utils.py:334-362).  ``sample_uniformly`` is dead code in the reference
    (it prints and exits, :337-339) and is not provided."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz

    def forward(self, xyz, new_xyz, features=None, idx=None):
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)  # (B, 3, npoint, nsample)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.radius
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        if self.ret_grouped_xyz:
            return new_features, grouped_xyz
        return new_features
