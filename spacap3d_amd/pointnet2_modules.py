"""Set-abstraction / feature-propagation modules -- counterpart of the two classes the reference model
actually uses from ``lib/pointnet2/pointnet2_modules.py`` (PointnetSAModuleVotes :165-276,
PointnetFPModule :361-421) and of ``SharedMLP`` (lib/pointnet2/pytorch_utils.py:11-36,67-120).

Parameter / buffer names reproduce the reference's state-dict layout
(``mlp_module.layer{i}.conv.weight``, ``mlp_module.layer{i}.bn.bn.{weight,bias,running_mean,...}``,
``mlp.layer{i}...``) so VoteNet checkpoints of the reference load unchanged.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils


class _BN2d(nn.Sequential):
    """pytorch_utils.py:39-58: a Sequential holding one BatchNorm2d named ``bn`` (weight 1, bias 0)."""

    def __init__(self, channels: int):
        super().__init__()
        self.add_module("bn", nn.BatchNorm2d(channels))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0.0)


class _ConvBNReLU2d(nn.Sequential):
    """pytorch_utils.py:67-120 with the arguments SharedMLP passes: 1x1 Conv2d (bias only without BN,
    kaiming-normal weight, :87-97) -> BatchNorm2d -> ReLU(inplace)."""

    def __init__(self, cin: int, cout: int, bn: bool):
        super().__init__()
        conv = nn.Conv2d(cin, cout, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), bias=not bn)
        nn.init.kaiming_normal_(conv.weight)
        if not bn:
            nn.init.constant_(conv.bias, 0.0)
        self.add_module("conv", conv)
        if bn:
            self.add_module("bn", _BN2d(cout))
        self.add_module("activation", nn.ReLU(inplace=True))
        self.has_bn = bn

    def forward(self, x, pool=False):
        """conv -> BN -> ReLU; in training mode the BN -> ReLU (-> max over the last dim when ``pool``) tail is one
        fused op of the backend.  Eval mode (running statistics) uses the stock modules."""
        from .backend import ops
        z = None
        if self.training or not torch.is_grad_enabled():
            f = getattr(ops(), "conv1x1", None)   # 1x1 convolution with the slab weight gradient (linear.Conv1x1)
            z = f(x, self.conv) if f is not None else None
        if z is None:
            z = self.conv(x)
        if self.has_bn and self.training:
            S = z.size(3) if pool else None
            if not pool or S in (16, 32, 64, 128):
                return ops().bn_relu_train(z, self.bn.bn, pool_S=S)
            return pointnet2_utils.group_max(ops().bn_relu_train(z, self.bn.bn))
        g = getattr(ops(), "bn_relu_eval", None) if (self.has_bn and z.is_cuda and not self.training) else None
        y = g(z, self.bn.bn) if g is not None else None    # inference: BatchNorm on its running statistics + ReLU, one launch
        if y is None:
            y = self.activation(self.bn(z) if self.has_bn else z)
        return pointnet2_utils.group_max(y) if pool else y


class SharedMLP(nn.Sequential):
    def __init__(self, args: List[int], *, bn: bool = False):
        super().__init__()
        for i in range(len(args) - 1):
            self.add_module(f"layer{i}", _ConvBNReLU2d(args[i], args[i + 1], bn))

    def forward(self, x, pool=False):
        """``pool``: also take the max over the last (sample) dimension after the final layer, fused into it."""
        layers = list(self.children())
        for i, layer in enumerate(layers):
            x = layer(x, pool=(pool and i == len(layers) - 1))
        return x


class PointnetSAModuleVotes(nn.Module):
    """FPS -> gather centres -> ball-query grouping -> SharedMLP -> max over the samples.
    Returns (new_xyz (B,npoint,3), new_features (B,C_out,npoint), inds (B,npoint) int32).
    Only max pooling is provided (the model uses nothing else; avg / rbf at :260-270 are unused)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pooling: str = "max", normalize_xyz: bool = False):
        super().__init__()
        assert pooling == "max", "only max pooling is on the SpaCap3D path"
        assert npoint is not None, "GroupAll is not on the SpaCap3D path"
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.use_xyz, self.normalize_xyz = use_xyz, normalize_xyz
        self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                                                     normalize_xyz=normalize_xyz)
        mlp_spec = list(mlp)
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3
        self.mlp_module = SharedMLP(mlp_spec, bn=bn)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, inds: torch.Tensor = None,
                idx: torch.Tensor = None, rows_idx: torch.Tensor = None, new_xyz: torch.Tensor = None):
        """``inds`` (B,npoint) / ``idx`` (B,npoint,nsample): optionally precomputed sampling and grouping indices
        (they depend on the coordinates only: detector.geometry_pyramid computes them ahead of the step);
        ``rows_idx``: optionally the inverted index of ``idx`` (sa_mlp.rows_index) for the fused op's backward;
        ``new_xyz``: optionally the sampled centres xyz[inds] themselves."""
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        else:
            assert inds.shape[1] == self.npoint
        from .backend import ops
        fused = getattr(ops(), "sa_mlp_train", None) if self.training else None
        if new_xyz is not None and not xyz.requires_grad:
            # precomputed centres xyz[inds] (detector.geometry_pyramid); never for coordinates that need a gradient
            assert new_xyz.shape == (xyz.shape[0], self.npoint, 3)
        elif fused is not None and xyz.is_cuda:
            # centres as a row gather of the (B,N,3) coordinates (same values as gather_operation on the transposed
            # copy, :239-241, without the two transposes)
            if not xyz.requires_grad and inds.dtype == torch.int32:
                from . import ext
                new_xyz = ext.gather_xyz(xyz.contiguous(), inds.contiguous())
            else:
                new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3))
        else:
            xyz_flipped = xyz.transpose(1, 2).contiguous()
            new_xyz = pointnet2_utils.gather_operation(xyz_flipped, inds).transpose(1, 2).contiguous()
        if idx is None:
            idx = pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
        else:
            assert idx.shape[1:] == (self.npoint, self.nsample)
        if fused is not None:
            # training step: grouping + SharedMLP + pooling as one point-major op of the backend (sa_mlp.py)
            out = fused(xyz, new_xyz, features, idx, self.mlp_module, self.radius if self.normalize_xyz else 1.0,
                        self.use_xyz, rows_idx)
            if out is not None:
                return new_xyz, out, inds
        fused_eval = getattr(ops(), "sa_mlp_eval", None) if (not self.training and not torch.is_grad_enabled()) else None
        if fused_eval is not None and xyz.is_cuda:
            # inference: the same fused kernels with the BatchNorm layers folded to their running statistics
            out = fused_eval(xyz, new_xyz, features, idx, self.mlp_module, self.radius if self.normalize_xyz else 1.0,
                             self.use_xyz)
            if out is not None:
                return new_xyz, out, inds
        if features is not None:
            features = features.contiguous()
        grouped_features, _grouped_xyz = self.grouper(xyz, new_xyz, features, idx=idx)  # (B, C+3, npoint, nsample)
        # SharedMLP + F.max_pool2d(x, [1, nsample]).squeeze(-1)  (pointnet2_modules.py:253-271)
        new_features = self.mlp_module(grouped_features, pool=True)             # (B, mlp[-1], npoint)
        return new_xyz, new_features, inds


class PointnetFPModule(nn.Module):
    """three_nn -> inverse-distance weights -> three_interpolate -> cat skip -> SharedMLP (:376-421)."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = SharedMLP(list(mlp), bn=bn)

    @staticmethod
    def neighbours(unknown, known):
        """(idx, weight) of the three nearest known points and their normalised inverse distances (:399-405)."""
        if unknown.is_cuda and not (unknown.requires_grad or known.requires_grad):
            from . import ext
            return ext.three_nn_weights(unknown.contiguous(), known.contiguous())   # search + weights in one launch
        dist, idx = pointnet2_utils.three_nn(unknown, known)
        dist_recip = 1.0 / (dist + 1e-8)
        norm = torch.sum(dist_recip, dim=2, keepdim=True)
        return idx, dist_recip / norm

    def forward(self, unknown, known, unknow_feats, known_feats, nn=None):
        """``nn``: optionally the precomputed ``neighbours(unknown, known)`` (coordinates only)."""
        if known is not None:
            idx, weight = nn if nn is not None else self.neighbours(unknown, known)
            if self.training and torch.is_grad_enabled() and unknow_feats is not None:
                # interpolation + concatenation with the skip features in one launch each way (no transposed copies, no cat)
                cat = pointnet2_utils.fp_concat_train(known_feats, idx, weight, unknow_feats)
                if cat is not None:
                    return self.mlp(cat.unsqueeze(-1)).squeeze(-1)
            if self.training and torch.is_grad_enabled():
                interpolated = pointnet2_utils.three_interpolate_train(known_feats, idx, weight)
            else:
                interpolated = pointnet2_utils.three_interpolate(known_feats.contiguous(), idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        new_features = torch.cat([interpolated, unknow_feats], dim=1) if unknow_feats is not None else interpolated
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)
