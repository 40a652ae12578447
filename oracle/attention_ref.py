"""Plain PyTorch fp32 restatement of the reference's ``attention()`` -- TEST INFRASTRUCTURE.

models/transformer_captioner.py:27-37:
    scores = Q K^T / sqrt(d_k); scores.masked_fill(mask == 0, -1e9); p = softmax(scores, -1);
    p = dropout(p); return p V, p
Used (a) by tests as the checker of the fused HIP attention kernel, (b) by bench.py's cpu_baseline leg.
Runs on any device torch supports; never imported by spacap3d_amd.
"""
import math

import torch
import torch.nn.functional as F


def attention_logits(query, key, mask=None, bias=None):
    d_k = query.size(-1)
    scores = torch.matmul(query, key.transpose(-2, -1)) / math.sqrt(d_k)
    if bias is not None:
        scores = scores + bias
    if mask is not None:
        scores = scores.masked_fill(mask == 0, -1e9)
    return scores


def attention(query, key, value, mask=None, dropout_p=0.0, training=False, need_p=True, bias=None):
    p_attn = F.softmax(attention_logits(query, key, mask, bias), dim=-1)
    if training and dropout_p > 0.0:
        p_attn = F.dropout(p_attn, dropout_p, True)
    return torch.matmul(p_attn, value), p_attn


def relation_feature(p_attn, value):
    """models/transformer_captioner.py:393-396, literally (repeat, product, two transposes, view)."""
    a = p_attn.unsqueeze(-1).repeat(1, 1, 1, 1, value.shape[-1])
    v = value.unsqueeze(-3)
    B, H, _, K, D = v.shape
    return (a * v).transpose(1, 2).transpose(2, 3).contiguous().view(B, K, K, H * D)


def layer_norm(x, a, b, eps=1e-6):
    """models/transformer_captioner.py:110-113, literally."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return a * (x - mean) / (std + eps) + b


def bn_relu_train(z, bn, pool_S=None):
    """BatchNorm (batch statistics) -> ReLU [-> max over the trailing samples] with stock torch ops: the tail of a
    SharedMLP layer (lib/pointnet2/pytorch_utils.py:11-36) and the pooling of pointnet2_modules.py:256-259."""
    y = F.relu(bn(z))
    if pool_S:
        y = F.max_pool2d(y, kernel_size=[1, y.size(3)]).squeeze(-1)
    return y


def relation_layer1(P, V, weight, bias):
    """relu(Linear(relation_feature(P, V))) as the reference computes it (:393-397 + first two modules of :319-326)."""
    return F.relu(F.linear(relation_feature(P, V), weight, bias))


class OracleBackend:
    """`backend` object for spacap3d_amd.backend.use_backend(): oracle ops + torch attention, CPU tensors."""

    name = "oracle"

    def __init__(self, openmp=False):
        from .ext_cpu import OracleExt
        self._ext = OracleExt(openmp=openmp)
        for n in ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
                  "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
                  "group_points_grad"):
            setattr(self, n, getattr(self._ext, n))
        self.attention = attention
        self.layer_norm = layer_norm
        self.relation_feature = relation_feature
        self.relation_layer1 = relation_layer1
        self.bn_relu_train = bn_relu_train

    # max over the samples of a group: F.max_pool2d(x, [1, S]) of pointnet2_modules.py:256-259 (first maximum wins)
    @staticmethod
    def group_max(x):
        v, i = torch.max(x, dim=3)
        return v, i.to(torch.uint8)

    @staticmethod
    def group_max_grad(grad_out, arg, S):
        gi = torch.zeros(*grad_out.shape, int(S), dtype=grad_out.dtype, device=grad_out.device)
        return gi.scatter_(3, arg.long().unsqueeze(-1), grad_out.unsqueeze(-1))
