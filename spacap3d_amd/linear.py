"""y = x W^T + b with a one-launch weight + bias gradient (csrc/sa_mlp.hip: linear_wgrad_kernel).

Counterpart of the ``nn.Linear`` projections / feed-forward layers of the reference Transformer
(models/transformer_captioner.py:63-99, 117-126).  Forward and dX stay ordinary BLAS GEMMs; the backward's
``dW = g^T x`` and ``db = sum_r g`` -- for <= 2 048 rows a memset + a split-K GEMM + a column-sum kernel of
~30 us of latency -- become one kernel producing per-slab partials of both plus one sum over the slabs.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from ._native import check, lib, sum_slabs


class FusedLinear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        CK, CP = weight.shape
        g2 = g.reshape(-1, CK)
        x2 = x.reshape(-1, CP)
        R = g2.shape[0]
        dx = (g2 @ weight).view_as(x) if ctx.needs_input_grad[0] else None
        nslab = int(lib.spacap_linear_wgrad_slabs(R, CK, CP)) if g2.is_cuda else 0
        if nslab == 0:
            return dx, g2.t() @ x2, g2.sum(0)
        g2, x2 = g2.contiguous(), x2.contiguous()
        with torch.cuda.device(g2.device):
            part = torch.empty(nslab, CK * CP + CK, dtype=torch.float32, device=g2.device)
            check(lib.spacap_linear_wgrad_f32(g2.data_ptr(), x2.data_ptr(), R, CK, CP, 1, part.data_ptr(),
                                              torch.cuda.current_stream(g2.device).cuda_stream), "spacap_linear_wgrad_f32")
            s = sum_slabs(part)
        return dx, s[:CK * CP].view(CK, CP), s[CK * CP:]


def linear(x, weight, bias):
    """F.linear with the fused weight/bias gradient (bias required)."""
    return FusedLinear.apply(x, weight, bias)
