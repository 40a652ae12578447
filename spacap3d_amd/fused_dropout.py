"""dropout(relu(x)) and res + dropout(y) as single launches each way (csrc/elementwise.hip).

Counterparts of ``self.dropout(F.relu(self.w_1(x)))`` (models/transformer_captioner.py:126) and of
``x + self.dropout(sublayer(self.norm(x)))`` (:115-123).  The keep mask comes from a counter hash (host seed per
call + the device-resident step counter of ``attention.rng_state``), regenerated in the backward.
"""
import torch
from torch.autograd import Function

from ._native import check, lib
from .attention import _next_seed, rng_state


def _st(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class ReluDropout(Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = x.contiguous()
        with torch.cuda.device(x.device):
            y = torch.empty_like(x)
            check(lib.spacap_relu_dropout_fwd_f32(x.data_ptr(), x.numel(), float(p), int(seed),
                                                  rng_state(x.device).data_ptr() if p > 0.0 else None, y.data_ptr(),
                                                  _st(x)), "spacap_relu_dropout_fwd_f32")
        ctx.save_for_backward(y)
        ctx.p = float(p)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        with torch.cuda.device(y.device):
            dx = torch.empty_like(y)
            check(lib.spacap_relu_dropout_bwd_f32(g.data_ptr(), y.data_ptr(), y.numel(), ctx.p, dx.data_ptr(), _st(y)),
                  "spacap_relu_dropout_bwd_f32")
        return dx, None, None


class DropoutAdd(Function):
    @staticmethod
    def forward(ctx, res, y, p, seed):
        res, y = res.contiguous(), y.contiguous()
        with torch.cuda.device(y.device):
            out = torch.empty_like(y)
            check(lib.spacap_dropout_add_fwd_f32(res.data_ptr(), y.data_ptr(), y.numel(), float(p), int(seed),
                                                 rng_state(y.device).data_ptr() if p > 0.0 else None, out.data_ptr(),
                                                 _st(y)), "spacap_dropout_add_fwd_f32")
        ctx.p, ctx.seed = float(p), int(seed)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        with torch.cuda.device(g.device):
            dy = torch.empty_like(g)
            check(lib.spacap_dropout_add_bwd_f32(g.data_ptr(), g.numel(), ctx.p, ctx.seed,
                                                 rng_state(g.device).data_ptr() if ctx.p > 0.0 else None, dy.data_ptr(),
                                                 _st(g)), "spacap_dropout_add_bwd_f32")
        return g, dy, None, None


def relu_dropout(x, p, training):
    """dropout(relu(x), p) -- plain relu when not training or p == 0."""
    if not training or p <= 0.0:
        return torch.relu(x)
    return ReluDropout.apply(x, float(p), _next_seed())


def dropout_add(res, y, p, training):
    """res + dropout(y, p) -- a plain add when not training or p == 0."""
    if not training or p <= 0.0 or res.shape != y.shape:
        return res + (torch.nn.functional.dropout(y, p, training) if training and p > 0.0 else y)
    return DropoutAdd.apply(res, y, float(p), _next_seed())
