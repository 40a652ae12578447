"""Train-mode BatchNorm + ReLU (+ max over the samples of a group) as one autograd op on libspacap_hip.so.

Counterpart of the tail of every ``SharedMLP`` layer of the reference (lib/pointnet2/pytorch_utils.py:11-36:
``Conv2d(1x1) -> BatchNorm2d -> ReLU(inplace)``) and of the pooling that follows the last one
(lib/pointnet2/pointnet2_modules.py:256-259).  See ``csrc/bn_relu.hip`` for the pass structure.
"""
import weakref

import torch
from torch.autograd import Function

from . import selections
from ._native import check, lib


def _ws(C, device):
    return torch.empty(int(lib.spacap_bn_workspace_bytes(C)), dtype=torch.uint8, device=device)


class BNReLU(Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, momentum, eps, pool_S):
        if not z.is_cuda:
            raise RuntimeError("CPU not supported")
        z = z.contiguous()
        B, C = z.shape[0], z.shape[1]
        L = z.numel() // (B * C)
        dev = z.device
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            stats = torch.empty(C, 2, dtype=torch.float32, device=dev)
            ws = _ws(C, dev)
            rm = running_mean.data_ptr() if running_mean is not None else None
            rv = running_var.data_ptr() if running_var is not None else None
            if not pool_S:
                # statistics + apply in one call (one launch when a channel has <= 32 768 elements)
                out = torch.empty_like(z)
                check(lib.spacap_bn_relu_train_f32(z.data_ptr(), B, C, L, float(eps), float(momentum), rm, rv, gamma.data_ptr(),
                                                   beta.data_ptr(), stats.data_ptr(), out.data_ptr(), ws.data_ptr(), st),
                      "spacap_bn_relu_train_f32")
                selections.visit("bn_relu", gammas=[gamma], zs=[z], stats=[stats], beta=beta)
                ctx.save_for_backward(z, stats, gamma, beta)
                ctx.pool_S = 0
                return out
            check(lib.spacap_bn_stats_f32(z.data_ptr(), B, C, L, float(eps), float(momentum), rm, rv,
                                          stats.data_ptr(), ws.data_ptr(), st), "spacap_bn_stats_f32")
            if pool_S:
                S = int(pool_S)
                P = L // S
                out = torch.empty(B, C, P, dtype=torch.float32, device=dev)
                arg = torch.empty(B, C, P, dtype=torch.uint8, device=dev)
                check(lib.spacap_bn_relu_max_f32(z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), B, C,
                                                 P, S, out.data_ptr(), arg.data_ptr(), st), "spacap_bn_relu_max_f32")
                ctx.save_for_backward(z, stats, gamma, beta, arg)
            else:
                out = torch.empty_like(z)
                check(lib.spacap_bn_relu_apply_f32(z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), B,
                                                   C, L, out.data_ptr(), st), "spacap_bn_relu_apply_f32")
                selections.visit("bn_relu", gammas=[gamma], zs=[z], stats=[stats], beta=beta)
                ctx.save_for_backward(z, stats, gamma, beta)
        ctx.pool_S = int(pool_S) if pool_S else 0
        return out

    @staticmethod
    def backward(ctx, d_out):
        z, stats, gamma, beta = ctx.saved_tensors[:4]
        B, C = z.shape[0], z.shape[1]
        L = z.numel() // (B * C)
        dev = z.device
        st = torch.cuda.current_stream(dev).cuda_stream
        d_out = d_out.contiguous()
        with torch.cuda.device(dev):
            dz = torch.empty_like(z)
            dg = torch.empty(C, dtype=torch.float32, device=dev)
            db = torch.empty(C, dtype=torch.float32, device=dev)
            ws = _ws(C, dev)
            if ctx.pool_S:
                arg = ctx.saved_tensors[4]
                S = ctx.pool_S
                check(lib.spacap_bn_relu_max_bwd_f32(z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                     d_out.data_ptr(), arg.data_ptr(), B, C, L // S, S, dz.data_ptr(),
                                                     dg.data_ptr(), db.data_ptr(), ws.data_ptr(), st),
                      "spacap_bn_relu_max_bwd_f32")
            else:
                check(lib.spacap_bn_relu_bwd_f32(z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                 d_out.data_ptr(), B, C, L, dz.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                                 ws.data_ptr(), st), "spacap_bn_relu_bwd_f32")
        return dz, dg, db, None, None, None, None, None


# BatchNorm modules whose training forward went through the fused ops (i.e. through bump_counter): only those may hand their
# counter to a training engine -- a layer on the stock nn.BatchNorm fallback increments its own buffer
import weakref
FUSED_SEEN = weakref.WeakSet()


def bump_counter(bn):
    """bn.num_batches_tracked += 1 -- unless a training engine has taken the counter over (the buffer then carries the
    attribute ``_spacap_deferred``: engine.Trainer._adopt_bn_counters bumps all of them with one add per step instead
    of one tiny launch per layer; with momentum set the counter never enters the arithmetic)."""
    FUSED_SEEN.add(bn)
    t = bn.num_batches_tracked
    if bn.track_running_stats and t is not None and not getattr(t, "_spacap_deferred", False):
        t.add_(1)


def bn_relu_train(z, bn, pool_S=None):
    """z (B,C,...) -> relu(batch_norm(z)) with batch statistics; with ``pool_S`` the trailing dimension (size
    pool_S) is max-reduced as well ((B,C,P,S) -> (B,C,P)).  ``bn`` is the torch BatchNorm module whose
    parameters / running statistics are used and updated (momentum semantics of torch.nn.BatchNorm)."""
    bump_counter(bn)
    momentum = 0.0 if bn.momentum is None else bn.momentum
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return BNReLU.apply(z, bn.weight, bn.bias, rm, rv, momentum, bn.eps, pool_S)


# ---- inference: relu(batch_norm(z)) on the running statistics, one launch of the library's apply kernel ------------------------
# (the stock modules run MIOpen's BatchNorm inference kernel + a ReLU: two launches per layer and the last library kernel of the
# inference forward).  The kernel takes (mean, 1 / sqrt(var + eps)) per channel: folded once per state of the module's buffers.
# The fold is kept per MODULE OBJECT (weakly) and is valid for the buffer OBJECTS it was made from at the versions they had: a
# training step's in-place update, load_state_dict (in-place copies) or .to() (new tensors) make a new fold.  Addresses, ids and
# version numbers alone are not identities -- a later module of the same shape reuses all three (seen: a stale fold handed to the
# next parametrisation of a test).
_EVAL_FOLD = weakref.WeakKeyDictionary()    # module -> (weak refs to the two buffers, their versions, eps, device, folded stats)


def bn_relu_eval(z, bn):
    """relu(bn(z)) for a BatchNorm module in eval mode (running statistics) on a float32 CUDA tensor z (B,C,...), without gradient
    recording; ``None`` when that does not apply (the caller then calls the modules)."""
    if not (z.is_cuda and z.dtype == torch.float32 and not torch.is_grad_enabled() and not bn.training and bn.track_running_stats
            and bn.running_mean is not None and bn.affine and z.dim() >= 2 and z.shape[1] == bn.num_features and z.numel() > 0):
        return None
    rm, rv = bn.running_mean, bn.running_var
    ent = _EVAL_FOLD.get(bn)
    # valid for THESE tensor objects at THESE versions (addresses and ids are reused by later modules: not identities)
    if ent is None or ent[0]() is not rm or ent[1]() is not rv or ent[2] != (rm._version, rv._version, float(bn.eps), z.device):
        with torch.cuda.device(z.device):
            stats = torch.stack([rm.float(), torch.rsqrt(rv.float() + bn.eps)], 1).contiguous()
        ent = _EVAL_FOLD[bn] = (weakref.ref(rm), weakref.ref(rv), (rm._version, rv._version, float(bn.eps), z.device), stats)
    stats = ent[3]
    z = z.contiguous()
    B, C = z.shape[0], z.shape[1]
    L = z.numel() // (B * C)
    with torch.cuda.device(z.device):
        out = torch.empty_like(z)
        check(lib.spacap_bn_relu_apply_f32(z.data_ptr(), stats.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), B, C, L,
                                           out.data_ptr(), torch.cuda.current_stream(z.device).cuda_stream), "spacap_bn_relu_apply_f32")
    return out
