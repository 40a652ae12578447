"""Fused scaled-dot-product attention on libspacap_hip.so -- boundary B2 of SURVEY.md section 8b.

Replaces the four-kernel Python ``attention()`` of the reference
(models/transformer_captioner.py:27-37:  QK^T/sqrt(d_k) -> masked_fill(mask==0, -1e9) -> softmax ->
dropout -> PV) with one launch of ``spacap_mha_fwd_f32`` and one of ``spacap_mha_bwd_f32``.

``attention(query, key, value, mask, dropout_p, training, need_p)`` returns ``(out, p_attn)`` like the
reference: ``out`` is (B,h,Lq,d_k) (a transposed view of a dense (B,Lq,h,d_k) buffer, so the caller's
``x.transpose(1,2).contiguous()`` is free) and ``p_attn`` the post-dropout (B,h,Lq,Lk) matrix the
reference stores as ``self.attn`` -- or None when ``need_p`` is False, in which case the L x L matrix
never leaves registers.  Gradients flow to q, k, v from both outputs (the relation head consumes
``p_attn`` of the last encoder layer, models/transformer_captioner.py:392-394).
"""
import math

import torch
from torch.autograd import Function

from ._native import check, lib, sum_slabs


def _strides3(t):
    assert t.stride(3) == 1, "last (d_k) dimension must be contiguous"
    return t.stride(0), t.stride(1), t.stride(2)


def _prep_mask(mask, B, Lq, Lk):
    """reference masks: (B,1,1,Lk) int64 key mask or (B,1,Lq,Lk) bool -> uint8 (B,Lq|1,Lk)."""
    if mask is None:
        return None, 0, 0
    # the same mask object goes through every layer of a stack: convert it once (2 launches saved per layer)
    for ent in _MASK_CACHE:
        if ent[0] is mask and ent[1] == (mask._version, B, Lq, Lk):
            return ent[2]
    res = _prep_mask_uncached(mask, B, Lq, Lk)
    _MASK_CACHE.insert(0, (mask, (mask._version, B, Lq, Lk), res))
    del _MASK_CACHE[4:]
    return res


_MASK_CACHE = []


def _prep_mask_uncached(mask, B, Lq, Lk):
    m = mask
    if m.dim() == 4:
        assert m.size(1) == 1, "per-head masks are not used by the reference"
        m = m[:, 0]
    assert m.dim() == 3 and m.size(-1) == Lk and m.size(1) in (1, Lq)
    if m.dtype == torch.uint8 and m.size(0) == B and m.is_contiguous():   # already in the kernels' form (caption_prep)
        return m, m.stride(0), (0 if m.size(1) == 1 else m.stride(1))
    m = (m != 0).to(torch.uint8)
    if m.size(0) != B:
        m = m.expand(B, -1, -1)
    m = m.contiguous()
    return m, m.stride(0), (0 if m.size(1) == 1 else m.stride(1))


# ---- dropout randomness ----------------------------------------------------------------------------------
# The keep mask is a counter hash of (seed, element).  `seed` is a host integer drawn per call from a Python
# counter seeded by torch's CPU generator; it would be frozen inside a captured hipGraph, so a device-resident
# word (`_RNG_STATE[device]`, bumped once per training step by an in-graph add) is mixed in by the kernel.
_RNG_STATE = {}
_CALL_COUNTER = [None]


def rng_state(device):
    device = torch.device(device)     # ("cuda:0" and torch.device("cuda", 0) must name the same word)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    st = _RNG_STATE.get(device)
    if st is None:
        st = torch.zeros(1, dtype=torch.int64, device=device)
        _RNG_STATE[device] = st
    return st


def advance_rng(device):
    """Call once per step (captured into the step's graph): every attention call then sees new dropout masks."""
    rng_state(device).add_(1)


def _next_seed():
    if _CALL_COUNTER[0] is None:
        _CALL_COUNTER[0] = int(torch.randint(0, 2 ** 31, (1,), device="cpu").item()) << 20
    _CALL_COUNTER[0] += 1
    return _CALL_COUNTER[0]


class FusedAttention(Function):
    @staticmethod
    def forward(ctx, q, k, v, mask_u8, mask_sb, mask_sq, bias, dropout_p, seed, need_p):
        if not q.is_cuda:
            raise RuntimeError("CPU not supported")
        B, h, Lq, dk = q.shape
        Lk = k.shape[2]
        if q.stride(3) != 1:
            q = q.contiguous()
        if k.stride(3) != 1:
            k = k.contiguous()
        if v.stride(3) != 1:
            v = v.contiguous()
        scale = 1.0 / math.sqrt(dk)
        with torch.cuda.device(q.device):
            out = torch.empty(B, Lq, h, dk, dtype=torch.float32, device=q.device)
            p = torch.empty(B, h, Lq, Lk, dtype=torch.float32, device=q.device) if need_p else None
            lse = torch.empty(B, h, Lq, 2, dtype=torch.float32, device=q.device)  # (row max, row sum)
            bs = (bias.stride(0), bias.stride(1), bias.stride(2)) if bias is not None else (0, 0, 0)
            check(lib.spacap_mha_fwd_f32(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), *_strides3(q), *_strides3(k), *_strides3(v),
                mask_u8.data_ptr() if mask_u8 is not None else None, mask_sb, mask_sq,
                bias.data_ptr() if bias is not None else None, *bs,
                B, h, Lq, Lk, dk, scale, float(dropout_p), int(seed),
                rng_state(q.device).data_ptr() if dropout_p > 0.0 else None,
                out.data_ptr(), p.data_ptr() if need_p else None, lse.data_ptr(),
                torch.cuda.current_stream(q.device).cuda_stream), "spacap_mha_fwd_f32")
        ctx.save_for_backward(q, k, v, mask_u8, bias, lse)
        ctx.set_materialize_grads(False)   # no zero tensor for the unused second output's gradient
        ctx.meta = (mask_sb, mask_sq, dropout_p, seed, scale, need_p)
        out_v = out.transpose(1, 2)
        if need_p:
            return out_v, p
        ctx.mark_non_differentiable(lse)
        return out_v, lse  # second output unused by callers when need_p is False

    @staticmethod
    def backward(ctx, d_out, d_p):
        q, k, v, mask_u8, bias, lse = ctx.saved_tensors
        mask_sb, mask_sq, dropout_p, seed, scale, need_p = ctx.meta
        B, h, Lq, dk = q.shape
        Lk = k.shape[2]
        if d_out is None:   # only the attention map was used downstream
            d_out = torch.zeros(B, h, Lq, dk, dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            d_out_c = d_out.transpose(1, 2).contiguous()  # (B, Lq, h, dk)
            d_p_c = d_p.contiguous() if (need_p and d_p is not None) else None
            dq = torch.empty(B, Lq, h, dk, dtype=torch.float32, device=q.device)
            dk_ = torch.empty(B, Lk, h, dk, dtype=torch.float32, device=q.device)
            dv = torch.empty(B, Lk, h, dk, dtype=torch.float32, device=q.device)
            ws = torch.empty(max(int(lib.spacap_mha_bwd_workspace_bytes(B, h, Lq)), 16), dtype=torch.uint8,
                             device=q.device)
            bs = (bias.stride(0), bias.stride(1), bias.stride(2)) if bias is not None else (0, 0, 0)
            check(lib.spacap_mha_bwd_f32(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), *_strides3(q), *_strides3(k), *_strides3(v),
                mask_u8.data_ptr() if mask_u8 is not None else None, mask_sb, mask_sq,
                bias.data_ptr() if bias is not None else None, *bs,
                B, h, Lq, Lk, dk, scale, float(dropout_p), int(seed),
                rng_state(q.device).data_ptr() if dropout_p > 0.0 else None, lse.data_ptr(),
                d_out_c.data_ptr(), d_p_c.data_ptr() if d_p_c is not None else None, ws.data_ptr(),
                dq.data_ptr(), dk_.data_ptr(), dv.data_ptr(), 0,
                torch.cuda.current_stream(q.device).cuda_stream), "spacap_mha_bwd_f32")
        return (dq.transpose(1, 2), dk_.transpose(1, 2), dv.transpose(1, 2), None, None, None, None, None, None,
                None)


class FusedSelfAttentionPacked(Function):
    """Self-attention on a packed projection qkv (B, L, 3*h*d_k) = [q | k | v] (one GEMM instead of three): the
    kernels read q, k, v through strides and the backward writes dq, dk, dv straight into one (B, L, 3*h*d_k)
    buffer, so the projection's backward is one dX GEMM, one dW GEMM and one bias sum instead of three each plus
    two accumulations.  Returns (out (B,L,h*d_k), p_attn (B,h,L,L) or the row statistics)."""

    @staticmethod
    def forward(ctx, qkv, h, mask_u8, mask_sb, mask_sq, dropout_p, seed, need_p):
        if not qkv.is_cuda:
            raise RuntimeError("CPU not supported")
        qkv = qkv.contiguous()
        B, L, three_hd = qkv.shape
        hd = three_hd // 3
        dk = hd // h
        scale = 1.0 / math.sqrt(dk)
        es = qkv.element_size()
        strides = (L * three_hd, dk, three_hd)
        with torch.cuda.device(qkv.device):
            out = torch.empty(B, L, hd, dtype=torch.float32, device=qkv.device)
            p = torch.empty(B, h, L, L, dtype=torch.float32, device=qkv.device) if need_p else None
            lse = torch.empty(B, h, L, 2, dtype=torch.float32, device=qkv.device)
            base = qkv.data_ptr()
            check(lib.spacap_mha_fwd_f32(
                base, base + hd * es, base + 2 * hd * es, *strides, *strides, *strides,
                mask_u8.data_ptr() if mask_u8 is not None else None, mask_sb, mask_sq, None, 0, 0, 0,
                B, h, L, L, dk, scale, float(dropout_p), int(seed),
                rng_state(qkv.device).data_ptr() if dropout_p > 0.0 else None,
                out.data_ptr(), p.data_ptr() if need_p else None, lse.data_ptr(),
                torch.cuda.current_stream(qkv.device).cuda_stream), "spacap_mha_fwd_f32")
        ctx.save_for_backward(qkv, mask_u8, lse)
        ctx.set_materialize_grads(False)   # no zero tensor for the unused second output's gradient
        ctx.meta = (h, mask_sb, mask_sq, dropout_p, seed, scale, need_p)
        if need_p:
            return out, p
        ctx.mark_non_differentiable(lse)
        return out, lse

    @staticmethod
    def backward(ctx, d_out, d_p):
        qkv, mask_u8, lse = ctx.saved_tensors
        h, mask_sb, mask_sq, dropout_p, seed, scale, need_p = ctx.meta
        B, L, three_hd = qkv.shape
        hd = three_hd // 3
        dk = hd // h
        es = qkv.element_size()
        strides = (L * three_hd, dk, three_hd)
        if d_out is None:   # only the attention map was used downstream
            d_out = torch.zeros(B, L, hd, dtype=torch.float32, device=qkv.device)
        with torch.cuda.device(qkv.device):
            d_out_c = d_out.contiguous()
            d_p_c = d_p.contiguous() if (need_p and d_p is not None) else None
            dqkv = torch.empty_like(qkv)
            ws = torch.empty(max(int(lib.spacap_mha_bwd_workspace_bytes(B, h, L)), 16), dtype=torch.uint8,
                             device=qkv.device)
            base, gb = qkv.data_ptr(), dqkv.data_ptr()
            check(lib.spacap_mha_bwd_f32(
                base, base + hd * es, base + 2 * hd * es, *strides, *strides, *strides,
                mask_u8.data_ptr() if mask_u8 is not None else None, mask_sb, mask_sq, None, 0, 0, 0,
                B, h, L, L, dk, scale, float(dropout_p), int(seed),
                rng_state(qkv.device).data_ptr() if dropout_p > 0.0 else None, lse.data_ptr(),
                d_out_c.data_ptr(), d_p_c.data_ptr() if d_p_c is not None else None, ws.data_ptr(),
                gb, gb + hd * es, gb + 2 * hd * es, three_hd,
                torch.cuda.current_stream(qkv.device).cuda_stream), "spacap_mha_bwd_f32")
        return dqkv, None, None, None, None, None, None, None


def self_attention_packed(qkv, h, mask=None, dropout_p=0.0, training=False, need_p=True):
    """qkv (B,L,3*h*d_k) packed [q | k | v] -> (out (B,L,h*d_k), p_attn (B,h,L,L) or None)."""
    B, L, _ = qkv.shape
    m, msb, msq = _prep_mask(mask, B, L, L)
    p = float(dropout_p) if training else 0.0
    seed = _next_seed() if p > 0.0 else 0
    out, second = FusedSelfAttentionPacked.apply(qkv, h, m, msb, msq, p, seed, need_p)
    return out, (second if need_p else None)


def attention(query, key, value, mask=None, dropout_p=0.0, training=False, need_p=True, bias=None):
    """(out (B,h,Lq,d_k), p_attn (B,h,Lq,Lk) or None); see the module docstring."""
    B, h, Lq, dk = query.shape
    Lk = key.shape[2]
    m, msb, msq = _prep_mask(mask, B, Lq, Lk)
    p = float(dropout_p) if training else 0.0
    seed = _next_seed() if p > 0.0 else 0
    out, second = FusedAttention.apply(query, key, value, m, msb, msq, bias, p, seed, need_p)
    return out, (second if need_p else None)


def _ln_backward(xc, a, stats, dyc, add, eps):
    """dx, da, db of the LayerNorm; the (da, db) column sums over the workgroup partials go through ``sum_slabs``
    (same 4-group order as the library's own reduction kernel) so that a training step can batch them with the
    weight-gradient sums (deferred_slab_sums)."""
    from ._native import sum_slabs
    D = xc.shape[-1]
    rows = xc.numel() // D
    with torch.cuda.device(xc.device):
        dx = torch.empty_like(xc)
        nbytes = int(lib.spacap_layernorm_bwd_workspace_bytes(rows, D))
        if D <= 512 and D % 2 == 0 and nbytes >= 8 * D:
            part = torch.empty(nbytes // (8 * D), 2 * D, dtype=torch.float32, device=xc.device)
            check(lib.spacap_layernorm_bwd_add_f32(xc.data_ptr(), a.data_ptr(), stats.data_ptr(), dyc.data_ptr(),
                                                   add.data_ptr() if add is not None else None, rows, D, eps,
                                                   dx.data_ptr(), None, None, part.data_ptr(),
                                                   torch.cuda.current_stream(xc.device).cuda_stream),
                  "spacap_layernorm_bwd_add_f32")
            s = sum_slabs(part, deferrable=True)
            return dx, s[:D], s[D:]
        da = torch.empty(D, dtype=torch.float32, device=xc.device)
        db = torch.empty(D, dtype=torch.float32, device=xc.device)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=xc.device)
        check(lib.spacap_layernorm_bwd_add_f32(xc.data_ptr(), a.data_ptr(), stats.data_ptr(), dyc.data_ptr(),
                                               add.data_ptr() if add is not None else None, rows, D, eps,
                                               dx.data_ptr(), da.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                               torch.cuda.current_stream(xc.device).cuda_stream),
              "spacap_layernorm_bwd_add_f32")
    return dx, da, db


class FusedLayerNorm(Function):
    """a * (x - mean) / (std_unbiased + eps) + b  (models/transformer_captioner.py:102-113) in one launch
    forward and two backward (spacap_layernorm_*_f32)."""

    @staticmethod
    def forward(ctx, x, a, b, eps):
        if not x.is_cuda:
            raise RuntimeError("CPU not supported")
        xc = x.contiguous()
        D = xc.shape[-1]
        rows = xc.numel() // D
        with torch.cuda.device(x.device):
            y = torch.empty_like(xc)
            stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
            check(lib.spacap_layernorm_fwd_f32(xc.data_ptr(), a.data_ptr(), b.data_ptr(), rows, D, float(eps),
                                               y.data_ptr(), stats.data_ptr(),
                                               torch.cuda.current_stream(x.device).cuda_stream), "spacap_layernorm_fwd_f32")
        ctx.save_for_backward(xc, a, stats)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, a, stats = ctx.saved_tensors
        dx, da, db = _ln_backward(xc, a, stats, dy.contiguous(), None, ctx.eps)
        return dx, da, db, None


def layer_norm(x, a, b, eps=1e-6):
    return FusedLayerNorm.apply(x, a, b, eps)


class FusedLayerNormResidual(Function):
    """(norm(x), x) for the pre-norm residual block ``x + dropout(sublayer(norm(x)))``
    (models/transformer_captioner.py:115-123): the second output is x itself, to be used as the residual operand, so
    that BOTH gradient paths into x arrive at this node and the backward kernel adds them while it writes dx
    (spacap_layernorm_bwd_add_f32) -- otherwise autograd sums them with one more pass per sub-layer."""

    @staticmethod
    def forward(ctx, x, a, b, eps):
        if not x.is_cuda:
            raise RuntimeError("CPU not supported")
        xc = x.contiguous()
        D = xc.shape[-1]
        rows = xc.numel() // D
        with torch.cuda.device(x.device):
            y = torch.empty_like(xc)
            stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
            check(lib.spacap_layernorm_fwd_f32(xc.data_ptr(), a.data_ptr(), b.data_ptr(), rows, D, float(eps),
                                               y.data_ptr(), stats.data_ptr(),
                                               torch.cuda.current_stream(x.device).cuda_stream), "spacap_layernorm_fwd_f32")
        ctx.save_for_backward(xc, a, stats)
        ctx.eps = float(eps)
        ctx.set_materialize_grads(False)
        return y, xc.view_as(xc)

    @staticmethod
    def backward(ctx, dy, dres):
        xc, a, stats = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None
        dx, da, db = _ln_backward(xc, a, stats, dy.contiguous(), dres.contiguous() if dres is not None else None, ctx.eps)
        return dx, da, db, None


def layer_norm_residual(x, a, b, eps=1e-6):
    """(norm(x), residual operand) -- see FusedLayerNormResidual."""
    return FusedLayerNormResidual.apply(x, a, b, eps)


class RelationFeature(Function):
    """R[b,i,j,h*D+d] = P[b,h,i,j] * V[b,h,j,d]  (models/transformer_captioner.py:393-396) in one launch each way."""

    @staticmethod
    def forward(ctx, P, V):
        if not P.is_cuda:
            raise RuntimeError("CPU not supported")
        P = P.contiguous()
        if V.stride(3) != 1 or any(s % 4 for s in V.stride()[:3]) or V.data_ptr() % 16:
            V = V.contiguous()
        B, H, K, D = V.shape
        with torch.cuda.device(P.device):
            R = torch.empty(B, K, K, H * D, dtype=torch.float32, device=P.device)
            check(lib.spacap_relation_feature_fwd_f32(P.data_ptr(), V.data_ptr(), V.stride(0), V.stride(1), V.stride(2),
                                                      B, H, K, D, R.data_ptr(),
                                                      torch.cuda.current_stream(P.device).cuda_stream),
                  "spacap_relation_feature_fwd_f32")
        ctx.save_for_backward(P, V)
        return R

    @staticmethod
    def backward(ctx, dR):
        P, V = ctx.saved_tensors
        B, H, K, D = V.shape
        dR = dR.contiguous()
        with torch.cuda.device(P.device):
            dP = torch.empty_like(P)
            dV = torch.empty(B, K, H, D, dtype=torch.float32, device=P.device)
            check(lib.spacap_relation_feature_bwd_f32(dR.data_ptr(), P.data_ptr(), V.data_ptr(), V.stride(0), V.stride(1),
                                                      V.stride(2), B, H, K, D, dP.data_ptr(), dV.data_ptr(),
                                                      torch.cuda.current_stream(P.device).cuda_stream),
                  "spacap_relation_feature_bwd_f32")
        return dP, dV.transpose(1, 2)


def relation_feature(P, V):
    return RelationFeature.apply(P, V)


class RelationLayer1(Function):
    """relu(b1 + sum_h P[b,h,i,j] * U[b,j,h,:])  -- see csrc/relation.hip."""

    @staticmethod
    def forward(ctx, P, U, b1):
        P, U = P.contiguous(), U.contiguous()
        B, H, K, _ = P.shape
        C = U.shape[-1]
        with torch.cuda.device(P.device):
            H1 = torch.empty(B, K, K, C, dtype=torch.float32, device=P.device)
            check(lib.spacap_relation_l1_fwd_f32(P.data_ptr(), U.data_ptr(), b1.data_ptr(), B, H, K, C, H1.data_ptr(),
                                                 torch.cuda.current_stream(P.device).cuda_stream), "spacap_relation_l1_fwd_f32")
        ctx.save_for_backward(P, U, H1)
        return H1

    @staticmethod
    def backward(ctx, dH1):
        P, U, H1 = ctx.saved_tensors
        B, H, K, _ = P.shape
        C = U.shape[-1]
        dH1 = dH1.contiguous()
        with torch.cuda.device(P.device):
            dP = torch.empty_like(P)
            dU = torch.empty(int(lib.spacap_relation_l1_isplit()), *U.shape, dtype=torch.float32, device=P.device)
            part = torch.empty(int(lib.spacap_relation_l1_blocks(B, K, C)), C, dtype=torch.float32, device=P.device)
            check(lib.spacap_relation_l1_bwd_f32(dH1.data_ptr(), H1.data_ptr(), P.data_ptr(), U.data_ptr(), B, H, K, C,
                                                 dP.data_ptr(), dU.data_ptr(), part.data_ptr(),
                                                 torch.cuda.current_stream(P.device).cuda_stream), "spacap_relation_l1_bwd_f32")
        return dP, sum_slabs(dU), sum_slabs(part)


def relation_layer1(P, V, weight, bias):
    """First Linear + ReLU of the relation MLP applied to the relation feature of (P, V), without forming the feature:
    P (B,h,K,K), V (B,h,K,d) (any strides), weight (C, h*d), bias (C) -> (B,K,K,C)."""
    if not P.is_cuda:
        raise RuntimeError("CPU not supported")
    B, H, K, D = V.shape
    C = weight.shape[0]
    if not lib.spacap_relation_l1_supported(H, K, C):
        # widths without a fused kernel (cfg5: d_model 512): the feature is formed (one launch) and multiplied by BLAS
        return torch.relu(torch.nn.functional.linear(RelationFeature.apply(P, V), weight, bias))
    U = torch.einsum("bhjd,ohd->bjho", V, weight.view(C, H, D))  # (B,K,H,C): tiny, autograd gives dV and dW1
    return RelationLayer1.apply(P, U, bias)
