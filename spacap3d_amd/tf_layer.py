"""Fused Transformer sub-layers (csrc/tf_layer.hip) as autograd nodes -- everything BETWEEN two attention calls of an
encoder / decoder stack is one launch (models/transformer_captioner.py: SublayerConnection :115-127, LayerNorm :102-113,
PositionwiseFeedForward :72-81, the output / packed q|k|v projections of MultiHeadedAttention :52-70, EncoderLayer
:180-191, DecoderLayer :209-225 in early-guide mode).

Per layer, forward:   attention -> [out-proj + dropout-add + LayerNorm] -> [w_1 + relu + dropout] ->
                      [w_2 + dropout-add + LayerNorm of the NEXT layer + its packed q|k|v projection]
          backward:   [dqkv Wqkv + LayerNorm' + residual + dropout'] -> [(. W_2) * relu/dropout mask] ->
                      [dhid W_1 + LayerNorm' + residual + dropout' + (. Wo)] -> attention backward (two launches)
The weight / bias / LayerNorm-parameter gradients are queued on the step's deferred batch (``_native.deferred_slab_sums``).

Node boundaries follow the BACKWARD kernels.  One private contract: the second output of ``AttnOutFfn1`` (the hidden
activations h) is consumed only by ``Ffn2Ln``, whose backward hands back the gradient w.r.t. the hidden layer's
PRE-activation (relu / dropout mask already applied, one fused launch); ``AttnOutFfn1.backward`` expects exactly that.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from ._native import TfRowsArgs, check, grad_slot, lib, linear_wgrad_partials, sum_slabs

D_MODEL = 128
# tests / A-B measurements switch the fused stacks off here (the per-operator path of transformer_captioner.py then runs)
SPLIT_MAX = int(os.environ.get("SPACAP_TF_SPLIT_MAX", "0"))   # lab knob: cap on the K slices of the split products
ENABLED = os.environ.get("SPACAP_TF_FUSED", "1") != "0"   # (A/B runs of bench.py)


def supported(d_model, d_ff):
    return d_model == D_MODEL and d_ff >= 128 and d_ff % 128 == 0


def _p(t):
    return t.data_ptr() if t is not None else None


def _rows(mode, R, dev, **kw):
    """One launch of the row-tile kernel (include/spacap_hip.h: spacap_tf_rows_args); tensors by keyword."""
    a = TfRowsArgs()
    a.mode, a.R = mode, R
    a.k1 = int(kw.pop("k1", 0))
    a.n2 = int(kw.pop("n2", 0))
    a.nparts = int(kw.pop("nparts", 0))
    a.lq = int(kw.pop("lq", 0))
    a.drop_p = float(kw.pop("drop_p", 0.0))
    a.eps = float(kw.pop("eps", 1e-6))
    a.seed = int(kw.pop("seed", 0))
    a.seed_dev = None
    if a.drop_p > 0.0:
        from .attention import rng_state
        a.seed_dev = rng_state(dev).data_ptr()
    for k, v in kw.items():
        setattr(a, k, _p(v))
    check(lib.spacap_tf_rows_f32(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream), "spacap_tf_rows_f32")


def _new(dev, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


def _parts(dev, R):
    return _new(dev, int(lib.spacap_tf_rows_parts(R)), 2 * D_MODEL)


def _split_product(a2, W, trans_w):
    """Partial sums over slices of k of a2 W^T (trans_w, W [128, K]) or a2 W (W [K, 128]) for a2 (R, K), K > 128:
    (parts (S, R, 128), S) -- the row kernel adds the slices in order (``nparts``).  The row kernel alone would run this
    product on R / 16 workgroups with K / 4 dependent steps each; split over K it fills the chip."""
    R, K = a2.shape
    dev = a2.device
    S = int(lib.spacap_tf_gemm_splits(R, K, D_MODEL))
    if SPLIT_MAX:
        S = max(d for d in range(1, min(S, SPLIT_MAX) + 1) if (K // 128) % d == 0)
    parts = _new(dev, S, R, D_MODEL)
    check(lib.spacap_tf_gemm_f32(a2.data_ptr(), W.data_ptr(), R, K, D_MODEL, 1 if trans_w else 0, S, parts.data_ptr(),
                                 torch.cuda.current_stream(dev).cuda_stream), "spacap_tf_gemm_f32")
    return parts, S


def _ln_param_grads(part):
    s = sum_slabs(part, deferrable=True)
    return s[:D_MODEL], s[D_MODEL:]


def _linear_grads(g2, x2, weight):
    """(dW, db) of y = x W^T + b from g2 (R, CK), x2 (R, CP): the batched weight-gradient kernel + one slab sum."""
    CK, CP = g2.shape[1], x2.shape[1]
    part = linear_wgrad_partials(g2, x2, True, deferrable=True)
    if part is None:
        return g2.t() @ x2, g2.sum(0)
    s = sum_slabs(part, deferrable=True, out=grad_slot(weight, CK * CP + CK))
    return s[:CK * CP].view(CK, CP), s[CK * CP:]


def _route_qkv(dw, db, rows):
    """gradients of the packed (3 d, d) weight / (3 d) bias as slices for the three nn.Linear parameters."""
    a, b = rows[0], rows[0] + rows[1]
    return (dw[:a], dw[a:b], dw[b:], db[:a], db[a:b], db[b:])


class LnQkv(Function):
    """(qkv, x) = (LayerNorm(x) Wqkv^T + bqkv, x): the FIRST layer's norm + packed projection.  The second output is x
    itself, to be used as the residual operand, so that both gradient paths into x arrive here and the backward kernel
    adds them while it writes dx.  ``routing`` = the three weights and three biases ``pw`` / ``pb`` are views over (so
    that autograd hands them their slices), or nothing when ``pw`` / ``pb`` take the gradient themselves."""

    @staticmethod
    def forward(ctx, x, ln_a, ln_b, eps, pw, pb, *routing):
        if not x.is_cuda:
            raise RuntimeError("CPU not supported")
        xc = x.contiguous()
        R, dev = xc.numel() // D_MODEL, xc.device
        N2 = pw.shape[0]
        with torch.cuda.device(dev):
            qkv, n, stats = _new(dev, *xc.shape[:-1], N2), _new(dev, R, D_MODEL), _new(dev, R, 2)
            _rows(0, R, dev, res=xc, ln_a=ln_a, ln_b=ln_b, eps=eps, n_out=n, stats=stats, w2=pw, bias2=pb, n2=N2, out2=qkv)
        ctx.save_for_backward(xc, ln_a, n, stats, pw)
        ctx.eps = float(eps)
        ctx.rows = [int(r.shape[0]) for r in routing[:3]]
        ctx.set_materialize_grads(False)
        return qkv, xc.view_as(xc)

    @staticmethod
    def backward(ctx, dqkv, dres):
        xc, ln_a, n, stats, pw = ctx.saved_tensors
        R, dev = xc.numel() // D_MODEL, xc.device
        nrout = 6 if ctx.rows else 0
        if dqkv is None:
            return (dres,) + (None,) * (5 + nrout)
        g2 = dqkv.reshape(R, -1).contiguous()
        with torch.cuda.device(dev):
            dx, part = _new(dev, *xc.shape), _parts(dev, R)
            _rows(1, R, dev, a1=g2, w1=pw, k1=g2.shape[1], x_ln=xc, stats=stats, ln_a=ln_a, eps=ctx.eps,
                  res=dres.contiguous() if dres is not None else None, x_out=dx, part=part)
            da, db = _ln_param_grads(part)
            dw, dbias = _linear_grads(g2, n, pw)
        if ctx.rows:
            return (dx, da, db, None, None, None) + _route_qkv(dw, dbias, ctx.rows)
        return dx, da, db, None, dw, dbias


# ---- split-bf16 feed-forward block for tall inputs (csrc/tf_layer.hip: tf_ffn_bf3_kernel) ---------------------------------------
# The kernel takes the layer's weights as pre-split bf16 piece images.  ``refresh_ffn_pieces(layers)`` (re)builds the images of a
# whole stack with ONE launch -- run_stack / greedy_decode call it at the start of every forward, i.e. after any optimizer update
# and before the backward of the same step -- and registers them under the address of w_1's weight; ``_ffn`` uses them when the
# input has more than FFN_BF3_MIN_ROWS rows and falls back to the fp32-MFMA kernel otherwise (few rows: latency bound either way).
FFN_BF3 = True
FFN_BF3_MIN_ROWS = 512
_FFN_PIECES = {}      # w_1.weight.data_ptr() -> (weakref to that parameter, bf16 piece images of the layer)


def refresh_ffn_pieces(layers):
    """One launch: the four split-bf16 images (W1, W2, W2^T, W1^T) of every layer's feed-forward weights, into a buffer that
    lives with the stack (so a captured hipGraph replays the launch on the same memory)."""
    import ctypes
    import weakref
    ffs = [l.feed_forward for l in layers]
    if not ffs or not FFN_BF3:
        return
    w1s, w2s = [f.w_1.weight for f in ffs], [f.w_2.weight for f in ffs]
    dff, dev = w1s[0].shape[0], w1s[0].device
    if not w1s[0].is_cuda or any(w.shape != (dff, D_MODEL) or not w.is_contiguous() for w in w1s) or \
            any(w.shape != (D_MODEL, dff) or not w.is_contiguous() for w in w2s):
        return
    per = int(lib.spacap_tf_ffn_pieces_elems(dff))
    buf = getattr(ffs[0], "_spacap_ffn_pieces", None)
    if buf is None or buf.shape != (len(ffs), per) or buf.device != dev:
        with torch.cuda.device(dev):
            buf = torch.empty(len(ffs), per, dtype=torch.bfloat16, device=dev)
        ffs[0]._spacap_ffn_pieces = buf
    n = len(ffs)
    arr = ctypes.c_void_p * n
    check(lib.spacap_tf_ffn_split_f32(arr(*[w.data_ptr() for w in w1s]), arr(*[w.data_ptr() for w in w2s]),
                                      arr(*[buf[i].data_ptr() for i in range(n)]), n, dff, torch.cuda.current_stream(dev).cuda_stream),
          "spacap_tf_ffn_split_f32")
    for dead in [k for k, (ref, _) in _FFN_PIECES.items() if ref() is None]:
        del _FFN_PIECES[dead]
    for i, w in enumerate(w1s):
        _FFN_PIECES[w.data_ptr()] = (weakref.ref(w), buf[i])


def _ffn(mode, x2, Wa, Wb, bias, y, dff, p, seed, dev):
    """One launch of the chained feed-forward kernel: (hid (R, dff), parts (dff / 128, R, 128))."""
    from .attention import rng_state
    R = x2.shape[0]
    hid, parts = _new(dev, R, dff), _new(dev, dff // 128, R, D_MODEL)
    st = torch.cuda.current_stream(dev).cuda_stream
    rs = rng_state(dev).data_ptr() if (p > 0.0 and mode == 0) else None
    w1 = Wa if mode == 0 else Wb
    ent = _FFN_PIECES.get(w1.data_ptr()) if (FFN_BF3 and R > FFN_BF3_MIN_ROWS) else None
    if ent is not None and ent[0]() is not None:
        check(lib.spacap_tf_ffn_bf3_f32(mode, x2.data_ptr(), ent[1].data_ptr(), _p(bias), _p(y), R, dff, float(p), int(seed), rs,
                                        hid.data_ptr(), parts.data_ptr(), st), "spacap_tf_ffn_bf3_f32")
        return hid, parts
    check(lib.spacap_tf_ffn_f32(mode, x2.data_ptr(), Wa.data_ptr(), Wb.data_ptr(), _p(bias), _p(y), R, dff, float(p), int(seed),
                                rs, hid.data_ptr(), parts.data_ptr(), st), "spacap_tf_ffn_f32")
    return hid, parts


class AttnOutFfn1(Function):
    """(x1, h, parts):  x1 = x + dropout(a Wo^T + bo);  h = dropout(relu(LayerNorm(x1) W1^T + b1));  parts = the partial sums
    of h W2^T over the 128-wide slices of d_ff (same launch as h: the hidden tile never leaves the chip between the two
    products).  backward(g_x1, g_hpre, g_parts): see the module docstring -- ``g_hpre`` is the gradient w.r.t. the hidden
    PRE-activation and ``g_parts`` the partial sums of g_hpre W1, both produced by ``Ffn2Ln.backward`` in one launch."""

    @staticmethod
    def forward(ctx, a, xres, Wo, bo, ln_a, ln_b, W1, b1, W2, eps, p_sub, p_ffn, seed1, seed2):
        if not a.is_cuda:
            raise RuntimeError("CPU not supported")
        ac, xr = a.contiguous(), xres.contiguous()
        R, dev = xr.numel() // D_MODEL, xr.device
        dff = W1.shape[0]
        with torch.cuda.device(dev):
            x1, n2, stats = _new(dev, *xr.shape), _new(dev, R, D_MODEL), _new(dev, R, 2)
            _rows(0, R, dev, a1=ac, w1=Wo, bias1=bo, k1=D_MODEL, drop_p=p_sub, seed=seed1, res=xr, x_out=x1, ln_a=ln_a,
                  ln_b=ln_b, eps=eps, n_out=n2, stats=stats)
            h, parts = _ffn(0, n2, W1, W2, b1, None, dff, p_ffn, seed2, dev)
        ctx.save_for_backward(ac, x1, n2, stats, ln_a, Wo, W1)
        ctx.meta = (float(eps), float(p_sub), int(seed1))
        ctx.set_materialize_grads(False)
        return x1, h.view(*xr.shape[:-1], dff), parts

    @staticmethod
    def backward(ctx, g_x1, g_hpre, g_parts):
        ac, x1, n2, stats, ln_a, Wo, W1 = ctx.saved_tensors
        eps, p_sub, seed1 = ctx.meta
        R, dev = x1.numel() // D_MODEL, x1.device
        dff = W1.shape[0]
        with torch.cuda.device(dev):
            dx1, dy1, da, part = _new(dev, *x1.shape), _new(dev, R, D_MODEL), _new(dev, *ac.shape), _parts(dev, R)
            kw = dict(x_ln=x1, stats=stats, ln_a=ln_a, eps=eps, res=g_x1.contiguous() if g_x1 is not None else None,
                      x_out=dx1, part=part, drop_p=p_sub, seed=seed1, n_out=dy1, w2=Wo, n2=D_MODEL, out2=da)
            dW1 = db1 = None
            if g_hpre is not None:
                gh = g_hpre.reshape(R, dff).contiguous()
                if g_parts is not None:
                    gp = g_parts.contiguous()
                    _rows(1, R, dev, a1=gp, nparts=gp.shape[0], **kw)
                else:
                    parts, S = _split_product(gh, W1, False)
                    _rows(1, R, dev, a1=parts, nparts=S, **kw)
                dW1, db1 = _linear_grads(gh, n2, W1)
            else:   # the hidden layer was not used downstream: only the residual path carries a gradient
                _rows(1, R, dev, g=torch.zeros(R, D_MODEL, dtype=torch.float32, device=dev), **kw)
            dln_a, dln_b = _ln_param_grads(part)
            dWo, dbo = _linear_grads(dy1, ac.reshape(R, D_MODEL), Wo)
        return da, dx1, dWo, dbo, dln_a, dln_b, dW1, db1, None, None, None, None, None, None


class AttnFfn1(Function):
    """``AttnOutFfn1`` with the self-attention in front of it inside the same node:  a = attention(qkv) (one launch);
    (x1, h, parts) as AttnOutFfn1.  Owning both lets the backward run the attention gradient as ONE launch: the row kernel
    that forms d a = dy Wo also emits delta[b, head, q] = sum_d a d a (the softmax-backward row term), after which the dQ
    and the dK / dV halves are independent (spacap_mha_bwd_delta_f32).  No attention matrix is returned: layers whose
    p_attn is consumed (the relation head's last encoder layer) use the separate attention node."""

    @staticmethod
    def forward(ctx, qkv, xres, mask_u8, mask_sb, mask_sq, heads, p_att, seed_att, Wo, bo, ln_a, ln_b, W1, b1, W2, eps, p_sub,
                p_ffn, seed1, seed2):
        if not qkv.is_cuda:
            raise RuntimeError("CPU not supported")
        import math
        from .attention import rng_state
        qc, xr = qkv.contiguous(), xres.contiguous()
        B, L, three = qc.shape
        hd = three // 3
        dk = hd // heads
        R, dev = B * L, xr.device
        dff = W1.shape[0]
        es = qc.element_size()
        strides = (L * three, dk, three)
        scale = 1.0 / math.sqrt(dk)
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            a, lse = _new(dev, B, L, hd), _new(dev, B, heads, L, 2)
            base = qc.data_ptr()
            check(lib.spacap_mha_fwd_f32(base, base + hd * es, base + 2 * hd * es, *strides, *strides, *strides, _p(mask_u8), mask_sb,
                                         mask_sq, None, 0, 0, 0, B, heads, L, L, dk, scale, float(p_att), int(seed_att),
                                         rng_state(dev).data_ptr() if p_att > 0.0 else None, a.data_ptr(), None, lse.data_ptr(), st),
                  "spacap_mha_fwd_f32")
            x1, n2, stats = _new(dev, *xr.shape), _new(dev, R, D_MODEL), _new(dev, R, 2)
            _rows(0, R, dev, a1=a, w1=Wo, bias1=bo, k1=D_MODEL, drop_p=p_sub, seed=seed1, res=xr, x_out=x1, ln_a=ln_a,
                  ln_b=ln_b, eps=eps, n_out=n2, stats=stats)
            h, parts = _ffn(0, n2, W1, W2, b1, None, dff, p_ffn, seed2, dev)
        ctx.save_for_backward(qc, mask_u8, lse, a, x1, n2, stats, ln_a, Wo, W1)
        ctx.meta = (float(eps), float(p_sub), int(seed1), int(heads), mask_sb, mask_sq, float(p_att), int(seed_att), scale)
        ctx.set_materialize_grads(False)
        return x1, h.view(*xr.shape[:-1], dff), parts

    @staticmethod
    def backward(ctx, g_x1, g_hpre, g_parts):
        from .attention import rng_state
        qc, mask_u8, lse, a, x1, n2, stats, ln_a, Wo, W1 = ctx.saved_tensors
        eps, p_sub, seed1, heads, mask_sb, mask_sq, p_att, seed_att, scale = ctx.meta
        B, L, three = qc.shape
        hd = three // 3
        dk = hd // heads
        R, dev = B * L, x1.device
        dff = W1.shape[0]
        es = qc.element_size()
        strides = (L * three, dk, three)
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            dx1, dy1, da, part = _new(dev, *x1.shape), _new(dev, R, D_MODEL), _new(dev, B, L, hd), _parts(dev, R)
            delta = _new(dev, B, heads, L)
            kw = dict(x_ln=x1, stats=stats, ln_a=ln_a, eps=eps, res=g_x1.contiguous() if g_x1 is not None else None,
                      x_out=dx1, part=part, drop_p=p_sub, seed=seed1, n_out=dy1, w2=Wo, n2=D_MODEL, out2=da, attn_out=a,
                      delta_out=delta, lq=L)
            dW1 = db1 = None
            if g_hpre is not None and g_parts is not None:
                gh, gp = g_hpre.reshape(R, dff).contiguous(), g_parts.contiguous()
                _rows(1, R, dev, a1=gp, nparts=gp.shape[0], **kw)
                dW1, db1 = _linear_grads(gh, n2, W1)
            elif g_hpre is not None:
                gh = g_hpre.reshape(R, dff).contiguous()
                parts, S = _split_product(gh, W1, False)
                _rows(1, R, dev, a1=parts, nparts=S, **kw)
                dW1, db1 = _linear_grads(gh, n2, W1)
            else:
                _rows(1, R, dev, g=torch.zeros(R, D_MODEL, dtype=torch.float32, device=dev), **kw)
            dln_a, dln_b = _ln_param_grads(part)
            dWo, dbo = _linear_grads(dy1, a.reshape(R, D_MODEL), Wo)
            dqkv = torch.empty_like(qc)
            base, gb = qc.data_ptr(), dqkv.data_ptr()
            # (the two halves of the attention gradient are independent once delta is known: one launch)
            check(lib.spacap_mha_bwd_delta_f32(base, base + hd * es, base + 2 * hd * es, *strides, *strides, *strides, _p(mask_u8),
                                               mask_sb, mask_sq, None, 0, 0, 0, B, heads, L, L, dk, scale, p_att, seed_att,
                                               rng_state(dev).data_ptr() if p_att > 0.0 else None, lse.data_ptr(), da.data_ptr(),
                                               delta.data_ptr(), gb, gb + hd * es, gb + 2 * hd * es, three, st),
                  "spacap_mha_bwd_delta_f32")
        return (dqkv, dx1, None, None, None, None, None, None, dWo, dbo, dln_a, dln_b, dW1, db1, None, None, None, None, None, None)


class Ffn2Ln(Function):
    """x2 = x1 + dropout(h W2^T + b2) -- ``parts`` = the partial sums of h W2^T from ``AttnOutFfn1`` --;
    n = LayerNorm(x2) with the NEXT layer's (or the stack's final) norm;  returns (x2, n Wqkv^T + bqkv) -- or (n,) when ``pw``
    is None (last layer: n is the stack's output).  The gradients returned for h / parts are the private pair described in
    the module docstring; ``p_ffn`` is the hidden layer's dropout probability, whose scale that mask needs."""

    @staticmethod
    def forward(ctx, h, parts, x1, W1, W2, b2, ln_a, ln_b, eps, p_sub, p_ffn, seed3, pw, pb, *routing):
        if not h.is_cuda:
            raise RuntimeError("CPU not supported")
        hc, xr, pc = h.contiguous(), x1.contiguous(), parts.contiguous()
        R, dev = xr.numel() // D_MODEL, xr.device
        last = pw is None
        with torch.cuda.device(dev):
            x2, n, stats = _new(dev, *xr.shape), _new(dev, *xr.shape), _new(dev, R, 2)
            kw = dict(a1=pc, nparts=pc.shape[0], bias1=b2, drop_p=p_sub, seed=seed3, res=xr, x_out=x2, ln_a=ln_a, ln_b=ln_b,
                      eps=eps, n_out=n, stats=stats)
            qkv = None
            if not last:
                qkv = _new(dev, *xr.shape[:-1], pw.shape[0])
                kw.update(w2=pw, bias2=pb, n2=pw.shape[0], out2=qkv)
            _rows(0, R, dev, **kw)
        ctx.save_for_backward(hc, x2, n, stats, ln_a, W1, W2, pw)
        ctx.meta = (float(eps), float(p_sub), float(p_ffn), int(seed3), last, [int(r.shape[0]) for r in routing[:3]])
        ctx.set_materialize_grads(False)
        return (n,) if last else (x2, qkv)

    @staticmethod
    def backward(ctx, *grads):
        hc, x2, n, stats, ln_a, W1, W2, pw = ctx.saved_tensors
        eps, p_sub, p_ffn, seed3, last, rows = ctx.meta
        R, dev = x2.numel() // D_MODEL, x2.device
        dff = W2.shape[1]
        with torch.cuda.device(dev):
            dx2, dy2, part = _new(dev, *x2.shape), _new(dev, R, D_MODEL), _parts(dev, R)
            kw = dict(x_ln=x2, stats=stats, ln_a=ln_a, eps=eps, x_out=dx2, part=part, drop_p=p_sub, seed=seed3, n_out=dy2)
            dw = dbias = None
            if last:
                g_mem = grads[0]
                _rows(1, R, dev, g=g_mem.contiguous() if g_mem is not None else torch.zeros_like(x2), **kw)
            else:
                g_x2, dqkv = grads
                gx = g_x2.contiguous() if g_x2 is not None else None
                if dqkv is not None:
                    g2 = dqkv.reshape(R, -1).contiguous()
                    _rows(1, R, dev, a1=g2, w1=pw, k1=g2.shape[1], res=gx, **kw)
                    dw, dbias = _linear_grads(g2, n.reshape(R, D_MODEL), pw)
                else:
                    _rows(1, R, dev, g=torch.zeros_like(x2), res=gx, **kw)
            dln_a, dln_b = _ln_param_grads(part)
            # dhid = (dy2 W2) where the hidden unit was active and kept (scaled by the hidden dropout) AND the partial sums of
            # dhid W1 over the slices of d_ff, in one launch
            hflat = hc.reshape(R, dff)
            dh, gparts = _ffn(1, dy2, W2, W1, None, hflat, dff, p_ffn, 0, dev)
            dW2, db2 = _linear_grads(dy2, hflat, W2)
        head = (dh.view_as(hc), gparts, dx2, None, dW2, db2, dln_a, dln_b, None, None, None, None)
        if last:
            return head + (None, None)
        if rows:
            return head + (None, None) + _route_qkv(dw, dbias, rows)
        return head + (dw, dbias)


# ---- a whole stack ---------------------------------------------------------------------------------------------------

def _packed(attn):
    """(packed_w (3d, d), packed_b (3d), routing): views over the flat optimizer buffer when the Trainer laid the three
    projections out back to back (spacap3d_amd/engine.py), else a concatenation that takes the gradient itself."""
    pk = getattr(attn, "_packed_qkv", None)
    lin = attn.linears
    if pk is not None and pk[0].data_ptr() == lin[0].weight.data_ptr():
        return pk[0], pk[1], tuple(l.weight for l in lin[:3]) + tuple(l.bias for l in lin[:3])
    return torch.cat([l.weight for l in lin[:3]], 0), torch.cat([l.bias for l in lin[:3]], 0), ()


def stack_supported(layers, x):
    """True when ``run_stack`` can run these layers: float32 CUDA rows of width 128, feed-forward width a multiple of 128,
    self-attention + feed-forward sub-layers only (encoder layers, or decoder layers in early-guide mode), contiguous
    parameters."""
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == D_MODEL and len(layers) > 0):
        return False
    for l in layers:
        ff, sa = l.feed_forward, l.self_attn
        if getattr(l, "early_guide", True) is not True or sa.h * sa.d_k != D_MODEL:
            return False
        if not supported(ff.w_1.in_features, ff.w_1.out_features) or ff.w_2.bias is None or ff.w_1.bias is None:
            return False
        if any(li.bias is None or not li.weight.is_contiguous() for li in sa.linears):
            return False
    return True


def decode_supported(layers, x, n_words):
    """True when ``greedy_decode`` can run: everything ``stack_supported`` asks for, plus what ``spacap_decode_attn_f32``
    requires (h = 8, d_k = 16, at most 32 cached positions) and one feed-forward width for the whole stack (one partial-sum
    buffer).  Anything else decodes through the cached per-operator path (``decode_incremental``)."""
    if not stack_supported(layers, x) or n_words + 1 > 32:
        return False
    dff = layers[0].feed_forward.w_1.out_features
    return all(l.self_attn.h == 8 and l.self_attn.d_k == 16 and l.feed_forward.w_1.out_features == dff for l in layers)


def run_stack(layers, final_norm, x, mask):
    """The N pre-norm layers (self-attention + feed-forward each) followed by ``final_norm``
    (models/transformer_captioner.py: Encoder :166-178 / Decoder :193-207 in early-guide mode)."""
    from .attention import _next_seed, _prep_mask, self_attention_packed

    def drop(m):
        return float(m.p) if m.training else 0.0

    def seed(p):
        return _next_seed() if p > 0.0 else 0

    sub = lambda l: (l.sublayer[0], l.sublayer[-1])
    l0 = layers[0]
    if x.numel() // x.shape[-1] > FFN_BF3_MIN_ROWS:
        refresh_ffn_pieces(layers)       # this forward's (and its backward's) split-bf16 weight images: one launch
    pw, pb, routing = _packed(l0.self_attn)
    qkv, xres = LnQkv.apply(x, l0.sublayer[0].norm.a_2, l0.sublayer[0].norm.b_2, l0.sublayer[0].norm.eps, pw, pb, *routing)
    out = None
    for i, l in enumerate(layers):
        sa, ff = l.self_attn, l.feed_forward
        s_att, s_ffn = sub(l)
        need_p = sa.keep_value if sa.store_attn is None else (sa.store_attn or sa.keep_value)
        p1, pf, p3 = drop(s_att.dropout), drop(ff.dropout), drop(s_ffn.dropout)
        wo = sa.linears[-1]
        tail = (wo.weight, wo.bias, s_ffn.norm.a_2, s_ffn.norm.b_2, ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, s_ffn.norm.eps, p1, pf)
        if not need_p and sa.d_k == 16 and sa.h == 8:
            # attention inside the node: its gradient is one launch (see AttnFfn1)
            B_, L_ = qkv.shape[0], qkv.shape[1]
            m8, msb, msq = _prep_mask(mask, B_, L_, L_)
            pa = float(sa.dropout.p) if sa.dropout.training else 0.0
            sa.attn = None
            x1, h, parts = AttnFfn1.apply(qkv, xres, m8, msb, msq, sa.h, pa, seed(pa), *tail, seed(p1), seed(pf))
        else:
            a, sa.attn = self_attention_packed(qkv, sa.h, mask=mask, dropout_p=sa.dropout.p, training=sa.dropout.training,
                                               need_p=need_p)
            if sa.keep_value:
                hd = sa.h * sa.d_k
                sa.value = qkv[..., 2 * hd:].view(qkv.shape[0], -1, sa.h, sa.d_k).transpose(1, 2)
            x1, h, parts = AttnOutFfn1.apply(a, xres, *tail, seed(p1), seed(pf))
        if i + 1 < len(layers):
            nxt = layers[i + 1]
            nn_ = nxt.sublayer[0].norm
            pw, pb, routing = _packed(nxt.self_attn)
            xres, qkv = Ffn2Ln.apply(h, parts, x1, ff.w_1.weight, ff.w_2.weight, ff.w_2.bias, nn_.a_2, nn_.b_2, nn_.eps, p3, pf,
                                     seed(p3), pw, pb, *routing)
        else:
            (out,) = Ffn2Ln.apply(h, parts, x1, ff.w_1.weight, ff.w_2.weight, ff.w_2.bias, final_norm.a_2, final_norm.b_2,
                                  final_norm.eps, p3, pf, seed(p3), None, None)
    return out


@torch.no_grad()
def greedy_decode(dec, generator, embed, pe, indicator, sos, n_words):
    """Greedy decoding of R sequences through the early-guide decoder stack ``dec`` with pre-allocated key / value caches
    (models/transformer_captioner.py:402-453; the reference re-runs encoder and decoder prefix for every word).  Position 0
    is the object indicator token (R, 128), position 1 the start symbol, then every chosen word is fed back; one token per
    sequence per step, four launches per layer and step (the training kernels with dropout off + spacap_decode_attn_f32).
    The word choice of a step -- vocabulary projection + arg-max + the chosen word's embedding row -- is one library call
    (csrc/tf_layer.hip: vocab_argmax_kernel / decode_next_kernel: no logits in HBM, no BLAS call).
    Returns the (R, n_words) int64 word ids."""
    import math
    layers = list(dec.layers)
    R, dev = indicator.shape[0], indicator.device
    T = n_words + 1
    st = torch.cuda.current_stream(dev).cuda_stream
    packs = [_packed(l.self_attn)[:2] for l in layers]
    h, dk = layers[0].self_attn.h, layers[0].self_attn.d_k
    dff = layers[0].feed_forward.w_1.out_features
    S = dff // 128
    scale = 1.0 / math.sqrt(dk)
    pieces = None
    if FFN_BF3 and R > FFN_BF3_MIN_ROWS:
        refresh_ffn_pieces(layers)
        pieces = [_FFN_PIECES.get(l.feed_forward.w_1.weight.data_ptr()) for l in layers]
        pieces = [e[1] for e in pieces] if all(e is not None for e in pieces) else None
    with torch.cuda.device(dev):
        kc = [_new(dev, R, T, D_MODEL) for _ in layers]
        vc = [_new(dev, R, T, D_MODEL) for _ in layers]
        qkv, a, x1, x2, n2, n = _new(dev, R, 3 * D_MODEL), _new(dev, R, D_MODEL), _new(dev, R, D_MODEL), _new(dev, R, D_MODEL), \
            _new(dev, R, D_MODEL), _new(dev, R, D_MODEL)
        parts = _new(dev, S, R, D_MODEL)
        ys = torch.empty(R, n_words, dtype=torch.long, device=dev)
        sqrt_d = math.sqrt(embed.d_model)
        x = indicator.contiguous()
        lut, gw, gb = embed.lut.weight.contiguous(), generator.proj.weight.contiguous(), generator.proj.bias.contiguous()
        V = gw.shape[0]
        from .linear import bf3_pieces
        gwp = bf3_pieces(gw)                            # the projection weight's three bf16 pieces, once per call
        pe_rows = pe[0, :T].contiguous()
        xw = _new(dev, R, D_MODEL)                      # the next step's input rows (written by the word kernel)
        ws = torch.empty(int(lib.spacap_decode_word_workspace_bytes(R, V)), dtype=torch.uint8, device=dev)
        for t in range(T):
            if t == 1:
                x = (lut[int(sos)] * sqrt_d + pe_rows[0]).expand(R, D_MODEL).contiguous()   # every sequence starts with <sos>
            elif t > 1:
                x = xw
            n0 = layers[0].sublayer[0].norm
            _rows(0, R, dev, res=x, ln_a=n0.a_2, ln_b=n0.b_2, eps=n0.eps, w2=packs[0][0], bias2=packs[0][1], n2=3 * D_MODEL, out2=qkv)
            xres = x
            for i, l in enumerate(layers):
                sa, ff, nf = l.self_attn, l.feed_forward, l.sublayer[-1].norm
                check(lib.spacap_decode_attn_f32(qkv.data_ptr(), kc[i].data_ptr(), vc[i].data_ptr(), R, h, dk, T, t, scale, a.data_ptr(),
                                                 st), "spacap_decode_attn_f32")
                _rows(0, R, dev, a1=a, w1=sa.linears[-1].weight, bias1=sa.linears[-1].bias, k1=D_MODEL, res=xres, x_out=x1,
                      ln_a=nf.a_2, ln_b=nf.b_2, eps=nf.eps, n_out=n2)
                if pieces is not None:
                    check(lib.spacap_tf_ffn_bf3_f32(0, n2.data_ptr(), pieces[i].data_ptr(), ff.w_1.bias.data_ptr(), None, R, dff, 0.0, 0,
                                                    None, None, parts.data_ptr(), st), "spacap_tf_ffn_bf3_f32")
                else:
                    check(lib.spacap_tf_ffn_f32(0, n2.data_ptr(), ff.w_1.weight.data_ptr(), ff.w_2.weight.data_ptr(),
                                                ff.w_1.bias.data_ptr(), None, R, dff, 0.0, 0, None, None, parts.data_ptr(), st),
                          "spacap_tf_ffn_f32")
                if i + 1 < len(layers):
                    nn_ = layers[i + 1].sublayer[0].norm
                    _rows(0, R, dev, a1=parts, nparts=S, bias1=ff.w_2.bias, res=x1, x_out=x2, ln_a=nn_.a_2, ln_b=nn_.b_2, eps=nn_.eps,
                          w2=packs[i + 1][0], bias2=packs[i + 1][1], n2=3 * D_MODEL, out2=qkv)
                    xres = x2
                else:
                    _rows(0, R, dev, a1=parts, nparts=S, bias1=ff.w_2.bias, res=x1, ln_a=dec.norm.a_2, ln_b=dec.norm.b_2,
                          eps=dec.norm.eps, n_out=n)
            if t >= 1:
                # ys[:, t - 1] = argmax_v (n W^T + b); xw = lut[word] sqrt(d) + pe[t]: the input of step t + 1
                check(lib.spacap_decode_word_f32(n.data_ptr(), gwp.data_ptr(), gb.data_ptr(), R, V, lut.data_ptr(), sqrt_d,
                                                 pe_rows[min(t, T - 1)].data_ptr(), ys.data_ptr(), n_words, t - 1, xw.data_ptr(),
                                                 ws.data_ptr(), st), "spacap_decode_word_f32")
    return ys
