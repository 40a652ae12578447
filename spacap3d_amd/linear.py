"""y = x W^T + b with a one-launch weight + bias gradient (csrc/sa_mlp.hip: linear_wgrad_kernel).

Counterpart of the ``nn.Linear`` projections / feed-forward layers of the reference Transformer
(models/transformer_captioner.py:63-99, 117-126).  Forward and dX run on the row-panel kernel up to 512 rows (the default
model's fused Transformer stacks, tf_layer.py, do not come through here at all; wider models' larger products still use the BLAS
GEMM, see _forward_product); the backward's
``dW = g^T x`` and ``db = sum_r g`` -- for <= 2 048 rows a memset + a split-K GEMM + a column-sum kernel of
~30 us of latency -- become one kernel producing per-slab partials of both plus one sum over the slabs.
"""
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from ._native import check, conv1x1_wgrad_partials, grad_slot, lib, linear_wgrad_partials, sum_slabs


def rows_product(a2, W, bias, trans_w):
    """out = a2 W^T (+ bias) when ``trans_w`` else a2 W, by the row-panel kernel (csrc/sa_mlp.hip: linear_rows_kernel);
    a2 (R, K) dense.  The caller checks ``use_rows``."""
    R, K = a2.shape
    CO = W.shape[0] if trans_w else W.shape[1]
    dev = a2.device
    with torch.cuda.device(dev):
        out = torch.empty(R, CO, dtype=torch.float32, device=dev)
        check(lib.spacap_linear_rows_f32(a2.data_ptr(), W.data_ptr(), bias.data_ptr() if bias is not None else None, R, K, CO,
                                         1 if trans_w else 0, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
              "spacap_linear_rows_f32")
    return out


def use_rows(R, K, CO):
    """Shapes where the row-panel kernel beats the BLAS GEMM (measured, tools/lab/linear_bench.py)."""
    return bool(lib.spacap_linear_rows_supported(R, K, CO)) and _ROWS_RULE(R, K, CO)


# the BLAS heuristics are erratic below a few hundred rows (R = 256: 20 - 60 us on one or two workgroups for the
# d_model-wide products, 4 - 6 us at R = 264 or 2 048); the row-panel kernel takes 3 - 7 us there
_ROWS_RULE = lambda R, K, CO: R <= 512


def bf3_pieces(W, trans=False):
    """The three bf16 pieces of W (N, K) -- or of W^T when ``trans`` (W is then (K, N)) -- as one bf16 tensor (3, N, K): the weight
    operand of ``bf3_product`` (csrc/gemm_bf3.hip: split once per call, 1.5 MB for a 512 x 512 matrix)."""
    W = W.contiguous()
    N, K = (W.shape[1], W.shape[0]) if trans else (W.shape[0], W.shape[1])
    dev = W.device
    with torch.cuda.device(dev):
        Wp = torch.empty(3, N, K, dtype=torch.bfloat16, device=dev)
        check(lib.spacap_gemm_bf3_split_w_f32(W.data_ptr(), W.shape[1], N, K, 1 if trans else 0, Wp.data_ptr(),
                                              torch.cuda.current_stream(dev).cuda_stream), "spacap_gemm_bf3_split_w_f32")
    return Wp


def bf3_product(a2, Wp, bias=None, relu=False, out=None):
    """out (R, N) = a2 (R, K) W^T (+ bias) (ReLU) with W given as its pieces (bf3_pieces): the tiled split-bf16 kernel
    (fp32-equivalent arithmetic on the bf16 matrix cores; K, N multiples of 128)."""
    R, K = a2.shape
    N = Wp.shape[1]
    assert a2.is_contiguous() and Wp.shape[2] == K
    dev = a2.device
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty(R, N, dtype=torch.float32, device=dev)
        check(lib.spacap_gemm_bf3_f32(a2.data_ptr(), K, Wp.data_ptr(), bias.data_ptr() if bias is not None else None, R, K, N,
                                      1 if relu else 0, out.data_ptr(), N, torch.cuda.current_stream(dev).cuda_stream),
              "spacap_gemm_bf3_f32")
    return out


def _use_bf3(R, K, N):
    return R >= 1024 and bool(lib.spacap_gemm_bf3_supported(K, N))


def _forward_product(x, weight, bias):
    """x W^T + b on this library's kernels: the row-panel kernel up to 512 rows of a d_model-sized projection, the tiled
    split-bf16 kernel for tall products of 128-multiples (the 512-wide model), the shape-agnostic fp32-MFMA row product
    otherwise.  (No BLAS call: tests/test_engine_gpu.py checks every BASELINE config's step for library kernels.)"""
    CO, K = weight.shape
    R = x.numel() // K
    if x.is_cuda and x.dtype == torch.float32 and weight.is_contiguous():
        x2 = x.reshape(R, K).contiguous()
        if use_rows(R, K, CO):
            out = rows_product(x2, weight, bias, True)
        elif _use_bf3(R, K, CO):
            out = bf3_product(x2, bf3_pieces(weight), bias)
        else:
            out = dense_product(x2, weight, True, bias=bias)
        return out.view(*x.shape[:-1], CO)
    return F.linear(x, weight, bias)


def _data_gradient(g2, weight):
    """g2 W for g2 (R, CK) dense and W (CK, CP); kernels as in _forward_product."""
    R, CK = g2.shape
    if g2.is_cuda and g2.dtype == torch.float32 and weight.is_contiguous():
        g2 = g2.contiguous()
        if use_rows(R, CK, weight.shape[1]):
            return rows_product(g2, weight, None, False)
        if _use_bf3(R, CK, weight.shape[1]):
            return bf3_product(g2, bf3_pieces(weight, trans=True))
        return dense_product(g2, weight, False)
    return g2 @ weight


class FusedLinear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return _forward_product(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        CK, CP = weight.shape
        g2 = g.reshape(-1, CK)
        x2 = x.reshape(-1, CP)
        R = g2.shape[0]
        dx = _data_gradient(g2, weight).view_as(x) if ctx.needs_input_grad[0] else None
        part = linear_wgrad_partials(g2.contiguous(), x2.contiguous(), True, deferrable=True) if g2.is_cuda else None
        if part is None:
            return dx, g2.t() @ x2, g2.sum(0)
        s = sum_slabs(part, deferrable=True, out=grad_slot(weight, CK * CP + CK))
        return dx, s[:CK * CP].view(CK, CP), s[CK * CP:]


def linear(x, weight, bias):
    """F.linear with the fused weight/bias gradient (bias required)."""
    return FusedLinear.apply(x, weight, bias)


class FFNTail(Function):
    """out = w_2(dropout(relu(h)))  (models/transformer_captioner.py:117-126, the part after w_1) as one autograd node:
    forward = the fused relu+dropout kernel + a BLAS GEMM; backward = one launch for d h (data gradient of w_2 with the
    relu / dropout mask applied in its epilogue, csrc/sa_mlp.hip: linear_dgrad_mask_kernel) + the one-launch weight /
    bias gradient."""

    @staticmethod
    def forward(ctx, h, weight, bias, p, seed):
        from .attention import rng_state
        h = h.contiguous()
        dev = h.device
        with torch.cuda.device(dev):
            y = torch.empty_like(h)
            check(lib.spacap_relu_dropout_fwd_f32(h.data_ptr(), h.numel(), float(p), int(seed),
                                                  rng_state(dev).data_ptr() if p > 0.0 else None, y.data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream), "spacap_relu_dropout_fwd_f32")
        ctx.save_for_backward(y, weight)
        ctx.p = float(p)
        return F.linear(y, weight, bias)   # K = d_ff: a long reduction, the BLAS split-K kernels are the right tool

    @staticmethod
    def backward(ctx, g):
        y, weight = ctx.saved_tensors
        CK, CP = weight.shape            # (d_model, d_ff)
        g2 = g.reshape(-1, CK).contiguous()
        y2 = y.reshape(-1, CP)
        R = g2.shape[0]
        dev = g2.device
        st = torch.cuda.current_stream(dev).cuda_stream
        scale = 1.0 / (1.0 - ctx.p)
        with torch.cuda.device(dev):
            dh = torch.empty_like(y2)
            check(lib.spacap_linear_dgrad_mask_f32(g2.data_ptr(), weight.contiguous().data_ptr(), y2.data_ptr(), scale, R, CK,
                                                   CP, dh.data_ptr(), st), "spacap_linear_dgrad_mask_f32")
            part = linear_wgrad_partials(g2, y2.contiguous(), True, deferrable=True)
            if part is None:
                dw, db = g2.t() @ y2, g2.sum(0)
            else:
                s = sum_slabs(part, deferrable=True, out=grad_slot(weight, CK * CP + CK))
                dw, db = s[:CK * CP].view(CK, CP), s[CK * CP:]
        return dh.view_as(y), dw, db, None, None


def ffn_tail(h, weight, bias, p, training):
    """w_2(dropout(relu(h))) -- ``None`` when the shapes have no fused kernel (d_model must be 128, d_ff a multiple of
    128); the caller then composes relu_dropout + linear."""
    if not h.is_cuda or weight.shape[0] != 128 or weight.shape[1] % 128 or bias is None:
        return None
    from .attention import _next_seed
    pp = float(p) if training else 0.0
    return FFNTail.apply(h, weight, bias, pp, _next_seed() if pp > 0.0 else 0)


class PackedLinear(Function):
    """y = x [W0; W1; W2]^T + [b0; b1; b2] where the three weights (and the three biases) are ADJACENT in memory, so
    ``packed_w`` / ``packed_b`` are plain views of them (spacap3d_amd/engine.py lays the parameters out that way in the
    flat optimizer buffer): the self-attention q | k | v projection without concatenating the weights every step.
    The parameters themselves are passed too, only so that autograd routes the gradient slices to them."""

    @staticmethod
    def forward(ctx, x, packed_w, packed_b, w0, w1, w2, b0, b1, b2):
        ctx.save_for_backward(x, packed_w)
        ctx.split = (w0.shape[0], w1.shape[0], w2.shape[0])
        return _forward_product(x, packed_w, packed_b)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        CK, CP = weight.shape
        g2 = g.reshape(-1, CK).contiguous()
        x2 = x.reshape(-1, CP).contiguous()
        R = g2.shape[0]
        dx = _data_gradient(g2, weight).view_as(x) if ctx.needs_input_grad[0] else None
        part = linear_wgrad_partials(g2, x2, True, deferrable=True)
        if part is None:
            dw, db = g2.t() @ x2, g2.sum(0)
        else:
            s = sum_slabs(part, deferrable=True, out=grad_slot(weight, CK * CP + CK))
            dw, db = s[:CK * CP].view(CK, CP), s[CK * CP:]
        a, b, _ = ctx.split
        return (dx, None, None, dw[:a], dw[a:a + b], dw[a + b:], db[:a], db[a:a + b], db[a + b:])


def packed_views(flat, params_w, params_b):
    """(packed_w, packed_b) views INTO ``flat`` over three weights / biases that sit back to back inside it, else None."""
    def adjacent(ps):
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        return all(p.is_contiguous() and lo <= p.data_ptr() < hi for p in ps) and \
            all(ps[i + 1].data_ptr() == ps[i].data_ptr() + ps[i].numel() * ps[i].element_size() for i in range(len(ps) - 1))
    if not (adjacent(params_w) and adjacent(params_b)) or any(p.shape[1:] != params_w[0].shape[1:] for p in params_w):
        return None
    rows, cols = sum(p.shape[0] for p in params_w), params_w[0].shape[1]
    ow = (params_w[0].data_ptr() - flat.data_ptr()) // flat.element_size()
    ob = (params_b[0].data_ptr() - flat.data_ptr()) // flat.element_size()
    return flat[ow:ow + rows * cols].view(rows, cols), flat[ob:ob + rows]


# wide relation head (d_model = 512): weight gradient by csrc/wgrad_bf3.inc instead of gemm_bf3_wgrad_kernel (lab switch)
WIDE_WGRAD_TR = os.environ.get("SPACAP_WIDE_WGRAD_TR", "1") != "0"


class Conv1x1(Function):
    """nn.Conv1d / nn.Conv2d with a 1x1 kernel on channel-major (B, C, N[, 1]) tensors -- the vote net and the
    feature-propagation MLPs (models/voting_module.py:33-60, lib/pointnet2/pointnet2_modules.py:376-421), the proposal head and the
    position embedding.  Forward (bias in the epilogue) and input gradient: csrc/conv1x1.hip (USE_OWN_CONV; point counts that are
    not a multiple of 64 fall back to the convolution library); weight gradient: csrc/sa_mlp.hip: conv1x1_wgrad_kernel + one sum
    over its slabs."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if USE_OWN_CONV:
            y = conv1x1_cm(0, weight, x, bias, weight.shape[0])
            if y is not None:
                return y
        return F.conv2d(x, weight, bias) if x.dim() == 4 else F.conv1d(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        B, CI = x.shape[0], x.shape[1]
        CO = weight.shape[0]
        N = x.numel() // (B * CI)
        g = g.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv1x1_cm(1, weight, g, None, CI) if USE_OWN_CONV else None
            if dx is None:
                dx = torch.ops.aten.convolution_backward(g, x, weight, None, [1] * (x.dim() - 2), [0] * (x.dim() - 2),
                                                         [1] * (x.dim() - 2), False, [0] * (x.dim() - 2), 1,
                                                         [True, False, False])[0]
        if not ctx.needs_input_grad[1]:    # frozen weights: the data gradient alone
            db = g.sum(dim=[0] + list(range(2, g.dim()))) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
            return dx, None, db
        if int(lib.spacap_conv1x1_wgrad_slabs(B, CO, CI, N)):
            part = conv1x1_wgrad_partials(g, x, B, CO, CI, N, deferrable=True, with_bias=ctx.has_bias)
            s = sum_slabs(part, deferrable=True)
            if part.shape[1] != CO * CI:      # the bias gradient rode along (a column of ones in the queued kernel)
                return dx, s[:CO * CI].view_as(weight), s[CO * CI:CO * CI + CO]
            dw = s.view_as(weight)
        else:   # point counts without a slab kernel (not a multiple of 32): the library's weight gradient
            dw = torch.ops.aten.convolution_backward(g, x, weight, None, [1] * (x.dim() - 2), [0] * (x.dim() - 2),
                                                     [1] * (x.dim() - 2), False, [0] * (x.dim() - 2), 1,
                                                     [False, True, False])[1]
        db = g.sum(dim=[0] + list(range(2, g.dim()))) if ctx.has_bias else None
        return dx, dw, db


# Forward and input gradient of these convolutions run on the library's own channel-major kernel (csrc/conv1x1.hip, bias in
# its epilogue; gated against float64 in tests/test_attention_gpu.py::test_conv1x1_channel_major_kernel).  Rounds 2 - 3 kept them
# on MIOpen / rocBLAS because the kernel's summation order moved a chaotic 5-step trajectory gate; that gate now freezes the
# discrete selections it cannot control (tests/test_engine_gpu.py) and a re-association passes it.  False restores the library
# calls (A/B measurements).
USE_OWN_CONV = True
# tests: the forward on the fp32-MFMA kernel (exact fp32 products) instead of the split-bf16 one -- for comparisons of two paths
# whose discrete selections (ReLU gates, pooling arg-max) must not see a 1e-6 difference in a pre-activation
CONV_FWD_EXACT_F32 = False


def conv1x1_cm(mode, weight, t, bias, M):
    """The library's own channel-major kernel (csrc/conv1x1.hip): forward (mode 0) or input gradient (mode 1) of a 1x1
    convolution on ``t`` (B, C, N[, 1]) contiguous; ``None`` for shapes it does not take (N not a multiple of 64)."""
    B, C = t.shape[0], t.shape[1]
    N = t.numel() // max(B * C, 1)
    CO, CI = weight.shape[0], weight.shape[1]
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and weight.is_contiguous() and B > 0 and
            lib.spacap_conv1x1_cm_supported(CI, CO, N)) or t.data_ptr() % 16:
        return None
    with torch.cuda.device(t.device):
        out = torch.empty((B, M) + tuple(t.shape[2:]), dtype=torch.float32, device=t.device)
        check(lib.spacap_conv1x1_cm_f32(2 if (mode == 0 and CONV_FWD_EXACT_F32) else mode, weight.data_ptr(), t.data_ptr(),
                                        bias.data_ptr() if bias is not None else None,
                                        B, CI, CO, N, out.data_ptr(), torch.cuda.current_stream(t.device).cuda_stream),
              "spacap_conv1x1_cm_f32")
    return out


def conv1x1(x, conv):
    """``conv(x)`` for a 1x1 nn.Conv1d / nn.Conv2d ``conv``; ``None`` when the shape has no kernel (the caller then calls the
    module).  Without gradient recording (the inference forward) only the forward kernel runs."""
    if not (x.is_cuda and x.dtype == torch.float32) or any(k != 1 for k in conv.kernel_size) or not x.is_contiguous():
        return None
    # the raw kernel returns a tensor without a grad_fn: only when NOTHING on this call can need a gradient (grad mode off, or
    # neither the input nor the parameters require one).  A frozen convolution behind a trainable layer goes through
    # Conv1x1.apply below, whose backward honours needs_input_grad.
    needs = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad
                                         or (conv.bias is not None and conv.bias.requires_grad))
    if not needs:
        return conv1x1_cm(0, conv.weight, x, conv.bias, conv.out_channels) if USE_OWN_CONV else None
    B, CI = x.shape[0], x.shape[1]
    N = x.numel() // max(B * CI, 1)
    if int(lib.spacap_conv1x1_wgrad_slabs(B, conv.out_channels, CI, N)) == 0:
        return None
    return Conv1x1.apply(x, conv.weight, conv.bias)


class RelationTail(Function):
    """pred = W3 relu(W2 hid1 + b2) + b3 on the B*K*K pair rows of the relation head
    (models/transformer_captioner.py:319-326, 392-397; hid1 = the first layer's ReLU output, 524 288 x 128 at the
    benchmark shape).  Forward: ONE kernel reads hid1 and writes hid2 and pred (csrc/sa_mlp.hip: sa_mid_fwd_kernel, TAIL).
    Backward: one streaming kernel gives dz2 = (dpred W3) * (hid2 > 0) and the partial sums of dW3, db2, db3
    (rel_tail_bwd_kernel); dhid1 = dz2 W2 and dW2 = dz2^T hid1 are MFMA-bound BLAS GEMMs (the latter cut into 64 row
    slabs: the BLAS heuristics do not split that reduction)."""

    SLABS = 64

    @staticmethod
    def forward(ctx, hid1, W2, b2, W3, b3):
        shape = hid1.shape
        h1 = hid1.reshape(-1, 128).contiguous()
        R = h1.shape[0]
        dev = h1.device
        W2c, W3c = W2.contiguous(), W3.contiguous()
        with torch.cuda.device(dev):
            hid2 = torch.empty_like(h1)
            pred = torch.empty(R, W3.shape[0], dtype=torch.float32, device=dev)
            check(lib.spacap_rel_tail_fwd_f32(h1.data_ptr(), W2c.data_ptr(), b2.data_ptr(), W3c.data_ptr(), b3.data_ptr(), R,
                                              hid2.data_ptr(), pred.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                  "spacap_rel_tail_fwd_f32")
        ctx.save_for_backward(h1, hid2, W2c, W3c)
        ctx.shape = shape
        return pred.view(*shape[:-1], W3.shape[0])

    @staticmethod
    def backward(ctx, g):
        h1, hid2, W2, W3 = ctx.saved_tensors
        NO = W3.shape[0]
        g2 = g.reshape(-1, NO).contiguous()
        R = g2.shape[0]
        dev = g2.device
        with torch.cuda.device(dev):
            nparts = int(lib.spacap_rel_tail_bwd_nparts(R))
            PW = NO * 128 + 128 + 16
            part = torch.empty(nparts, PW, dtype=torch.float32, device=dev)
            dz2 = torch.empty_like(hid2)
            check(lib.spacap_rel_tail_bwd_f32(g2.data_ptr(), W3.data_ptr(), hid2.data_ptr(), R, dz2.data_ptr(), part.data_ptr(),
                                              torch.cuda.current_stream(dev).cuda_stream), "spacap_rel_tail_bwd_f32")
            s = sum_slabs(part, deferrable=True)
            dW3, db2, db3 = s[:NO * 128].view(NO, 128), s[NO * 128:NO * 128 + 128], s[NO * 128 + 128:NO * 128 + 128 + NO]
            dh1 = None
            if ctx.needs_input_grad[0]:
                if R >= 49152 and lib.spacap_gemm_rows_supported(128, 128):
                    # dz2 W2 on the streaming split-bf16 kernel (fp32-equivalent; 170 us in BLAS -> ~125 us at 524 288 rows)
                    dh1 = torch.empty_like(hid2)
                    W2t = W2.t().contiguous()
                    check(lib.spacap_gemm_rows_f32(dz2.data_ptr(), W2t.data_ptr(), R, 128, 128, dh1.data_ptr(),
                                                   torch.cuda.current_stream(dev).cuda_stream), "spacap_gemm_rows_f32")
                    dh1 = dh1.view(ctx.shape)
                else:
                    dh1 = (dz2 @ W2).view(ctx.shape)
            S = RelationTail.SLABS
            if R % S == 0 and R >= 64 * S:
                dW2 = sum_slabs(torch.bmm(dz2.view(S, R // S, 128).transpose(1, 2), h1.view(S, R // S, 128)).view(S, -1),
                                deferrable=True).view(128, 128)
            else:
                dW2 = dz2.t() @ h1
        return dh1, dW2, db2, dW3, db3


def relation_tail(hid1, lin2, lin3):
    """``lin3(relu(lin2(hid1)))`` for the relation head's 128 -> 128 -> 9 tail; ``None`` for other shapes."""
    if not hid1.is_cuda or hid1.dtype != torch.float32 or tuple(lin2.weight.shape) != (128, 128) or \
            tuple(lin3.weight.shape) != (9, 128) or lin2.bias is None or lin3.bias is None or hid1.shape[-1] != 128:
        return None
    return RelationTail.apply(hid1, lin2.weight, lin2.bias, lin3.weight, lin3.bias)


class RelationHead(Function):
    """The whole relation head (models/transformer_captioner.py:319-326, 392-397) as one kernel each way
    (csrc/relation_fused.hip): pred = W3 relu(W2 relu(b1 + sum_h P U) + b2) + b3 on the B*K*K pairs, with
    U[b,j,h,:] = V[b,h,j,:] W1[:, 16h:16h+16]^T formed by the caller.  Neither the pair feature nor the first hidden layer
    nor any gradient of pair size exists in HBM: the forward stores hid2 (the backward's one large input) and pred; the
    backward returns dP, the dU partials and per-workgroup partial sums of dW2, dW3, db1, db2, db3."""

    @staticmethod
    def forward(ctx, P, U, b1, W2, b2, W3, b3):
        P, U, W2c, W3c = P.contiguous(), U.contiguous(), W2.contiguous(), W3.contiguous()
        B, H, K, _ = P.shape
        dev = P.device
        with torch.cuda.device(dev):
            hid2 = torch.empty(B, K, K, 128, dtype=torch.float32, device=dev)
            pred = torch.empty(B, K, K, 9, dtype=torch.float32, device=dev)
            check(lib.spacap_relation_fused_fwd_f32(P.data_ptr(), U.data_ptr(), b1.data_ptr(), W2c.data_ptr(), b2.data_ptr(),
                                                    W3c.data_ptr(), b3.data_ptr(), B, K, hid2.data_ptr(), pred.data_ptr(),
                                                    torch.cuda.current_stream(dev).cuda_stream), "spacap_relation_fused_fwd_f32")
        ctx.save_for_backward(P, U, b1, W2c, W3c, hid2)
        return pred

    @staticmethod
    def backward(ctx, g):
        P, U, b1, W2, W3, hid2 = ctx.saved_tensors
        B, H, K, _ = P.shape
        dev = P.device
        g = g.contiguous()
        with torch.cuda.device(dev):
            PW = int(lib.spacap_relation_fused_part_floats())
            nparts = int(lib.spacap_relation_fused_nparts(B, K))
            zslots = int(lib.spacap_relation_fused_zsplit(B, K, nparts))
            part = torch.empty(nparts, PW, dtype=torch.float32, device=dev)
            dP = torch.empty_like(P)
            dU = torch.empty(zslots, *U.shape, dtype=torch.float32, device=dev)
            check(lib.spacap_relation_fused_bwd_f32(g.data_ptr(), hid2.data_ptr(), P.data_ptr(), U.data_ptr(), b1.data_ptr(),
                                                    W2.data_ptr(), W3.data_ptr(), B, K, nparts, zslots, dP.data_ptr(), dU.data_ptr(),
                                                    part.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                  "spacap_relation_fused_bwd_f32")
            s = sum_slabs(part, deferrable=True)
        o = 128 * 128
        dW2, dW3 = s[:o].view(128, 128), s[o:o + 9 * 128].view(9, 128)
        o += 9 * 128
        return dP, sum_slabs(dU), s[o:o + 128], dW2, s[o + 128:o + 256], dW3, s[o + 256:o + 265]


class RelationWide(Function):
    """The relation head at widths the one-kernel form has no kernel for (the 512-wide / 32-head stress configuration;
    models/transformer_captioner.py:319-326, 392-397 at d_model = 512), composed of this library's kernels:
        hid1 = relu(b1 + sum_h P U)            csrc/gemm_bf3.hip: rel_wide_l1_* (the pair feature is never formed: U = per-head first
                                               Linear of V; one workgroup per key column on the matrix cores, P read transposed)
        hid2 = relu(hid1 W2^T + b2)            csrc/gemm_bf3.hip (tiled split-bf16 product, bias + ReLU in its epilogue)
        pred = hid2 W3^T + b3                  csrc/dense_rows.hip
    backward: dz2 / dW3 / db2 / db3 in one pass over hid2 (rel_wide_tail_bwd_kernel), dW2 = dz2^T hid1 and dhid1 = dz2 W2 as
    split-bf16 products, then the first layer's backward (masks by hid1 > 0 itself).  Three tensors of pair size live between
    forward and backward (hid1, hid2; dz2 in the backward; dhid1 reuses hid2's memory)."""

    @staticmethod
    def forward(ctx, P, U, b1, W2, b2, W3, b3):
        P, U, W2c, W3c = P.contiguous(), U.contiguous(), W2.contiguous(), W3.contiguous()
        B, H, K, _ = P.shape
        C = U.shape[-1]
        dev = P.device
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            Pt = torch.empty(B, K, H, K, dtype=torch.float32, device=dev)      # Pt[b,j,h,i] = P[b,h,i,j]: a key column's block contiguous
            check(lib.spacap_rel_wide_transpose_f32(P.data_ptr(), Pt.data_ptr(), B, H, K, 1, st), "spacap_rel_wide_transpose_f32")
            hid1 = torch.empty(B * K * K, C, dtype=torch.float32, device=dev)
            check(lib.spacap_rel_wide_l1_fwd_f32(Pt.data_ptr(), U.data_ptr(), b1.data_ptr(), B, H, K, C, hid1.data_ptr(), st),
                  "spacap_rel_wide_l1_fwd_f32")
            hid2 = bf3_product(hid1, bf3_pieces(W2c), b2, relu=True)
            pred = dense_product(hid2, W3c, True, bias=b3)
        ctx.save_for_backward(Pt, U, W2c, W3c, hid1, hid2)
        ctx.spent = False
        return pred.view(B, K, K, W3c.shape[0])

    @staticmethod
    def backward(ctx, g):
        if ctx.spent:    # (dhid1 is written into hid2's memory: a second backward over a retained graph would read garbage)
            raise RuntimeError("RelationWide: its backward can run once per forward (retain_graph is not supported)")
        ctx.spent = True
        Pt, U, W2, W3, hid1, hid2 = ctx.saved_tensors
        B, K, H, _ = Pt.shape
        C = U.shape[-1]
        NO = W3.shape[0]
        R = B * K * K
        dev = Pt.device
        st = torch.cuda.current_stream(dev).cuda_stream
        g2 = g.reshape(R, NO).contiguous()
        with torch.cuda.device(dev):
            nparts = int(lib.spacap_rel_wide_tail_nparts(R))
            part = torch.empty(nparts, NO * C + C + 16, dtype=torch.float32, device=dev)
            dz2 = torch.empty_like(hid2)
            check(lib.spacap_rel_wide_tail_bwd_f32(g2.data_ptr(), W3.data_ptr(), hid2.data_ptr(), R, C, nparts, dz2.data_ptr(),
                                                   part.data_ptr(), st), "spacap_rel_wide_tail_bwd_f32")
            s = sum_slabs(part, deferrable=True)
            dW3, db2, db3 = s[:NO * C].view(NO, C), s[NO * C:NO * C + C], s[NO * C + C:NO * C + C + NO]
            nslab = int(lib.spacap_gemm_bf3_wgrad_slabs(R, C, C))
            pw = torch.empty(nslab, C * C, dtype=torch.float32, device=dev)
            if WIDE_WGRAD_TR:
                # the Linear layers' split-bf16 weight-gradient kernel (row-major images read by transposing LDS reads:
                # csrc/wgrad_bf3.inc) instead of gemm_bf3_wgrad_kernel's images staged transposed with two-byte LDS writes
                check(lib.spacap_linear_wgrad_nslab_f32(dz2.data_ptr(), hid1.data_ptr(), R, C, C, 0, nslab, pw.data_ptr(), st),
                      "spacap_linear_wgrad_nslab_f32")
            else:
                check(lib.spacap_gemm_bf3_wgrad_f32(dz2.data_ptr(), C, hid1.data_ptr(), C, R, C, C, nslab, pw.data_ptr(), st),
                      "spacap_gemm_bf3_wgrad_f32")
            dW2 = sum_slabs(pw, deferrable=True).view(C, C)
            dh1 = bf3_product(dz2, bf3_pieces(W2, trans=True), out=hid2)      # hid2 is dead: its memory takes dhid1
            del dz2
            dPt, dU = torch.empty_like(Pt), torch.empty_like(U)
            pb = torch.empty(B * K, C, dtype=torch.float32, device=dev)
            check(lib.spacap_rel_wide_l1_bwd_f32(dh1.data_ptr(), hid1.data_ptr(), Pt.data_ptr(), U.data_ptr(), B, H, K, C, dPt.data_ptr(),
                                                 dU.data_ptr(), pb.data_ptr(), st), "spacap_rel_wide_l1_bwd_f32")
            dP = torch.empty(B, H, K, K, dtype=torch.float32, device=dev)
            check(lib.spacap_rel_wide_transpose_f32(dPt.data_ptr(), dP.data_ptr(), B, H, K, 0, st), "spacap_rel_wide_transpose_f32")
        return dP, dU, sum_slabs(pb), dW2, db2, dW3, db3


def relation_head_wide(P, V, lin1, lin2, lin3):
    """``relation_head`` for widths without the one-kernel form; ``None`` when these kernels do not cover the shape either."""
    B, H, K, D = V.shape
    C = lin1.weight.shape[0]
    if not P.is_cuda or P.dtype != torch.float32 or lin1.bias is None or lin2.bias is None or lin3.bias is None or \
            tuple(lin1.weight.shape) != (C, H * D) or tuple(lin2.weight.shape) != (C, C) or lin3.weight.shape[1] != C or \
            lin3.weight.shape[0] != 9 or not lib.spacap_rel_wide_l1_supported(H, K, C) or \
            not lib.spacap_gemm_bf3_supported(C, C) or not lib.spacap_rel_wide_tail_supported(C):
        return None
    U = RelationU.apply(V, lin1.weight)
    return RelationWide.apply(P, U, lin1.bias, lin2.weight, lin2.bias, lin3.weight, lin3.bias)


def relation_head(P, V, lin1, lin2, lin3):
    """``lin3(relu(lin2(relu(lin1(feature(P, V))))))`` -> (B,K,K,9) through ``RelationHead``; ``None`` when the shape has
    no fused kernel (the caller composes relation_layer1 / relation_tail instead).  P (B,h,K,K), V (B,h,K,d)."""
    B, H, K, D = V.shape
    if not P.is_cuda or P.dtype != torch.float32 or lin1.bias is None or lin2.bias is None or lin3.bias is None or \
            tuple(lin1.weight.shape) != (128, H * D) or tuple(lin2.weight.shape) != (128, 128) or \
            not lib.spacap_relation_fused_supported(H, K, 128, lin3.weight.shape[0]) or lin3.weight.shape[1] != 128:
        return relation_head_wide(P, V, lin1, lin2, lin3) if torch.is_grad_enabled() or P.is_cuda else None
    U = RelationU.apply(V, lin1.weight)   # (B,K,H,C): the first Linear applied per head to the value vectors
    return RelationHead.apply(P, U, lin1.bias, lin2.weight, lin2.bias, lin3.weight, lin3.bias)


# ---- dense row products of any shape (csrc/dense_rows.hip): what used to be rocBLAS GEMMs inside the step ----------------------

def _dense(dev, a_ptr, lda, W_ptr, ldw, trans_w, bias_ptr, R, K, CO, out_ptr, ldo, a_rows=(0, 0, 0), o_rows=(0, 0, 0), o_zero=0,
           slices=1, slice_stride=0, batch=0, a_z=0, w_z=0):
    check(lib.spacap_dense_rows_f32(a_ptr, lda, a_rows[0], a_rows[1], a_rows[2], W_ptr, ldw, 1 if trans_w else 0, bias_ptr, R, K, CO,
                                    out_ptr, ldo, o_rows[0], o_rows[1], o_rows[2], int(o_zero), int(slices), int(slice_stride),
                                    int(batch), int(a_z), int(w_z), torch.cuda.current_stream(dev).cuda_stream),
          "spacap_dense_rows_f32")


def dense_product(a2, W, trans_w, bias=None, col0=0, ncols=None):
    """out = a2 Wv^T (+ bias) when ``trans_w`` else a2 Wv, with Wv = W[:, col0 : col0 + ncols] a COLUMN SLICE of the dense 2-D
    matrix W (no copy: the kernel takes the row stride).  a2 (R, K): rows of K contiguous floats at ANY uniform row stride (a
    column window of wider rows is read in place).  Any sizes."""
    R, K = a2.shape
    ldw = W.shape[1]
    ncols = ldw - col0 if ncols is None else ncols
    CO = W.shape[0] if trans_w else ncols
    assert (K == 1 or a2.stride(1) == 1) and W.is_contiguous() and (ncols == K if trans_w else W.shape[0] == K)
    lda = a2.stride(0) if R > 1 else K
    dev = a2.device
    with torch.cuda.device(dev):
        out = torch.empty(R, CO, dtype=torch.float32, device=dev)
        _dense(dev, a2.data_ptr(), lda, W.data_ptr() + 4 * col0, ldw, trans_w, bias.data_ptr() if bias is not None else None, R, K, CO,
               out.data_ptr(), CO)
    return out


class VocabProjection(Function):
    """logits[b, t, :] = n[b, t + skip, :] W^T + bias, t < L - skip: the caption head's vocabulary projection
    (models/transformer_captioner.py:93-100 ``Generator.proj``) applied to positions skip.. of every sequence
    (:373-379: the decoder's output without the object-indicator position) -- read in place instead of through a slice copy,
    and with the data gradient written straight into the full (B, L, D) layout (position < skip: zeros).  Forward, data
    gradient (split over the 3 001-long reduction, slices added in order) and weight / bias gradient are csrc/dense_rows.hip."""

    @staticmethod
    def forward(ctx, n, weight, bias, skip):
        if not n.is_cuda:
            raise RuntimeError("CPU not supported")
        n, w = n.contiguous(), weight.contiguous()
        B, L, D = n.shape
        V, T = w.shape[0], L - skip
        dev = n.device
        with torch.cuda.device(dev):
            logits = torch.empty(B, T, V, dtype=torch.float32, device=dev)
            _dense(dev, n.data_ptr(), D, w.data_ptr(), D, True, bias.data_ptr() if bias is not None else None, B * T, D, V,
                   logits.data_ptr(), V, a_rows=(T, L * D, skip) if skip else (0, 0, 0))
        ctx.save_for_backward(n, w)
        ctx.skip, ctx.has_bias = int(skip), bias is not None
        return logits

    @staticmethod
    def backward(ctx, g):
        n, w = ctx.saved_tensors
        skip = ctx.skip
        B, L, D = n.shape
        V, T = w.shape[0], L - skip
        R = B * T
        dev = n.device
        g = g.contiguous()
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            dn = None
            if ctx.needs_input_grad[0]:
                S = int(lib.spacap_dense_rows_slices(R, V, D))
                dn = torch.empty(B, L, D, dtype=torch.float32, device=dev)
                rows = (T, L * D, skip) if skip else (0, 0, 0)
                if S > 1:
                    parts = torch.empty(S, B * L * D, dtype=torch.float32, device=dev)
                    _dense(dev, g.data_ptr(), V, w.data_ptr(), D, False, None, R, V, D, parts.data_ptr(), D, o_rows=rows,
                           o_zero=1 if skip else 0, slices=S, slice_stride=B * L * D)
                    check(lib.spacap_dense_sum_slices_f32(parts.data_ptr(), S, B * L * D, B * L * D, dn.data_ptr(), st),
                          "spacap_dense_sum_slices_f32")
                else:
                    _dense(dev, g.data_ptr(), V, w.data_ptr(), D, False, None, R, V, D, dn.data_ptr(), D, o_rows=rows,
                           o_zero=1 if skip else 0)
            dW = torch.empty(V, D, dtype=torch.float32, device=dev)
            db = torch.empty(V, dtype=torch.float32, device=dev) if ctx.has_bias else None
            # (launched here, not queued with the other optimizer-only work: the incoming gradient is the caption loss's own buffer,
            # which does not outlive this node -- queued, the kernel read what the decoder's backward had written over it)
            check(lib.spacap_dense_wgrad_small_f32(g.data_ptr(), V, n.data_ptr(), D, T if skip else 0, L * D, skip, R, V, D,
                                                   dW.data_ptr(), db.data_ptr() if db is not None else None, st),
                  "spacap_dense_wgrad_small_f32")
        return dn, dW, db, None


def vocab_projection(n, lin, skip):
    """``lin(n[:, skip:, :])`` for the vocabulary ``nn.Linear`` ``lin``; None when the tensors are not float32 CUDA."""
    if not (n.is_cuda and n.dtype == torch.float32 and n.dim() == 3 and lin.weight.shape[1] == n.shape[-1] and n.shape[1] > skip):
        return None
    return VocabProjection.apply(n, lin.weight, lin.bias, skip)


class RelationU(Function):
    """U[b, j, h, :] = W1[:, d h : d h + d] V[b, h, j, :]: the relation head's first Linear applied per head to the value vectors
    (models/transformer_captioner.py:319-326 with the pair feature of :393-397 factored, see relation_head).  V is the (B, h, K, d)
    VIEW of the packed q | k | v projection (row stride 3 h d): read in place.  h independent products of one launch each way;
    the weight gradient is h diagonal blocks of dU^T V (per-slab partials in dW1's own layout + one deferrable sum)."""

    @staticmethod
    def forward(ctx, V, W1):
        B, H, K, D = V.shape
        C = W1.shape[0]
        Vr = V.transpose(1, 2)                      # (B, K, H, D)
        if not (Vr.stride(3) == 1 and Vr.stride(2) == D and Vr.stride(0) == K * Vr.stride(1)):
            Vr = Vr.contiguous()
        W1c = W1.contiguous()
        lda = Vr.stride(1)
        dev = V.device
        with torch.cuda.device(dev):
            U = torch.empty(B, K, H, C, dtype=torch.float32, device=dev)
            _dense(dev, Vr.data_ptr(), lda, W1c.data_ptr(), H * D, True, None, B * K, D, C, U.data_ptr(), H * C, slices=H,
                   slice_stride=C, batch=1, a_z=D, w_z=D)
        ctx.save_for_backward(Vr, W1c)
        return U

    @staticmethod
    def backward(ctx, dU):
        Vr, W1 = ctx.saved_tensors
        B, K, H, D = Vr.shape
        C = W1.shape[0]
        R = B * K
        dev = dU.device
        dU = dU.contiguous()
        lda = Vr.stride(1)
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            dV = None
            if ctx.needs_input_grad[0]:
                dVr = torch.empty(B, K, H, D, dtype=torch.float32, device=dev)
                _dense(dev, dU.data_ptr(), H * C, W1.data_ptr(), H * D, False, None, R, C, D, dVr.data_ptr(), H * D, slices=H,
                       slice_stride=D, batch=1, a_z=C, w_z=D)
                dV = dVr.transpose(1, 2)
            nslab = int(lib.spacap_dense_wgrad_blocks_slabs(R))
            part = torch.empty(nslab, C * H * D, dtype=torch.float32, device=dev)
            check(lib.spacap_dense_wgrad_blocks_f32(dU.data_ptr(), H * C, Vr.data_ptr(), lda, R, H, C, D, nslab, part.data_ptr(), st),
                  "spacap_dense_wgrad_blocks_f32")
            dW1 = sum_slabs(part, deferrable=True).view(C, H * D)
        return dV, dW1
