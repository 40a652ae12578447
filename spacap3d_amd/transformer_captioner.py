"""Spatiality-guided Transformer captioner -- counterpart of ``models/transformer_captioner.py``.

Same module tree / parameter names as the reference (so its state dicts load), same ``data_dict``
contract.  The hot spots run on the HIP library:
  * ``attention()`` (reference :27-37) -> one fused kernel (``spacap3d_amd.attention``); the L x L matrix
    ``p_attn`` is materialised only for layers whose ``self.attn`` is consumed (``store_attn``), i.e. the
    last encoder layer (relation head, :392-394) -- or for every layer when ``store_attn_all`` is set,
    which reproduces the reference's always-on ``self.attn`` (used by its eval-time attention dumps);
  * the relation feature  R[b,i,j,(h,d)] = P[b,h,i,j] * V[b,h,j,d]  (:393-396) is formed by one broadcast
    product (the reference materialises an extra ``repeat`` copy of P first).
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .backend import ops
from .loss_helper import nn_distance

MAX_DES_LEN = 30  # lib/config.py:55
PROPOSAL_FEATURE_DIM = 128  # models/proposal_module.py:39 (vote aggregation mlp [..., 128])


def subsequent_mask(size, device=None):
    """(1, size, size) bool, True on and below the diagonal (:16-20)."""
    return torch.ones(1, size, size, dtype=torch.bool, device=device).tril()


def clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def attention(query, key, value, mask=None, dropout=None, need_p=True):
    """Reference signature (:27) plus ``need_p``; returns (P V, p_attn)."""
    p = dropout.p if dropout is not None else 0.0
    training = dropout.training if dropout is not None else False
    return ops().attention(query, key, value, mask=mask, dropout_p=p, training=training, need_p=need_p)


def _linear(x, lin):
    """``lin(x)`` through the backend's Linear op (one-launch weight + bias gradient) when it has one."""
    f = getattr(ops(), "linear", None)
    if f is None or lin.bias is None or not x.is_cuda:
        return lin(x)
    return f(x, lin.weight, lin.bias)


def _linear_wb(x, w, b):
    f = getattr(ops(), "linear", None)
    return f(x, w, b) if (f is not None and x.is_cuda) else F.linear(x, w, b)


def _sum_slabs(t):
    if not t.is_cuda:
        return t.sum(0)
    from ._native import sum_slabs
    return sum_slabs(t.contiguous())


class _TallLinear(torch.autograd.Function):
    """y = x W^T + b for inputs with very many rows (the relation head runs its MLP on B*K*K = 524 288 pair
    features).  Forward and dX are ordinary GEMMs; the weight gradient dW = G^T X reduces over all rows into a
    128 x 128 output, which the BLAS heuristics run as a handful of tiles with no split over K (1.08 ms per
    layer on MI355X); here the rows are cut into 64 slabs, multiplied as one batched GEMM and summed
    (0.15 ms, fixed summation order)."""

    SLABS = 64

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        rows, S = g2.shape[0], _TallLinear.SLABS
        dx = (g2 @ weight).view_as(x) if ctx.needs_input_grad[0] else None
        if rows % S == 0 and rows >= 64 * S:
            dw = _sum_slabs(torch.bmm(g2.view(S, rows // S, -1).transpose(1, 2), x2.view(S, rows // S, -1)))
        else:
            dw = g2.t() @ x2
        # column sums of a very tall, narrow matrix: torch's reduction takes 0.66 ms for 524 288 x 9; two stages 17 us
        db = g2.view(1024, rows // 1024, -1).sum(1).sum(0) if rows % 1024 == 0 and rows >= 65536 else g2.sum(0)
        return dx, dw, db


def tall_linear(x, lin):
    return _TallLinear.apply(x, lin.weight, lin.bias)


class MultiHeadedAttention(nn.Module):
    def __init__(self, h, d_model, dropout=0.1, keep_value=False, store_attn=None):
        super().__init__()
        assert d_model % h == 0
        self.d_k = d_model // h
        self.h = h
        self.linears = clones(nn.Linear(d_model, d_model), 4)
        self.attn = None
        self.value = None
        self.dropout = nn.Dropout(p=dropout)
        self.keep_value = keep_value
        # None: follow the module-wide default (see TransformerDecoderModel.store_attn_all)
        self.store_attn = store_attn

    def forward(self, query, key, value, mask=None):
        nb = query.size(0)
        packed = getattr(ops(), "self_attention_packed", None) if (query is key and key is value) else None
        if packed is not None and query.is_cuda:
            # self-attention: q, k, v from ONE GEMM over the concatenated weights; the kernels read the packed result
            # through strides and write one packed gradient (see attention.FusedSelfAttentionPacked)
            hd = self.h * self.d_k
            pk = getattr(self, "_packed_qkv", None)
            if pk is not None and pk[0].data_ptr() == self.linears[0].weight.data_ptr():
                # the three weights are adjacent in the optimizer's flat buffer (engine.py): no concatenation
                from .linear import PackedLinear
                qkv = PackedLinear.apply(query, pk[0], pk[1], *[l.weight for l in self.linears[:3]],
                                         *[l.bias for l in self.linears[:3]])
            else:
                w = torch.cat([l.weight for l in self.linears[:3]], dim=0)
                b = torch.cat([l.bias for l in self.linears[:3]], dim=0)
                qkv = _linear_wb(query, w, b)
            need_p = self.keep_value if self.store_attn is None else (self.store_attn or self.keep_value)
            p = self.dropout.p
            x, self.attn = packed(qkv, self.h, mask=mask, dropout_p=p, training=self.dropout.training, need_p=need_p)
            if self.keep_value:
                self.value = qkv[..., 2 * hd:].view(nb, -1, self.h, self.d_k).transpose(1, 2)
            return _linear(x, self.linears[-1])
        if mask is not None:
            mask = mask.unsqueeze(1)
        query, key, value = [l(x).view(nb, -1, self.h, self.d_k).transpose(1, 2)
                             for l, x in zip(self.linears, (query, key, value))]
        need_p = self.keep_value if self.store_attn is None else (self.store_attn or self.keep_value)
        x, self.attn = attention(query, key, value, mask=mask, dropout=self.dropout, need_p=need_p)
        if self.keep_value:
            self.value = value
        x = x.transpose(1, 2).contiguous().view(nb, -1, self.h * self.d_k)
        return _linear(x, self.linears[-1])


    def forward_incremental(self, x_new, cache, mask=None):
        """Self-attention for the newest token(s) only: project q/k/v of ``x_new`` (B, t_new, d), append k/v to
        ``cache`` (a dict holding 'k' and 'v' of the previous positions), attend over everything cached.  ``mask``
        (B, t_new, t_total) is only needed when several new tokens are fed at once (the first step)."""
        nb = x_new.size(0)
        q, k, v = [l(x_new).view(nb, -1, self.h, self.d_k).transpose(1, 2) for l in self.linears[:3]]
        if cache.get("k") is not None:
            k = torch.cat([cache["k"], k], dim=2)
            v = torch.cat([cache["v"], v], dim=2)
        cache["k"], cache["v"] = k, v
        m = mask.unsqueeze(1) if mask is not None else None
        x, _ = attention(q, k.contiguous(), v.contiguous(), mask=m, dropout=self.dropout, need_p=False)
        x = x.transpose(1, 2).contiguous().view(nb, -1, self.h * self.d_k)
        return self.linears[-1](x)


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_model, d_ff, dropout=0.1):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        h = _linear(x, self.w_1)
        tail = getattr(ops(), "ffn_tail", None)
        if tail is not None and h.is_cuda:
            out = tail(h, self.w_2.weight, self.w_2.bias, self.dropout.p, self.dropout.training)
            if out is not None:
                return out
        f = getattr(ops(), "relu_dropout", None)
        h = f(h, self.dropout.p, self.dropout.training) if (f is not None and h.is_cuda) else self.dropout(F.relu(h))
        return _linear(h, self.w_2)


class Embeddings(nn.Module):
    def __init__(self, d_model, vocab):
        super().__init__()
        self.lut = nn.Embedding(vocab, d_model)
        self.d_model = d_model

    def forward(self, x):
        return self.lut(x) * math.sqrt(self.d_model)


class Generator(nn.Module):
    def __init__(self, d_model, vocab):
        super().__init__()
        self.proj = nn.Linear(d_model, vocab)

    def forward(self, x):
        return F.log_softmax(self.proj(x), dim=-1)


class LayerNorm(nn.Module):
    """a * (x - mean) / (std_unbiased + eps) + b   (:102-113) -- NOT nn.LayerNorm."""

    def __init__(self, features, eps=1e-6):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x):
        return ops().layer_norm(x, self.a_2, self.b_2, self.eps)


class SublayerConnection(nn.Module):
    def __init__(self, size, dropout):
        super().__init__()
        self.norm = LayerNorm(size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, sublayer):
        nr = getattr(ops(), "layer_norm_residual", None)
        if nr is not None and x.is_cuda and torch.is_grad_enabled() and x.requires_grad:
            # norm and residual operand from one node: its backward adds the two gradient paths into x itself
            normed, x = nr(x, self.norm.a_2, self.norm.b_2, self.norm.eps)
            y = sublayer(normed)
        else:
            y = sublayer(self.norm(x))
        f = getattr(ops(), "dropout_add", None)
        if f is not None and y.is_cuda:
            return f(x, y, self.dropout.p, self.dropout.training)
        return x + self.dropout(y)


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).unsqueeze(1).float()
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x, src_pos=None):
        return self.dropout(x + self.pe[:, :x.size(1)])


class PositionalEncodingLearned(nn.Module):
    def __init__(self, input_channel, d_model=128):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            nn.Conv1d(input_channel, d_model, kernel_size=1), nn.BatchNorm1d(d_model), nn.ReLU(inplace=True),
            nn.Conv1d(d_model, d_model, kernel_size=1))

    def forward(self, x, xyz):
        h = self.position_embedding_head
        t = xyz.transpose(1, 2).contiguous()
        if self.training and t.is_cuda and getattr(ops(), "bn_relu_train", None) is not None:
            conv = getattr(ops(), "conv1x1", None)

            def c(m, v):   # 1x1 convolution whose weight gradient joins the step's deferred batch (linear.Conv1x1)
                y = conv(v, m) if conv is not None else None
                return m(v) if y is None else y
            t = c(h[3], ops().bn_relu_train(c(h[0], t), h[1]))   # Conv1d -> [BatchNorm1d -> ReLU as one fused op] -> Conv1d
        elif t.is_cuda and not torch.is_grad_enabled() and getattr(ops(), "conv1x1", None) is not None:
            conv = ops().conv1x1     # inference forward: the two 1x1 convolutions on the library's kernel, BatchNorm on running statistics

            def c(m, v):
                y = conv(v, m)
                return m(v) if y is None else y
            g = getattr(ops(), "bn_relu_eval", None)
            u = c(h[0], t)
            v = g(u, h[1]) if g is not None else None
            t = c(h[3], h[2](h[1](u)) if v is None else v)
        else:
            t = h(t)
        return x + t.transpose(1, 2).contiguous()


class Encoder(nn.Module):
    def __init__(self, layer, N):
        super().__init__()
        self.layers = clones(layer, N)
        self.norm = LayerNorm(layer.size)

    def forward(self, x, mask):
        st = getattr(ops(), "tf_stack", None)
        if st is not None and st.stack_supported(self.layers, x):
            return st.run_stack(self.layers, self.norm, x, mask)   # 4 launches per layer (spacap3d_amd/tf_layer.py)
        for layer in self.layers:
            x = layer(x, mask)
        return self.norm(x)


class EncoderLayer(nn.Module):
    def __init__(self, size, self_attn, feed_forward, dropout):
        super().__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.sublayer = clones(SublayerConnection(size, dropout), 2)
        self.size = size

    def forward(self, x, mask):
        x = self.sublayer[0](x, lambda y: self.self_attn(y, y, y, mask))
        return self.sublayer[1](x, self.feed_forward)


class Decoder(nn.Module):
    def __init__(self, layer, N):
        super().__init__()
        self.layers = clones(layer, N)
        self.norm = LayerNorm(layer.size)

    def forward(self, x, memory, src_mask, tgt_mask, obj_indicator=None):
        if obj_indicator is not None:
            x = torch.cat((obj_indicator, x), dim=1)
        st = getattr(ops(), "tf_stack", None)
        if st is not None and st.stack_supported(self.layers, x):   # early guide: self-attention + feed-forward only
            return st.run_stack(self.layers, self.norm, x, tgt_mask)
        for layer in self.layers:
            x = layer(x, memory, src_mask, tgt_mask)
        return self.norm(x)


class DecoderLayer(nn.Module):
    def __init__(self, size, self_attn, src_attn, feed_forward, dropout, early_guide=True):
        super().__init__()
        self.size = size
        self.self_attn = self_attn
        self.src_attn = src_attn
        self.feed_forward = feed_forward
        self.early_guide = early_guide
        self.sublayer = clones(SublayerConnection(size, dropout), 3)

    def forward(self, x, memory, src_mask, tgt_mask):
        m = memory
        x = self.sublayer[0](x, lambda y: self.self_attn(y, y, y, tgt_mask))
        if not self.early_guide:
            x = self.sublayer[1](x, lambda y: self.src_attn(y, m, m, src_mask))
        return self.sublayer[2](x, self.feed_forward)


def decode_incremental(decoder, x_new, caches, memory=None, src_mask=None, mask=None):
    """Run ``x_new`` (B, t_new, d) -- the positions not seen yet -- through the decoder stack, reusing the cached
    keys / values of earlier positions (one dict per layer).  Pre-LN layers with a causal mask make this exactly
    the last rows of the full recomputation the reference performs at every step (:435-438)."""
    x = x_new
    for layer, cache in zip(decoder.layers, caches):
        x = x + layer.sublayer[0].dropout(layer.self_attn.forward_incremental(layer.sublayer[0].norm(x), cache, mask))
        if not layer.early_guide:
            x = layer.sublayer[1](x, lambda y: layer.src_attn(y, memory, memory, src_mask))
        x = layer.sublayer[2](x, layer.feed_forward)
    return decoder.norm(x)


class _Identity(nn.Module):
    def forward(self, x, *a):
        return x


class EncoderDecoder(nn.Module):
    def __init__(self, encoder, decoder, src_embed, tgt_embed, generator, early_guide=True):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.src_embed = src_embed
        self.tgt_embed = tgt_embed
        self.generator = generator
        self.early_guide = early_guide

    def forward(self, src, tgt, src_mask, tgt_mask, obj_indicator=None, src_pos=None, obj_idx=None, memory=None):
        if memory is None:
            if self.encoder is None:
                memory = self.src_embed(src, src_pos) if src_pos is not None else src
            else:
                memory = self.encode(src, src_pos, src_mask)
        return self.decode(memory, src_mask, tgt, tgt_mask, obj_indicator=obj_indicator, obj_idx=obj_idx)

    def encode(self, src, src_pos, src_mask):
        return self.encoder(self.src_embed(src, src_pos), src_mask)

    def decode(self, memory, src_mask, tgt, tgt_mask, obj_indicator=None, obj_idx=None):
        if memory.shape[0] != tgt.shape[0]:  # inference: B*K sequences over B scenes (:252-257)
            assert memory.shape[0] * memory.shape[1] == tgt.shape[0]
            B, K, _ = memory.shape
            obj_indicator = obj_indicator + memory.reshape(B * K, -1).unsqueeze(1)
            if self.early_guide:
                # the reference materialises repeat_interleave(memory, K) = (B*K, K, d); in early-guide mode
                # the decoder never reads it (no cross-attention, :223-224), so it is not built here
                memory = None
        if obj_idx is not None:  # training (:260-261)
            obj_indicator = obj_indicator + torch.gather(
                memory, 1, obj_idx.repeat(1, memory.size(-1)).unsqueeze(1))
        if self.early_guide:
            return self.decoder(self.tgt_embed(tgt), memory, src_mask, tgt_mask, obj_indicator=obj_indicator)
        return self.decoder(self.tgt_embed(tgt), obj_indicator, None, tgt_mask, None)


class TransformerDecoderModel(nn.Module):
    def __init__(self, vocabulary, N, h, d_model, d_ff, transformer_dropout, bn_momentum=0.1, src_pos_type=None,
                 use_transformer_encoder=False, early_guide=True, check_relation=False, store_attn_all=False):
        super().__init__()
        self.word_to_idx = vocabulary["word2idx"]
        self.src_pos_type = src_pos_type
        self.vocabulary = vocabulary
        self.use_transformer_encoder = use_transformer_encoder
        self.check_relation = check_relation
        self.early_guide = early_guide
        self.store_attn_all = store_attn_all
        self.model = self.make_model(len(vocabulary["word2idx"]), N=N, h=h, d_model=d_model, d_ff=d_ff,
                                     dropout=transformer_dropout, bn_momentum=bn_momentum,
                                     src_pos_type=src_pos_type, use_transformer_encoder=use_transformer_encoder,
                                     early_guide=early_guide)
        # DEVIATION (BASELINE configs[4] "d_model=512"): the reference has no input projection, so
        # its encoder only runs with d_model == 128 (the proposal feature width, :163-164).  For other widths the 128-d
        # proposal tokens go through one Linear(128, d_model) first; with d_model == 128 (every reference configuration)
        # the module does not exist and the state-dict layout is the reference's.
        self.token_proj = nn.Linear(PROPOSAL_FEATURE_DIM, d_model) if d_model != PROPOSAL_FEATURE_DIM else None
        if check_relation:
            self.relation_proposal = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(),
                                                   nn.Linear(d_model, d_model), nn.ReLU(), nn.Linear(d_model, 9))
        if store_attn_all:
            for mod in self.modules():
                if isinstance(mod, MultiHeadedAttention):
                    mod.store_attn = True

    def make_model(self, tgt_vocab, N=6, h=8, d_model=512, d_ff=2048, dropout=0.1, bn_momentum=0.1,
                   src_pos_type=None, use_transformer_encoder=False, early_guide=True):
        c = copy.deepcopy
        attn = MultiHeadedAttention(h, d_model)
        ff = PositionwiseFeedForward(d_model, d_ff, dropout)
        position = PositionalEncoding(d_model, dropout)
        if src_pos_type is not None:
            src_position = PositionalEncodingLearned(3 if src_pos_type in ("xyz", "center") else 6, d_model)
        else:
            src_position = c(position)
        encoder = None
        if use_transformer_encoder:
            # only the LAST encoder layer's P and V are read (relation head), the reference keeps all six
            layer = EncoderLayer(d_model, MultiHeadedAttention(h, d_model, keep_value=False), c(ff), dropout)
            encoder = Encoder(layer, N)
            encoder.layers[-1].self_attn.keep_value = self.check_relation
        model = EncoderDecoder(
            encoder,
            Decoder(DecoderLayer(d_model, c(attn), c(attn), c(ff), dropout, early_guide), N),
            src_position if use_transformer_encoder else _Identity(),
            nn.Sequential(Embeddings(d_model, tgt_vocab), c(position)),
            Generator(d_model, tgt_vocab), early_guide=early_guide)
        for p in model.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in model.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.momentum = bn_momentum
        return model

    def _prepare_feature(self, seq):
        seq = seq[:, :-1] if self.early_guide else seq[:, 1:-1]
        seq_mask = (seq > 0).unsqueeze(-2)
        seq_mask = seq_mask & subsequent_mask(seq.size(-1), device=seq.device)
        return (seq[:, 1:] if self.early_guide else seq), seq_mask

    def forward(self, data_dict, is_eval=False):
        return self.forward_eval(data_dict) if is_eval else self.forward_train(data_dict)

    def _src_pos(self, ep):
        if self.src_pos_type == "xyz":
            return ep["aggregated_vote_xyz"]
        if self.src_pos_type == "center":
            return ep["center"]
        if self.src_pos_type == "loc":
            return torch.cat([ep["center"], ep["pred_size"]], -1)
        return None

    def relation_feature(self):
        """R[b,i,j,h*16+d] = P[b,h,i,j] * V[b,h,j,d] of the last encoder layer (:393-396)."""
        sa = self.model.encoder.layers[-1].self_attn
        return ops().relation_feature(sa.attn, sa.value)   # (B,h,K,K), (B,h,K,d_k) -> (B,K,K,h*d_k)

    def forward_train(self, ep):
        src = ep["aggregated_vote_features"]
        if self.token_proj is not None:
            src = _linear(src, self.token_proj)
        src_pos = self._src_pos(ep)
        prep = getattr(ops(), "caption_prep", None) if (src.is_cuda and self.early_guide and self.model.encoder is not None) else None
        src_mask = ep["bbox_mask"].unsqueeze(1)
        x0 = out_full = None
        if prep is not None:
            # encoder first, then nearest proposal + object indicator + token embedding + positional encoding + dropout + mask
            # as one launch (csrc/caption_prep.hip) and the decoder stack on its output
            memory = self.model.encode(src, src_pos, src_mask)
            if self.check_relation:
                self._relation_head_forked(ep)
            te = self.model.tgt_embed
            got = prep(ep["aggregated_vote_xyz"], ep["ref_center_label"], src, memory, ep["lang_label"], te[0], te[1])
            if got is not None:
                x0, m8, idx1, dist, good, pred_ious = got
                ep["match_idx"] = idx1
                out_full = self.model.decoder(x0, None, None, m8)
                out = out_full[:, 1:, :]
        if x0 is None:
            _, _, target_ious, idx = nn_distance(ep["aggregated_vote_xyz"], ep["ref_center_label"].unsqueeze(1))
            ep["match_idx"] = idx.squeeze(1)
            ref_obj_feature = torch.gather(src, 1, idx.repeat(1, src.size(-1)).unsqueeze(1))
            seq, seq_mask = self._prepare_feature(ep["lang_label"])
            if prep is None:
                memory = None
                if self.model.encoder is not None:
                    memory = self.model.encode(src, src_pos, src_mask)
                    if self.check_relation:
                        self._relation_head(ep)  # reads the last encoder layer only
            out = self.model(src=src, tgt=seq, src_mask=src_mask, tgt_mask=seq_mask,
                             obj_indicator=ref_obj_feature, src_pos=src_pos,
                             obj_idx=idx if self.use_transformer_encoder else None, memory=memory)
            out = out[:, 1:, :] if self.early_guide else out
            good = (target_ious > -1).squeeze(1)
            # mean over the good boxes without a host sync (the reference branches on .sum() > 0, :385)
            pred_ious = (target_ious.squeeze(1) * good).sum() / good.sum().clamp(min=1)
        fused = getattr(ops(), "caption_head_loss", None) if (out.is_cuda and self.training and "lang_ids" in ep) else None
        if fused is not None and ep["lang_ids"].shape[1] >= out.shape[1] + 1:
            # vocabulary projection, then log-softmax + the caption loss / accuracy in one op (fused_losses.CaptionHeadLoss);
            # loss_helper.get_scene_cap_loss picks the pair up instead of recomputing it from the log-probabilities
            # (the projection reads positions 1.. of the decoder output in place and writes its data gradient into the full
            # layout: no slice copy, no padding of the gradient -- linear.VocabProjection)
            vp = getattr(ops(), "vocab_projection", None) if out_full is not None else None
            logits = vp(out_full, self.model.generator.proj, 1) if vp is not None else None
            if logits is None:
                logits = self.model.generator.proj(out)
            ep["lang_cap"], cl, ca, ep["_cap_vec"] = fused(logits, ep["lang_ids"], good)
            ep["_cap_loss"] = (cl, ca)
        else:
            ep["lang_cap"] = self.model.generator(out)
        ep["pred_ious"] = pred_ious
        ep["good_bbox_masks"] = good
        if self.check_relation and "relation_pred" not in ep:
            self._relation_head(ep)
        return ep

    # The relation head reads the last encoder layer only and the decoder reads the encoder's output only: the two are independent
    # until the losses are added.  `fork_relation` (set by engine.Trainer for its captured step) runs the head on a stream of its own;
    # autograd then runs its backward on that stream as well, beside the decoder's backward (the engine orders the streams where
    # gradients cross).  Values are those of the serial order: no kernel changes, only placement.  The head's persistent grids
    # leave the decoder's launches room (spacap_relation_fused_leave_cus).
    fork_relation = False

    def _relation_head_forked(self, ep):
        src = ep["aggregated_vote_features"]
        if not (self.fork_relation and src.is_cuda and torch.is_grad_enabled()):
            return self._relation_head(ep)
        dev = src.device
        from .engine import _role_stream
        rs = _role_stream(dev, "relation")
        cur = torch.cuda.current_stream(dev)
        rs.wait_stream(cur)
        sa = self.model.encoder.layers[-1].self_attn
        with torch.cuda.stream(rs):
            self._relation_head(ep)
        # The caching allocator must know that both streams touch these -- INSIDE a capture as well: a block freed on the stream it
        # was allocated on is handed to the next allocation of that stream at once (the graph's private pool included), while the
        # other stream's kernel may still be reading it (seen: NaN losses at 2 scenes per GPU, the relation head's backward reading
        # an attention map whose memory the encoder's backward had already been given).  A recorded block is held back instead.
        for t in (sa.attn, sa.value):
            if torch.is_tensor(t):
                t.record_stream(rs)
        ep["relation_pred"].record_stream(cur)
        ep["_rel_stream"] = rs    # loss_helper.get_scene_cap_loss joins the streams before it reads relation_pred

    def _relation_head(self, ep):
        """relation_pred (B,K,K,9) from the last encoder layer's attention map and values (:392-397)."""
        rp = self.relation_proposal  # Linear-ReLU-Linear-ReLU-Linear (:319-326)
        sa = self.model.encoder.layers[-1].self_attn
        # the whole head as one kernel each way (csrc/relation_fused.hip); other widths compose the layers below
        head = getattr(ops(), "relation_head", None)
        pred = head(sa.attn, sa.value, rp[0], rp[2], rp[4]) if head is not None else None
        if pred is None:
            # feature (P (x) V) + first Linear + ReLU in one kernel: the (B,K,K,128) feature is never formed
            hid = ops().relation_layer1(sa.attn, sa.value, rp[0].weight, rp[0].bias)
            tail = getattr(ops(), "relation_tail", None)
            pred = tail(hid, rp[2], rp[4]) if (tail is not None and torch.is_grad_enabled()) else None
            if pred is None:
                pred = tall_linear(F.relu(tall_linear(hid, rp[2])), rp[4])
        ep["relation_pred"] = pred

    def forward_eval(self, ep, use_cache=True):
        """Greedy decoding of B*K captions (:402-453).  The reference re-runs the 6-layer encoder AND the whole
        decoder prefix at each of the 31 steps; the encoder output does not depend on the words (computed once),
        and with ``use_cache`` the decoder keeps the keys / values of earlier positions so that each step only
        processes the newest token (16x fewer token-layer evaluations).  ``use_cache=False`` recomputes the prefix
        as the reference does (kept for parity tests)."""
        obj_features = ep["aggregated_vote_features"]
        if self.token_proj is not None:
            obj_features = self.token_proj(obj_features)
        B, K, _ = obj_features.shape
        src_pos = self._src_pos(ep)
        if not self.use_transformer_encoder:
            src = torch.repeat_interleave(obj_features, K, dim=0)
            if src_pos is not None:
                src_pos = torch.repeat_interleave(src_pos, K, dim=0)
        else:
            src = obj_features
        src_mask = ep["bbox_mask"].unsqueeze(1)
        if self.model.encoder is None:
            memory = self.model.src_embed(src, src_pos) if src_pos is not None else src
        else:
            memory = self.model.encode(src, src_pos, src_mask)
        obj_flat = obj_features.reshape(B * K, -1)
        ys = torch.full((B * K, 1), self.word_to_idx["sos"], dtype=torch.long, device=obj_features.device)
        if not use_cache:
            for _ in range(MAX_DES_LEN + 1):
                L = ys.size(1) + 1 if self.early_guide else ys.size(1)
                out = self.model(src, ys, src_mask, subsequent_mask(L, device=ys.device),
                                 obj_flat.unsqueeze(1), src_pos=src_pos, memory=memory)
                next_word = self.model.generator(out[:, -1, :]).argmax(dim=-1)
                ys = torch.cat([ys, next_word.unsqueeze(1)], dim=1)
            ep["lang_cap"] = ys[:, 1:].view(B, K, -1)
            return ep
        # -- cached path -----------------------------------------------------------------------------------
        dec = self.model.decoder
        caches = [dict() for _ in dec.layers]
        if memory.shape[0] != B * K:  # encoder ran once per scene: the indicator adds that proposal's memory row
            indicator = obj_flat.unsqueeze(1) + memory.reshape(B * K, -1).unsqueeze(1)
            dec_memory = None
        else:
            indicator, dec_memory = obj_flat.unsqueeze(1), memory
        embed, pos = self.model.tgt_embed[0], self.model.tgt_embed[1]
        st = getattr(ops(), "tf_stack", None)
        if self.early_guide and not self.training and st is not None \
                and st.decode_supported(dec.layers, indicator.squeeze(1), MAX_DES_LEN + 1):
            # pre-allocated key / value caches, one token per sequence and step, four launches per layer (tf_layer.greedy_decode)
            words = st.greedy_decode(dec, self.model.generator, embed, pos.pe, indicator.squeeze(1), self.word_to_idx["sos"],
                                     MAX_DES_LEN + 1)
            ep["lang_cap"] = words.view(B, K, -1)
            return ep
        if self.early_guide:
            # positions: 0 = object indicator, 1.. = words (sinusoid added to the words only, as tgt_embed does)
            x_new = torch.cat((indicator, pos.dropout(embed(ys) + pos.pe[:, :1])), dim=1)
            mask = subsequent_mask(2, device=ys.device).expand(B * K, -1, -1)
            cross_mem, cross_mask = dec_memory, src_mask
        else:
            x_new = pos.dropout(embed(ys) + pos.pe[:, :1])
            mask = None
            cross_mem, cross_mask = indicator, None  # late guide: cross-attention over the 1-token indicator (:266)
        for step in range(MAX_DES_LEN + 1):
            out = decode_incremental(dec, x_new, caches, memory=cross_mem, src_mask=cross_mask, mask=mask)
            next_word = self.model.generator(out[:, -1, :]).argmax(dim=-1)
            ys = torch.cat([ys, next_word.unsqueeze(1)], dim=1)
            t = ys.size(1) - 1  # index of the newest word within the word sequence
            x_new = pos.dropout(embed(next_word.unsqueeze(1)) + pos.pe[:, t:t + 1])
            mask = None
        ep["lang_cap"] = ys[:, 1:].view(B, K, -1)
        return ep
