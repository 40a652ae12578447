"""The captioner's decoder input for a training step as one op (csrc/caption_prep.hip).

Reference: models/transformer_captioner.py:350-367 (nearest proposal to the referred object, object indicator, teacher-forcing
tokens and mask), :246-249 (indicator + encoder output of that proposal), :129-137 / :150-161 (embedding * sqrt(d_model) +
positional encoding, dropout), :193-199 (indicator prepended).  Early-guide mode with the Transformer encoder only."""
import torch
from torch.autograd import Function

from ._native import check, lib
from .attention import _next_seed, rng_state

class CaptionPrep(Function):
    @staticmethod
    def forward(ctx, xyz, ref, src, memory, tok, emb, pe, p, seed):
        dev = src.device
        B, K, D = src.shape
        T, V = tok.shape[1], emb.shape[0]
        L = T - 1
        xyz, ref, srcc, tok, embc, pe = xyz.contiguous(), ref.contiguous(), src.contiguous(), tok.contiguous(), emb.contiguous(), pe.contiguous()
        mem = memory.contiguous() if memory is not None else None
        with torch.cuda.device(dev):
            x0 = torch.empty(B, L, D, dtype=torch.float32, device=dev)
            mask = torch.empty(B, L, L, dtype=torch.uint8, device=dev)
            idx = torch.empty(B, dtype=torch.int64, device=dev)
            dist = torch.empty(B, dtype=torch.float32, device=dev)
            good = torch.empty(B, dtype=torch.bool, device=dev)
            pred = torch.empty(1, dtype=torch.float32, device=dev)
            ticket = torch.empty(1, dtype=torch.int32, device=dev)   # last-block ticket of THIS launch (zeroed by the entry point)
            check(lib.spacap_caption_prep_fwd_f32(xyz.data_ptr(), ref.data_ptr(), srcc.data_ptr(), mem.data_ptr() if mem is not None else None,
                                                  tok.data_ptr(), embc.data_ptr(), pe.data_ptr(), B, K, D, T, V, float(p), int(seed),
                                                  rng_state(dev).data_ptr(), x0.data_ptr(), mask.data_ptr(), idx.data_ptr(),
                                                  dist.data_ptr(), good.data_ptr(), pred.data_ptr(), ticket.data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream), "spacap_caption_prep_fwd_f32")
        ctx.save_for_backward(tok, idx)
        ctx.dims = (B, K, D, T, V, float(p), int(seed), memory is not None)
        ctx.mark_non_differentiable(mask, idx, dist, good, pred)
        ctx.set_materialize_grads(False)   # no zero tensors for the five index / scalar outputs' gradients
        return x0, mask, idx, dist, good, pred

    @staticmethod
    def backward(ctx, g, *_):
        tok, idx = ctx.saved_tensors
        B, K, D, T, V, p, seed, has_mem = ctx.dims
        dev = g.device
        g = g.contiguous()
        need_rows = ctx.needs_input_grad[2] or (has_mem and ctx.needs_input_grad[3])
        with torch.cuda.device(dev):
            d_rows = torch.empty(B, K, D, dtype=torch.float32, device=dev) if need_rows else None
            d_emb = torch.empty(V, D, dtype=torch.float32, device=dev)
            check(lib.spacap_caption_prep_bwd_f32(g.data_ptr(), tok.data_ptr(), idx.data_ptr(), B, K, D, T, V, p, seed,
                                                  rng_state(dev).data_ptr(), d_rows.data_ptr() if need_rows else None,
                                                  d_emb.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                  "spacap_caption_prep_bwd_f32")
        return (None, None, d_rows if ctx.needs_input_grad[2] else None, d_rows if (has_mem and ctx.needs_input_grad[3]) else None,
                None, d_emb, None, None, None)


def caption_prep(xyz, ref, src, memory, tok, embed, pos):
    """``embed``: the Embeddings module, ``pos``: the PositionalEncoding module (sinusoidal table + dropout).  Returns
    (x0 (B,L,D), mask uint8 (B,L,L), match_idx (B), dist (B), good (B) bool, pred_ious ()), or ``None`` when the op does not
    apply (CPU tensors, other dtypes, a table shorter than the sequence)."""
    T = tok.shape[1]
    if not src.is_cuda or src.dtype != torch.float32 or tok.dtype != torch.int64 or T < 2 or pos.pe.shape[1] < T - 2 or \
            embed.lut.weight.shape[1] != src.shape[-1]:
        return None
    xyz = xyz.detach()   # the match is an arg-min and the distance only feeds the reported ``pred_ious``: no gradient path
    p = float(pos.dropout.p) if pos.dropout.training else 0.0
    x0, mask, idx, dist, good, pred = CaptionPrep.apply(xyz, ref, src, memory, tok, embed.lut.weight, pos.pe[0], p,
                                                        _next_seed() if p > 0.0 else 0)
    return x0, mask, idx, dist, good, pred[0]
