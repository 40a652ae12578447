"""Per-kernel timings of the fused Transformer sub-layer kernels (csrc/tf_layer.hip) at the cfg2 shapes: encoder rows
R = 8 x 256, decoder rows R = 8 x 32, d_model 128, d_ff 2048.  HIP events around ITERS back-to-back launches.
    python tools/lab/tf_bench.py [ITERS]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spacap3d_amd._native import TfRowsArgs, check, lib  # noqa: E402

DEV = torch.device("cuda:0")
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def t(fn, what, flops=None, bytes_=None):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / ITERS
    extra = ""
    if flops:
        extra += f"  {flops / us * 1e-6:7.1f} TFLOP/s"
    if bytes_:
        extra += f"  {bytes_ / us * 1e-6:6.2f} TB/s"
    print(f"{what:58s} {us:8.1f} us{extra}", flush=True)
    return us


def rows(mode, R, **kw):
    a = TfRowsArgs()
    a.mode, a.R = mode, R
    for k in ("k1", "n2", "nparts"):
        setattr(a, k, int(kw.pop(k, 0)))
    a.drop_p, a.eps, a.seed, a.seed_dev = float(kw.pop("drop_p", 0.1)), 1e-6, 1234, None
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if v is not None else None)
    st = torch.cuda.current_stream().cuda_stream
    return lambda: check(lib.spacap_tf_rows_f32(ctypes.byref(a), st), "rows")


def mha(B, L, p=0.1):
    import math
    h, dk, hd = 8, 16, 128
    st = torch.cuda.current_stream().cuda_stream
    qkv, dout = torch.randn(B, L, 3 * hd, device=DEV), torch.randn(B, L, hd, device=DEV)
    mask = (torch.rand(B, 1, L, device=DEV) > 0.3).to(torch.uint8)
    mask[..., 0] = 1
    rng = torch.zeros(1, dtype=torch.int64, device=DEV)
    out, lse = torch.empty(B, L, hd, device=DEV), torch.empty(B, h, L, 2, device=DEV)
    strides = (L * 3 * hd, dk, 3 * hd)
    base = qkv.data_ptr()
    args = (base, base + hd * 4, base + 2 * hd * 4, *strides, *strides, *strides, mask.data_ptr(), L, 0, None, 0, 0, 0, B, h, L, L, dk,
            1.0 / math.sqrt(dk), p, 77, rng.data_ptr())
    g, ws = torch.empty_like(qkv), torch.empty(B * h * L, device=DEV)
    gp = (g.data_ptr(), g.data_ptr() + hd * 4, g.data_ptr() + 2 * hd * 4, 3 * hd, st)
    fl = 4.0 * B * h * L * L * dk
    t(lambda: check(lib.spacap_mha_fwd_f32(*args, out.data_ptr(), None, lse.data_ptr(), st), "f"), f"mha fwd B={B} L={L}", fl)
    t(lambda: check(lib.spacap_mha_bwd_f32(*args, lse.data_ptr(), dout.data_ptr(), None, ws.data_ptr(), *gp), "b"),
      f"mha bwd, two launches B={B} L={L}", 2.5 * fl)
    t(lambda: check(lib.spacap_mha_bwd_delta_f32(*args, lse.data_ptr(), dout.data_ptr(), ws.data_ptr(), *gp), "b"),
      f"mha bwd, one launch (delta given) B={B} L={L}", 2.5 * fl)


def main():
    r = lambda *s: torch.randn(*s, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    dff = 2048
    mha(8, 256)
    mha(8, 32)
    for R in (2048, 256):
        print(f"--- R = {R}")
        x, n, a_, h = r(R, 128), r(R, 128), r(R, 128), torch.relu(r(R, dff))
        W1, b1, W2, b2 = r(dff, 128) * 0.1, r(dff), r(128, dff) * 0.05, r(128)
        Wo, bo, pw, pb = r(128, 128) * 0.1, r(128), r(384, 128) * 0.1, r(384)
        la, lb = r(128), r(128)
        xo, no, stats, qkv = r(R, 128), r(R, 128), r(R, 2), r(R, 384)
        hout, dh = torch.empty(R, dff, device=DEV), torch.empty(R, dff, device=DEV)
        g128, gdff = r(R, 128), r(R, dff)
        part = torch.empty(int(lib.spacap_tf_rows_parts(R)), 256, device=DEV)
        f1 = 2.0 * R * 128 * dff
        t(lambda: check(lib.spacap_tf_ffn1_f32(n.data_ptr(), W1.data_ptr(), b1.data_ptr(), R, dff, 0.1, 7, None, hout.data_ptr(), st), "f"),
          "ffn1: relu-dropout(n W1^T + b1)", f1)
        t(lambda: check(lib.spacap_linear_dgrad_mask_f32(g128.data_ptr(), W2.data_ptr(), h.data_ptr(), 1.1, R, 128, dff, dh.data_ptr(), st), "m"),
          "linear_dgrad_mask (sa_mlp.hip)", f1)
        t(lambda: check(lib.spacap_tf_dgrad_mask_f32(g128.data_ptr(), W2.data_ptr(), h.data_ptr(), 1.1, R, 128, dff, dh.data_ptr(), st), "m"),
          "tf_dgrad_mask", f1)
        pt = torch.empty(dff // 128, R, 128, device=DEV)
        t(lambda: check(lib.spacap_tf_ffn_f32(0, n.data_ptr(), W1.data_ptr(), W2.data_ptr(), b1.data_ptr(), None, R, dff, 0.1, 7, None,
                                              hout.data_ptr(), pt.data_ptr(), st), "c"), "chained ffn fwd (h + 16 partial sums)", 2 * f1)
        t(lambda: check(lib.spacap_tf_ffn_f32(1, g128.data_ptr(), W2.data_ptr(), W1.data_ptr(), None, h.data_ptr(), R, dff, 0.1, 0, None,
                                              dh.data_ptr(), pt.data_ptr(), st), "c"), "chained ffn bwd (dhid + 16 partial sums)", 2 * f1)
        for S in (16,):
            parts = torch.empty(S, R, 128, device=DEV)
            t(lambda: check(lib.spacap_tf_gemm_f32(h.data_ptr(), W2.data_ptr(), R, dff, 128, 1, S, parts.data_ptr(), st), "g"),
              f"split product h W2^T, {S} slices", f1)
            t(lambda: check(lib.spacap_tf_gemm_f32(gdff.data_ptr(), W1.data_ptr(), R, dff, 128, 0, S, parts.data_ptr(), st), "g"),
              f"split product dhid W1, {S} slices", f1)
            t(rows(0, R, a1=parts, nparts=S, bias1=b2, res=x, x_out=xo, ln_a=la, ln_b=lb, n_out=no, stats=stats, w2=pw, bias2=pb, n2=384,
                   out2=qkv), f"rows fwd: sum {S} slices + residual + LN + qkv")
            t(rows(1, R, a1=parts, nparts=S, x_ln=x, stats=stats, ln_a=la, res=g128, x_out=xo, part=part, n_out=no, w2=Wo, n2=128, out2=qkv),
              f"rows bwd: sum {S} slices + LN' + residual + dropout' + (. Wo)")
        t(rows(0, R, res=x, ln_a=la, ln_b=lb, n_out=no, stats=stats, w2=pw, bias2=pb, n2=384, out2=qkv), "rows fwd: LN + qkv")
        t(rows(0, R, a1=a_, w1=Wo, bias1=bo, k1=128, res=x, x_out=xo, ln_a=la, ln_b=lb, n_out=no, stats=stats),
          "rows fwd: out-proj + residual + LN")
        t(rows(0, R, a1=h, w1=W2, bias1=b2, k1=dff, res=x, x_out=xo, ln_a=la, ln_b=lb, n_out=no, stats=stats, w2=pw, bias2=pb, n2=384,
               out2=qkv), "rows fwd: in-kernel K = 2048 product + residual + LN + qkv", f1)
        t(rows(1, R, a1=qkv, w1=pw, k1=384, x_ln=x, stats=stats, ln_a=la, res=g128, x_out=xo, part=part, n_out=no),
          "rows bwd: dqkv Wqkv + LN' + residual + dropout'")
        t(rows(1, R, a1=gdff, w1=W1, k1=dff, x_ln=x, stats=stats, ln_a=la, res=g128, x_out=xo, part=part, n_out=no, w2=Wo, n2=128,
               out2=qkv), "rows bwd: in-kernel K = 2048 product + ...", f1)


if __name__ == "__main__":
    main()
