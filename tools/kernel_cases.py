"""The hot kernels of the training step as stand-alone launches at their cfg2 shapes (8 scenes x 40 000 points,
256 proposals), through the C ABI on synthetic device buffers -- measurement tooling shared by

  * bench.py             live HIP-event timing of the roofline kernels right after the timed steps,
  * tools/pmc_kernels.py the same launches under `rocprofv3 --pmc ...` (FETCH_SIZE / WRITE_SIZE / MFMA counters).

Every case states its ALGORITHMIC work per launch: flops and bytes.
  shared-MLP layer  flops 2*cin*cout*R, bytes 4*R*(cin+cout)
  FPS               bytes B*(m-1)*N*20 (streamed model), rounds m-1 (latency chain)
  MHA per layer     flops 4*B*h*Lq*Lk*d_k, bytes 4*B*L*4*h*d_k (+ 4*B*h*Lq*Lk when P is emitted)
  relation head     per layer flops 2*B*K*K*cin*cout, bytes 4*B*K*K*(cin+cout)
"""
import torch

from spacap3d_amd._native import check, lib

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (about 6.3 TB/s achievable)
PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD (= the fp32 vector peak)
PEAK_MFMA_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak (sustained by a pure MFMA loop on this pool: 2 100, tools/lab/clock_cal.hip)


def _st(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _rand(*shape, dev):
    return torch.randn(*shape, dtype=torch.float32, device=dev)


def _stats(C, dev):
    st = torch.empty(C, 4, dtype=torch.float32, device=dev)
    st[:, 0] = 0.05 * torch.randn(C, device=dev)
    st[:, 1] = 1.0 + 0.1 * torch.rand(C, device=dev)
    st[:, 2] = st[:, 1] * (1.0 + 0.1 * torch.rand(C, device=dev))
    st[:, 3] = 0.1 * torch.randn(C, device=dev)
    return st


def _part(C, dev):
    return torch.empty(int(lib.spacap_sa_nparts()) * 2 * C, dtype=torch.float64, device=dev)


def sa_mid_fwd(R, cin, cout, dev, label):
    zin, st, W = _rand(R, cin, dev=dev), _stats(cin, dev), _rand(cout, cin, dev=dev) * 0.1
    zout, part = torch.empty(R, cout, dtype=torch.float32, device=dev), _part(max(cin, cout), dev)

    def run():
        check(lib.spacap_sa_mid_fwd_f32(zin.data_ptr(), st.data_ptr(), W.data_ptr(), R, cin, cout, zout.data_ptr(),
                                        part.data_ptr(), _st(dev)), "sa_mid_fwd")
    import os
    split = os.environ.get("SPACAP_SA_F32MFMA", "0") in ("", "0") and cout % 128 == 0   # the library's default for these shapes
    return dict(name=f"sa_mid_fwd {cin}->{cout} R={R} ({label})", kernel="sa_mid_fwd", run=run, bf16_products=6 if split else 0,
                flops=2.0 * cin * cout * R, bytes=4.0 * R * (cin + cout), keep=(zin, st, W, zout, part),
                what=f"{label}: z_out = relu(bn(z_in)) W^T + batch statistics of z_out, {R} rows, {cin} -> {cout} channels")


def sa_mid_fwd_l1in(R, dev, label):
    """SA1 layer 2 as the step runs it since round 3: z1 rebuilt from the rows' four inputs (16 B per row) instead of read."""
    rel4, W1, st, W = _rand(R, 4, dev=dev), _rand(64, 4, dev=dev), _stats(64, dev), _rand(64, 64, dev=dev) * 0.1
    zout, part = torch.empty(R, 64, dtype=torch.float32, device=dev), _part(64, dev)

    def run():
        check(lib.spacap_sa_mid_fwd_l1in_f32(rel4.data_ptr(), W1.data_ptr(), 4, 1, st.data_ptr(), W.data_ptr(), R, zout.data_ptr(),
                                             part.data_ptr(), _st(dev)), "sa_mid_fwd_l1in")
    return dict(name=f"sa_mid_fwd 64->64 R={R} ({label}, z1 rebuilt)", kernel="sa_mid_fwd", run=run, bf16_products=0,
                flops=2.0 * 64 * 64 * R, bytes=4.0 * R * (4 + 64), keep=(rel4, W1, st, W, zout, part),
                what=f"{label}: z2 = relu(bn(W1 in)) W2^T + batch statistics, {R} rows; reads 16 B per row (the row's inputs), "
                     "writes z2: z1 (268 MB at SA1) is neither written nor read")


def sa_dgrad(R, ck, cp, pooled, S, dev, label):
    G = R // S
    dy = _rand(G if pooled else R, ck, dev=dev)
    arg = torch.randint(0, S, (G, ck), dtype=torch.uint8, device=dev) if pooled else None
    zk, zp = _rand(R, ck, dev=dev), _rand(R, cp, dev=dev)
    coef, stp, W = _stats(ck, dev), _stats(cp, dev), _rand(ck, cp, dev=dev) * 0.1
    dyp, part = torch.empty(R, cp, dtype=torch.float32, device=dev), _part(max(ck, cp), dev)

    def run():
        check(lib.spacap_sa_dgrad_f32(dy.data_ptr(), arg.data_ptr() if pooled else None, S if pooled else 0, zk.data_ptr(),
                                      coef.data_ptr(), W.data_ptr(), zp.data_ptr(), stp.data_ptr(), R, ck, cp, dyp.data_ptr(),
                                      part.data_ptr(), _st(dev)), "sa_dgrad")
    # reads: z_k (dense dz is rebuilt from it), z_prev (ReLU mask), dy (dense) or the pooled gradient; writes dy_prev
    byts = 4.0 * R * (ck + 2 * cp) + (4.0 * G * ck + G * ck if pooled else 4.0 * R * ck)
    import os
    split = (os.environ.get("SPACAP_SA_F32MFMA", "0") in ("", "0") and ck in (128, 256)
             and cp % 64 == 0 and R >= 49152)   # the library's default for these shapes (csrc/sa_bf3_dgrad.inc)
    return dict(name=f"sa_dgrad {ck}->{cp} R={R} {'pooled' if pooled else 'dense'} ({label})", kernel="sa_dgrad", run=run,
                bf16_products=6 if split else 0, flops=2.0 * ck * cp * R, bytes=byts, keep=(dy, arg, zk, zp, coef, stp, W, dyp, part),
                what=f"{label}: dy_prev = (dz W) * relu'(bn(z_prev)) + BN sums, {R} rows")


def sa_wgrad(R, ck, cp, pooled, S, dev, label):
    G = R // S
    dy = _rand(G if pooled else R, ck, dev=dev)
    arg = torch.randint(0, S, (G, ck), dtype=torch.uint8, device=dev) if pooled else None
    zk, zp = _rand(R, ck, dev=dev), _rand(R, cp, dev=dev)
    coef, stp = _stats(ck, dev), _stats(cp, dev)
    pw = torch.empty(int(lib.spacap_sa_wgrad_slabs(R, ck, cp, 1 if pooled else 0)), ck, cp, dtype=torch.float32, device=dev)

    def run():
        check(lib.spacap_sa_wgrad_f32(dy.data_ptr(), arg.data_ptr() if pooled else None, S if pooled else 0, zk.data_ptr(),
                                      coef.data_ptr(), zp.data_ptr(), stp.data_ptr(), R, ck, cp, pw.data_ptr(), _st(dev)), "sa_wgrad")
    byts = 4.0 * R * (ck + cp) + (4.0 * G * ck + G * ck if pooled else 4.0 * R * ck) + 4.0 * pw.numel()
    return dict(name=f"sa_wgrad {ck}x{cp} R={R} {'pooled' if pooled else 'dense'} ({label})", kernel="sa_wgrad_kernel", run=run,
                flops=2.0 * ck * cp * R, bytes=byts, keep=(dy, arg, zk, zp, coef, stp, pw),
                what=f"{label}: dW = dz^T relu(bn(z_prev)) per row slab, {R} rows")


def sa_wgrad_pool(R, c2, c3, S, dev, label):
    """The pooled layer's weight gradient from z2 alone (csrc/sa_l3bwd.inc: sa_wgrad_pool_kernel): sparse term + Gram matrix."""
    G = R // S
    dym, arg = _rand(G, c3, dev=dev), torch.randint(0, S, (G, c3), dtype=torch.uint8, device=dev)
    z2, coef, st2 = _rand(R, c2, dev=dev), _stats(c3, dev), _stats(c2, dev)
    npw, nfl = int(lib.spacap_sa_wgrad_pool_parts(R, c2, c3, S)), int(lib.spacap_sa_l3bwd_part_floats(c2, c3))
    pw = torch.empty(npw, nfl, dtype=torch.float32, device=dev)

    def run():
        check(lib.spacap_sa_wgrad_pool_f32(dym.data_ptr(), arg.data_ptr(), S, coef.data_ptr(), z2.data_ptr(), st2.data_ptr(), R, c3, c2,
                                           pw.data_ptr(), _st(dev)), "sa_wgrad_pool")
    # matrix work actually issued: the Gram matrix's upper class pairs (3/4 at C2 = 64, 5/8 at C2 = 128) + the sparse FMAs
    frac = 0.75 if c2 == 64 else 0.625
    return dict(name=f"sa_wgrad_pool {c2}->{c3} R={R} ({label})", kernel="sa_wgrad_pool_kernel", run=run,
                flops=2.0 * R * c2 * c2 * frac + 2.0 * G * c3 * c2, bytes=4.0 * (R * c2 + G * c3 * 1.25 + npw * nfl),
                keep=(dym, arg, z2, coef, st2, pw),
                what=f"{label}: dW3 partials = (g d)^T a2 (one row per group and channel) + Gram a2^T a2 + colsum a2 from ONE pass "
                     f"over z2 ({c2} floats per row; the dense kernel streams z3 and z2 = {c2 + c3} floats per row), {npw} workgroups")


def sa_mid_fwd_pool(R, cin, cout, S, dev, label, store):
    """The pooled last layer's forward with / without storing its output (the z3-free backward needs none of it)."""
    zin, st, W, g3 = _rand(R, cin, dev=dev), _stats(cin, dev), _rand(cout, cin, dev=dev) * 0.1, torch.ones(cout, dtype=torch.float32, device=dev)
    zout = torch.empty(R, cout, dtype=torch.float32, device=dev) if store else None
    nsub = R // min(S, 32)
    cv, ci = torch.empty(nsub, cout, 2, dtype=torch.float32, device=dev), torch.empty(nsub, cout, 2, dtype=torch.uint8, device=dev)
    part = _part(max(cin, cout), dev)

    def run():
        check(lib.spacap_sa_mid_fwd_pool_f32(zin.data_ptr(), st.data_ptr(), W.data_ptr(), g3.data_ptr(), R, cin, cout, S,
                                             zout.data_ptr() if store else None, part.data_ptr(), cv.data_ptr(), ci.data_ptr(), _st(dev)),
              "sa_mid_fwd_pool")
    return dict(name=f"sa_mid_fwd_pool {cin}->{cout} R={R} ({label}, {'z3 stored' if store else 'z3 not stored'})", kernel="sa_mid_fwd_bf3s",
                run=run, bf16_products=6, flops=2.0 * cin * cout * R, bytes=4.0 * R * (cin + (cout if store else 0)) + 10.0 * nsub * cout,
                keep=(zin, st, W, g3, zout, cv, ci, part),
                what=f"{label}: last layer + statistics + pooling candidates, output {'stored' if store else 'NOT stored'}")


def rel_fused(B, K, mode, dev):
    """The relation head as one launch each way (csrc/relation_fused.hip) on B * K * K proposal pairs."""
    H = 8
    P, U = torch.softmax(_rand(B, H, K, K, dev=dev), -1), _rand(B, K, H, 128, dev=dev) * 0.3
    b1, W2, b2 = _rand(128, dev=dev) * 0.1, _rand(128, 128, dev=dev) * 0.1, _rand(128, dev=dev) * 0.1
    W3, b3 = _rand(9, 128, dev=dev) * 0.1, _rand(9, dev=dev)
    R = B * K * K
    hid2, pred = torch.empty(R, 128, dtype=torch.float32, device=dev), torch.empty(R, 9, dtype=torch.float32, device=dev)
    check(lib.spacap_relation_fused_fwd_f32(P.data_ptr(), U.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3.data_ptr(),
                                            b3.data_ptr(), B, K, hid2.data_ptr(), pred.data_ptr(), _st(dev)), "rel_fused_fwd")
    if mode == 0:
        def run():
            check(lib.spacap_relation_fused_fwd_f32(P.data_ptr(), U.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                                                    W3.data_ptr(), b3.data_ptr(), B, K, hid2.data_ptr(), pred.data_ptr(), _st(dev)),
                  "rel_fused_fwd")
        return dict(name=f"rel_fused_fwd B={B} K={K} (relation head, {R} pairs)", kernel="rel_fused_fwd_kernel", run=run,
                    flops=2.0 * R * (8 * 128 + 128 * 128 + 128 * 9), flops_split=2.0 * R * 128 * 128,
                    bytes=4.0 * R * (8 + 128 + 9), bf16_products=6,
                    keep=(P, U, b1, W2, b2, W3, b3, hid2, pred),
                    what="pair feature + Linear-ReLU-Linear-ReLU-Linear in one launch: reads P (8 values per pair), writes hid2 and "
                         "pred; the 128 x 128 layer as split-bf16, the per-key and 9-wide products as fp32 MFMA")
    dpred = _rand(R, 9, dev=dev)
    nparts = int(lib.spacap_relation_fused_nparts(B, K))
    zs = int(lib.spacap_relation_fused_zsplit(B, K, nparts))
    dP, dU = torch.empty_like(P), torch.empty(zs, B, K, H, 128, dtype=torch.float32, device=dev)
    part = torch.empty(nparts, int(lib.spacap_relation_fused_part_floats()), dtype=torch.float32, device=dev)

    def run():
        check(lib.spacap_relation_fused_bwd_f32(dpred.data_ptr(), hid2.data_ptr(), P.data_ptr(), U.data_ptr(), b1.data_ptr(),
                                                W2.data_ptr(), W3.data_ptr(), B, K, nparts, zs, dP.data_ptr(), dU.data_ptr(),
                                                part.data_ptr(), _st(dev)), "rel_fused_bwd")
    return dict(name=f"rel_fused_bwd B={B} K={K} (relation head, {R} pairs)", kernel="rel_fused_bwd_kernel", run=run,
                flops=2.0 * R * (3 * 8 * 128 + 2 * 128 * 128 + 3 * 128 * 9), flops_split=2.0 * R * 2 * 128 * 128,
                bytes=4.0 * R * (128 + 9 + 8 + 8), bf16_products=6,
                keep=(P, U, b1, W2, W3, hid2, dpred, dP, dU, part),
                what="reads hid2, dpred and P, recomputes hid1, writes dP; dU and the parameter sums stay on chip until the end "
                     "(dhid1 and dW2 as split-bf16)")


def mha_fwd(B, h, L, dk, dev, need_p):
    hd = h * dk
    qkv = _rand(B, L, 3 * hd, dev=dev)
    mask = (torch.rand(B, 1, L, device=dev) > 0.3).to(torch.uint8)
    mask[..., 0] = 1
    out = torch.empty(B, L, h, dk, dtype=torch.float32, device=dev)
    p = torch.empty(B, h, L, L, dtype=torch.float32, device=dev) if need_p else None
    stats = torch.empty(B, h, L, 2, dtype=torch.float32, device=dev)
    sb, sh, sl = L * 3 * hd, dk, 3 * hd
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * hd, qkv.data_ptr() + 8 * hd

    def run():
        check(lib.spacap_mha_fwd_f32(q, k, v, sb, sh, sl, sb, sh, sl, sb, sh, sl, mask.data_ptr(), L, 0, None, 0, 0, 0, B, h, L, L,
                                     dk, 1.0 / dk ** 0.5, 0.1, 1234, None, out.data_ptr(), p.data_ptr() if need_p else None,
                                     stats.data_ptr(), _st(dev)), "mha_fwd")
    return dict(name=f"mha_fwd B={B} h={h} L={L} d_k={dk}{' +P' if need_p else ''}", kernel="mha_fwd", run=run,
                flops=4.0 * B * h * L * L * dk, bytes=4.0 * B * L * 4 * hd + (4.0 * B * h * L * L if need_p else 0.0),
                keep=(qkv, mask, out, p, stats), what="encoder self-attention layer (QK^T, mask, softmax, dropout, PV) in one launch")


def tf_ffn(R, dff, mode, dev):
    """The Transformer feed-forward block as one chained launch, encoder shape: the kernel the step runs for that many rows --
    split-bf16 products with pre-split operand-order weight images above 512 rows (csrc/tf_layer.hip: tf_ffn_bf3_kernel), the
    fp32-MFMA kernel below (tf_ffn_kernel)."""
    import ctypes
    x, W1, W2 = _rand(R, 128, dev=dev), _rand(dff, 128, dev=dev) * 0.1, _rand(128, dff, dev=dev) * 0.05
    b1, y = _rand(dff, dev=dev) * 0.1, _rand(R, dff, dev=dev).relu_()
    hid, part = torch.empty(R, dff, dtype=torch.float32, device=dev), torch.empty(dff // 128, R, 128, dtype=torch.float32, device=dev)
    bf3 = R > 512
    pieces = None
    if bf3:
        pieces = torch.empty(int(lib.spacap_tf_ffn_pieces_elems(dff)), dtype=torch.bfloat16, device=dev)
        arr = ctypes.c_void_p * 1
        check(lib.spacap_tf_ffn_split_f32(arr(W1.data_ptr()), arr(W2.data_ptr()), arr(pieces.data_ptr()), 1, dff, _st(dev)), "tf_ffn_split")

    def run():
        if bf3:
            check(lib.spacap_tf_ffn_bf3_f32(mode, x.data_ptr(), pieces.data_ptr(), b1.data_ptr() if mode == 0 else None,
                                            None if mode == 0 else y.data_ptr(), R, dff, 0.1, 7 if mode == 0 else 0, None,
                                            hid.data_ptr(), part.data_ptr(), _st(dev)), "tf_ffn_bf3")
        elif mode == 0:
            check(lib.spacap_tf_ffn_f32(0, x.data_ptr(), W1.data_ptr(), W2.data_ptr(), b1.data_ptr(), None, R, dff, 0.1, 7, None,
                                        hid.data_ptr(), part.data_ptr(), _st(dev)), "tf_ffn")
        else:
            check(lib.spacap_tf_ffn_f32(1, x.data_ptr(), W2.data_ptr(), W1.data_ptr(), None, y.data_ptr(), R, dff, 0.1, 0, None,
                                        hid.data_ptr(), part.data_ptr(), _st(dev)), "tf_ffn")
    byts = 4.0 * (R * 128 + R * dff * (1 if mode == 0 else 2) + (dff // 128) * R * 128 + 2 * 128 * dff)
    arith = "split-bf16 products" if bf3 else "fp32 MFMA"
    d = dict(name=f"tf_ffn {'fwd' if mode == 0 else 'bwd'} R={R} d_ff={dff} (encoder feed-forward block)",
             kernel="tf_ffn_bf3_kernel" if bf3 else "tf_ffn_kernel",
             run=run, flops=4.0 * R * 128 * dff, bytes=byts, keep=(x, W1, W2, b1, y, hid, part, pieces),
             what=f"hidden = dropout(relu(n W1^T + b1)) stored + partial sums of hidden W2^T per 128-wide slice of d_ff, one launch "
                  f"({arith})" if mode == 0 else
                  f"dhidden = (dy W2) * mask stored + partial sums of dhidden W1 per slice of d_ff, one launch ({arith})")
    if bf3:
        d["bf16_products"] = 6
    return d


def tf_rows(R, dff, dev):
    """The row-tile kernel between two attention calls (csrc/tf_layer.hip: tf_rows_kernel<FWD>) at its largest launch of a step:
    the feed-forward block's second half of an encoder layer -- add the d_ff / 128 partial products of w_2, dropout, residual,
    LayerNorm, then the NEXT layer's packed q|k|v projection (128 -> 384)."""
    import ctypes
    from spacap3d_amd._native import TfRowsArgs
    nparts = dff // 128
    part, res = _rand(nparts, R, 128, dev=dev), _rand(R, 128, dev=dev)
    ln_a, ln_b, W2, b2 = 1 + 0.1 * _rand(128, dev=dev), 0.1 * _rand(128, dev=dev), 0.1 * _rand(384, 128, dev=dev), 0.1 * _rand(384, dev=dev)
    x_out, n_out, stats = torch.empty(R, 128, device=dev), torch.empty(R, 128, device=dev), torch.empty(R, 2, device=dev)
    out2 = torch.empty(R, 384, device=dev)
    seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    a = TfRowsArgs()
    a.mode, a.R, a.k1, a.n2, a.nparts, a.lq, a.drop_p, a.eps, a.seed = 0, R, 0, 384, nparts, 0, 0.1, 1e-6, 11
    a.seed_dev = seed_dev.data_ptr()
    a.a1, a.res, a.x_out, a.ln_a, a.ln_b, a.n_out, a.stats = (part.data_ptr(), res.data_ptr(), x_out.data_ptr(), ln_a.data_ptr(),
                                                               ln_b.data_ptr(), n_out.data_ptr(), stats.data_ptr())
    a.w2, a.bias2, a.out2 = W2.data_ptr(), b2.data_ptr(), out2.data_ptr()

    def run():
        check(lib.spacap_tf_rows_f32(ctypes.byref(a), _st(dev)), "tf_rows")
    byts = 4.0 * (nparts * R * 128 + R * 128 * 3 + R * 2 + R * 384 + 384 * 128)
    return dict(name=f"tf_rows fwd R={R} (w_2 partials of d_ff={dff} -> residual -> LayerNorm -> q|k|v 128->384)", kernel="tf_rows_kernel",
                run=run, flops=2.0 * R * 128 * 384, bytes=byts, keep=(part, res, ln_a, ln_b, W2, b2, x_out, n_out, stats, out2, seed_dev, a),
                what=f"{R // 16} workgroups of 16 rows: {nparts} partial products added in order, dropout + residual + LayerNorm (two "
                     "barriers), then 16 x 384 outputs per workgroup as fp32 MFMA; 52 such launches per step (12 us average): a chain "
                     "of dependent phases on half the chip's CUs -- latency-bound, priced here against the HBM roof its bytes imply")


def fps(B, N, m, dev):
    from spacap3d_amd import synthetic as S
    xyz = S.scene_batch(B, N, use_height=False, seed=1000).to(dev)
    ws = torch.empty(max(int(lib.spacap_fps_workspace_bytes(B, N)), 16), dtype=torch.uint8, device=dev)
    idx = torch.empty(B, m, dtype=torch.int32, device=dev)

    def run():
        check(lib.spacap_fps_f32(xyz.data_ptr(), B, N, m, ws.data_ptr(), idx.data_ptr(), _st(dev)), "fps")
    return dict(name=f"fps B={B} {N}->{m} (SA1 sampling)", kernel="fps_bucket_kernel", run=run, flops=10.0 * B * (m - 1) * N,
                bytes=20.0 * B * (m - 1) * N, compulsory_bytes=float(B * (12 * N + 4 * m)), rounds=m - 1, keep=(xyz, ws, idx),
                what="furthest point sampling, one workgroup per scene, m-1 sequentially dependent rounds")


def sa_dgrad_wgrad_l1in(B, N, S, dev, label):
    """SA1 layer 2 backward as the step runs it since round 6: data gradient + BN sums + the first layer's three sums + the
    layer's own weight gradient from ONE read of dy2 / z2 (csrc/sa_mlp.hip, sa_dgrad_kernel<.., L1, WG>)."""
    R = B * N * S
    dy, zk, rel4 = _rand(R, 64, dev=dev), _rand(R, 64, dev=dev), _rand(R, 4, dev=dev)
    coef, stp, W2, W1 = _stats(64, dev), _stats(64, dev), _rand(64, 64, dev=dev) * 0.1, _rand(64, 4, dev=dev)
    nparts = int(lib.spacap_sa_nparts())
    part, pl1 = _part(64, dev), torch.empty(nparts, 64 * 8 + 4, dtype=torch.float32, device=dev)
    pw = torch.empty(int(lib.spacap_sa_dgrad_wgrad_l1in_slabs(R)), 64, 64, dtype=torch.float32, device=dev)

    def run():
        check(lib.spacap_sa_dgrad_wgrad_l1in_f32(dy.data_ptr(), zk.data_ptr(), coef.data_ptr(), W2.data_ptr(), rel4.data_ptr(),
                                                 W1.data_ptr(), 4, 1, stp.data_ptr(), B, N, S, part.data_ptr(), pl1.data_ptr(),
                                                 pw.data_ptr(), _st(dev)), "sa_dgrad_wgrad_l1in")
    return dict(name=f"sa_dgrad+wgrad 64->64 R={R} ({label}, z1 rebuilt)", kernel="sa_dgrad", run=run, bf16_products=0,
                flops=2.0 * 2 * 64 * 64 * R, bytes=4.0 * R * (64 + 64 + 4), keep=(dy, zk, rel4, coef, stp, W2, W1, part, pl1, pw),
                what=f"{label}: dy1 = (dz2 W2) relu'(bn(z1)) reduced to the first layer's three sums, and dW2 = dz2^T a1, {R} rows; "
                     "reads dy2, z2 and 16 B of inputs per row, writes partials only")


def linear_wgrad(R, ck, cp, dev, label):
    """torch.nn.Linear weight + bias gradient (csrc/wgrad_bf3.inc): split-bf16 products from transposing LDS reads."""
    g, x = _rand(R, ck, dev=dev), _rand(R, cp, dev=dev)
    ns = int(lib.spacap_linear_wgrad_slabs(R, ck, cp))
    part = torch.empty(ns, ck * cp + ck, dtype=torch.float32, device=dev)

    def run():
        check(lib.spacap_linear_wgrad_f32(g.data_ptr(), x.data_ptr(), R, ck, cp, 1, part.data_ptr(), _st(dev)), "linear_wgrad")
    import os
    split = os.environ.get("SPACAP_SA_F32MFMA", "0") in ("", "0")
    return dict(name=f"linear_wgrad {ck}x{cp} R={R} ({label})", kernel="linear_wgrad", run=run, bf16_products=6 if split else 0,
                flops=2.0 * ck * cp * R, bytes=4.0 * (R * (ck + cp) + part.numel()), keep=(g, x, part),
                what=f"{label}: dW = g^T x, db = colsum g per row slab ({ns} slabs), {R} rows")


def fps_small(B, N, m, dev, label):
    from spacap3d_amd import synthetic as S
    xyz = S.scene_batch(B, N, use_height=False, seed=1001).to(dev)[..., :3].contiguous()
    ws = torch.empty(max(int(lib.spacap_fps_workspace_bytes(B, N)), 16), dtype=torch.uint8, device=dev)
    idx = torch.empty(B, m, dtype=torch.int32, device=dev)

    def run():
        check(lib.spacap_fps_f32(xyz.data_ptr(), B, N, m, ws.data_ptr(), idx.data_ptr(), _st(dev)), "fps")
    return dict(name=f"fps B={B} {N}->{m} ({label})", kernel="fps_small_kernel", run=run, flops=10.0 * B * (m - 1) * N,
                bytes=20.0 * B * (m - 1) * N, compulsory_bytes=float(B * (12 * N + 4 * m)), rounds=m - 1, keep=(xyz, ws, idx),
                what="furthest point sampling of a sampled level: coordinates resident in LDS, one barrier per round")


def cases(dev, B=8):
    """name -> case dict, cfg2 shapes with B scenes."""
    R1, R2, R3, R4, RA = B * 2048 * 64, B * 1024 * 32, B * 512 * 16, B * 256 * 16, B * 256 * 16
    return [
        lambda: sa_mid_fwd(R2, 128, 256, dev, "SA2 layer 3"),
        lambda: sa_mid_fwd(R2, 128, 128, dev, "SA2 layer 2"),
        lambda: sa_mid_fwd(R1, 64, 64, dev, "SA1 layer 2"),
        lambda: sa_mid_fwd_l1in(R1, dev, "SA1 layer 2"),
        lambda: sa_mid_fwd(R1, 64, 128, dev, "SA1 layer 3"),
        lambda: sa_dgrad(R2, 256, 128, True, 32, dev, "SA2 layer 3"),
        lambda: sa_dgrad(R2, 128, 128, False, 32, dev, "SA2 layer 2"),
        lambda: sa_dgrad(R1, 128, 64, True, 64, dev, "SA1 layer 3"),
        lambda: sa_wgrad(R2, 256, 128, True, 32, dev, "SA2 layer 3"),
        lambda: sa_wgrad(R2, 128, 128, False, 32, dev, "SA2 layer 2"),
        lambda: sa_wgrad(R1, 128, 64, True, 64, dev, "SA1 layer 3"),
        lambda: sa_wgrad(R1, 64, 64, False, 64, dev, "SA1 layer 2"),
        lambda: sa_wgrad_pool(R1, 64, 128, 64, dev, "SA1 layer 3"),
        lambda: sa_wgrad_pool(R2, 128, 256, 32, dev, "SA2 layer 3"),
        lambda: sa_mid_fwd_pool(R1, 64, 128, 64, dev, "SA1 layer 3", True),
        lambda: sa_mid_fwd_pool(R1, 64, 128, 64, dev, "SA1 layer 3", False),
        lambda: rel_fused(B, 256, 0, dev),
        lambda: rel_fused(B, 256, 1, dev),
        lambda: mha_fwd(B, 8, 256, 16, dev, True),
        lambda: tf_ffn(B * 256, 2048, 0, dev),
        lambda: tf_ffn(B * 256, 2048, 1, dev),
        lambda: tf_rows(B * 256, 2048, dev),
        lambda: fps(B, 40000, 2048, dev),
        lambda: sa_dgrad_wgrad_l1in(B, 2048, 64, dev, "SA1 layer 2"),
        lambda: linear_wgrad(B * 256, 2048, 128, dev, "FFN first linear"),
        lambda: fps_small(B, 2048, 1024, dev, "SA2 sampling"),
    ]


def time_case(case, iters=20, warm=3):
    """Average duration (us) of `iters` back-to-back launches between two HIP events on the launch stream."""
    for _ in range(warm):
        case["run"]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        case["run"]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def roofline_entry(case, us, pmc=None):
    """The bench line's roofline object for one case.  The roof is the one that binds the IMPLEMENTED arithmetic:
      * fp32-MFMA kernels: matrix ceiling 157.3 TFLOP/s, ridge 157.3 / 8 = 19.7 flop/B;
      * split-bf16 kernels (6 bf16 products per fp32 product): matrix ceiling 2 500 / 6 = 417 TFLOP/s fp32-equivalent,
        ridge 417 / 8 = 52 flop/B -- the shared-MLP layers (AI = cin cout / (2 (cin + cout)) <= 43 flop/B) are HBM-bound;
    below the ridge the entry is priced in bytes / s against 8 TB/s, above it in flop/s against the matrix ceiling.  The
    fp32-equivalent flop rate stays in the entry as a secondary figure.  FPS is a latency chain (on-chip resident)."""
    split = bool(case.get("bf16_products"))
    ceiling = PEAK_MFMA_BF16_TFLOPS / case["bf16_products"] if split else PEAK_MFMA_F32_TFLOPS
    if split and case.get("flops_split") is not None:
        # mixed kernels (relation head): `flops_split` of the flops as split-bf16, the rest as fp32 MFMA -- the ceiling is
        # total flops / (time of each part at its own peak)
        fs = case["flops_split"]
        ceiling = case["flops"] / (fs / ceiling + (case["flops"] - fs) / PEAK_MFMA_F32_TFLOPS)
    if "rounds" in case:
        ent = dict(bound="latency", kernel=case["name"], achieved=us / case["rounds"], peak=None, unit="us/round", frac=None,
                   streamed_model_GBs=case["bytes"] / us * 1e-3, streamed_model_frac_of_hbm=case["bytes"] / us * 1e-3 / PEAK_HBM_GBS)
    elif case["flops"] / case["bytes"] > ceiling * 1e3 / PEAK_HBM_GBS:
        a = case["flops"] / us * 1e-6
        ent = dict(bound="mfma", kernel=case["name"], achieved=a, peak=ceiling, unit="TFLOP/s", frac=a / ceiling)
    else:
        a = case["bytes"] / us * 1e-3
        ent = dict(bound="hbm", kernel=case["name"], achieved=a, peak=PEAK_HBM_GBS, unit="GB/s", frac=a / PEAK_HBM_GBS)
    if "rounds" not in case:
        ent["arithmetic_intensity_flop_per_byte"] = case["flops"] / case["bytes"]
        ent["ridge_flop_per_byte"] = ceiling * 1e3 / PEAK_HBM_GBS
        ent["fp32_equivalent_TFLOPs"] = case["flops"] / us * 1e-6
        ent["frac_of_fp32_mfma_peak"] = ent["fp32_equivalent_TFLOPs"] / PEAK_MFMA_F32_TFLOPS
    ent.update(launch_us=us, algorithmic_flops=case["flops"], algorithmic_bytes=case["bytes"], what=case["what"])
    if case.get("bf16_products"):
        # fp32 results from bf16 matrix instructions: every fp32 product is evaluated as 6 exact bf16 x bf16 products with
        # fp32 accumulation (csrc/sa_bf3.inc).  `achieved` stays the ALGORITHMIC fp32 flops of the layer / time and `peak`
        # the fp32-MFMA peak it replaces; the matrix pipe itself executes 6x those flops in bf16:
        ent["implementation"] = "split-bf16: 6 bf16 MFMA products per fp32 product, fp32 accumulate (fp32-equivalent result)"
        ent["mfma_bf16_TFLOPs"] = case["bf16_products"] * case.get("flops_split", case["flops"]) / us * 1e-6
        ent["mfma_bf16_frac_of_peak"] = ent["mfma_bf16_TFLOPs"] / PEAK_MFMA_BF16_TFLOPS
    rec = (pmc or {}).get(case["name"])
    ent["traffic"] = rec.get("hbm_bytes") if rec else None
    if rec:
        for k in ("fetch_bytes_corrected", "write_bytes", "mfma_util", "clock_GHz", "mfma_flops_counted", "mfma_bf16_flops_counted", "profiled_us", "source"):
            if k in rec:
                ent[k] = rec[k]
    return ent
