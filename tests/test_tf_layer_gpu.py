"""Fused Transformer sub-layer kernels (csrc/tf_layer.hip, spacap3d_amd/tf_layer.py) against float64 PyTorch
restatements of the reference modules (models/transformer_captioner.py: LayerNorm :102-113, SublayerConnection :115-127,
PositionwiseFeedForward :72-81, EncoderLayer :180-191, DecoderLayer :209-225) and against this repository's own
per-operator path.  fp32 MFMA arithmetic: tolerances are fp32 rounding (1e-5 of the tensor's scale).  Dropout cannot match
another RNG stream: parity runs with p = 0, dropout through its invariants (keep rate, forward / backward mask agreement
by a directional derivative under replayed seeds)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def tf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spacap3d_amd import tf_layer
    return tf_layer


def ln64(x, a, b, eps=1e-6):
    mu = x.mean(-1, keepdim=True)
    sd = x.std(-1, keepdim=True)
    return a * (x - mu) / (sd + eps) + b


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _rand(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("R", [16, 256, 250, 2048])
def test_ln_qkv_forward_backward(tf, R):
    x = _rand(R, 128, seed=1).requires_grad_()
    a, b = (_rand(128, seed=2) * 0.3 + 1).requires_grad_(), _rand(128, seed=3).requires_grad_()
    pw, pb = _rand(384, 128, seed=4, scale=0.1).requires_grad_(), _rand(384, seed=5).requires_grad_()
    qkv, xres = tf.LnQkv.apply(x, a, b, 1e-6, pw, pb)
    gq, gr = _rand(R, 384, seed=6), _rand(R, 128, seed=7)
    (qkv * gq).sum().add((xres * gr).sum()).backward()
    x6, a6, b6, w6, pb6 = (t.detach().double().requires_grad_() for t in (x, a, b, pw, pb))
    q6 = ln64(x6, a6, b6) @ w6.t() + pb6
    (q6 * gq.double()).sum().add((x6 * gr.double()).sum()).backward()
    assert rel(qkv, q6) < 2e-6 and torch.equal(xres, x)
    for got, want, name in ((x.grad, x6.grad, "dx"), (a.grad, a6.grad, "da"), (b.grad, b6.grad, "db"),
                            (pw.grad, w6.grad, "dW"), (pb.grad, pb6.grad, "dbias")):
        assert rel(got, want) < 1e-5, name


@pytest.mark.parametrize("R,dff", [(256, 2048), (2048, 2048), (40, 256)])
def test_attn_out_ffn1_and_ffn2_chain(tf, R, dff):
    """AttnOutFfn1 -> Ffn2Ln (with and without the next projection) forward values and every gradient."""
    t = lambda *s, seed, scale=1.0: _rand(*s, seed=seed, scale=scale).requires_grad_()
    a_, x_ = t(R, 128, seed=1), t(R, 128, seed=2)
    Wo, bo = t(128, 128, seed=3, scale=0.1), t(128, seed=4)
    l2a, l2b = ((_rand(128, seed=5) * 0.3 + 1).requires_grad_(), t(128, seed=6))
    W1, b1 = t(dff, 128, seed=7, scale=0.1), t(dff, seed=8, scale=0.1)
    W2, b2 = t(128, dff, seed=9, scale=0.05), t(128, seed=10)
    l3a, l3b = ((_rand(128, seed=11) * 0.3 + 1).requires_grad_(), t(128, seed=12))
    pw, pb = t(384, 128, seed=13, scale=0.1), t(384, seed=14)
    leaves = [a_, x_, Wo, bo, l2a, l2b, W1, b1, W2, b2, l3a, l3b, pw, pb]
    g1, g2, g3 = _rand(R, 128, seed=20), _rand(R, 384, seed=21), _rand(R, 128, seed=22)
    for last in (False, True):
        for p in leaves:
            p.grad = None
        x1, h, parts = tf.AttnOutFfn1.apply(a_, x_, Wo, bo, l2a, l2b, W1, b1, W2, 1e-6, 0.0, 0.0, 0, 0)
        if last:
            (mem,) = tf.Ffn2Ln.apply(h, parts, x1, W1, W2, b2, l3a, l3b, 1e-6, 0.0, 0.0, 0, None, None)
            (mem * g3).sum().backward()
        else:
            x2, qkv = tf.Ffn2Ln.apply(h, parts, x1, W1, W2, b2, l3a, l3b, 1e-6, 0.0, 0.0, 0, pw, pb)
            ((x2 * g1).sum() + (qkv * g2).sum()).backward()
        L = [p.detach().double().requires_grad_() for p in leaves]
        a6, x6, Wo6, bo6, l2a6, l2b6, W16, b16, W26, b26, l3a6, l3b6, pw6, pb6 = L
        x16 = x6 + a6 @ Wo6.t() + bo6
        h6 = torch.relu(ln64(x16, l2a6, l2b6) @ W16.t() + b16)
        x26 = x16 + h6 @ W26.t() + b26
        n6 = ln64(x26, l3a6, l3b6)
        if last:
            (n6 * g3.double()).sum().backward()
            assert rel(mem, n6) < 3e-6
        else:
            q6 = n6 @ pw6.t() + pb6
            ((x26 * g1.double()).sum() + (q6 * g2.double()).sum()).backward()
            assert rel(x2, x26) < 3e-6 and rel(qkv, q6) < 3e-6
        assert rel(x1, x16) < 2e-6 and rel(h, h6) < 3e-6
        n = len(leaves) - (2 if last else 0)
        names = "a x Wo bo l2a l2b W1 b1 W2 b2 l3a l3b pw pb".split()
        for got, want, name in zip(leaves[:n], L[:n], names):
            assert rel(got.grad, want.grad) < 2e-5, (name, last)


def _stack(kind, N=2, dff=512, p=0.0, seed=0):
    from spacap3d_amd import transformer_captioner as T
    torch.manual_seed(seed)
    attn = T.MultiHeadedAttention(8, 128, dropout=p)
    ff = T.PositionwiseFeedForward(128, dff, p)
    if kind == "enc":
        m = T.Encoder(T.EncoderLayer(128, copy.deepcopy(attn), copy.deepcopy(ff), p), N)
    else:
        m = T.Decoder(T.DecoderLayer(128, copy.deepcopy(attn), copy.deepcopy(attn), copy.deepcopy(ff), p, True), N)
    for q in m.parameters():
        if q.dim() > 1:
            torch.nn.init.xavier_uniform_(q)
        else:
            torch.nn.init.normal_(q, 0.0 if q.abs().max() == 0 else 1.0, 0.1)
    return m.to(DEV)


def _run(m, kind, x, mask):
    return m(x, mask) if kind == "enc" else m(x, None, None, mask)


@pytest.mark.parametrize("kind,B,L", [("enc", 2, 256), ("dec", 3, 32), ("enc", 1, 64), ("enc", 8, 256)])   # (8 x 256 rows: split-bf16 feed-forward)
def test_stack_matches_the_per_operator_path(tf, kind, B, L):
    m = _stack(kind).train()
    x = _rand(B, L, 128, seed=5)
    if kind == "enc":
        mask = (torch.rand(B, 1, L, generator=torch.Generator().manual_seed(1)) > 0.3).long().to(DEV)
        mask[..., 0] = 1
    else:
        mask = ((torch.rand(B, 1, L, generator=torch.Generator().manual_seed(1)) > 0.2)
                & torch.ones(1, L, L, dtype=torch.bool).tril()).to(DEV)
    g = _rand(B, L, 128, seed=6)
    res = {}
    for fused in (True, False):
        tf.ENABLED = fused
        try:
            for q in m.parameters():
                q.grad = None
            xi = x.clone().requires_grad_()
            out = _run(m, kind, xi, mask)
            (out * g).sum().backward()
            res[fused] = (out.detach(), xi.grad, [q.grad.clone() if q.grad is not None else None for q in m.parameters()])
        finally:
            tf.ENABLED = True
    assert rel(res[True][0], res[False][0]) < 1e-5
    assert rel(res[True][1], res[False][1]) < 5e-5
    gmax = max(float(b.abs().max()) for b in res[False][2] if b is not None)
    for (name, _), a, b in zip(m.named_parameters(), res[True][2], res[False][2]):
        assert (a is None) == (b is None), name
        if a is None:
            continue   # (the decoder's unused cross-attention in early-guide mode)
        # (the key bias has an analytically zero gradient -- softmax is invariant to it -- so only rounding noise is left:
        # errors are measured against the larger of the tensor's own scale and 1e-3 of the largest gradient)
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-3 * gmax)
        assert err < 1e-4, (name, err)


@pytest.mark.parametrize("R,K,N,trans,S", [(256, 2048, 128, True, 16), (2048, 2048, 128, False, 8), (100, 256, 128, True, 2),
                                           (70, 384, 256, False, 1), (64, 128, 128, True, 1)])
def test_split_product_and_masked_gradient(tf, R, K, N, trans, S):
    from spacap3d_amd._native import check, lib
    a = _rand(R, K, seed=1)
    W = _rand(N, K, seed=2, scale=0.1) if trans else _rand(K, N, seed=2, scale=0.1)
    out = torch.empty(S, R, N, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    check(lib.spacap_tf_gemm_f32(a.data_ptr(), W.data_ptr(), R, K, N, 1 if trans else 0, S, out.data_ptr(), st), "gemm")
    want = a.double() @ (W.double().t() if trans else W.double())
    assert rel(out.double().sum(0), want) < 2e-6
    KS = K // S
    for sl in range(S):   # every slice holds exactly its part of k
        ws = W.double()[:, sl * KS:(sl + 1) * KS].t() if trans else W.double()[sl * KS:(sl + 1) * KS]
        assert rel(out[sl], a.double()[:, sl * KS:(sl + 1) * KS] @ ws) < 2e-6
    assert 1 <= lib.spacap_tf_gemm_splits(R, K, N) <= K // 128 and (K // 128) % lib.spacap_tf_gemm_splits(R, K, N) == 0
    if K == 128 or not trans:
        g = _rand(R, 128, seed=3)
        Wm = _rand(128, 512, seed=4, scale=0.1)
        y = torch.relu(_rand(R, 512, seed=5))
        dx = torch.empty(R, 512, device=DEV)
        check(lib.spacap_tf_dgrad_mask_f32(g.data_ptr(), Wm.data_ptr(), y.data_ptr(), 1.25, R, 128, 512, dx.data_ptr(), st), "mask")
        assert rel(dx, (g.double() @ Wm.double()) * 1.25 * (y > 0)) < 2e-6


def test_fused_stack_launch_count(tf):
    """4 kernels per layer forward (+1 for the first norm / projection): the point of the fusion."""
    m = _stack("enc", N=3).eval()
    x = _rand(2, 256, 128, seed=5)
    with torch.no_grad():
        m(x, None)
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            m(x, None)
            torch.cuda.synchronize()
    ev = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
    if not ev:
        pytest.skip("the profiler reported no device events")
    ours = sum(e.count for e in ev if "tf_" in e.key or "mha_" in e.key)
    other = sum(e.count for e in ev) - ours
    # (outside a Trainer the three projection weights are not adjacent in memory: two concatenations per layer)
    assert ours == 1 + 4 * 3 and other <= 2 * 3, [(e.key[:60], e.count) for e in ev]


@pytest.mark.parametrize("kind", ["enc", "dec"])
def test_dropout_masks_agree_between_forward_and_backward(tf, kind):
    """Directional derivative under REPLAYED seeds: the same masks are drawn when the host call counter is rewound and the
    device step word is left alone; a backward that regenerated a different mask is off by O(1)."""
    from spacap3d_amd import attention as att
    p = 0.25
    m = _stack(kind, N=2, dff=256, p=p).train()
    for mod in m.modules():
        if hasattr(mod, "keep_value"):
            mod.dropout.p = 0.0      # attention-probability dropout has its own test (test_attention_gpu.py)
    B, L = 2, 32
    x = _rand(B, L, 128, seed=5)
    g = _rand(B, L, 128, seed=6)
    d = _rand(B, L, 128, seed=7)
    att._next_seed()
    saved, word = att._CALL_COUNTER[0], att.rng_state(DEV).clone()
    c0 = 20261003 << 20    # fixed masks: a finite difference through ReLU / dropout kinks must not depend on what ran before
    att.rng_state(DEV).zero_()   # (nor on how many training steps earlier tests took: the device step word enters the hash)

    def f(xx):
        att._CALL_COUNTER[0] = c0
        return (_run(m, kind, xx, None) * g).sum()

    xi = x.clone().requires_grad_()
    f(xi).backward()
    want = float((xi.grad * d).sum())
    eps = 1e-2
    with torch.no_grad():
        num = float((f(x + eps * d).double() - f(x - eps * d).double()) / (2 * eps))
    att._CALL_COUNTER[0] = saved
    att.rng_state(DEV).copy_(word)
    assert abs(num - want) <= 3e-2 * max(abs(want), 1.0), (num, want)
    # keep rate / scaling of the hidden layer's dropout
    with torch.no_grad():
        n2 = _rand(512, 128, seed=9)
        W1, b1 = _rand(256, 128, seed=10, scale=0.1), _rand(256, seed=11)
        from spacap3d_amd._native import check, lib
        h0, h1 = torch.empty(512, 256, device=DEV), torch.empty(512, 256, device=DEV)
        st = torch.cuda.current_stream().cuda_stream
        check(lib.spacap_tf_ffn1_f32(n2.data_ptr(), W1.data_ptr(), b1.data_ptr(), 512, 256, 0.0, 0, None, h0.data_ptr(), st), "ffn1")
        check(lib.spacap_tf_ffn1_f32(n2.data_ptr(), W1.data_ptr(), b1.data_ptr(), 512, 256, p, 1234, None, h1.data_ptr(), st), "ffn1")
        act = h0 > 0
        kept = (h1 > 0) & act
        rate = float(kept.sum()) / float(act.sum())
        assert abs(rate - (1 - p)) < 0.02, rate
        assert rel(h1[kept], h0[kept] / (1 - p)) < 1e-6
        assert rel(h0, torch.relu(n2.double() @ W1.double().t() + b1.double())) < 2e-6


def test_decode_attention_step_over_the_cache(tf):
    """spacap_decode_attn_f32: appending the new token's k, v and attending over positions 0..t equals the last row of causal
    attention over the whole prefix (what the reference recomputes at every word, models/transformer_captioner.py:435-438)."""
    from spacap3d_amd._native import check, lib
    R, T, h, dk = 37, 32, 8, 16
    kc, vc = torch.zeros(R, T, 128, device=DEV), torch.zeros(R, T, 128, device=DEV)
    out = torch.empty(R, 128, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for t in range(T):
        qkv = _rand(R, 384, seed=100 + t)
        rows.append(qkv)
        check(lib.spacap_decode_attn_f32(qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(), R, h, dk, T, t, 0.25, out.data_ptr(), st), "dec")
        allr = torch.stack(rows, 1).double()                                   # (R, t+1, 384)
        q = allr[:, -1, :128].view(R, h, 1, dk)
        k = allr[:, :, 128:256].view(R, t + 1, h, dk).transpose(1, 2)
        v = allr[:, :, 256:].view(R, t + 1, h, dk).transpose(1, 2)
        p = torch.softmax(q @ k.transpose(-1, -2) * 0.25, -1)
        want = (p @ v).transpose(1, 2).reshape(R, 128)
        assert rel(out, want) < 3e-6, t
        assert torch.equal(kc[:, t], qkv[:, 128:256]) and torch.equal(vc[:, t], qkv[:, 256:])


@pytest.mark.parametrize("B,L,mask_kind,p", [(2, 256, "key", 0.0), (3, 32, "causal", 0.0), (2, 256, "key", 0.2), (1, 96, None, 0.0)])
def test_single_launch_attention_backward_equals_the_two_launch_form(tf, B, L, mask_kind, p):
    """spacap_mha_bwd_delta_f32 (delta = sum_d out d_out handed in, dQ and dK/dV halves in one launch) against
    spacap_mha_bwd_f32 (two launches, delta formed by the first): same kernels' arithmetic, so bitwise equal gradients; the
    handed-in delta itself equals the first form's up to fp32 summation order."""
    import math
    from spacap3d_amd._native import check, lib
    h, dk, hd = 8, 16, 128
    qkv = _rand(B, L, 3 * hd, seed=1)
    dout = _rand(B, L, hd, seed=2)
    if mask_kind == "key":
        mask = (torch.rand(B, 1, L, generator=torch.Generator().manual_seed(3)) > 0.3).to(torch.uint8).to(DEV)
        mask[..., 0] = 1
        msb, msq = L, 0
    elif mask_kind == "causal":
        mask = torch.ones(B, L, L, dtype=torch.uint8).tril().to(DEV).contiguous()
        msb, msq = L * L, L
    else:
        mask, msb, msq = None, 0, 0
    st = torch.cuda.current_stream().cuda_stream
    strides = (L * 3 * hd, dk, 3 * hd)
    base = qkv.data_ptr()
    mp = mask.data_ptr() if mask is not None else None
    rng = torch.zeros(1, dtype=torch.int64, device=DEV)
    rp = rng.data_ptr() if p > 0 else None
    out, lse = torch.empty(B, L, hd, device=DEV), torch.empty(B, h, L, 2, device=DEV)
    args = (base, base + hd * 4, base + 2 * hd * 4, *strides, *strides, *strides, mp, msb, msq, None, 0, 0, 0, B, h, L, L, dk,
            1.0 / math.sqrt(dk), p, 77, rp)
    check(lib.spacap_mha_fwd_f32(*args, out.data_ptr(), None, lse.data_ptr(), st), "fwd")
    g1, g2 = torch.empty_like(qkv), torch.empty_like(qkv)
    ws = torch.empty(B * h * L, device=DEV)
    check(lib.spacap_mha_bwd_f32(*args, lse.data_ptr(), dout.data_ptr(), None, ws.data_ptr(), g1.data_ptr(), g1.data_ptr() + hd * 4,
                                 g1.data_ptr() + 2 * hd * 4, 3 * hd, st), "bwd")
    delta = (out * dout).view(B, L, h, dk).sum(-1).permute(0, 2, 1).contiguous()
    assert rel(delta.view(-1), ws) < 2e-5
    check(lib.spacap_mha_bwd_delta_f32(*args, lse.data_ptr(), dout.data_ptr(), ws.data_ptr(), g2.data_ptr(), g2.data_ptr() + hd * 4,
                                       g2.data_ptr() + 2 * hd * 4, 3 * hd, st), "bwd delta")
    assert torch.equal(g1, g2)
    check(lib.spacap_mha_bwd_delta_f32(*args, lse.data_ptr(), dout.data_ptr(), delta.data_ptr(), g2.data_ptr(),
                                       g2.data_ptr() + hd * 4, g2.data_ptr() + 2 * hd * 4, 3 * hd, st), "bwd delta")
    assert rel(g2, g1) < 2e-5


@pytest.mark.parametrize("h", [4, 2])
def test_eval_with_other_head_counts_decodes_through_the_cached_operator_path(tf, h):
    """The fused decode step needs h = 8, d_k = 16 (spacap_decode_attn_f32); a model built with another head count (the
    constructor accepts any h dividing 128, models/transformer_captioner.py:268-287; the attention kernels take d_k = 16, 32, 64) must still decode -- through the cached
    per-operator path -- and give the captions of the reference-style loop that recomputes the prefix."""
    from spacap3d_amd import synthetic as S
    from spacap3d_amd.engine import synthetic_batch
    from spacap3d_amd.spacapnet import build_default
    torch.manual_seed(0)
    model = build_default(vocab_size=120, num_proposal=32, N=2, h=h, d_ff=256).to(DEV).eval()
    dec = model.caption.model.decoder
    x = torch.zeros(4, 128, device=DEV)
    assert tf.stack_supported(dec.layers, x) and not tf.decode_supported(dec.layers, x, 31)
    data = synthetic_batch(2, 4096, DEV, seed=1, vocab=120)
    with torch.no_grad():
        d = model(dict(data), is_eval=True)
        d2 = model.caption.forward_eval(dict(d), use_cache=False)
    assert d["lang_cap"].shape == d2["lang_cap"].shape == (2, 32, 31)
    assert (d["lang_cap"] == d2["lang_cap"]).float().mean() > 0.995


@pytest.mark.parametrize("R,V", [(2048, 3001), (37, 40), (16, 64), (300, 1000)])
def test_decode_word_choice_without_logits(R, V):
    """spacap_decode_word_f32 (csrc/tf_layer.hip: vocab_argmax_kernel + decode_next_kernel): the greedy word of every sequence
    = arg-max of x W^T + b (models/transformer_captioner.py:441-447 on the Generator of :93-100; first maximum on ties, as
    torch.max), written into the caption, and the next input row lut[word] sqrt(d) + pe.  Against float64 logits: the chosen
    word's logit is within fp32 rounding of the maximum (an exact tie in float64 picks the smaller index)."""
    import math
    from spacap3d_amd._native import check, lib
    g = torch.Generator().manual_seed(R + V)
    x = torch.randn(R, 128, generator=g).to(DEV)
    W, b = (0.3 * torch.randn(V, 128, generator=g)).to(DEV), torch.randn(V, generator=g).to(DEV)
    W[7] = W[3]
    b[7] = b[3]                                      # two identical words: the first one must win wherever they lead
    x[0] = 0.0
    b[3] = b[7] = 50.0                               # ... which they do for row 0 (all-zero input: logits = bias)
    lut, pe = torch.randn(V, 128, generator=g).to(DEV), torch.randn(128, generator=g).to(DEV)
    ys = torch.full((R, 5), -1, dtype=torch.long, device=DEV)
    xn = torch.empty(R, 128, device=DEV)
    ws = torch.empty(int(lib.spacap_decode_word_workspace_bytes(R, V)), dtype=torch.uint8, device=DEV)
    scale = math.sqrt(128.0)
    from spacap3d_amd.linear import bf3_pieces
    Wp = bf3_pieces(W)
    check(lib.spacap_decode_word_f32(x.data_ptr(), Wp.data_ptr(), b.data_ptr(), R, V, lut.data_ptr(), scale, pe.data_ptr(), ys.data_ptr(), 5, 2,
                                     xn.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "spacap_decode_word_f32")
    word = ys[:, 2]
    assert int(word[0]) == 3 and bool((ys[:, [0, 1, 3, 4]] == -1).all())
    logits = x.double() @ W.double().t() + b.double()
    best = logits.max(1).values
    chosen = logits.gather(1, word.view(-1, 1)).squeeze(1)
    assert float((best - chosen).max()) < 1e-4 * float(logits.abs().max())
    assert float((word == logits.argmax(1)).double().mean()) > 0.995
    assert torch.allclose(xn, lut[word] * scale + pe, rtol=0, atol=1e-5)


@pytest.mark.parametrize("R,dff,p", [(2048, 2048, 0.0), (1000, 256, 0.0), (2048, 2048, 0.1)])
def test_feed_forward_block_on_split_bf16_products(tf, R, dff, p):
    """csrc/tf_layer.hip: tf_ffn_bf3_kernel (both chained products of models/transformer_captioner.py:72-81 as bf16 x 3, weights from
    the pre-split piece images of spacap_tf_ffn_split_f32) against the fp32-MFMA kernel it replaces for tall inputs: hidden
    layer, the partial sums of the second product, and the same for the backward direction -- fp32-equivalent (1e-5 of scale,
    identical ReLU / dropout masks up to units within rounding of zero), and against float64 for the forward."""
    import types
    W1, b1 = _rand(dff, 128, seed=7, scale=0.1), _rand(dff, seed=8, scale=0.1)
    W2 = _rand(128, dff, seed=9, scale=0.05)
    x, dy = _rand(R, 128, seed=1), _rand(R, 128, seed=2)
    layer = types.SimpleNamespace(feed_forward=types.SimpleNamespace(w_1=types.SimpleNamespace(weight=W1), w_2=types.SimpleNamespace(weight=W2)))
    outs = {}
    for bf3 in (False, True):
        tf._FFN_PIECES.clear()
        tf.FFN_BF3 = bf3
        try:
            if bf3:
                tf.refresh_ffn_pieces([layer])
                assert W1.data_ptr() in tf._FFN_PIECES
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                h, parts = tf._ffn(0, x, W1, W2, b1, None, dff, p, 12345, DEV)
                dh, gparts = tf._ffn(1, dy, W2, W1, None, h, dff, p, 0, DEV)
                torch.cuda.synchronize()
            names = [e.key for e in prof.key_averages()]
            if names:
                assert any("tf_ffn_bf3_kernel" in n for n in names) == bf3, names
        finally:
            tf.FFN_BF3 = True
            tf._FFN_PIECES.clear()
        outs[bf3] = (h, parts.sum(0), dh, gparts.sum(0))
    for name, a, b in zip(("hid", "w2 product", "dhid", "w1 product"), outs[True], outs[False]):
        # a hidden unit within fp32 rounding of zero may be gated differently by the two arithmetics: compare where both agree
        if name in ("hid", "dhid"):
            same = (outs[True][0] > 0) == (outs[False][0] > 0)
            assert float(same.float().mean()) > 0.9999
            a, b = a * same, b * same
            assert rel(a, b) < 1e-5, (name, rel(a, b))
        else:
            assert rel(a, b) < 2e-4, (name, rel(a, b))    # (sums over d_ff that include the few differently gated units)
    if p == 0.0:
        h64 = torch.relu(x.double() @ W1.double().t() + b1.double())
        assert rel(outs[True][0], h64) < 3e-6 and rel(outs[True][1], h64 @ W2.double().t()) < 3e-6
