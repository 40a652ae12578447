"""csrc/gemm_bf3.hip against float64: the tiled split-bf16 products that take the 512-wide relation head of the stress
configuration (BASELINE.json config 5; models/transformer_captioner.py:319-326, 392-398 at d_model = 512, h = 32) and the
Transformer's Linear layers at that width off the BLAS library.  Split-bf16 is fp32-equivalent: the bar is the one of the
fp32-MFMA kernels, 1e-5 of the result's scale (a single-bf16 product would sit at 4e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def lin():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spacap3d_amd import linear
    return linear


@pytest.mark.parametrize("R,K,N,trans,relu", [(4096, 512, 512, False, True), (1000, 128, 256, False, False), (131, 256, 128, True, False),
                                              (8192, 2048, 512, True, False), (20000, 512, 2048, False, True), (1, 128, 128, False, False)])
def test_tiled_split_bf16_product(lin, R, K, N, trans, relu):
    g = torch.Generator().manual_seed(R + K + N)
    a = torch.randn(R, K, generator=g).to(DEV)
    W = (torch.randn(K, N, generator=g) if trans else torch.randn(N, K, generator=g)).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV) if relu else None
    out = lin.bf3_product(a, lin.bf3_pieces(W, trans=trans), bias, relu=relu)
    ref = a.double() @ (W.double() if trans else W.double().t())
    if bias is not None:
        ref = ref + bias.double()
    if relu:
        ref = ref.clamp_min(0)
    assert out.shape == ref.shape and _rel(out, ref) < 1e-5, _rel(out, ref)
    # the pieces reproduce the weight to 2^-24
    Wp = lin.bf3_pieces(W, trans=trans).double().sum(0)
    assert _rel(Wp, W.double().t() if trans else W.double()) < 2e-7


@pytest.mark.parametrize("R,N,K", [(65536, 512, 512), (5000, 128, 256), (33, 256, 128)])
def test_tiled_split_bf16_weight_gradient(R, N, K):
    from spacap3d_amd._native import check, lib
    g = torch.Generator().manual_seed(R + N)
    G, X = torch.randn(R, N, generator=g).to(DEV), (torch.randn(R, K, generator=g) + 0.2).to(DEV)
    for ns in {int(lib.spacap_gemm_bf3_wgrad_slabs(R, N, K)), 1, 5}:
        part = torch.full((ns, N * K), float("nan"), device=DEV)
        check(lib.spacap_gemm_bf3_wgrad_f32(G.data_ptr(), N, X.data_ptr(), K, R, N, K, ns, part.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "spacap_gemm_bf3_wgrad_f32")
        ref = G.double().t() @ X.double()
        assert _rel(part.double().sum(0).view(N, K), ref) < (1e-5 if ns > 1 or R < 10000 else 1e-4), ns


@pytest.mark.parametrize("C,H,seed", [(512, 32, 0), (512, 32, 1), (512, 32, 2), (256, 16, 0), (128, 8, 0)])
def test_wide_relation_head_matches_float64(lin, C, H, seed):
    """RelationWide (hid1 kernel -> split-bf16 layer 2 -> 9-wide layer 3; backward: tail pass, split-bf16 weight / data gradients,
    first-layer backward) against the float64 composition of models/transformer_captioner.py:392-397 + :319-326."""
    B, K, D = 2, 64, 16
    torch.manual_seed(seed)              # (the Linear layers draw their weights from the global generator)
    g = torch.Generator().manual_seed(C + seed)
    P = torch.softmax(torch.randn(B, H, K, K, generator=g), -1).to(DEV).requires_grad_(True)
    V = torch.randn(B, H, K, D, generator=g).to(DEV).requires_grad_(True)
    l1, l2, l3 = torch.nn.Linear(H * D, C).to(DEV), torch.nn.Linear(C, C).to(DEV), torch.nn.Linear(C, 9).to(DEV)
    assert H * D == C
    pred = lin.relation_head_wide(P, V, l1, l2, l3)
    assert pred is not None and pred.shape == (B, K, K, 9)
    wsum = torch.randn(pred.shape, generator=g).to(DEV)
    (pred * wsum).sum().backward()
    got = [pred, P.grad, V.grad] + [p.grad for m in (l1, l2, l3) for p in (m.weight, m.bias)]
    P64, V64 = P.detach().double().requires_grad_(True), V.detach().double().requires_grad_(True)
    ms = [torch.nn.Linear(m.in_features, m.out_features).to(DEV).double() for m in (l1, l2, l3)]
    for m64, m in zip(ms, (l1, l2, l3)):
        m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    feat = (P64.unsqueeze(-1) * V64.unsqueeze(-3)).transpose(1, 2).transpose(2, 3).contiguous().view(B, K, K, H * D)
    ref = ms[2](torch.relu(ms[1](torch.relu(ms[0](feat)))))
    (ref * wsum.double()).sum().backward()
    want = [ref, P64.grad, V64.grad] + [p.grad for m in ms for p in (m.weight, m.bias)]
    names = ["pred", "dP", "dV", "dW1", "db1", "dW2", "db2", "dW3", "db3"]
    errs = {n: _rel(a, b) for n, a, b in zip(names, got, want)}
    print("wide relation head vs float64:", {n: f"{e:.1e}" for n, e in errs.items()})
    # 8 M ReLU gates: one that sits within fp32 rounding of zero and resolves the other way than in float64 moves single
    # entries of dP / dV by its whole summand -- those two are held at the flip level, everything else at fp32 level
    for n, e in errs.items():
        assert e < (2e-3 if n in ("dP", "dV") else 3e-5), (n, e)
