"""csrc/dense_rows.hip against float64: row products of any shape (the first-layer feature product of the set-abstraction
modules with its column-sliced weight, lib/pointnet2/pointnet2_modules.py:241-259; the vocabulary projection and its
gradients, models/transformer_captioner.py:93-100, :373-379; the relation head's per-head value projection, :319-326) -- the
products that were rocBLAS calls inside the training step until round 4.  fp32 MFMA arithmetic: 1e-5 of the result's scale."""
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


@pytest.mark.parametrize("R,K,CO,col0,trans", [(16384, 128, 128, 3, True), (8192, 256, 128, 3, True), (4096, 128, 256, 3, False),
                                               (1000, 7, 64, 3, True), (320000, 132, 64, 3, True), (777, 64, 7, 3, False),
                                               (248, 128, 3001, 0, True), (33, 3001, 128, 0, False), (1, 5, 3, 0, True)])
def test_row_product_with_a_column_sliced_weight(lin, R, K, CO, col0, trans):
    g = torch.Generator().manual_seed(R + K)
    a = torch.randn(R, K, generator=g).to(DEV)
    W = torch.randn(CO if trans else K, col0 + (K if trans else CO), generator=g).to(DEV)
    bias = torch.randn(CO, generator=g).to(DEV) if trans else None
    out = lin.dense_product(a, W, trans, bias=bias, col0=col0)
    Wv = W[:, col0:].double()
    ref = a.double() @ (Wv.t() if trans else Wv)
    if bias is not None:
        ref = ref + bias.double()
    assert out.shape == ref.shape and _rel(out, ref) < 1e-5


@pytest.mark.parametrize("B,L,D,V,skip", [(8, 32, 128, 3001, 1), (3, 9, 128, 517, 1), (2, 5, 128, 64, 0), (16, 32, 128, 3001, 1)])
def test_vocabulary_projection_reads_and_writes_in_place(lin, B, L, D, V, skip):
    """VocabProjection = Linear over positions skip.. of every sequence: values and all three gradients against float64
    autograd of the sliced composition (models/transformer_captioner.py:373-379 + :93-100); the data gradient of the
    skipped positions is exactly zero."""
    g = torch.Generator().manual_seed(V)
    n = torch.randn(B, L, D, generator=g).to(DEV).requires_grad_(True)
    lin_m = torch.nn.Linear(D, V).to(DEV)
    out = lin.vocab_projection(n, lin_m, skip)
    n64 = n.detach().double().requires_grad_(True)
    w64, b64 = lin_m.weight.detach().double().requires_grad_(True), lin_m.bias.detach().double().requires_grad_(True)
    ref = torch.nn.functional.linear(n64[:, skip:, :], w64, b64)
    assert out.shape == ref.shape and _rel(out, ref) < 1e-5
    go = torch.randn(out.shape, generator=g).to(DEV)
    out.backward(go)
    ref.backward(go.double())
    assert _rel(n.grad, n64.grad) < 1e-5 and _rel(lin_m.weight.grad, w64.grad) < 1e-5 and _rel(lin_m.bias.grad, b64.grad) < 1e-5
    if skip:
        assert float(n.grad[:, :skip].abs().max()) == 0.0
    # twice the same backward: bit-identical (the K slices are added in a fixed order)
    n2 = n.detach().clone().requires_grad_(True)
    lin_m.zero_grad()
    lin.vocab_projection(n2, lin_m, skip).backward(go)
    assert torch.equal(n2.grad, n.grad)


@pytest.mark.parametrize("B,K", [(8, 256), (2, 40), (16, 512)])
def test_relation_value_projection_per_head(lin, B, K):
    """RelationU against float64 autograd of the einsum it replaces, on the strided view of a packed q | k | v projection."""
    H, D, C = 8, 16, 128
    g = torch.Generator().manual_seed(K)
    qkv = torch.randn(B, K, 3 * H * D, generator=g).to(DEV).requires_grad_(True)
    W1 = (0.1 * torch.randn(C, H * D, generator=g)).to(DEV).requires_grad_(True)
    V = qkv[..., 2 * H * D:].view(B, K, H, D).transpose(1, 2)
    U = lin.RelationU.apply(V, W1)
    q64, w64 = qkv.detach().double().requires_grad_(True), W1.detach().double().requires_grad_(True)
    V64 = q64[..., 2 * H * D:].view(B, K, H, D).transpose(1, 2)
    ref = torch.einsum("bhjd,ohd->bjho", V64, w64.view(C, H, D))
    assert U.shape == ref.shape and _rel(U, ref) < 1e-5
    go = torch.randn(U.shape, generator=g).to(DEV)
    U.backward(go)
    ref.backward(go.double())
    assert _rel(qkv.grad, q64.grad) < 1e-5 and _rel(W1.grad, w64.grad) < 1e-5


@pytest.mark.parametrize("R,M,N", [(320000, 64, 7), (320000, 64, 132), (5000, 64, 1), (1234, 100, 145), (37, 3, 16), (40000, 64, 17)])
def test_tall_weight_gradient_of_a_narrow_product(R, M, N):
    """spacap_dense_wgrad_tall_f32: dW = G^T X over many rows into a small, odd-width matrix -- the feature columns of SA1's first
    layer at 7 / 132 input channels (BASELINE configs 3 / 4; lib/pointnet2/pytorch_utils.py:11-36), which round 4 still sent to
    torch.bmm.  Per-slab partial results summed in slab order; against float64, 1e-5 of the result's scale (one fp32
    accumulation per slab of a few hundred rows)."""
    from spacap3d_amd._native import check, lib, sum_slabs
    g = torch.Generator().manual_seed(R + N)
    G = torch.randn(R, M, generator=g).to(DEV)
    X = (torch.randn(R, N, generator=g) + 0.3).to(DEV)
    nslab = int(lib.spacap_dense_wgrad_tall_slabs(R, M, N))
    assert nslab >= 1
    for ns in {nslab, 1, 3}:
        part = torch.full((ns, M * N), float("nan"), device=DEV)
        check(lib.spacap_dense_wgrad_tall_f32(G.data_ptr(), M, X.data_ptr(), N, R, M, N, ns, part.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "spacap_dense_wgrad_tall_f32")
        got = part.double().sum(0).view(M, N)
        ref = G.double().t() @ X.double()
        assert _rel(got, ref) < (1e-5 if ns > 1 or R < 50000 else 2e-4), (ns, _rel(got, ref))
    # strided operands (a column window of wider rows)
    Gw, Xw = torch.randn(R, M + 5, generator=g).to(DEV), torch.randn(R, N + 3, generator=g).to(DEV)
    part = torch.empty(nslab, M * N, device=DEV)
    check(lib.spacap_dense_wgrad_tall_f32(Gw.data_ptr(), M + 5, Xw.data_ptr(), N + 3, R, M, N, nslab, part.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "spacap_dense_wgrad_tall_f32")
    ref = Gw[:, :M].double().t() @ Xw[:, :N].double()
    assert _rel(part.double().sum(0).view(M, N), ref) < 1e-5
