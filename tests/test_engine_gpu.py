"""Training engine on the GPU: the hipGraph replay path and the side-stream sampling prefetch must compute what
the plain eager step computes (same losses step by step), and two eager runs must agree (no races)."""
import copy

import pytest
import torch

from spacap3d_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(seed=0):
    from spacap3d_amd.engine import Trainer
    from spacap3d_amd.spacapnet import build_default
    torch.manual_seed(seed)
    model = build_default(vocab_size=200, num_proposal=64, N=2, d_ff=256).to(DEV).train()
    for m in model.modules():  # dropout off: the comparison must be deterministic
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def _run(mode, steps=6):
    from spacap3d_amd.engine import Trainer, synthetic_batch
    model = _make()
    # tiny learning rate: the first Adam steps at the reference's 1e-3 are chaotic on a repeated synthetic batch
    # (two identical eager runs drift apart by 3 % after three steps), which would hide real discrepancies
    tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-6, split_optimizer=(mode == "graph-split"),
                 multi_stream=not mode.endswith("single-stream"))
    data = synthetic_batch(2, 4096, DEV, seed=3, vocab=200)
    nxt = data if mode in ("prefetch", "graph", "graph-split", "graph-single-stream") else None
    losses = [float(tr.step(data, next_data=nxt))]
    if mode in ("graph", "graph-split", "graph-single-stream"):
        # enable_graph runs `warmup` real optimizer steps itself; account for them
        assert tr.enable_graph(data, warmup=2), tr.graph_error
        losses += [None, None]
    while len(losses) < steps:
        losses.append(float(tr.step(data, next_data=nxt)))
    return losses


def test_eager_is_repeatable_and_prefetch_graph_agree():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    a = _run("eager")
    b = _run("eager")
    c = _run("prefetch")
    g = _run("graph")
    gs = _run("graph-split")  # what a multi-rank run does: fwd+bwd in the graph, gradient packing + Adam outside
    # every mode above runs the relation head and the detection losses as side-stream branches
    # (spacap3d_amd/streams.py); these two keep the whole step on one stream
    e1 = _run("eager-single-stream")
    g1 = _run("graph-single-stream")
    assert all(x == x and abs(x) < 1e5 for x in a)
    for name, other in (("eager-again", b), ("prefetch", c), ("graph", g), ("graph-split", gs),
                        ("eager-single-stream", e1), ("graph-single-stream", g1)):
        for i, (x, y) in enumerate(zip(a, other)):
            if y is None:
                continue
            assert abs(x - y) <= 2e-3 * abs(x) + 1e-4, (name, i, a, other)


def test_flat_adam_matches_torch_adam():
    """spacap3d_amd/optim.py (one launch over a flat parameter buffer) vs torch.optim.Adam with the reference's
    settings (scripts/train.py:262), 5 steps of random gradients on oddly shaped tensors."""
    from spacap3d_amd.distributed import FlatGradBucket
    from spacap3d_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(128, 128), (7,), (3, 5, 2), (2048, 128), (1,), (259, 128, 1, 1)]
    pa = [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    ref = torch.optim.Adam(pb, lr=1e-3, weight_decay=1e-5)
    bucket = FlatGradBucket(pa, views=False)
    opt = FlatAdam(bucket, lr=1e-3, weight_decay=1e-5)
    for p, q in zip(pa, pb):
        assert torch.equal(p, q)            # re-pointing at the flat buffer preserved the values
    for step in range(5):
        grads = [torch.randn(*s, generator=g).to(DEV) * (10.0 ** (step - 2)) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        bucket.pack()
        opt.step()
        ref.step()
        for p, q in zip(pa, pb):
            assert torch.allclose(p, q, rtol=2e-6, atol=2e-7), (step, float((p - q).abs().max()))
