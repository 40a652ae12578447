"""World-size-2 CPU (gloo) tests of the data-parallel path: scene sharding, parameter broadcast, and the single
flat-bucket gradient all-reduce; two ranks on half batches must produce the gradients / weights of one rank on the
full batch for a batch-separable model (BatchNorm statistics are per rank by design, as under DataParallel)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spacap3d_amd.distributed import FlatGradBucket, broadcast_parameters, shard_scenes


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _make_model(seed=100 + rank)  # ranks start DIFFERENT; broadcast must align them
        broadcast_parameters(model)
        bucket = FlatGradBucket(model.parameters())
        opt = torch.optim.Adam(bucket.params, lr=1e-2)
        g = torch.Generator().manual_seed(0)
        X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
        mine = list(shard_scenes(8, rank, world))
        for _ in range(3):
            bucket.zero()
            loss = ((model(X[mine]) - Y[mine]) ** 2).mean()
            loss.backward()
            bucket.all_reduce_mean()
            opt.step()
        out_q.put((rank, [p.detach().numpy().copy() for p in model.parameters()], bucket.flat.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_single_process_full_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict()
    for _ in range(2):
        r, params, flat = q.get(timeout=120)
        results[r] = (params, flat)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, full batch, starting from rank 0's weights
    model = _make_model(seed=100)
    bucket = FlatGradBucket(model.parameters())
    opt = torch.optim.Adam(bucket.params, lr=1e-2)
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    for _ in range(3):
        bucket.zero()
        ((model(X) - Y) ** 2).mean().backward()
        bucket.all_reduce_mean()  # no-op without a process group
        opt.step()
    for r in (0, 1):
        for a, b in zip(results[r][0], model.parameters()):
            torch.testing.assert_close(torch.from_numpy(a), b.detach(), rtol=1e-5, atol=1e-6)
    assert (results[0][1] == results[1][1]).all()  # both ranks hold the same reduced gradient bucket


def test_flat_bucket_views_and_zero():
    m = _make_model(0)
    b = FlatGradBucket(m.parameters())
    n = sum(p.numel() for p in m.parameters())
    assert n <= b.flat.numel() <= n + 3 * len(b.params)                  # every tensor starts 16-byte aligned
    assert all(o % 4 == 0 for o in b.offsets) and all(v.data_ptr() % 16 == 0 for v in b.views)
    m(torch.ones(2, 6)).sum().backward()
    assert float(b.flat.abs().sum()) > 0
    for p in m.parameters():
        assert p.grad.data_ptr() >= b.flat.data_ptr()
    b.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in m.parameters())


def test_shard_scenes_partitions_the_batch():
    seen = []
    for r in range(4):
        seen += list(shard_scenes(64, r, 4))
    assert seen == list(range(64))
    with pytest.raises(AssertionError):
        shard_scenes(10, 0, 4)


def test_flat_bucket_pack_mode_matches_view_mode():
    """views=False: autograd assigns gradients, pack() gathers them; same flat content as the view mode."""
    a, b = _make_model(3), _make_model(3)
    ba, bb = FlatGradBucket(a.parameters(), views=True), FlatGradBucket(b.parameters(), views=False)
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(1))
    for m, bk in ((a, ba), (b, bb)):
        bk.zero()
        m(x).square().sum().backward()
        bk.pack()
    assert torch.equal(ba.flat, bb.flat)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bb.params, bb.views))
    bb.zero()
    assert all(p.grad is None for p in bb.params)


def _launch_workers(device, nproc=2, extra=(), timeout=900):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "tests", "dist_worker.py"), "--device", device,
           *extra]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:] + out.stdout[-1000:]
    assert out.stdout.count("DIST_OK") == nproc, out.stdout[-2000:]


def test_real_training_step_on_two_gloo_ranks():
    """The real Trainer step (CPU checker backend) on 2 ranks launched as the driver launches bench.py: reduced bucket =
    mean of the local buckets, bit-identical on both ranks; parameters stay bit-identical after the optimizer steps."""
    _launch_workers("cpu")


@pytest.mark.gpu
def test_real_training_step_on_two_rccl_ranks():
    """Same checks over RCCL with one GPU per rank (eager step, then the hipGraph + outside-graph all-reduce path);
    skipped on boxes with a single GPU."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    _launch_workers("cuda")
    _launch_workers("cuda", extra=("--graph", "--steps", "3"))
    _launch_workers("cuda", extra=("--graph", "--steps", "3", "--overlap"))


@pytest.mark.gpu
def test_overlapped_exchange_on_two_ranks_sharing_the_gpu():
    """The overlapped tail with a REAL process group on a one-GPU box: two ranks on cuda:0 over gloo, hipGraph step, the
    captioner's slice all-reduced on the communication stream behind the device-side wait, the error word's MAX all-reduce,
    the detector's slice on the main stream -- reduced buckets and parameters bit-identical on both ranks, no timeout."""
    _launch_workers("cuda", extra=("--graph", "--steps", "3", "--overlap", "--share-gpu"))
    _launch_workers("cuda", extra=("--graph", "--steps", "3", "--share-gpu"))


def test_flat_bucket_keeps_the_captioner_parameters_as_one_suffix():
    """The overlapped gradient exchange (engine.Trainer._boundary / _overlapped_tail) all-reduces the flat bucket in two slices:
    [detector | captioner].  That needs the captioner's parameters -- the packed q | k | v groups included, which _group_qkv
    moves to the end -- to form ONE suffix of the bucket, 16-byte aligned like every tensor of it."""
    import torch
    from spacap3d_amd import synthetic as S
    from spacap3d_amd.engine import Trainer, synthetic_batch
    from spacap3d_amd.spacapnet import build_default
    from oracle.attention_ref import OracleBackend
    from spacap3d_amd import backend
    torch.manual_seed(0)
    with backend.use_backend(OracleBackend()):
        model = build_default(vocab_size=60, num_proposal=16, N=1, d_ff=64).train()
        tr = Trainer(model, S.mean_size_arr().numpy())
        tr._setup(synthetic_batch(1, 1024, "cpu", seed=0, vocab=60))
    names = {id(p): n for n, p in model.named_parameters()}
    i0 = tr._cap_start
    ps = tr.bucket.params
    assert i0 is not None and 0 < i0 < len(ps)
    assert all(names[id(p)].startswith("caption.") for p in ps[i0:]) and not any(names[id(p)].startswith("caption.") for p in ps[:i0])
    assert tr.bucket.offsets[i0] % 4 == 0
    # the slices cover the bucket exactly
    assert tr.bucket.flat[:tr.bucket.offsets[i0]].numel() + tr.bucket.flat[tr.bucket.offsets[i0]:].numel() == tr.bucket.flat.numel()
