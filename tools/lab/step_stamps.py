"""Lab: where the time of a pipelined, graph-replayed training step goes -- device timestamps (100 MHz wall clock) written by
one-thread kernels captured into the step at phase boundaries, forward and backward (an identity autograd node stamps when
the backward passes it).  Unlike a rocprofv3 kernel trace this does not serialise the side-stream sampling chain.
    python tools/lab/step_stamps.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spacap3d_amd import synthetic as S  # noqa: E402
from spacap3d_amd._native import check, lib  # noqa: E402
from spacap3d_amd.engine import Trainer, synthetic_batch  # noqa: E402
from spacap3d_amd.spacapnet import build_default  # noqa: E402

DEV = torch.device("cuda:0")
NAMES = []
BUF = torch.zeros(4096, dtype=torch.int64, device=DEV)


def stamp(name):
    if name not in NAMES:
        NAMES.append(name)
    i = NAMES.index(name)
    check(lib.spacap_lab_stamp(BUF.data_ptr() + 8 * i, torch.cuda.current_stream(DEV).cuda_stream), "stamp")


class Mark(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, name):
        ctx.name = name
        stamp("f:" + name)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        stamp("b:" + ctx.name)
        return g, None


def hook(x, name):
    """Stamps without an autograd node of their own (HOOKS=1): a forward stamp and a tensor hook for the backward."""
    if not os.environ.get("HOOKS"):
        return Mark.apply(x, name)
    stamp("f:" + name)
    if x.requires_grad:
        x.register_hook(lambda g: stamp("b:" + name))
    return x


def stamp_every_call():
    """CALLS=1: a stamp in front of every C-ABI call of the step (names numbered in call order): the duration of each of this
    library's kernels as it runs beside the side stream (plus whatever tensor operations follow it)."""
    from spacap3d_amd import _native
    count = [0]
    for name in _native.SIGNATURES:
        if name.startswith(("spacap_lab", "spacap_sa_reserve", "spacap_stream_delay")) or "supported" in name or "nparts" in name \
                or "slabs" in name or "zsplit" in name or "_parts" in name or "workspace" in name or "floats" in name or "isplit" in name \
                or "blocks" in name:
            continue
        fn = getattr(lib, name)

        only = [t for t in os.environ.get("ONLY", "").split(",") if t]

        def wrapped(*a, fn=fn, name=name):
            if not torch.cuda.is_current_stream_capturing():
                return fn(*a)
            count[0] += 1
            if only and not any(t in name for t in only):
                return fn(*a)
            stamp(f"{count[0]:04d} {name[7:]}")
            r = fn(*a)
            if only:   # ONLY=substr,substr: stamps around the named calls and nowhere else (a handful of extra launches per step)
                stamp(f"{count[0]:04d} {name[7:]} END")
            return r
        setattr(lib, name, wrapped)


def main():
    if os.environ.get("CALLS"):
        stamp_every_call()
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    torch.manual_seed(0)
    model = build_default().to(DEV).train()
    # phase boundaries: wrap the sub-modules' forwards
    bb, vg, pr, cap = model.backbone_net, model.vgen, model.proposal, model.caption
    f_bb = bb.forward

    def bb_forward(d):
        stamp("f:start")
        d = f_bb(d)
        for k in ("sa1_features", "sa2_features", "sa3_features", "sa4_features"):
            if d[k].requires_grad:
                d[k].register_hook(lambda g, k=k: stamp("b:" + k[:3]))
        d["fp2_features"] = hook(d["fp2_features"], "backbone")
        return d
    if os.environ.get("NO_MARKS"):
        bb_forward = f_bb
    bb.forward = bb_forward
    f_pr = pr.forward

    def pr_forward(xyz, features, d):
        d = f_pr(xyz, features, d)
        d["aggregated_vote_features"] = hook(d["aggregated_vote_features"], "proposal")
        return d
    if not os.environ.get("NO_MARKS"):
        pr.forward = pr_forward
    enc = cap.model.encode

    def encode(*a, **k):
        return hook(enc(*a, **k), "encoder")
    if not os.environ.get("NO_MARKS"):
        cap.model.encode = encode
    rel = cap._relation_head

    def relation(ep):
        rel(ep)
        ep["relation_pred"] = hook(ep["relation_pred"], "relation")
    if not os.environ.get("NO_MARKS"):
        cap._relation_head = relation
    dec = cap.model.decoder.forward

    def decoder(*a, **k):
        return hook(dec(*a, **k), "decoder")
    if not os.environ.get("NO_MARKS"):
        cap.model.decoder.forward = decoder
    tr = Trainer(model, S.mean_size_arr().numpy())
    if os.environ.get("GEOM") == "0":
        tr.prefetch_geometry = False      # sampling chain only on the side stream; groupings / neighbour searches inside the step
    if os.environ.get("SKEW"):
        tr.prefetch_skew_us = int(os.environ["SKEW"])
    f_loss = tr.loss

    def loss(d):
        out = f_loss(d)
        stamp("f:loss")
        return out
    if not os.environ.get("NO_MARKS"):
        tr.loss = loss
        f_opt = tr._optimizer_step

        def opt(src):
            stamp("b:done (deferred weight gradients flushed)")
            f_opt(src)
            stamp("opt:done")
        tr._optimizer_step = opt
        from spacap3d_amd import _native
        f_dq = _native.deferred_slab_sums

        import contextlib

        @contextlib.contextmanager
        def dq():
            with f_dq() as q:
                yield q
                stamp("b:autograd done")
        import spacap3d_amd.engine as E
        if hasattr(E, "deferred_slab_sums"):
            E.deferred_slab_sums = dq
    data = synthetic_batch(8, 40000, DEV, seed=1000)
    tr.step(data, next_data=data)
    assert tr.enable_graph(data), tr.graph_error
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    one = lambda: tr.step(data, next_data=data)
    if os.environ.get("NOSIDE"):   # the pyramid computed once and re-attached: no side-stream work beside the step
        tr.prefetch(data); torch.cuda.synchronize()
        saved = data["_fps_prefetch"]
        def one():
            data["_fps_prefetch"] = saved
            tr.step(data, next_data=None)
    for _ in range(10):
        one()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        one()
    e1.record()
    torch.cuda.synchronize()
    print(f"{e0.elapsed_time(e1) / steps:.3f} ms/step over {steps} steps")
    if not NAMES:
        return
    t = BUF.cpu().tolist()
    rows = sorted((t[i], n) for i, n in enumerate(NAMES))
    t0 = rows[0][0]
    prev = t0
    for v, n in rows:
        print(f"  {n:46s} at {(v - t0) / 100.0:8.1f} us   (+{(v - prev) / 100.0:7.1f})")
        prev = v
    if os.environ.get("CALLS"):   # time from each call to the next stamp, summed per entry point
        import collections
        tot, cnt = collections.Counter(), collections.Counter()
        for (v, n), (v2, _) in zip(rows, rows[1:]):
            key = n.split(" ", 1)[1] if n[:4].isdigit() else n
            tot[key] += (v2 - v) / 100.0
            cnt[key] += 1
        print("--- per entry point: us until the next stamp (the kernel itself + tensor operations that follow it)")
        for k, v in tot.most_common(40):
            print(f"  {k:46s} {v:8.1f} us  {cnt[k]:3d} calls")


if __name__ == "__main__":
    main()
