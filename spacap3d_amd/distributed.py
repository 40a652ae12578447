"""Data parallelism for the training step: one process per GPU, scenes sharded by rank, ONE collective per
step -- a single all-reduce of every gradient in one flat fp32 bucket (37 MB for the default model) over
RCCL / xGMI.  The reference's only multi-GPU mode is ``torch.nn.DataParallel``
(scripts/train.py:198-200: one process, replicate / scatter / gather every step); scenes are independent
units, so no other exchange is needed.  BatchNorm statistics stay per rank, as under DataParallel.

``FlatGradBucket`` makes every parameter's ``.grad`` a view into one contiguous buffer, so the all-reduce
needs no packing copies and ``zero()`` is a single memset.  Works with any backend (``nccl`` = RCCL on ROCm,
``gloo`` for the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend: str = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them).
    Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # SPACAP_DIST_BACKEND=gloo: test knob (several ranks sharing one GPU, where RCCL refuses to start)
            backend = os.environ.get("SPACAP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_scenes(global_batch: int, rank: int, world: int):
    """Even split of the scene indices of one global batch: rank r gets [r*B/world, (r+1)*B/world)."""
    assert global_batch % world == 0, "global batch must divide evenly over ranks"
    per = global_batch // world
    return range(rank * per, (rank + 1) * per)


@torch.no_grad()
def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Rank-0 parameters and buffers to every rank (once, at start)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


class FlatGradBucket:
    """All gradients of ``params`` in one flat buffer.

    Two modes:
      * ``views=True`` (default): every ``p.grad`` is a view into the flat buffer; autograd accumulates into it
        (one small add kernel per parameter per step), ``zero()`` is a single memset, the all-reduce needs no packing.
      * ``views=False``: gradients are produced by autograd as ordinary tensors (``zero()`` sets them to None, so
        autograd assigns instead of adding: ~190 launches fewer per step); ``pack()`` gathers them into the flat
        buffer with one multi-tensor copy right before the all-reduce and re-points ``p.grad`` at the reduced
        views.  A single-rank run never needs to pack at all."""

    def __init__(self, params, views: bool = True):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        self.views_mode = views
        dev, dt = self.params[0].device, self.params[0].dtype
        # every tensor starts on a 16-byte boundary of the flat buffer (kernels that write gradients in place, and the
        # flat optimizer, use 16-byte accesses); the padding elements stay zero and ride along in the all-reduce
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off = (off + p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(off, dtype=dt, device=dev)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]
        if views:
            for p, v in zip(self.params, self.views):
                p.grad = v

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def zero(self):
        if self.views_mode:
            self.flat.zero_()
        else:
            for p in self.params:
                p.grad = None

    def pack(self, sources=None):
        """views=False only: copy the freshly produced gradients into the flat buffer (one foreach copy) and point
        ``p.grad`` at the flat views.  ``sources``: the tensors holding the new gradients when they are not
        ``p.grad`` any more (a replayed hipGraph keeps writing into the tensors autograd allocated at capture)."""
        if self.views_mode:
            return
        if sources is None:
            sources = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        # gradients that were produced directly inside the flat buffer (engine.Trainer registers [dW | db] slots for
        # the Linear layers, _native.GRAD_SLOTS) need no copy
        pairs = [(v, s) for v, s in zip(self.views, sources) if s.data_ptr() != v.data_ptr()]
        if pairs and pairs[0][0].is_cuda:
            from ._native import copy_batched   # one launch per 120 tensors (the library's multi-tensor copy: ~4 x 20 us)
            copy_batched([v for v, _ in pairs], [s.contiguous() for _, s in pairs])
        elif pairs:
            torch._foreach_copy_([v for v, _ in pairs], [s for _, s in pairs])
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce(self, sources=None, force_pack=False, average=True):
        """The step's single collective: SUM of the flat bucket over the ranks.  Returns the factor that turns the result into
        the mean (1 / world): with ``average=True`` it has already been applied to the bucket (one more pass over it) and 1.0
        is returned; with ``average=False`` the caller folds it into its own pass (FlatAdam's ``grad_scale``).  No-op in a
        single process (unless ``force_pack``, used by tests to exercise the multi-rank code path on one GPU)."""
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if multi or force_pack:
            self.pack(sources)
        if not multi:
            return 1.0
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        world = dist.get_world_size()
        if average:
            self.flat.div_(world)
            return 1.0
        return 1.0 / world

    def all_reduce_mean(self, sources=None, force_pack=False):
        self.all_reduce(sources=sources, force_pack=force_pack, average=True)


def used_parameters(module: torch.nn.Module, loss: torch.Tensor):
    """Parameters that receive a gradient from ``loss`` (the early-guide decoder never touches its
    cross-attention blocks, models/transformer_captioner.py:223-224, so 60 tensors of the default model get
    none; the reference's Adam skips them because their .grad stays None)."""
    params = [p for p in module.parameters() if p.requires_grad]
    grads = torch.autograd.grad(loss, params, allow_unused=True, retain_graph=False)
    return [p for p, g in zip(params, grads) if g is not None]
