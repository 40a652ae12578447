"""Minimal training engine for the hot path: forward of SpaCapNet + total loss + backward + the single
gradient all-reduce + Adam -- what ``Solver._forward/_compute_loss/_backward`` do per iteration in the
reference (lib/solver.py:343-385), without its host synchronisations (``.item()`` x9, timers,
CUDA_LAUNCH_BLOCKING=1) and without logging / checkpointing (out of scope).
"""
import os

import torch

from . import synthetic as S
from .detector import geometry_pyramid, sampling_pyramid
from .distributed import FlatGradBucket, broadcast_parameters, used_parameters
from .loss_helper import get_scene_cap_loss, start_detection_losses


# One stream per (device, role) for the whole process.  torch hands out pool streams round robin and the ROCm runtime maps them onto
# a handful of hardware queues: a Trainer that made its own side / capture / communication streams could, as the N-th Trainer of a
# process, get a side stream that shares a hardware queue with the stream its step is replayed on -- the "hidden" sampling chain
# then serialises with the step (measured: BASELINE config 4 as the third configuration of one bench.py process, 12.7 instead of
# 9.0 ms per step).  Fixed roles, created once in a fixed order, keep the placement the same for every Trainer / Evaluator.
_STREAMS = {}


def _role_stream(device, role):
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _STREAMS:
        dev = torch.device("cuda", key[1])
        # (lab switch: a high-priority side stream changes nothing in the usual placement and makes the bad one worse,
        # tools/lab/quick_config_steps.py)
        prio = int(os.environ.get("SPACAP_SIDE_PRIORITY", "0"))
        _STREAMS[key] = {r: torch.cuda.Stream(device=dev, priority=prio if r == "side" else 0)
                         for r in ("side", "capture", "comm", "wgrad", "relation")}
    return _STREAMS[key][role]


class Trainer:
    def __init__(self, model: torch.nn.Module, mean_size_arr, lr: float = 1e-3, weight_decay: float = 1e-5,
                 use_relation: bool = True, split_optimizer: bool = False,
                 adam_eps: float = 1e-8):
        self.model = model
        self.mean_size_arr = mean_size_arr
        self.use_relation = use_relation
        self.lr, self.weight_decay, self.adam_eps = lr, weight_decay, adam_eps
        self.bucket = None
        self.optimizer = None
        self.side_stream = None
        self.prefetch_geometry = True   # False: prefetch the sampling indices only (sampling_pyramid)
        self.prefetch_skew_us = 80      # pause of the side stream before the pyramid's graph (see prefetch; 0: +0.65 ms, 20: +0.04 ms)
        self.graph = None          # captured hipGraph of one training step (see enable_graph)
        self.graph_error = None
        self._static = None
        self._static_loss = None
        self._eager_steps = 0
        self._eager_checks = 2     # first steps: verify the deferred-gradient contract (see _core)
        self._graph_grads = None
        self._capture_stream = None
        self.last_losses = {}
        self._bn_counters = None
        self._grad_slots = {}
        self.grad_scale = 1.0
        self._recapture = False
        self._overlap_armed = False
        # tests: behave as a multi-rank run does (optimizer + gradient packing outside the graph) on one GPU
        self.split_optimizer = split_optimizer
        self.prefetch_graph = os.environ.get("SPACAP_PREFETCH_GRAPH", "1") != "0"   # the side-stream pyramid as one graph launch
        self.fork_relation = os.environ.get("SPACAP_FORK_RELATION", "1") != "0"     # relation head beside the decoder (_fork_relation)
        self.flush_mid = os.environ.get("SPACAP_FLUSH_MID", "1") != "0"             # captioner weight gradients beside the detector's backward
        self._flush_mid, self._mid_done = False, 0
        self._mid_stream = None
        # multi-rank tail: SPACAP_OVERLAP_ALLREDUCE=1 all-reduces the captioner's slice of the flat gradient bucket on a
        # communication stream while the detector's backward is still running (see _boundary / _optimizer_step).  OFF by
        # default -- one all-reduce after the whole backward -- until the RCCL leg has passed the bitwise bucket test on a
        # box with two GPUs (tests/dist_worker.py; no such box in this pool so far: only the gloo shared-GPU leg has run).
        self.overlap_allreduce = os.environ.get("SPACAP_OVERLAP_ALLREDUCE", "0") == "1"
        self.overlap_timeout_ms = 20000   # a boundary that never arrives within this is an error, never a fall-through
        self._cap_start = None      # index into bucket.params where the captioner's parameters begin (a suffix of the bucket)
        self._comm_stream = None
        # int64 [4] on the device (allocated in _setup, outside any capture): [0] = armed steps begun, [1] = "captioner
        # gradients packed" flag = the value of [0] its step published, [2] = STICKY error word (a stream wait timed out:
        # spacap_stream_wait_ge; max over the ranks), [3] unused
        self._sig = None
        self._sig_host = 0          # host mirror of _sig[0]
        self._sig_resync = False    # an exception may have separated the two: re-read the device word at the next tail
        self._err_host = None       # pinned int64 [1]: the error word as of the previous overlapped tail (+ its event)
        self._err_event = None
        self._boundary_done = False
        self._armed_step = False
        self.boundary_launches = 0  # (tests: how many times the boundary actions ran)
        self.force_comm_wait = False  # (tests: queue the communication stream's wait on one rank too)
        if next(model.parameters()).is_cuda:
            # process-wide kernel setting, owned by the newest Trainer: no CUs left out until this one prefetches
            from ._native import check, lib
            check(lib.spacap_sa_reserve_cus(0), "spacap_sa_reserve_cus")
            self._fork_relation(next(model.parameters()).device)   # (here, not at capture: eager and replayed steps then launch the same grids)
        self._prefetch_graph_key = self._prefetch_graph_obj = self._prefetch_in = self._prefetch_out = None
        broadcast_parameters(model)

    # -- sampling-pyramid prefetch ---------------------------------------------------------------------------
    # The furthest-point sampling of the backbone is a chain of ~4 000 sequentially dependent rounds that
    # occupies one CU per scene (8 of 256 CUs at cfg2) for ~5 ms and depends on the input coordinates only.
    # prefetch() runs it for the NEXT batch on a side HIP stream while the current step's dense work fills the
    # other CUs; step() then consumes the indices through the modules' `inds` argument.  Each step still
    # computes exactly one pyramid; only its placement in time changes.
    def prefetch(self, next_data):
        pc = next_data["point_clouds"]
        if not pc.is_cuda:
            return
        if self.side_stream is None:
            self.side_stream = _role_stream(pc.device, "side")
            # the sampling chain holds one CU per scene while the backbone's forward runs beside it: the forward layer
            # kernels size their persistent grids to the rest of the chip (csrc/sa_mlp.hip: spacap_sa_reserve_cus)
            from ._native import check, lib
            check(lib.spacap_sa_reserve_cus(min(64, int(pc.shape[0]))), "spacap_sa_reserve_cus")
        cur = torch.cuda.current_stream(pc.device)
        self.side_stream.wait_stream(cur)
        fn = geometry_pyramid if self.prefetch_geometry else sampling_pyramid
        if self.graph is not None and self.prefetch_graph and not torch.cuda.is_current_stream_capturing():
            # beside a replayed step the pyramid is ONE graph launch as well: ~60 eager launches per step from the host cost
            # the main stream ~0.27 ms (tools/lab/step_without_side_stream.py).  Static input / outputs: the step copies
            # the outputs into its own static buffers before the next prefetch may overwrite them (wait_stream above).
            key = (tuple(pc.shape), fn is geometry_pyramid)
            if self._prefetch_graph_key != key:
                with torch.cuda.stream(self.side_stream), torch.no_grad():
                    self._prefetch_in = pc[..., :3].contiguous().clone()
                    for _ in range(2):
                        fn(self._prefetch_in)
                self.side_stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(g, stream=self.side_stream):
                    self._prefetch_out = fn(self._prefetch_in)
                self._prefetch_graph_obj, self._prefetch_graph_key = g, key
                self.side_stream.wait_stream(cur)
            with torch.cuda.stream(self.side_stream), torch.no_grad():
                self._prefetch_in.copy_(pc[..., :3], non_blocking=True)
                # The pyramid and the step's graph are released by the same event (the copy into the static buffers).  When
                # both become runnable at the same instant the step takes 8.63 ms instead of 8.11 (cfg2, same box, bench.py;
                # pauses of 5 .. 200 us all give 8.10 - 8.13, a pause on the main stream instead does the same), so the side
                # stream pauses first.
                skew = int(self.prefetch_skew_us)
                if skew > 0:
                    from ._native import check, lib
                    check(lib.spacap_stream_delay(skew, self.side_stream.cuda_stream), "spacap_stream_delay")
                self._prefetch_graph_obj.replay()
                ev = torch.cuda.Event()
                ev.record(self.side_stream)
            next_data["_fps_prefetch"] = (self._prefetch_out, ev)
            return
        with torch.cuda.stream(self.side_stream), torch.no_grad():
            # sampling indices + ball-query groupings + interpolation weights: all functions of the coordinates
            pyr = fn(pc[..., :3].contiguous())
            ev = torch.cuda.Event()
            ev.record(self.side_stream)
        next_data["_fps_prefetch"] = (pyr, ev)

    @staticmethod
    def _consume_prefetch(data_dict):
        pre = data_dict.pop("_fps_prefetch", None)
        if pre is None:
            return data_dict
        pyr, ev = pre
        cur = torch.cuda.current_stream(pyr[0].device)
        cur.wait_event(ev)
        for t in pyr:
            t.record_stream(cur)
        d = dict(data_dict)
        d["fps_pyramid"] = pyr
        return d

    def loss(self, data_dict):
        # the detection losses only need the proposal module's outputs: started right after it
        kw = dict(num_heading_bin=S.NUM_HEADING_BIN, num_size_cluster=S.NUM_SIZE_CLUSTER, mean_size_arr=self.mean_size_arr)
        need = ("vote_label", "center_label")
        early = (lambda d: start_detection_losses(d, **kw)) if all(k in data_dict for k in need) else None
        d = self.model(dict(data_dict), after_proposal=early)
        det_vec = d.get("_det_vec")
        d = get_scene_cap_loss(d, use_relation=self.use_relation, **kw)
        if self._overlap_armed or self._flush_mid:
            # The backward runs the captioner's nodes first (they were created last), then the detection losses, then the
            # detector.  A hook on the first detector-side tensor therefore fires exactly when every captioner gradient exists.
            marks = [t for t in (det_vec, d.get("aggregated_vote_features"), d.get("aggregated_vote_xyz"))
                     if torch.is_tensor(t) and t.requires_grad]
            for t in marks:
                t.register_hook(self._boundary_hook if self._overlap_armed else self._mid_flush_hook)
            # (A second flush when the backward enters SA4 -- the vote / proposal / feature-propagation nets' convolution weight
            # gradients beside SA4 - SA2's backward -- was registered on the typed channel-major VIEW of sa4_features for most of
            # round 6 and never fired: the consumers take the point-major tensor behind it.  Registered on that tensor it fires and
            # costs 0.75 ms beside the sampling chain (7.13 -> 7.85 ms same box; without the chain 6.76 = 6.76): the SA modules'
            # persistent backward kernels are sized for the CUs the chain leaves, a third party on the chip sends their last
            # workgroups into a second round.  Not registered.  A flush of the proposal / vote nets' convolution weight gradients beside
            # the feature-propagation modules' backward, JOINED before SA4's backward starts, costs 1.05 ms with the chain beside
            # the step and 0.08 ms without: 0.38 ms with GPU_MAX_HW_QUEUES=8 or 16 and 6 ms with 6 -- the step's graph, its forked
            # branches and the pyramid's graph share the runtime's four hardware queues, and where a further branch lands decides.)
        return d

    # -- the captioner's weight gradients beside the detector's backward --------------------------------------------------
    # When the backward reaches the detector, every weight gradient of the captioner (~50 Linear layers: one batched launch of
    # 0.12 ms + its slab sums) is queued and nothing but the optimizer will read it.  What follows on the step's stream -- the
    # backward of the proposal / vote / feature-propagation nets and of SA4 / SA3 -- is ~0.5 ms of short launches that leave most
    # of the chip idle: the queue is flushed THERE, on a stream of its own, instead of after the whole backward.
    def _mid_flush_hook(self, grad):
        if self._flush_mid and not self._mid_done:    # (registered on up to three boundary tensors: the first to arrive flushes)
            self._mid_done = 1
            from . import _native
            dq = _native._DEFERRED
            if dq is not None and grad.is_cuda:
                dev = grad.device
                ws = _role_stream(dev, "wgrad")
                cur = torch.cuda.current_stream(dev)
                ws.wait_stream(cur)
                # (the caching allocator must know both streams touch these, inside a capture as well: see
                # TransformerDecoderModel._relation_head_forked)
                for j in dq.jobs + dq.conv_jobs:
                    for t in (j[0], j[1], j[3]):
                        t.record_stream(ws)
                for part, out in dq.items:
                    part.record_stream(ws)
                    out.record_stream(ws)
                with torch.cuda.stream(ws):
                    dq.flush()
                self._mid_stream = ws
        return None

    def _setup(self, data_dict):
        """First step: discover which parameters the loss reaches, then lay their gradients out in one flat
        bucket and build Adam over exactly those (scripts/train.py:262: Adam lr 1e-3, weight_decay 1e-5)."""
        # (this discovery pass is not a training step: BatchNorm running statistics / batch counters it moves are restored)
        bufs = [(b, b.detach().clone()) for b in self.model.buffers()]
        d = self.loss(data_dict)
        used = used_parameters(self.model, d["loss"])
        with torch.no_grad():
            for b, saved in bufs:
                b.copy_(saved)
        # gradients are assigned by autograd (no per-parameter accumulate kernels) and packed into the flat
        # bucket only when there is something to all-reduce
        used = self._group_qkv(used)
        self._adopt_bn_counters()
        self.bucket = FlatGradBucket(used, views=False)
        kw = dict(lr=self.lr, weight_decay=self.weight_decay, eps=self.adam_eps)
        if used[0].is_cuda:
            from .optim import FlatAdam   # one launch over a flat parameter buffer (spacap3d_amd/optim.py)
            self.optimizer = FlatAdam(self.bucket, **kw)
            self._attach_packed_qkv()
            self._register_grad_slots()
        else:
            self.optimizer = torch.optim.Adam(used, **kw)
        self._find_captioner_suffix()
        if used[0].is_cuda and self._sig is None:
            # here and not at first use: a zero-fill issued inside enable_graph's capture would be replayed every step
            self._sig = torch.zeros(4, dtype=torch.int64, device=used[0].device)
            # the backward's seed (see _seed_grad): allocated and filled here, outside any capture, for the same reason
            self._one = torch.ones_like(d["loss"].detach())
            self._comm_stream = _role_stream(used[0].device, "comm")
            self._err_host = torch.zeros(1, dtype=torch.int64).pin_memory()

    def _adopt_bn_counters(self):
        """The ``num_batches_tracked`` buffers of the BatchNorm layers whose training forward runs through the fused ops
        (fused_bn.bump_counter) become views of ONE int64 buffer that the step bumps once: ~25 one-element add launches
        per step fewer.  Values, names and state_dict entries are unchanged.  (Only inside this Trainer's steps: a
        train-mode forward outside it no longer advances these counters.)"""
        from .detector import ProposalModule, VotingModule
        from .pointnet2_modules import _BN2d
        from .transformer_captioner import PositionalEncodingLearned
        bns = []
        for m in self.model.modules():
            if isinstance(m, _BN2d):
                bns.append(m.bn)
            elif isinstance(m, VotingModule):
                bns += [m.bn1, m.bn2]
            elif isinstance(m, ProposalModule):
                bns += [l for l in m.proposal if isinstance(l, torch.nn.BatchNorm1d)]
            elif isinstance(m, PositionalEncodingLearned):
                bns.append(m.position_embedding_head[1])
        # ... and only those the discovery pass of _setup actually saw on the fused path: a layer that fell back to the stock
        # nn.BatchNorm (unsupported widths, a backend without the fused op) increments its own buffer and would count twice
        from .fused_bn import FUSED_SEEN
        bns = [b for b in bns if b.track_running_stats and b.num_batches_tracked is not None and b.momentum is not None
               and b in FUSED_SEEN]
        if not bns or not bns[0].num_batches_tracked.is_cuda:
            return
        flat = torch.stack([b.num_batches_tracked.detach() for b in bns]).contiguous()
        for i, b in enumerate(bns):
            view = flat[i]
            view._spacap_deferred = True
            b._buffers["num_batches_tracked"] = view
        self._bn_counters = flat

    def _attention_modules(self):
        from .transformer_captioner import MultiHeadedAttention
        return [m for m in self.model.modules() if isinstance(m, MultiHeadedAttention)]

    def _group_qkv(self, used):
        """Order the flat buffer so that the q, k, v weights of every attention module are adjacent (then their
        biases): the packed projection reads them as ONE (3*d, d) matrix without a concatenation per step."""
        pos = {id(p): i for i, p in enumerate(used)}
        moved, groups = set(), []
        for m in self._attention_modules():
            ws = [l.weight for l in m.linears[:3]]
            bs = [l.bias for l in m.linears[:3]]
            if all(id(p) in pos for p in ws + bs):
                groups.append(ws + bs)
                moved.update(id(p) for p in ws + bs)
        rest = [p for p in used if id(p) not in moved]
        return rest + [p for g in groups for p in g]

    def _register_grad_slots(self):
        """[dW | db] of every nn.Linear whose OWN bias directly follows its weight in the flat bucket (matched by module,
        not by shape), and of the packed q | k | v groups (three weights, then their three biases) -> the matching slice of
        the flat gradient buffer.  The table belongs to this Trainer and is only active around its backward
        (_native.grad_slots in _core): the deferred slab sums then write there and the gradient pack skips them."""
        self._grad_slots = {}
        ps, flat, offs = self.bucket.params, self.bucket.flat, self.bucket.offsets
        pos = {id(p): i for i, p in enumerate(ps)}
        for m in self.model.modules():
            if isinstance(m, torch.nn.Linear) and m.bias is not None:
                iw, ib = pos.get(id(m.weight)), pos.get(id(m.bias))
                if iw is not None and ib == iw + 1 and offs[ib] == offs[iw] + m.weight.numel():
                    self._grad_slots[m.weight.data_ptr()] = flat[offs[iw]:offs[iw] + m.weight.numel() + m.bias.numel()]
        for m in self._attention_modules():
            pk = getattr(m, "_packed_qkv", None)
            ws, bs = [l.weight for l in m.linears[:3]], [l.bias for l in m.linears[:3]]
            if pk is None or not all(id(t) in pos for t in ws + bs):
                continue
            i0 = pos[id(ws[0])]
            n = sum(t.numel() for t in ws + bs)
            if [pos[id(t)] for t in ws + bs] == list(range(i0, i0 + 6)) and offs[i0 + 5] + bs[2].numel() == offs[i0] + n:
                self._grad_slots[pk[0].data_ptr()] = flat[offs[i0]:offs[i0] + n]

    def _attach_packed_qkv(self):
        from .linear import packed_views
        for m in self._attention_modules():
            pk = packed_views(self.optimizer.flat_p, [l.weight for l in m.linears[:3]], [l.bias for l in m.linears[:3]])
            m._packed_qkv = pk

    def _core(self, data_dict, with_optimizer=True):
        """zero grads -> forward -> loss -> backward [-> all-reduce -> Adam]; no host sync, capturable."""
        pc = data_dict["point_clouds"]
        if pc.is_cuda:
            from .attention import advance_rng
            advance_rng(pc.device)  # new attention-dropout masks every step, also under graph replay
        self.bucket.zero()
        if self._bn_counters is not None:
            self._bn_counters.add_(1)    # every BatchNorm layer's num_batches_tracked, one launch (see _adopt_bn_counters)
        self._overlap_armed = self._overlap_possible(pc)
        if pc.is_cuda and (self.fork_relation and self.use_relation and not self._overlap_armed) != getattr(self, "_fork_on", False):
            self._fork_relation(pc.device, armed=self._overlap_armed)   # (a test switched the overlapped exchange on / off after construction)
        self._armed_step = self._overlap_armed   # (persists through graph replays: the tail counts the step, see _optimizer_step)
        self._boundary_done = False
        self._mid_done, self._mid_stream = 0, None
        self._flush_mid = self.flush_mid and pc.is_cuda and not self._overlap_armed
        if self._overlap_armed:
            self._sig[:1].add_(1)        # the step number the boundary will publish
        d = self.loss(data_dict)
        if pc.is_cuda:
            # the ~70 weight-gradient slab sums of the backward are only read by the optimizer: queue them and run them
            # as ONE launch when the backward is over (spacap3d_amd/_native.py: deferred_slab_sums)
            from ._native import deferred_slab_sums, grad_slots
            # (only when autograd will ASSIGN the gradients: accumulating into an existing .grad reads them at once -- and
            # must not be handed views of the flat bucket either, so the slot table is only active in this branch)
            if all(p.grad is None for p in self.bucket.params):
                with grad_slots(getattr(self, "_grad_slots", None)), deferred_slab_sums() as dq:
                    d["loss"].backward(self._seed_grad(d["loss"]))
                    if self._mid_stream is not None:    # the captioner's weight gradients ran beside the detector's backward
                        torch.cuda.current_stream(pc.device).wait_stream(self._mid_stream)
                        self._mid_stream = None
                    if self._eager_checks > 0:
                        # A queued sum is unfilled until the flush: every byte of it must have reached parameters' .grad
                        # untouched (a parameter consumed by two autograd nodes, or an AccumulateGrad that clones, would
                        # have read it).  Compared by address RANGE: a storage pointer would match any view of the bucket.
                        self._eager_checks -= 1
                        spans = sorted((g.data_ptr(), g.data_ptr() + g.numel() * g.element_size())
                                       for g in (p.grad for p in self.model.parameters()) if g is not None)
                        merged = []
                        for lo, hi in spans:
                            if merged and lo <= merged[-1][1]:
                                merged[-1][1] = max(merged[-1][1], hi)
                            else:
                                merged.append([lo, hi])
                        def covered(o):   # (a few alignment-padding elements inside a sum are nobody's gradient)
                            lo, hi = o.data_ptr(), o.data_ptr() + o.numel() * o.element_size()
                            got = sum(max(0, min(hi, b) - max(lo, a)) for a, b in merged)
                            return got >= (hi - lo) - 15 * o.element_size()
                        lost = [tuple(o.shape) for o in dq.outputs() if not covered(o)]
                        if lost:
                            raise RuntimeError(f"deferred weight-gradient sums did not land in a leaf .grad: {lost}")
            else:
                d["loss"].backward()
        else:
            d["loss"].backward()
        self._overlap_armed = False
        if with_optimizer:
            self._optimizer_step(None)
        # the loss terms of this step as device scalars (no host sync; under graph replay: the static result tensors)
        self.last_losses = {k: d[k].detach() for k in ("loss", "vote_loss", "objectness_loss", "box_loss", "sem_cls_loss",
                                                       "cap_loss", "relation_loss") if k in d and torch.is_tensor(d[k])}
        return d["loss"].detach()

    def _seed_grad(self, loss):
        """The backward's seed d loss / d loss = 1 as a tensor that lives with the Trainer (autograd otherwise fills a fresh
        ones_like(loss) every step: one more launch inside the captured step)."""
        one = getattr(self, "_one", None)
        if one is None or one.device != loss.device or one.dtype != loss.dtype or one.shape != loss.shape:
            one = self._one = torch.ones_like(loss)
        return one

    def _optimizer_step(self, sources):
        """gradient all-reduce (multi-rank) + Adam.  FlatAdam reads the packed gradients of the flat bucket and takes the
        1 / world of the mean as its gradient scale: no separate division pass over the 36 MB bucket."""
        flat = not isinstance(self.optimizer, torch.optim.Optimizer)
        if getattr(self, "_armed_step", False):
            if self._sig_resync:
                # a step that raised between the device-side increment and this point left the mirror behind: a wait for the
                # stale value would pass at once.  Re-read the device word (it already counts the step being finished).
                torch.cuda.synchronize(self._sig.device)
                self._sig_host = int(self._sig[0].item())
                self._sig_resync = False
            else:
                self._sig_host += 1    # mirrors the device-side step counter: one increment per EXECUTED armed step, boundary or not
        if flat and self._boundary_done:
            # the captioner's slice was packed (and published) in the middle of the backward: its all-reduce goes to the
            # communication stream behind a wait for that flag, the detector's slice follows on this stream
            self._overlapped_tail(sources)
            return
        scale = self.bucket.all_reduce(sources=sources, force_pack=self.split_optimizer or flat, average=not flat)
        self.grad_scale = scale            # bucket.flat * grad_scale = the mean gradient the optimizer applied
        if flat:
            self.optimizer.step(grad_scale=scale)
        else:
            self.optimizer.step()

    # -- all-reduce overlapped with the backward ---------------------------------------------------------------------
    # The flat bucket holds the detector's parameters first and the captioner's last (_find_captioner_suffix).  The backward
    # finishes the captioner long before the backbone (it is ~40 % of the backward's time), so its slice can travel while
    # the detector's backward still runs: at the boundary (a tensor hook, see loss()) the step flushes the deferred weight
    # gradients queued so far, copies the captioner's stray gradients into the bucket and publishes the step number in a
    # device word; the host, after launching the step, queues "wait for that word, all-reduce the captioner's slice" on a
    # communication stream.  All of this is captured into the step's hipGraph like any other kernel; the collective
    # itself stays outside (RCCL calls are not captured).  Values are those of the serial tail: an all-reduce is elementwise.
    def _find_captioner_suffix(self):
        names = {id(p): n for n, p in self.model.named_parameters()}
        ps = self.bucket.params
        i = len(ps)
        while i > 0 and names.get(id(ps[i - 1]), "").startswith("caption."):
            i -= 1
        ok = 0 < i < len(ps) and not any(names.get(id(p), "").startswith("caption.") for p in ps[:i])
        self._cap_start = i if ok else None

    def _overlap_possible(self, pc):
        if not (self.overlap_allreduce and pc.is_cuda and self._cap_start is not None and self.bucket is not None
                and not self.bucket.views_mode and not isinstance(self.optimizer, torch.optim.Optimizer)):
            return False
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not (multi or self.split_optimizer):
            return False
        return self._sig is not None      # (allocated by _setup)

    def _boundary_hook(self, grad):
        if self._overlap_armed and not self._boundary_done:
            self._boundary_done = True
            self._boundary()
        return None

    def _boundary(self):
        from . import _native
        from ._native import check, copy_batched, lib
        dq = _native._DEFERRED
        if dq is not None:
            dq.flush()             # the captioner's weight gradients and slab sums (one batched launch each)
        b, i0 = self.bucket, self._cap_start
        pairs = [(v, p.grad) for p, v in zip(b.params[i0:], b.views[i0:]) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if pairs:
            copy_batched([v for v, _ in pairs], [g.contiguous() for _, g in pairs])
        # a parameter this step produced no gradient for contributes zeros, as FlatGradBucket.pack writes them (its slice
        # still holds the previous step's reduced gradient: bucket.zero() does not clear the flat buffer in this mode)
        none = [v for p, v in zip(b.params[i0:], b.views[i0:]) if p.grad is None]
        if none:
            torch._foreach_zero_(none)
        dev = b.flat.device
        check(lib.spacap_stream_signal(self._sig[1:2].data_ptr(), self._sig[:1].data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
              "spacap_stream_signal")
        self.boundary_launches += 1

    def _overlapped_tail(self, sources):
        import torch.distributed as dist
        from ._native import check, copy_batched, lib
        b, i0 = self.bucket, self._cap_start
        dev = b.flat.device
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        off = b.offsets[i0]
        main = torch.cuda.current_stream(dev)
        err = self._sig[2:3]
        if multi or self.force_comm_wait:
            comm = self._comm_stream
            with torch.cuda.stream(comm):
                # Nothing else orders this stream against the step: the wait IS the dependency.  If it times out, the
                # collective behind it runs on a half-written slice -- the device cannot un-queue it -- so a timeout (a) sets
                # the sticky error word, (b) reaches every rank (MAX all-reduce of the word), (c) turns this step's Adam
                # off on the device (skip_word) and (d) raises on the host at the next step() / check_health().
                check(lib.spacap_stream_wait_ge(self._sig[1:2].data_ptr(), self._sig_host, int(self.overlap_timeout_ms),
                                                err.data_ptr(), comm.cuda_stream), "spacap_stream_wait_ge")
                if multi:
                    dist.all_reduce(b.flat[off:], op=dist.ReduceOp.SUM)
                    dist.all_reduce(err, op=dist.ReduceOp.MAX)
        # the detector's stray gradients (everything the captured step did not produce inside the bucket)
        if sources is None:
            sources = [p.grad for p in b.params]
        pairs = [(v, s_) for v, s_ in zip(b.views[:i0], sources[:i0]) if s_ is not None and s_.data_ptr() != v.data_ptr()]
        if pairs:
            copy_batched([v for v, _ in pairs], [s_.contiguous() for _, s_ in pairs])
        none = [v for v, s_ in zip(b.views[:i0], sources[:i0]) if s_ is None]
        if none:
            torch._foreach_zero_(none)     # (as FlatGradBucket.pack: no gradient this step = zeros, not last step's)
        for p, v in zip(b.params, b.views):
            p.grad = v
        if multi:
            dist.all_reduce(b.flat[:off], op=dist.ReduceOp.SUM)
        if multi or self.force_comm_wait:
            main.wait_stream(self._comm_stream)
        self.grad_scale = 1.0 / dist.get_world_size() if multi else 1.0
        self.optimizer.step(grad_scale=self.grad_scale, skip_word=err)
        # the error word travels to pinned host memory behind Adam; step() looks at it once the copy has landed
        self._err_host.copy_(err, non_blocking=True)
        self._err_event = torch.cuda.Event()
        self._err_event.record(main)

    def check_health(self, wait=True):
        """Raises if a stream wait of the overlapped gradient exchange ever timed out on any rank (the sticky error word; the
        optimizer update of that step was skipped on the device).  ``wait=False``: only looks at what has already reached
        the host (step() does this every time, one step behind, without stalling the stream)."""
        if self._sig is None or self._err_event is None:
            return
        if wait:
            self._err_event.synchronize()
        elif not self._err_event.query():
            return
        bad = int(self._err_host[0])
        if bad:
            raise RuntimeError(
                f"overlapped gradient all-reduce: the wait for step {bad}'s captioner gradients timed out after "
                f"{self.overlap_timeout_ms} ms on some rank; that step's optimizer update was skipped on every rank and "
                "the exchange cannot be trusted any more (set SPACAP_OVERLAP_ALLREDUCE=0 for the serial tail)")

    # -- hipGraph mode ------------------------------------------------------------------------------------------
    # One training step is ~800 kernel launches (1 900 before the fused operators), most of them microseconds long; eager
    # PyTorch needs ~13 us of host time per launch, which caps the step at ~25 ms whatever the GPU does.  The
    # step (zero-grad, forward, loss, backward, Adam) is therefore captured once into a hipGraph over static
    # input buffers and replayed; per step the host then only copies the batch (and the prefetched sampling
    # pyramid) into the static buffers, launches the graph, and starts the next pyramid on the side stream.
    # With more than one rank the gradient all-reduce and Adam stay outside the graph (RCCL calls are not
    # captured); single-rank runs capture them too.
    def enable_graph(self, example, warmup=3):
        """Capture the step for batches shaped like ``example`` (which must carry a prefetched pyramid if
        prefetching is used).  Falls back to eager mode (and records why) if capture fails.
        NOTE: the ``warmup`` iterations are real optimizer steps on ``example`` (see _capture); pass ``warmup=0`` after
        at least one eager step() if the trajectory must not contain them."""
        import torch.distributed as dist
        dev = example["point_clouds"].device
        if dev.type != "cuda":
            return False
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self._graph_with_opt = (world == 1) and not self.split_optimizer
        try:
            if self.bucket is None:
                self._setup({k: v for k, v in example.items() if k != "_fps_prefetch"})
            ex = self._consume_prefetch(dict(example))
            static = {}
            for k, v in ex.items():
                if k == "fps_pyramid":
                    static[k] = [t.clone() for t in v]
                elif torch.is_tensor(v):
                    static[k] = v.clone()
            # Autograd's AccumulateGrad nodes remember the stream they were created on; the warm-up below and the
            # capture must therefore share ONE side stream, and nothing may keep an older graph (created on the
            # default stream) alive -- the attention modules stash graph-attached tensors (`attn`, `value`).
            for mod in self.model.modules():
                if hasattr(mod, "attn") and hasattr(mod, "keep_value"):
                    mod.attn, mod.value = None, None
            self._capture(static, warmup)
            return True
        except Exception as e:  # noqa: BLE001 -- any capture failure means "stay eager"
            self.graph, self._static, self._static_loss = None, None, None
            import traceback
            self.graph_error = f"{type(e).__name__}: {str(e)[:300]}\n" + "".join(traceback.format_tb(e.__traceback__)[-6:])
            torch.cuda.synchronize(dev)
            return False

    # CUs the relation head's grids leave to the decoder running beside it (below 56 the two chains serialise again: measured)
    RELATION_LEAVE_CUS = int(os.environ.get("SPACAP_RELATION_LEAVE_CUS", "64"))

    def _fork_relation(self, dev, armed=None):
        """Inside this Trainer's steps the relation head (forward 0.16 ms, backward 0.46 ms of persistent workgroups) and the caption
        decoder (two chains of ~50 latency-bound launches on a few CUs) run on two streams: see
        TransformerDecoderModel.fork_relation.  Not with the overlapped gradient exchange (its boundary hook assumes one stream
        has produced every captioner gradient)."""
        from ._native import check, lib
        from .transformer_captioner import TransformerDecoderModel
        on = self.fork_relation and self.use_relation and not (self.overlap_allreduce if armed is None else armed)
        self._fork_on = on
        for mod in self.model.modules():
            if isinstance(mod, TransformerDecoderModel):
                mod.fork_relation = on
        check(lib.spacap_relation_fused_leave_cus(self.RELATION_LEAVE_CUS if on else 0), "spacap_relation_fused_leave_cus")

    def _capture(self, static, warmup):
        """``warmup`` REAL training steps on the static batch (they update the parameters, the BatchNorm statistics
        and the optimizer state exactly like step() would: callers that count steps must count them), then the
        capture of one more step, which is not executed until the first replay."""
        dev = static["point_clouds"].device
        if self._capture_stream is None:
            self._capture_stream = _role_stream(dev, "capture")
        s = self._capture_stream
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._core(static, self._graph_with_opt)
                if not self._graph_with_opt:
                    self._optimizer_step(None)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            loss = self._core(static, self._graph_with_opt)
        self.graph, self._static, self._static_loss = g, static, loss
        # the tensors autograd assigned as gradients during capture: every replay rewrites them in place
        self._graph_grads = [p.grad for p in self.bucket.params]
        self._recapture = False

    def set_hyper(self, lr=None, bn_momentum=None):
        """Change the learning rate and / or the BatchNorm momentum (what the reference's Solver does every epoch with
        StepLR / BNMomentumScheduler, lib/solver.py:228-235).  These scalars are kernel ARGUMENTS, i.e. frozen into a
        captured hipGraph: when a graph is active it is re-captured (same static buffers, no warm-up steps) at the
        next step."""
        if lr is not None:
            self.lr = float(lr)
            if isinstance(self.optimizer, torch.optim.Optimizer):
                for grp in self.optimizer.param_groups:
                    grp["lr"] = self.lr
            elif self.optimizer is not None:
                self.optimizer.lr = self.lr
        if bn_momentum is not None:
            for m in self.model.modules():
                if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                    m.momentum = float(bn_momentum)
        if self.graph is not None:
            self._recapture = True

    def _graph_step(self, data_dict, next_data):
        if self._recapture:
            static, self.graph = self._static, None
            for mod in self.model.modules():
                if hasattr(mod, "attn") and hasattr(mod, "keep_value"):
                    mod.attn, mod.value = None, None
            try:
                self._capture(static, warmup=0)
            except Exception as e:  # noqa: BLE001 -- as in enable_graph: a failed capture means "stay eager", not a dead Trainer
                self.graph, self._static, self._static_loss, self._recapture = None, None, None, False
                import traceback
                self.graph_error = f"{type(e).__name__}: {str(e)[:300]}\n" + "".join(traceback.format_tb(e.__traceback__)[-6:])
                torch.cuda.synchronize(data_dict["point_clouds"].device)
                return self.step(data_dict, next_data)
        pre = data_dict.pop("_fps_prefetch", None)
        static_pyr = self._static.get("fps_pyramid")
        dsts, srcs = [], []
        if static_pyr is not None:
            # The captured graph READS the pyramid (sampling / grouping / interpolation indices) from static buffers
            # and never computes it.  A batch that arrives without a prefetched pyramid (first batch after
            # enable_graph, epoch boundary, caller without next_data) gets one computed here, in line, on the
            # current stream -- never the previous batch's indices.
            if pre is not None:
                pyr, ev = pre
                torch.cuda.current_stream(pyr[0].device).wait_event(ev)
            else:
                with torch.no_grad():
                    fn = geometry_pyramid if len(static_pyr) > 4 else sampling_pyramid
                    pyr = fn(data_dict["point_clouds"][..., :3].contiguous())
            if len(pyr) != len(static_pyr):
                raise RuntimeError(f"prefetched pyramid has {len(pyr)} tensors, the captured graph expects "
                                   f"{len(static_pyr)} (prefetch_geometry changed after enable_graph?)")
            dsts, srcs = list(static_pyr), list(pyr)
            for src in pyr:
                src.record_stream(torch.cuda.current_stream(src.device))
        elif pre is not None:
            # graph captured WITHOUT a pyramid: it samples / groups inside the replay; the prefetched one is not needed
            torch.cuda.current_stream(pre[0][0].device).wait_event(pre[1])
        for k, dst in self._static.items():
            if k != "fps_pyramid" and k in data_dict and data_dict[k] is not dst:
                dsts.append(dst)
                srcs.append(data_dict[k])
        if dsts:   # the batch and its pyramid into the graph's static buffers: ONE launch (the library's multi-tensor copy
            # needs one ~20 us launch per dtype); tensors that differ in dtype / layout from their buffer go the library's way
            same = [(d, s_) for d, s_ in zip(dsts, srcs) if d.dtype == s_.dtype and d.shape == s_.shape and s_.is_contiguous()
                    and d.is_contiguous()]
            rest = [(d, s_) for d, s_ in zip(dsts, srcs) if not (d.dtype == s_.dtype and d.shape == s_.shape and s_.is_contiguous()
                                                                and d.is_contiguous())]
            if same:
                from ._native import copy_batched
                copy_batched([d for d, _ in same], [s_ for _, s_ in same])
            groups = {}
            for dst, src in rest:
                g = groups.setdefault((dst.dtype, src.dtype), ([], []))
                g[0].append(dst)
                g[1].append(src)
            for gd, gs in groups.values():
                torch._foreach_copy_(gd, gs, non_blocking=True)
        if next_data is not None:
            self.prefetch(next_data)
        self.graph.replay()
        if not self._graph_with_opt:
            self._optimizer_step(self._graph_grads)
        return self._static_loss

    def step(self, data_dict, next_data=None):
        """One full training step; returns the (device) loss tensor, no host sync.  ``next_data``: the batch of
        the following step, whose sampling pyramid is started on the side stream first."""
        self.check_health(wait=False)
        try:
            if self.graph is not None:
                return self._graph_step(data_dict, next_data)
            if self.bucket is None:
                self._setup({k: v for k, v in data_dict.items() if k != "_fps_prefetch"})
            data_dict = self._consume_prefetch(data_dict)  # pops what the previous step prefetched for this batch
            if next_data is not None:
                self.prefetch(next_data)                   # may be the same dict object: order matters
            return self._core(data_dict)
        except BaseException:
            self._sig_resync = self._sig is not None   # the device-side step counter may be ahead of its host mirror now
            raise


class Evaluator:
    """The inference forward (``model(data, is_eval=True)``: detector, encoder once, greedy decoding with key / value caches;
    models/SpaCapNet.py:47-85 with models/transformer_captioner.py:402-453) for a STREAM of batches: the sampling / grouping
    pyramid of the next batch depends on its coordinates only and is computed on a side stream while the current batch
    decodes -- what ``Trainer.prefetch`` does for training steps.  In a single forward the 4.5 ms sampling chain (2 047 + 1 023 +
    511 + 255 dependent rounds on one workgroup per scene) is on the critical path; with a batch in flight it is not.  Values
    are those of the plain forward (same indices, same kernels)."""

    def __init__(self, model, graph=True):
        self.model = model
        self.side_stream = None
        self.graph = graph          # the pyramid as ONE graph launch per batch (static input / outputs) instead of ~60 eager launches
        self._g = self._g_key = self._g_in = self._g_out = None

    def prefetch(self, next_data):
        pc = next_data["point_clouds"]
        if not pc.is_cuda:
            return
        if self.side_stream is None:
            self.side_stream = _role_stream(pc.device, "side")
        cur = torch.cuda.current_stream(pc.device)
        self.side_stream.wait_stream(cur)
        if self.graph:
            key = tuple(pc.shape)
            if self._g_key != key:
                with torch.cuda.stream(self.side_stream), torch.no_grad():
                    self._g_in = pc[..., :3].contiguous().clone()
                    for _ in range(2):
                        geometry_pyramid(self._g_in)
                self.side_stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(g, stream=self.side_stream):
                    self._g_out = geometry_pyramid(self._g_in)
                self._g, self._g_key = g, key
                self.side_stream.wait_stream(cur)
            with torch.cuda.stream(self.side_stream), torch.no_grad():
                self._g_in.copy_(pc[..., :3], non_blocking=True)
                self._g.replay()
                # (the graph's static outputs are overwritten by the next replay, which may start before the forward that consumes
                # this pyramid has finished: hand out a copy -- one batched launch)
                from ._native import copy_batched
                pyr = [torch.empty_like(t) for t in self._g_out]
                copy_batched(pyr, list(self._g_out))
                ev = torch.cuda.Event()
                ev.record(self.side_stream)
            next_data["_fps_prefetch"] = (pyr, ev)
            return
        with torch.cuda.stream(self.side_stream), torch.no_grad():
            pyr = geometry_pyramid(pc[..., :3].contiguous())
            ev = torch.cuda.Event()
            ev.record(self.side_stream)
        next_data["_fps_prefetch"] = (pyr, ev)

    @torch.no_grad()
    def __call__(self, data_dict, next_data=None):
        d = Trainer._consume_prefetch(dict(data_dict))
        data_dict.pop("_fps_prefetch", None)
        if next_data is not None:
            self.prefetch(next_data)
        return self.model(d, is_eval=True)


def synthetic_batch(batch: int, n_points: int, device, seed: int = 0, vocab: int = 3001, use_color=False,
                    use_normal=False, use_multiview=False, use_height=True):
    d = {"point_clouds": S.scene_batch(batch, n_points, use_color=use_color, use_normal=use_normal,
                                       use_multiview=use_multiview, use_height=use_height, seed=seed)}
    d.update(S.labels(batch, n_points, vocab=vocab, seed=seed))
    return {k: v.to(device) for k, v in d.items()}
