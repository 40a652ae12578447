"""Training engine on the GPU: the hipGraph replay path and the side-stream sampling prefetch must compute what
the plain eager step computes (same losses step by step), and two eager runs must agree (no races)."""
import copy
import os

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
    tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-6, split_optimizer=(mode == "graph-split"))
    data = synthetic_batch(2, 4096, DEV, seed=3, vocab=200)
    nxt = data if mode in ("prefetch", "graph", "graph-split") else None
    losses = [float(tr.step(data, next_data=nxt))]
    if mode in ("graph", "graph-split"):
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
    assert all(x == x and abs(x) < 1e5 for x in a)
    for name, other in (("eager-again", b), ("prefetch", c), ("graph", g), ("graph-split", gs)):
        for i, (x, y) in enumerate(zip(a, other)):
            if y is None:
                continue
            assert abs(x - y) <= 2e-3 * abs(x) + 1e-4, (name, i, a, other)


def test_forked_placement_of_the_relation_head_and_of_the_weight_gradients_changes_no_value():
    """Round 6: inside a Trainer's steps the relation head runs on a stream of its own beside the caption decoder
    (models/transformer_captioner.py:392-398 against :193-225: independent until the losses are added) and the captioner's
    queued weight gradients are flushed beside the detector's backward (engine.Trainer._mid_flush_hook).  Placement only: with
    both switched off the same kernels run in one chain.  Losses step by step and the parameters after the run must agree to
    rounding (the head's persistent grids leave CUs to the decoder, which regroups its partial sums: not bit for bit)."""
    from spacap3d_amd.engine import Trainer, synthetic_batch
    data = synthetic_batch(2, 4096, DEV, seed=3, vocab=200)
    runs = []
    for forked in (True, False):
        model = _make()
        tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-6)
        tr.fork_relation = tr.flush_mid = forked
        tr._fork_relation(torch.device(DEV))
        losses = [float(tr.step(data, next_data=data))]
        assert tr.enable_graph(data, warmup=1), tr.graph_error
        for _ in range(3):
            losses.append(float(tr.step(data, next_data=data)))
        torch.cuda.synchronize()
        from spacap3d_amd.transformer_captioner import TransformerDecoderModel
        assert all(m.fork_relation == forked for m in model.modules() if isinstance(m, TransformerDecoderModel))
        runs.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()}))
    (la, pa), (lb, pb) = runs
    for i, (x, y) in enumerate(zip(la, lb)):
        assert abs(x - y) <= 1e-5 * abs(x) + 1e-6, (i, la, lb)
    for n in pa:   # (Adam normalises the step: a parameter that started at zero moves by ~lr per step whatever its gradient's size,
        # so rounding-level gradient differences show at the scale of lr; bounded by the distance the steps can cover)
        d = float((pa[n] - pb[n]).abs().max())
        assert d <= 1.2e-5 + 1e-5 * float(pb[n].abs().max()), (n, d)    # 2 lr per step (a sign flip of a near-zero gradient), 5 steps


def test_graph_step_never_reuses_the_previous_batch_geometry():
    """A graph captured WITH a prefetched pyramid reads sampling / grouping indices from static buffers.  Feeding it a
    NEW batch that carries no prefetch (epoch boundary, caller without next_data) must compute that batch's pyramid in
    line -- not replay the previous batch's indices on the new point cloud."""
    from spacap3d_amd.engine import Trainer, synthetic_batch

    def run(graph):
        model = _make()
        tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-6)
        a = synthetic_batch(2, 4096, DEV, seed=3, vocab=200)
        b = synthetic_batch(2, 4096, DEV, seed=4, vocab=200)
        tr.step(a, next_data=a)
        if graph:
            assert tr.enable_graph(a, warmup=2), tr.graph_error
        else:
            tr.step(a), tr.step(a)
        la = float(tr.step(dict(a)))              # no prefetch attached
        lb = float(tr.step(dict(b)))              # different scenes, no prefetch attached
        lb2 = float(tr.step(dict(b), next_data=b))
        lb3 = float(tr.step(b))                   # consumes the prefetch
        return la, lb, lb2, lb3

    e, g = run(False), run(True)
    assert abs(e[0] - e[1]) > 1e-3 * abs(e[0]), "the two batches must differ for this test to mean anything"
    for x, y in zip(e, g):
        assert abs(x - y) <= 2e-3 * abs(x) + 1e-4, (e, g)


def test_hyperparameter_change_reaches_a_captured_graph():
    """lr and the BatchNorm momentum are kernel arguments, frozen into the hipGraph: Trainer.set_hyper re-captures."""
    from spacap3d_amd.engine import Trainer, synthetic_batch
    model = _make()
    tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-3)
    data = synthetic_batch(2, 4096, DEV, seed=3, vocab=200)
    tr.step(data, next_data=data)
    assert tr.enable_graph(data, warmup=1), tr.graph_error
    tr.step(data, next_data=data)
    w = model.backbone_net.sa2.mlp_module.layer1.conv.weight
    bn = model.backbone_net.sa2.mlp_module.layer1.bn.bn
    before, rm = w.detach().clone(), bn.running_mean.clone()
    tr.step(data, next_data=data)
    assert not torch.equal(w, before) and not torch.equal(bn.running_mean, rm)
    tr.set_hyper(lr=0.0, bn_momentum=0.0)
    before, rm = w.detach().clone(), bn.running_mean.clone()
    l1 = float(tr.step(data, next_data=data))
    l2 = float(tr.step(data, next_data=data))
    assert torch.equal(w, before), "lr = 0 must freeze the parameters (weight decay scales with lr too)"
    assert torch.equal(bn.running_mean, rm), "momentum = 0 must freeze the running statistics"
    assert l1 == l1 and abs(l1 - l2) <= 1e-3 * abs(l1)
    tr.set_hyper(lr=1e-3)
    tr.step(data, next_data=data)
    assert not torch.equal(w, before)


def _anchored_batch():
    """cfg1-sized batch whose GT boxes sit on the initial model's proposals (S.anchor_boxes_on_proposals): a stable set
    of positive proposals, so the loss is a smooth function of the step instead of switching terms on and off."""
    from oracle.attention_ref import OracleBackend
    from spacap3d_amd import backend
    from spacap3d_amd.engine import synthetic_batch
    from spacap3d_amd.spacapnet import build_default
    data = synthetic_batch(2, 4096, "cpu", seed=3, vocab=200)
    with backend.use_backend(OracleBackend()), torch.no_grad():
        torch.manual_seed(0)
        probe = build_default(vocab_size=200, num_proposal=64, N=2, d_ff=256).train()(dict(data))
    data = S.anchor_boxes_on_proposals(data, probe["aggregated_vote_xyz"])
    # the votes that become proposals are pinned to the initial model's choice: FPS of the predicted vote positions is
    # a chaotic discrete function of the weights (detector.ProposalModule.forward), which would turn 1e-6 of fp32
    # summation-order noise into different proposals within two optimizer steps
    data["proposal_inds"] = probe["aggregated_vote_inds"].clone()
    return data


def _fresh_model(device):
    from spacap3d_amd.spacapnet import build_default
    torch.manual_seed(0)
    model = build_default(vocab_size=200, num_proposal=64, N=2, d_ff=256)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model.to(device).train()


def test_five_step_trajectory_matches_the_oracle_backend():
    """The same Trainer code, the same initial weights and batch: 5 optimizer steps on the HIP product path (fused
    kernels, FlatAdam) against 5 steps on the CPU checker (oracle ops + torch CPU + torch.optim.Adam), dropout off,
    lr 1e-4, proposal sampling indices pinned (see _anchored_batch).  Per-step losses within 1e-3 relative: gradients,
    Adam and the BatchNorm statistics all feed the next step's loss, so this pins the whole update, not just one forward.
    Adam's eps is 1e-3 here: with the default 1e-8 the first steps are sign(g) updates, which turn every gradient entry
    that is fp32 noise around zero (biases in front of a BatchNorm, key biases, ...) into a full +-lr step of random
    sign -- 9.3 M coordinated +-1e-4 steps put the loss in its quadratic regime, where those random signs show up at the
    1e-2 level (measured: 25.31 vs 25.53 after one step).  eps = 1e-3 scales such entries by |g| / eps instead."""
    from oracle.attention_ref import OracleBackend
    from spacap3d_amd import backend
    from spacap3d_amd.engine import Trainer
    data = _anchored_batch()

    def run(device, be, steps=5, jitter=0.0):
        with backend.use_backend(be):
            model = _fresh_model(device)
            if jitter:   # the same weights up to a relative perturbation of fp32-rounding size
                g = torch.Generator().manual_seed(7)
                with torch.no_grad():
                    for p in model.parameters():
                        p.mul_(1.0 + jitter * torch.randn(p.shape, generator=g).to(p.device))
            tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-4, adam_eps=1e-3)
            d = {k: v.to(device) for k, v in data.items()}
            out = []
            for _ in range(steps):
                tr.step(d)
                out.append({k: float(v) for k, v in tr.last_losses.items()})
            return out

    cpu = run("cpu", OracleBackend())
    gpu = run(DEV, backend.HipBackend())
    # the trajectory's own noise floor: the HIP path again from weights perturbed by 1e-7 relative (less than one fp32
    # rounding of a summation order).  Whatever two such runs differ by is not a property of any kernel: the band of a term
    # at a step is never tighter than three times that difference, so that a re-association of a sum (a different but equally
    # correct kernel) cannot fail this gate while a wrong gradient (errors of 1e-3 and up, growing step by step) still does.
    gpu2 = run(DEV, backend.HipBackend(), jitter=1e-7)
    # Step 0 (same weights): every term within 1e-3.  Later steps: the two runs' forward values differ by ~2e-5 (fp32
    # through ~15 BatchNorm'd layers), so a ReLU whose input lies within that of zero resolves differently; at this size
    # the proposal head sees only 2 x 64 positions, and ONE such flip was measured to change the gradient entering the
    # backbone by 2.4 % (tools/lab/head_bisect.py: a single element of 16 384, every other channel agrees to 5e-5).  The
    # terms far from the flip (vote / caption / relation) are held to 6e-3; objectness (thresholded labels), box, class and
    # the total to 5e-2.
    for k in cpu[0]:
        assert abs(cpu[0][k] - gpu[0][k]) <= 1e-3 * max(abs(cpu[0][k]), 1e-2), (0, k, cpu[0], gpu[0])
    tight = ("vote_loss", "cap_loss", "relation_loss")
    for i, (a, b) in enumerate(zip(cpu, gpu)):
        for k in a:
            # (the runs drift apart step by step once a selection differs: the band doubles after the third step)
            # (round 6: 6e-3, was 5e-3.  The split-bf16 weight-gradient / convolution kernels of this round re-round sums at the
            # 1e-6 level; through the selections above that moved the caption term of step 2 from just inside to just outside the old
            # band -- 3.5318 against the checker's 3.5140 = 5.08e-3, where a 1e-7 jitter of the weights alone moves it by 7.5e-4
            # and the fp32-MFMA build of the same step gives 3.5324)
            tol = (6e-3 if k in tight else 5e-2) * (1 if i < 3 else 3)
            band = tol * max(abs(a[k]), 1e-2)
            floor = min(3.0 * abs(b[k] - gpu2[i][k]), 2.0 * band)   # the run-to-run spread may widen the band, never past 2x
            assert abs(a[k] - b[k]) <= max(band, floor), (i, k, a, b, gpu2[i])
    # everything goes down on both (anchored boxes, pinned proposals: a stable set of positives)
    for run_ in (cpu, gpu):
        assert run_[-1]["loss"] < run_[0]["loss"] and run_[-1]["cap_loss"] < run_[0]["cap_loss"] \
            and run_[-1]["vote_loss"] < run_[0]["vote_loss"], run_


def test_twenty_steps_reduce_the_loss():
    """Optimisation sanity at the reference's settings (Adam lr 1e-3, wd 1e-5, scripts/train.py:262) with hipGraph
    replay and the side-stream geometry prefetch on, on a fixed batch.  The caption and objectness terms must go down
    (CPU checker run: 3.64 -> 0.03 and 0.33 -> 0.07); the vote labels are noise and the box / class terms are means over
    the (changing) set of positive proposals, which switch on and off from step to step, so the total is only required
    to stay finite."""
    from spacap3d_amd.engine import Trainer
    data = {k: v.to(DEV) for k, v in _anchored_batch().items()}
    tr = Trainer(_fresh_model(DEV), S.mean_size_arr().numpy(), lr=1e-3)
    tr.step(data, next_data=data)
    first = {k: float(v) for k, v in tr.last_losses.items()}
    assert tr.enable_graph(data, warmup=1), tr.graph_error
    hist = []
    for _ in range(20):
        tr.step(data, next_data=data)
        hist.append({k: float(v) for k, v in tr.last_losses.items()})
    assert all(v == v and abs(v) < 1e4 for h in hist for v in h.values()), hist
    assert hist[-1]["cap_loss"] < 0.2 * first["cap_loss"], (first, hist[-1])
    assert hist[-1]["objectness_loss"] < 0.6 * first["objectness_loss"], (first, hist[-1])


def test_geometry_pyramid_equals_what_the_modules_compute_themselves():
    """detector.geometry_pyramid (sampling indices, ball-query groupings, interpolation neighbours: everything the
    trainer prefetches on the side stream) fed through the backbone must give bit-identical features to the backbone
    computing them in line; 9 000 points so that SA1 takes the cell-grid ball query."""
    from spacap3d_amd.detector import Pointnet2Backbone, geometry_pyramid, sampling_pyramid
    torch.manual_seed(0)
    net = Pointnet2Backbone(input_feature_dim=1).to(DEV).train()
    pc = S.scene_batch(2, 9000, seed=5).to(DEV)
    pyr = geometry_pyramid(pc[..., :3].contiguous())
    assert len(pyr) == 19 and pyr[15].shape == (2, 2048, 3) and pyr[4].shape == (2, 2048, 64) and pyr[11].shape == (2, 1024, 3) and pyr[12].dtype == torch.uint8
    outs = []
    for p in (None, sampling_pyramid(pc[..., :3].contiguous()), pyr):
        d = {"point_clouds": pc}
        if p is not None:
            d["fps_pyramid"] = p
        with torch.no_grad():
            out = net(d)
        outs.append((out["fp2_features"].clone(), out["sa1_inds"].clone(), out["sa4_features"].clone()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_gradient_slices_published_mid_backward_give_the_serial_tail():
    """The multi-rank tail overlapped with the backward (engine.Trainer._boundary / _overlapped_tail): at the captioner /
    detector boundary of the backward the step flushes the deferred weight gradients, packs the captioner's slice of the flat
    bucket and publishes the step number; the tail packs the detector's slice.  On one GPU (split_optimizer: the multi-rank code
    path without a process group): after every step -- eager and replayed -- the flat bucket must hold, bit for bit, the FINAL
    gradient of every parameter (a captioner gradient produced or changed after the boundary would leave a stale slice), and
    the losses follow the serial tail's (two separate runs are not bitwise repeatable: the tolerance of the other mode
    comparisons).  SURVEY.md section 8e; reference: one gradient exchange per step, scripts/train.py:198-200."""
    from spacap3d_amd.engine import Trainer, synthetic_batch
    losses = {}
    for overlap in (True, False):
        model = _make()
        tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-6, split_optimizer=True)
        tr.overlap_allreduce = overlap
        data = synthetic_batch(2, 4096, DEV, seed=3, vocab=200)
        finals = []
        if overlap:
            real = tr._overlapped_tail

            def spy(sources, real=real, tr=tr, finals=finals):
                src = sources if sources is not None else [p.grad for p in tr.bucket.params]
                finals.append([g.detach().clone() if g is not None else None for g in src])
                real(sources)
            tr._overlapped_tail = spy
        out = [float(tr.step(data, next_data=data)) for _ in range(2)]
        if overlap:
            assert len(finals) == 2 and tr.boundary_launches == 2
            for want, v in zip(finals[-1], tr.bucket.views):
                if want is not None:
                    assert torch.equal(want, v)
        assert tr.enable_graph(data, warmup=1), tr.graph_error
        out += [float(tr.step(data, next_data=data)) for _ in range(3)]
        torch.cuda.synchronize()
        if overlap:
            assert tr._cap_start is not None and 0 < tr._cap_start < len(tr.bucket.params)
            assert tr.boundary_launches == 4                      # two eager steps, the warm-up step, the capture
            nz = 0
            for want, v in zip(tr._graph_grads, tr.bucket.views):  # the replayed step's own gradient tensors
                assert torch.equal(want, v)
                nz += int(want.abs().max() > 0)
            assert nz > len(tr.bucket.views) // 2
        else:
            assert tr.boundary_launches == 0
        losses[overlap] = out
    for i, (x, y) in enumerate(zip(losses[True], losses[False])):
        assert abs(x - y) <= 2e-3 * abs(x) + 1e-4, (i, losses)


def test_timed_out_gradient_wait_raises_and_never_applies_the_update():
    """The overlapped exchange's stream wait (csrc/elementwise.hip: wait_ge_kernel) must not fall through silently: armed for a
    step number nobody will publish, it (a) leaves that number in the sticky error word, (b) turns the step's Adam off ON THE
    DEVICE -- parameters and moments bit-identical to before the step -- and (c) makes check_health() and the next step()
    raise.  One GPU, split_optimizer (the multi-rank code path without a process group) with the communication stream's wait
    forced on.  SURVEY.md section 8e; reference: one gradient exchange per step, scripts/train.py:198-200."""
    from spacap3d_amd.engine import Trainer, synthetic_batch
    model = _make()
    tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-3, split_optimizer=True)
    tr.overlap_allreduce, tr.force_comm_wait, tr.overlap_timeout_ms = True, True, 2000
    data = synthetic_batch(2, 4096, DEV, seed=3, vocab=200)
    for _ in range(2):                                   # healthy steps: the wait is satisfied by the boundary's signal
        tr.step(data, next_data=data)
    tr.check_health()
    assert tr.boundary_launches == 2 and int(tr._sig[2]) == 0
    before = [t.clone() for t in (tr.optimizer.flat_p, tr.optimizer.m, tr.optimizer.v)]
    tr._sig_host += 7                                    # the host now waits for a step number the device will not reach
    tr.overlap_timeout_ms = 50
    tr.step(data, next_data=data)                        # (returns: the failure is on the device, one step behind on the host)
    torch.cuda.synchronize()
    assert int(tr._sig[2]) == tr._sig_host               # sticky word = the value the wait gave up on
    for a, b in zip(before, (tr.optimizer.flat_p, tr.optimizer.m, tr.optimizer.v)):
        assert torch.equal(a, b)                         # the update was skipped, not applied to half-written gradients
    with pytest.raises(RuntimeError, match="timed out"):
        tr.check_health()
    with pytest.raises(RuntimeError, match="timed out"):
        tr.step(data, next_data=data)


def test_step_counter_survives_an_aborted_step():
    """An exception between the device-side increment of the step counter and the optimizer tail used to leave the host's
    mirror one behind, so the NEXT wait passed at once on the old flag value.  The mirror is re-read from the device after any
    failed step: the following steps wait for the right number (no timeout, updates applied)."""
    from spacap3d_amd.engine import Trainer, synthetic_batch
    model = _make()
    tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-3, split_optimizer=True)
    tr.overlap_allreduce, tr.force_comm_wait, tr.overlap_timeout_ms = True, True, 3000
    data = synthetic_batch(2, 4096, DEV, seed=3, vocab=200)
    tr.step(data, next_data=data)
    real = tr.loss

    def broken(d):
        real(d)
        raise ValueError("injected")
    tr.loss = broken
    with pytest.raises(ValueError):
        tr.step(data, next_data=data)
    tr.loss = real
    torch.cuda.synchronize()
    assert int(tr._sig[0]) == tr._sig_host + 1 and tr._sig_resync
    p0 = tr.optimizer.flat_p.clone()
    for _ in range(2):
        tr.step(data, next_data=data)
    tr.check_health()
    assert int(tr._sig[0]) == tr._sig_host == int(tr._sig[1]) and int(tr._sig[2]) == 0
    assert not torch.equal(p0, tr.optimizer.flat_p)


def test_overlapped_tail_zeroes_the_slices_of_parameters_without_a_gradient():
    """with autograd-assigned gradients the flat bucket is not cleared between steps; a used parameter that gets no
    gradient in some step must contribute zeros (as FlatGradBucket.pack writes them), not the previous step's reduced values."""
    from spacap3d_amd.engine import Trainer, synthetic_batch
    model = _make()
    tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-6, split_optimizer=True)
    tr.overlap_allreduce = True
    data = synthetic_batch(2, 4096, DEV, seed=3, vocab=200)
    tr.step(data, next_data=data)
    i0 = tr._cap_start
    victims = [0, i0 - 1, i0, len(tr.bucket.params) - 1]       # two on the detector's side of the bucket, two on the captioner's
    assert all(float(tr.bucket.views[i].abs().max()) > 0 for i in victims)
    real_tail, real_boundary = tr._overlapped_tail, tr._boundary

    def boundary():
        for i in victims[2:]:
            tr.bucket.params[i].grad = None
        real_boundary()

    def tail(sources):
        for i in victims[:2]:
            tr.bucket.params[i].grad = None
        real_tail(sources)
    tr._boundary, tr._overlapped_tail = boundary, tail
    tr.step(data, next_data=data)
    torch.cuda.synchronize()
    assert tr.boundary_launches == 2
    for i in victims:
        assert float(tr.bucket.views[i].abs().max()) == 0.0
    others = [v for i, v in enumerate(tr.bucket.views) if i not in victims]
    assert sum(int(v.abs().max() > 0) for v in others) > len(others) // 2


def test_bench_ends_within_seconds_when_one_rank_dies_at_start():
    """`python bench.py --gpus 2` (the driver's form) with rank 1 exiting before it joins the process group: the launcher polls
    all ranks, terminates rank 0 (which would sit in init_process_group for the store's 10 - 30 min timeout), prints every
    rank's last stderr lines and exits non-zero."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SPACAP_SHARE_GPU="1", SPACAP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SPACAP_BENCH_FAIL_RANK="1")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--batch", "2", "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    took = time.time() - t0
    assert out.returncode != 0
    assert "rank(s) failed first (rank, exit code): [(1, 1)]" in out.stderr, out.stderr[-2000:]
    assert "---- rank 0 (exit code -" in out.stderr, out.stderr[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert took < 30, took


def test_evaluator_stream_of_batches_gives_the_plain_forward():
    """engine.Evaluator: the next batch's sampling / grouping pyramid is computed on a side stream while the current batch
    decodes; captions, boxes and sampled indices of every batch equal the plain ``model(data, is_eval=True)`` forward
    (models/SpaCapNet.py:47-85, models/transformer_captioner.py:402-453)."""
    from spacap3d_amd.engine import Evaluator, synthetic_batch
    model = _make().eval()
    batches = [synthetic_batch(2, 4096, DEV, seed=s, vocab=200) for s in (1, 2, 3)]
    with torch.no_grad():
        want = [model(dict(b), is_eval=True) for b in batches]
    ev = Evaluator(model)
    got = []
    for i, b in enumerate(batches):
        got.append(ev(b, next_data=batches[i + 1] if i + 1 < len(batches) else None))
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        for k in ("sa1_inds", "sa2_inds", "aggregated_vote_inds", "lang_cap", "bbox_mask"):
            assert torch.equal(w[k], g[k]), k
        assert torch.equal(w["bbox_corner"], g["bbox_corner"])


def test_bench_runs_with_two_ranks_sharing_the_gpu():
    """The multi-rank path of bench.py end to end on a one-GPU box: two ranks launched exactly as the driver does
    (torch.distributed.run), both mapped onto cuda:0 and talking gloo instead of RCCL (test knobs SPACAP_SHARE_GPU /
    SPACAP_DIST_BACKEND).  Covers: scene sharding by rank, parameter broadcast, hipGraph of forward + backward with the
    gradient packing, flat all-reduce and Adam outside it, barrier + max-over-ranks timing, one JSON line from rank 0."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SPACAP_SHARE_GPU="1", SPACAP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--batch", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["final_loss"] == rec["final_loss"]
    assert abs(rec["value"] - 2 * 2 / (rec["ms_per_step"] * 1e-3)) < 1e-6 * rec["value"]   # whole-job scenes / s


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (the form the driver uses): bench.py starts two fresh rank processes
    itself before touching the GPU (reference: the one-command multi-GPU mode of scripts/train.py:198-200), rank 0 prints
    one line with n_gpus 2.  Both ranks share cuda:0 over gloo here (1-GPU box)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SPACAP_SHARE_GPU="1", SPACAP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "2",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["config"]["parallelism"] == "dp2"
    assert abs(rec["value"] - 2 * 2 / (rec["ms_per_step"] * 1e-3)) < 1e-6 * rec["value"]


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


@pytest.mark.parametrize("num_proposal", [256, 512, 40])
def test_fused_detection_losses_match_the_torch_composition(num_proposal):
    """csrc/losses.hip (3 launches forward, 1 backward) against the op-by-op composition of
    spacap3d_amd/loss_helper.py (= lib/loss_helper.py:35-197): eight loss values, integer labels bit-exact, gradients
    w.r.t. the proposal head output, the centres and the votes."""
    from spacap3d_amd import backend
    from spacap3d_amd.engine import synthetic_batch
    from spacap3d_amd.loss_helper import start_detection_losses
    from spacap3d_amd.spacapnet import build_default
    torch.manual_seed(1)
    model = build_default(vocab_size=200, num_proposal=num_proposal, N=1, d_ff=64).to(DEV).train()
    data = synthetic_batch(3, 8192, DEV, seed=5, vocab=200)
    hip = backend.ops()
    saved = hip.detection_losses
    msa = S.mean_size_arr().numpy()
    results = []
    for fused in (True, False):
        torch.manual_seed(2)
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        d = model.backbone_net(dict(data))
        xyz, feats = d["fp2_xyz"], d["fp2_features"]
        d["seed_inds"], d["seed_xyz"], d["seed_features"] = d["fp2_inds"], xyz, feats
        vx, vf = model.vgen(xyz, feats)
        vf = vf.div(torch.norm(vf, p=2, dim=1).unsqueeze(1))
        vx = vx.detach().requires_grad_(True)
        d["vote_xyz"], d["vote_features"] = vx, vf
        d = model.proposal(vx, vf.detach(), d)
        net = d["_proposal_net"].detach().requires_grad_(True)
        # rebuild the decoded views from the leaf so both paths differentiate w.r.t. the same tensors
        model.proposal.decode_scores(net.transpose(2, 1), d)
        cen = d["center"].detach().requires_grad_(True)
        d["center"] = cen
        net2 = d["_proposal_net"]
        try:
            hip.detection_losses = saved if fused else None
            start_detection_losses(d, S.NUM_HEADING_BIN, S.NUM_SIZE_CLUSTER, msa)
        finally:
            hip.detection_losses = saved
        t = d["_detection_losses"]
        losses = torch.stack([t[0], t[1], t[5], t[6], t[7], t[8], t[9], t[10]])
        w = torch.tensor([1.0, 0.5, 1.0, 0.1, 1.0, 0.1, 1.0, 0.1], device=DEV)
        (losses * w).sum().backward()
        gnet = net.grad if net.grad is not None else torch.zeros_like(net)
        results.append((losses.detach(), t[2], t[3], t[4], gnet.clone(), cen.grad.clone(), vx.grad.clone()))
    (la, laba, maska, oaa, gna, gca, gva), (lb, labb, maskb, oab, gnb, gcb, gvb) = results
    assert torch.equal(laba, labb) and torch.equal(maska, maskb)
    # padded ground-truth rows are identical, so a proposal nearest to one of them is tied between all of them: the
    # fused kernel takes the first, torch.min an unspecified one -- compare the assignment where it names a real box
    real_a = torch.gather(data["box_label_mask"], 1, oaa) > 0
    real_b = torch.gather(data["box_label_mask"], 1, oab) > 0
    assert torch.equal(real_a, real_b) and torch.equal(oaa[real_a], oab[real_b])
    assert 0 < int(laba.sum()) < laba.numel()
    assert torch.allclose(la, lb, rtol=2e-5, atol=1e-6), (la, lb)
    for name, a, b in (("dnet", gna, gnb), ("dcenter", gca, gcb), ("dvote", gva, gvb)):
        scale = float(b.abs().max())
        assert scale > 1e-6, name
        assert float((a - b).abs().max()) / scale < 2e-5, (name, float((a - b).abs().max()), scale)


def test_fused_relation_loss_matches_the_torch_composition():
    """csrc/losses.hip rel_* kernels against loss_helper.compute_relation_loss (= lib/loss_helper.py:240-289)."""
    from spacap3d_amd.fused_losses import relation_losses
    from spacap3d_amd.loss_helper import compute_relation_loss
    g = torch.Generator().manual_seed(4)
    B, K, M = 3, 64, 128
    d = {"relation_pred": (torch.randn(B, K, K, 9, generator=g) * 2).to(DEV),
         "object_assignment": torch.randint(0, 20, (B, K), generator=g).to(DEV),
         "objectness_label": torch.randint(0, 2, (B, K), generator=g).to(DEV),
         "box_label_mask_int": (torch.arange(M)[None, :] < torch.tensor([[12], [20], [5]])).long().to(DEV)}
    for a in "xyz":
        d[f"{a}_label"] = torch.randint(0, 3, (B, M, M), generator=g).to(DEV)
    d["relation_pred"][0, 0, 1, 0:3] = 1.5          # a three-way tie: argmax must pick the first class
    pa = d["relation_pred"].clone().requires_grad_(True)
    pb = d["relation_pred"].clone().requires_grad_(True)
    ra = relation_losses(dict(d, relation_pred=pa))
    rb = compute_relation_loss(dict(d, relation_pred=pb))
    w = {"x_loss": 0.3, "y_loss": 1.0, "z_loss": 2.0}
    sum(ra[k] * v for k, v in w.items()).backward()
    sum(rb[k] * v for k, v in w.items()).backward()
    for k in ("x_loss", "y_loss", "z_loss", "x_acc", "y_acc", "z_acc"):
        assert torch.allclose(ra[k], rb[k], rtol=1e-5, atol=1e-7), (k, float(ra[k]), float(rb[k]))
    assert float(pb.grad.abs().max()) > 0
    assert torch.allclose(pa.grad, pb.grad, rtol=1e-4, atol=1e-9)
    # no selected pair at all: losses 0, gradients 0 (n is clamped to 1)
    d0 = dict(d, objectness_label=torch.zeros_like(d["objectness_label"]), relation_pred=pa.detach().clone().requires_grad_(True))
    r0 = relation_losses(d0)
    assert float(r0["x_loss"]) == 0.0 and float(r0["z_acc"]) == 0.0


def test_gradient_slots_are_scoped_to_the_step_and_accumulation_outside_it_is_plain():
    """The [dW | db] slots of the flat bucket (_native.GRAD_SLOTS) are only active around the backward inside
    Trainer._core.  A backward run OUTSIDE it with .grad still set (manual accumulation) must see ordinary gradients: two
    identical accumulated passes give exactly twice one pass (a slot view aliasing .grad would give four times)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spacap3d_amd import _native
    from spacap3d_amd.engine import Trainer, synthetic_batch
    model = _make()
    tr = Trainer(model, S.mean_size_arr().numpy(), lr=1e-6)
    data = synthetic_batch(2, 4096, DEV, seed=3, vocab=200)
    tr.step(data)
    tr.step(data)
    assert _native.GRAD_SLOTS == {} and tr._grad_slots, "slot table must be inactive outside the step"
    some = next(iter(m for m in model.modules() if isinstance(m, torch.nn.Linear) and m.weight.data_ptr() in tr._grad_slots))
    assert _native.grad_slot(some.weight, some.weight.numel() + some.bias.numel()) is None
    # a second Trainer does not disturb the first one's table
    other = Trainer(_make(1), S.mean_size_arr().numpy(), lr=1e-6)
    other.step(data)
    assert tr._grad_slots and all(k in tr._grad_slots for k in [some.weight.data_ptr()])
    # manual accumulation outside the step (dropout is off and train-mode BatchNorm uses batch statistics: both passes
    # evaluate the same function)
    params = [p for p in tr.bucket.params]
    for p in params:
        p.grad = None
    tr.loss(dict(data))["loss"].backward()
    g1 = [p.grad.clone() for p in params]
    tr.loss(dict(data))["loss"].backward()      # accumulates into the existing .grad
    names = {id(p): n for n, p in model.named_parameters()}
    bad = []
    gmax = max(float(a.abs().max()) for a in g1)
    for p, a in zip(params, g1):   # (an aliased slot would give 4x: the bar only has to separate 2x from that; biases in
        err = float((p.grad - 2 * a).abs().max())   # front of a BatchNorm have an analytically zero gradient: noise only)
        if err > 1e-3 * max(float(2 * a.abs().max()), 1e-3 * gmax):
            bad.append((names[id(p)], err, float(a.abs().max()), float((p.grad / (a + 1e-30)).median())))
    assert not bad, bad[:8]


NO_LIBRARY_CASES = {
    # BASELINE.json configs at small batch / few points: widths (which decide the kernels) as in bench.py's CFG table
    "cfg2": dict(feats=dict(), model=dict(), points=8192),
    "cfg3": dict(feats=dict(use_color=True, use_normal=True), model=dict(), points=8192),
    "cfg4": dict(feats=dict(use_multiview=True, use_normal=True), model=dict(), points=8192),
    "cfg5": dict(feats=dict(), model=dict(d_model=512, h=32, num_proposal=512), points=8192),
}

@pytest.mark.parametrize("cfg", sorted(NO_LIBRARY_CASES))
def test_no_library_gemm_or_convolution_kernel_inside_a_training_step(cfg, monkeypatch):
    """Every dense product of the training step is one of this repository's kernels -- no rocBLAS (`Cijk_*`), MIOpen
    (`naive_conv*`, `miopen*`) or hipBLASLt kernel is launched between the start of a step and the end of its optimizer update
    (vote / proposal / FP / position nets: models/voting_module.py:28-61, models/proposal_module.py:46-54,
    lib/pointnet2/pointnet2_modules.py:376-421, models/transformer_captioner.py:149-164; vocabulary projection :93-100; the SA
    modules' first-layer feature product and its weight gradient at 1 / 7 / 132 input channels).  Checked on the kernel names of
    one eager step under torch.profiler, for the model of every BASELINE config."""
    from torch.profiler import ProfilerActivity, profile
    from spacap3d_amd.engine import Trainer, synthetic_batch
    from spacap3d_amd.spacapnet import build_default
    case = NO_LIBRARY_CASES[cfg]
    # single-chain placement (which kernels run does not depend on where they run): on some boxes the tracer returns NO kernel of
    # the forked relation branch for a whole process -- ten profiled steps in a row without `rel_wide_l1_bwd_kernel`, while the
    # head's weights kept moving -- and this test's presence checks then fail for a reason that is not its subject
    monkeypatch.setenv("SPACAP_FORK_RELATION", "0")
    monkeypatch.setenv("SPACAP_FLUSH_MID", "0")
    torch.manual_seed(0)
    kw = dict(vocab_size=3001, num_proposal=256, input_feature_dim=S.num_extra_channels(**case["feats"]))
    kw.update(case["model"])
    model = build_default(**kw).to(DEV).train()     # the benchmark's model (widths matter here)
    tr = Trainer(model, S.mean_size_arr().numpy())
    data = synthetic_batch(2, case["points"], DEV, seed=1, **case["feats"])
    for _ in range(2):
        tr.step(data, next_data=data)
    torch.cuda.synchronize()
    expect = ["conv1x1_cm_kernel", "dense_rows_kernel"] + (["dense_wgrad_tall_kernel"] if cfg in ("cfg3", "cfg4") else []) \
        + (["gemm_bf3_kernel", "linear_wgrad_bf3_kernel", "rel_wide_l1_bwd_kernel"] if cfg == "cfg5" else [])
    names = set()
    for attempt in range(10):   # (the tracer often returns an incomplete event list for a step -- seen: a forked branch's
        with profile(activities=[ProfilerActivity.CUDA]) as prof:   # kernels missing: the library check applies to every profiled
            tr.step(data, next_data=data)                           # step, the presence checks to their union)
            torch.cuda.synchronize()
        got = [e.key for e in prof.key_averages() if e.device_type is not None and "cuda" in str(e.device_type).lower()]
        bad = [n for n in got if n.startswith("Cijk_") or "naive_conv" in n or "miopen" in n.lower() or "hipblaslt" in n.lower()
               or "rocblas" in n.lower()]
        assert not bad, bad
        names |= set(got)
        if len(names) > 80 and all(any(e in n for n in names) for e in expect):
            break
    assert len(names) > 80, len(names)      # the profiler saw the step's kernels (distinct names: ~95 of ~320 launches)
    assert any("conv1x1_cm_kernel" in n for n in names) and any("dense_rows_kernel" in n for n in names)
    if cfg in ("cfg3", "cfg4"):
        assert any("dense_wgrad_tall_kernel" in n for n in names)
    if cfg == "cfg5":   # the 512-wide relation head and Linear layers: tiled split-bf16 products (csrc/gemm_bf3.hip)
        # (the wide head's weight gradient: the Linear layers' split-bf16 kernel since round 6, csrc/wgrad_bf3.inc)
        assert any("gemm_bf3_kernel" in n for n in names) and any("linear_wgrad_bf3_kernel" in n for n in names)
        if not any("rel_wide_l1_bwd_kernel" in n for n in names) and os.environ.get("SPACAP_TEST_DUMP"):
            open(os.environ["SPACAP_TEST_DUMP"], "w").write("\n".join(sorted(names)))
        assert any("rel_wide_l1_bwd_kernel" in n for n in names), sorted(n[:60] for n in names if "rel_" in n or "gemm_bf3" in n)
