"""The host-side mirror (spacap3d_amd.*) against golden vectors produced by RUNNING THE REFERENCE'S PYTHON
(tests/golden/make_fixtures.py, generated in the build container; the reference itself never travels).

Two legs over the same assertions:
  * ``oracle`` (CPU, runs everywhere): host logic + CPU oracle ops  -> checks the port of the glue;
  * ``hip``    (-m gpu): host logic + HIP kernels through the C ABI  -> checks the product path.
Integer outputs (FPS / ball-query derived indices, assignments, greedy captions) must be identical.  Float
outputs: 2e-4 on the CPU leg (same torch CPU kernels as the fixture generator); 3e-4 of the tensor's scale (+ 4e-4
relative) on the GPU leg, where every GEMM of the ~30-layer network runs with a different fp32 summation order and
train-mode BatchNorm renormalises the differences (measured on MI355X: <= 1e-4 of scale, tools/lab/golden_err.py).
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from detweights import fill_  # noqa: E402

from spacap3d_amd import backend, synthetic as S  # noqa: E402
from spacap3d_amd.loss_helper import get_scene_cap_loss  # noqa: E402
from spacap3d_amd.spacapnet import SpaCapNet  # noqa: E402

G = os.path.join(HERE, "golden")
TOL = {"cpu": (2e-4, 2e-5), "cuda:0": (4e-4, 3e-4)}
_leg = {"device": "cpu"}


def _backend(kind):
    if kind == "oracle":
        from oracle.attention_ref import OracleBackend
        _leg["device"] = "cpu"
        return OracleBackend(), "cpu"
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _leg["device"] = "cuda:0"
    return backend.HipBackend(), "cuda:0"


LEGS = [pytest.param("oracle", id="oracle-cpu"), pytest.param("hip", id="hip-gpu", marks=pytest.mark.gpu)]


def _close(got, want, name, rtol=None, atol=None):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    r0, a0 = TOL[_leg["device"]]
    loosen = r0 / 2e-4  # explicit per-call tolerances are stated for the CPU leg and scaled for the GPU leg
    rtol = r0 if rtol is None else rtol * loosen
    atol = a0 if atol is None else atol * loosen
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol * scale, err_msg=name)


def _build(fx, device):
    V = int(fx["cfg_V"])
    model = SpaCapNet(num_class=S.NUM_CLASS, vocabulary=S.make_vocabulary(V), num_heading_bin=1,
                      num_size_cluster=18, mean_size_arr=fx["mean_size_arr"], input_feature_dim=1,
                      num_proposal=int(fx["cfg_P"]), N=int(fx["cfg_layers"]), h=8, d_model=128,
                      d_ff=int(fx["cfg_d_ff"]), transformer_dropout=0.0, src_pos_type="xyz",
                      use_transformer_encoder=True, early_guide=True, check_relation=True)
    fill_(model, seed=1)
    for mod in model.modules():  # as in make_fixtures.py: the attention dropout (default 0.1) is switched off
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return model.to(device)


def _inputs(fx, device):
    d = {"point_clouds": torch.from_numpy(fx["point_clouds"]).to(device)}
    for k in fx.files:
        if k.startswith("label_"):
            d[k[6:]] = torch.from_numpy(fx[k]).to(device)
    return d


@pytest.mark.parametrize("kind", LEGS)
def test_train_step_matches_reference(kind):
    be, device = _backend(kind)
    fx = np.load(os.path.join(G, "train_step_cfg1.npz"))
    with backend.use_backend(be):
        model = _build(fx, device).train()
        d = model(_inputs(fx, device))
        d = get_scene_cap_loss(d, use_relation=True, mean_size_arr=fx["mean_size_arr"])
        d["loss"].backward()
    # integer outputs: exact
    for k in ("sa1_inds", "sa2_inds", "fp2_inds", "aggregated_vote_inds", "match_idx", "object_assignment",
              "objectness_label", "bbox_mask"):
        got = d[k].detach().cpu().numpy()
        assert np.array_equal(got, fx["out_" + k]), k
    # float outputs
    for k in fx.files:
        if not k.startswith("out_"):
            continue
        name = k[4:]
        if name.endswith("__flat7"):
            name = name[:-7]
            got = d[name].detach().cpu().numpy()
            got = got.reshape(got.shape[0], -1)[:, ::7]
        else:
            got = d[name].detach().cpu().numpy()
        if got.dtype.kind in "iu" or got.dtype == np.bool_:
            continue
        _close(got, fx[k], name)
    for k in fx.files:
        if k.startswith("loss_"):
            _close(float(d[k[5:]]), fx[k], k, rtol=5e-4)
    enc_last = model.caption.model.encoder.layers[-1].self_attn
    _close(enc_last.attn[:, ::4], fx["attn_last_enc"], "attn_last_enc")
    _close(enc_last.value, fx["value_last_enc"], "value_last_enc")
    params = dict(model.named_parameters())
    for k in fx.files:
        if k.startswith("grad_") and k != "grad_absent":
            g = params[k[5:]].grad.detach().cpu().numpy().reshape(-1)
            g = g[::3] if g.size > 4096 else g
            if device == "cpu" or (k.startswith("grad_caption.") and "relation" not in k):
                _close(g, fx[k], k, rtol=2e-3, atol=2e-4)
            else:
                # Detector / relation-head gradients on the GPU leg.  What makes this fixture ill-conditioned is ONE
                # discrete step: the vote-aggregation FPS runs on network outputs (vote_xyz), and a 3e-6 relative
                # weight perturbation on the CPU leg alone already changes which votes become proposals (aggregated
                # features move by 100 % of their scale).  With the sampled indices identical -- asserted above --
                # the gradients agree to fp32 rounding noise amplified by the gates downstream (ReLU, the max-pool's arg-max
                # routing) and the BatchNorm backward's cancellations.  How large that noise is was measured by moving ONE
                # fp32 statistic of SA1's first BatchNorm by one unit in the last place (profiles/r04d_golden_sensitivity.txt,
                # tools/lab/golden_err.py): l2 of relation_proposal.0.weight 2.8e-3 -> 9.0e-3, of sa1 layer 0 1.5e-3 ->
                # 2.4e-3; with that layer's statistics in closed form (sa_mlp.L1_MOMENTS: as close to float64 as the summed
                # form, one ulp apart from it) sa1 layer 0 lands at 1.14e-2, fp1 layer 0 at 6.2e-3, the relation head at
                # 1.5e-3; linf <= 1.7e-2 throughout.  The bars are 1.5 x the largest of those realisations: what this
                # comparison can resolve.  The gate that HOLDS these gradients against the reference is
                # test_train_step_gradients_with_the_reference_selections below (the reference's own selections forced:
                # every detector gradient within 6e-4); this free-run bar only bounds what flipped near-ties can do.
                linf = np.abs(g - fx[k]).max() / (np.abs(fx[k]).max() + 1e-12)
                l2 = np.linalg.norm(g - fx[k]) / (np.linalg.norm(fx[k]) + 1e-12)
                assert linf < 2.6e-2 and l2 < 1.7e-2, (k, linf, l2)
    absent = sorted(n for n, p in model.named_parameters() if p.grad is None)
    assert absent == list(fx["grad_absent"])


# l2 bars of the forced-selection step.  Measured on MI355X (round 5): every one of the 73 detector gradients within 3.0e-4 of
# the reference with all gates pinned -> 6e-4: twice the measurement, so a 5x regression (1.5e-3) fails.
BAR_PINNED = 6e-4
# Default mode: SA1's first pre-activation is rebuilt, not stored, so that layer's ~3 300 listed near-tie gates decide freely (and
# its statistics come from the closed form, sa_mlp.L1_MOMENTS); everything else pinned.  Measured: worst 2.9e-3 (free run: 1.1e-2).
BAR_SA1_L0_FREE = 6e-3


class _ForceSelections:
    """selections.HOOK that overwrites, inside the HIP operators' forward, every near-tie ReLU gate and max-pool arg-max with
    what the reference's run chose (tests/golden/train_step_cfg1_selections.npz): the pre-activation of a gate that came out
    on the other side is moved to +-DELTA around the gate (a change of <= tau + DELTA in a unit-scale quantity, at a few
    thousand of millions of elements), the arg-max map takes the reference's sample index."""
    DELTA = 2e-5

    def __init__(self, model, sel):
        self.sel, self.files = sel, set(sel.files)
        self.names = {id(p): n[:-len(".weight")] for n, p in model.named_parameters() if n.endswith(".weight")}
        self.flipped, self.listed, self.rerouted, self.seen, self.unpinned = 0, 0, 0, set(), set()

    def _gate(self, z, elem, ch, mean, scale, shift, want_pos):
        """z.view(-1)[elem] belongs to channel ch; bn = (z - mean[ch]) * scale[ch] + shift[ch] must be > 0 iff want_pos."""
        zf = z.view(-1)
        cur = (zf[elem] - mean[ch]) * scale[ch] + shift[ch]
        bad = ((cur > 0) != want_pos) & (scale[ch] != 0)
        tgt = torch.where(want_pos, torch.full_like(cur, self.DELTA), torch.full_like(cur, -self.DELTA))
        zf[elem[bad]] = (mean[ch] + (tgt - shift[ch]) / scale[ch])[bad]
        self.flipped += int(bad.sum())
        self.listed += int(elem.numel())

    def __call__(self, kind, gammas, zs, stats, **kw):
        dev = zs[-1].device
        for k, (g, z, st) in enumerate(zip(gammas, zs, stats)):
            name = self.names[id(g)]
            if "near_" + name + "_idx" not in self.files:    # (the captioner's position head: not a detector layer)
                continue
            self.seen.add(name)
            idx = torch.from_numpy(self.sel["near_" + name + "_idx"].astype(np.int64)).to(dev)
            pos = torch.from_numpy(self.sel["near_" + name + "_pos"]).to(dev).bool()
            shp = [int(v) for v in self.sel["shape_" + name]]              # the reference's (B, C, P, S) / (B, C, L)
            C = shp[1]
            inner = int(np.prod(shp[2:]))
            b, c, rest = idx // (C * inner), (idx // inner) % C, idx % inner
            if z is None:        # SA1's first layer in the default mode: rebuilt from the rows' inputs, never stored -- its gates stay free
                self.unpinned.add(name)
                continue
            if kind == "sa":     # point-major rows (b, p, s) x C; stats rows (mean, 1/std, gamma/std, beta)
                assert z.shape == (shp[0] * inner, C), (name, z.shape, shp)
                self._gate(z, (b * inner + rest) * C + c, c, st[:, 0], st[:, 2], st[:, 3], pos)
            else:                # channel-major like the reference; stats rows (mean, 1/std)
                assert z.numel() == int(np.prod(shp)), (name, z.shape, shp)
                self._gate(z, idx, c, st[:, 0], st[:, 1] * g.detach(), kw["beta"].detach(), pos)
        if kind == "sa":
            B, N, S = kw["dims"]
            arg, out, zmax, z3, st3 = kw["arg"], kw["out"], kw["zmax"], zs[2], stats[2]
            mod = self.names[id(gammas[2])][:-len(".layer2.bn.bn")]
            pidx = torch.from_numpy(self.sel["pool_" + mod + "_idx"].astype(np.int64)).to(dev)
            parg = torch.from_numpy(self.sel["pool_" + mod + "_arg"]).to(dev)
            Bs, C3, P, S_ = [int(v) for v in self.sel["poolshape_" + mod]]
            assert (Bs, P, S_, C3) == (B, N, S, arg.shape[2])
            b, c, p = pidx // (C3 * P), (pidx // P) % C3, pidx % P
            self.rerouted += int((arg[b, p, c] != parg).sum())
            arg[b, p, c] = parg
            # every (group, channel) a gate or a route may have touched: output and arg-max pre-activation from the final z3 / arg
            rows = (torch.arange(B * N, device=dev).view(B, N, 1) * S + arg.long())          # (B, N, C3): row of z3
            zsel = torch.gather(z3.view(B * N * S, C3), 0, rows.view(-1, C3)).view(B, N, C3)
            new_out = torch.relu((zsel - st3[:, 0]) * st3[:, 2] + st3[:, 3])
            touched = torch.zeros(B, N, C3, dtype=torch.bool, device=dev)
            touched[b, p, c] = True
            nm = self.names[id(gammas[2])]
            i3 = torch.from_numpy(self.sel["near_" + nm + "_idx"].astype(np.int64)).to(dev)
            inner = P * S
            touched[i3 // (C3 * inner), (i3 % inner) // S, (i3 // inner) % C3] = True
            out[touched] = new_out[touched]
            if zmax is not None:
                zmax[touched] = zsel[touched]


@pytest.mark.gpu
@pytest.mark.parametrize("stored_z1", [True, False], ids=["all-gates-pinned", "default-sa1-layer0-free"])
def test_train_step_gradients_with_the_reference_selections(stored_z1):
    """The cfg1 golden training step on the HIP path with the REFERENCE'S discrete selections forced: the vote-aggregation
    sampling (`proposal_inds`), every near-tie ReLU gate and every near-tie max-pool route of the detector
    (tests/golden/train_step_cfg1_selections.npz, written by the reference's own run; spacap3d_amd/selections.py).  The step
    is then a smooth function of the weights and EVERY detector gradient is held against the reference's numbers at
    l2 <= 1e-3 -- the bar a free run cannot be given (a flipped near-tie re-routes a whole summand; see the free test's
    comment).  Reference: lib/pointnet2/pytorch_utils.py:11-36 (Conv -> BN -> ReLU), pointnet2_modules.py:256-259 (max-pool)."""
    from spacap3d_amd import sa_mlp, selections
    be, device = _backend("hip")
    fx = np.load(os.path.join(G, "train_step_cfg1.npz"))
    sel = np.load(os.path.join(G, "train_step_cfg1_selections.npz"))
    keep = sa_mlp.RECOMPUTE_Z1
    try:
        if stored_z1:
            sa_mlp.RECOMPUTE_Z1 = False  # SA1's first pre-activation stored (the rebuilt form is bit-identical: test_sa_mlp_gpu.py)
        with backend.use_backend(be):
            model = _build(fx, device).train()
            force = _ForceSelections(model, sel)
            selections.HOOK = force
            inp = _inputs(fx, device)
            inp["proposal_inds"] = torch.from_numpy(sel["aggregated_vote_inds"]).to(device)
            d = model(inp)
            selections.HOOK = None
            d = get_scene_cap_loss(d, use_relation=True, mean_size_arr=fx["mean_size_arr"])
            d["loss"].backward()
    finally:
        selections.HOOK, sa_mlp.RECOMPUTE_Z1 = None, keep
    want_layers = {k[5:-4] for k in sel.files if k.startswith("near_") and k.endswith("_idx")}
    assert force.seen == want_layers, sorted(want_layers ^ force.seen)           # every listed layer ran through a pinned operator
    # forcing must be a correction of near-ties, not a rewrite: a small share of the listed gates actually flipped
    assert force.flipped <= 0.05 * force.listed and force.rerouted <= 2000, (force.flipped, force.listed, force.rerouted)
    for k in ("sa1_inds", "sa2_inds", "aggregated_vote_inds", "match_idx", "object_assignment", "objectness_label", "bbox_mask"):
        assert np.array_equal(d[k].detach().cpu().numpy(), fx["out_" + k]), k
    params = dict(model.named_parameters())
    worst = {}
    top = max(float(np.linalg.norm(sel[k])) for k in sel.files if k.startswith("grad_"))
    for k in sel.files:
        if k.startswith("grad_"):
            name = k[5:]
            g = params[name].grad.detach().cpu().numpy().reshape(-1)[::int(sel["gradstep_" + name])]
            want = sel[k]
            if np.linalg.norm(want) < 1e-5 * top:
                # analytically zero (a convolution bias in front of a train-mode BatchNorm: vgen.conv1 / conv2): rounding noise
                # on both sides, compared on the scale of the real gradients
                assert np.linalg.norm(g) < 1e-5 * top, (name, float(np.linalg.norm(g)))
                continue
            l2 = np.linalg.norm(g - want) / (np.linalg.norm(want) + 1e-30)
            worst[name] = l2
    print("forced selections:", force.flipped, "of", force.listed, "gates flipped,", force.rerouted, "routes changed; worst l2:",
          sorted(((round(float(v), 6), n) for n, v in worst.items()), reverse=True)[:8])
    assert force.unpinned == (set() if stored_z1 else {"backbone_net.sa1.mlp_module.layer0.bn.bn"}), force.unpinned
    bar = BAR_PINNED if stored_z1 else BAR_SA1_L0_FREE
    bad = {n: v for n, v in worst.items() if not v < bar}
    assert not bad, (bad, force.flipped, force.rerouted)


@pytest.mark.parametrize("kind", LEGS)
def test_eval_greedy_decoding_matches_reference(kind):
    be, device = _backend(kind)
    fx = np.load(os.path.join(G, "train_step_cfg1.npz"))
    ev = np.load(os.path.join(G, "eval_greedy_cfg1.npz"))
    with backend.use_backend(be), torch.no_grad():
        model = _build(fx, device).eval()
        d = model(_inputs(fx, device), is_eval=True)
    _close(d["aggregated_vote_features"], ev["aggregated_vote_features"], "aggregated_vote_features")
    _close(d["bbox_corner"], ev["bbox_corner"], "bbox_corner")
    assert d["bbox_corner"].dtype == torch.float64
    assert np.array_equal(d["bbox_mask"].cpu().numpy(), ev["bbox_mask"])
    caps = d["lang_cap"].cpu().numpy()
    assert caps.shape == ev["lang_cap"].shape
    # greedy arg-max chains: identical unless two words tie to within float noise (none do on this fixture)
    assert (caps == ev["lang_cap"]).mean() > 0.999
    # the same captions must come out of the reference-style loop that recomputes the whole prefix at every step
    with backend.use_backend(be), torch.no_grad():
        d2 = model.caption.forward_eval(dict(d), use_cache=False)
    assert (d2["lang_cap"].cpu().numpy() == ev["lang_cap"]).mean() > 0.999


@pytest.mark.parametrize("kind", LEGS)
def test_sa_and_fp_modules_match_reference(kind):
    from spacap3d_amd.pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes
    be, device = _backend(kind)
    fx = np.load(os.path.join(G, "sa_fp_modules.npz"))
    with backend.use_backend(be):
        sa = PointnetSAModuleVotes(npoint=128, radius=0.4, nsample=16, mlp=[6, 16, 32], use_xyz=True,
                                   normalize_xyz=True)
        fill_(sa, seed=2)
        sa = sa.to(device).train()
        xyz = torch.from_numpy(fx["xyz"]).to(device)
        feats = torch.from_numpy(fx["feats"]).to(device).requires_grad_(True)
        new_xyz, new_feats, inds = sa(xyz, feats)
        new_feats.sum().backward()
        grouped, gxyz = sa.grouper(xyz, new_xyz, feats.detach())
        fp = PointnetFPModule(mlp=[32 + 6, 24])
        fill_(fp, seed=3)
        fp = fp.to(device).train()
        fp_out = fp(xyz, new_xyz, feats.detach(), new_feats.detach())
    assert np.array_equal(inds.cpu().numpy(), fx["inds"])
    assert np.array_equal(new_xyz.detach().cpu().numpy(), fx["new_xyz"])
    if device == "cpu":
        assert np.array_equal(gxyz.cpu().numpy(), fx["grouped_xyz"])   # exact: gather, subtract, divide
        assert np.array_equal(grouped.detach().cpu().numpy(), fx["grouped"])
    else:
        # torch's GPU kernel for `tensor / python_scalar` multiplies by the reciprocal (1 ulp from the CPU's
        # true division); the gather and the subtraction are exact
        np.testing.assert_allclose(gxyz.cpu().numpy(), fx["grouped_xyz"], rtol=3e-7, atol=0)
        np.testing.assert_allclose(grouped.detach().cpu().numpy(), fx["grouped"], rtol=3e-7, atol=0)
    _close(new_feats, fx["new_feats"], "new_feats")
    _close(feats.grad, fx["feats_grad"], "feats_grad", rtol=1e-3, atol=1e-4)
    _close(fp_out, fx["fp_out"], "fp_out")
