"""Round-2 golden vectors (tests/golden/make_fixtures_r2.py: the reference's own Python run in the build container):

  * the reference's ``PointnetSAModuleVotes`` at the MLP shapes the FUSED shared-MLP kernels (csrc/sa_mlp.hip) cover,
    incl. the ``Y = F W1[:, 3:]`` first-layer path at 64 output channels with 7 and 132 input channels (BASELINE
    configs 3 and 4) -- on the GPU leg the test asserts that the fused path is the one that ran;
  * full training steps with input_feature_dim 7 (cfg3) and 132 (cfg4);
  * the captioner's non-default branches: --late_guide (cross-attention over the 1-token memory,
    models/transformer_captioner.py:223-224,266), --no_relation, --no_enc and the README's base model.

Legs as in test_golden.py: ``oracle`` (CPU, host glue + C oracle ops) and ``hip`` (-m gpu, the product path).
Float comparisons use two figures: max |err| / max |want| (``linf``) and ||err|| / ||want|| (``l2``).
"""
import os
import sys

import numpy as np
import pytest
import torch
from spacap3d_amd.layout import point_major_of

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from detweights import _uniform, fill_  # noqa: E402

from spacap3d_amd import backend, synthetic as S  # noqa: E402
from spacap3d_amd.loss_helper import get_scene_cap_loss  # noqa: E402
from spacap3d_amd.spacapnet import SpaCapNet  # noqa: E402

G = os.path.join(HERE, "golden")
LEGS = [pytest.param("oracle", id="oracle-cpu"), pytest.param("hip", id="hip-gpu", marks=pytest.mark.gpu)]


def _backend(kind):
    if kind == "oracle":
        from oracle.attention_ref import OracleBackend
        return OracleBackend(), "cpu"
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return backend.HipBackend(), "cuda:0"


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _errs(got, want):
    got, want = _np(got).astype(np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    d = got - want
    return float(np.abs(d).max() / (np.abs(want).max() + 1e-30)), float(np.linalg.norm(d) / (np.linalg.norm(want) + 1e-30))


def _check(got, want, name, linf, l2):
    a, b = _errs(got, want)
    assert a <= linf and b <= l2, f"{name}: linf {a:.3e} (allowed {linf:.1e}), l2 {b:.3e} (allowed {l2:.1e})"


def hash_tensor(shape, key, seed=0, scale=1.0, shift=0.0):
    n = int(np.prod(shape))
    return (torch.from_numpy(_uniform(n, key, seed)).view(*shape) * scale + shift).contiguous()


# ---- fused SA modules -------------------------------------------------------------------------------------------------
SA_CASES = {   # as in make_fixtures_r2.py
    "sa1_c1": (4096, 256, 0.3, 64, [1, 64, 64, 128], False),
    "sa1_c7": (4096, 256, 0.3, 64, [7, 64, 64, 128], True),
    "sa1_c132": (4096, 128, 0.3, 64, [132, 64, 64, 128], False),
    "sa2": (2048, 256, 0.4, 32, [128, 128, 128, 256], True),
    "sa3": (1024, 128, 0.8, 16, [256, 128, 128, 256], True),
    "agg": (1024, 64, 0.3, 16, [256, 128, 128, 128], True),
}
# (output, parameter gradients, input-feature gradient): linf / l2 bounds per leg.  CPU leg = same torch CPU kernels as
# the generator.  GPU leg, outputs: fp32 GEMMs with another summation order under train-mode BatchNorm.  GPU leg,
# gradients: a max-pool arg-max (or ReLU sign) whose two candidates are within fp32 noise selects differently than in
# the generator's run; ONE such flip among the ~G*C selections re-routes one of the ~G summands of a weight-gradient
# row (1/64 of that row at G = 4096, ~1e-3 of the matrix) -- measured on MI355X with tools/lab/sa1_c132_err.py: cases
# without a flip agree to 1e-6 (l2), cases with flips to 3e-5 .. 4e-3, for the fused kernels and for the per-operator
# torch path alike.  The bounds admit a few flips; tests/test_sa_mlp_gpu.py and tests/test_configs_gpu.py hold the
# flip-free float64 comparisons.
SA_TOL = {"cpu": ((2e-5, 2e-6), (2e-4, 2e-5), (2e-4, 2e-5)), "cuda:0": ((2e-4, 2e-5), (2e-2, 3e-3), (2e-2, 3e-3))}


@pytest.mark.parametrize("name", list(SA_CASES))
@pytest.mark.parametrize("kind", LEGS)
def test_sa_module_at_fused_shapes_matches_reference(kind, name):
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    be, device = _backend(kind)
    fx = np.load(os.path.join(G, "sa_modules_fused.npz"))
    N, npoint, radius, ns, mlp, need_grad = SA_CASES[name]
    xyz = S.scene_batch(2, N, use_height=False, seed=100 + len(name)).to(device)
    feats = hash_tensor((2, mlp[0], N), "feats_" + name, seed=4, scale=2.0).to(device)
    wout = hash_tensor((2, mlp[-1], npoint), "wout_" + name, seed=5, scale=2.0).to(device)
    with backend.use_backend(be):
        sa = PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=ns, mlp=list(mlp), use_xyz=True, normalize_xyz=True)
        fill_(sa, seed=7)
        sa = sa.to(device).train()
        feats.requires_grad_(need_grad)
        new_xyz, new_feats, inds = sa(xyz, feats)
        if kind == "hip":
            from spacap3d_amd import sa_mlp
            assert sa_mlp.supported(sa.mlp_module, ns), "this shape must have fused kernels"
            assert point_major_of(new_feats) is not None, "the fused shared-MLP path did not run"
        (new_feats * wout).sum().backward()
    (o_linf, o_l2), (g_linf, g_l2), (f_linf, f_l2) = SA_TOL[device]
    assert np.array_equal(_np(inds), fx[name + "_inds"])
    _check(_np(new_feats).reshape(2, -1)[:, ::3], fx[name + "_new_feats__flat3"], name + " new_feats", o_linf, o_l2)
    if need_grad:
        _check(_np(feats.grad).reshape(2, -1)[:, ::17], fx[name + "_feats_grad__flat17"], name + " feats_grad", f_linf, f_l2)
    for k, p in sa.named_parameters():
        g = _np(p.grad)
        g = g.reshape(-1)[::3] if g.size > 4096 else g
        _check(g, fx[name + "_grad_" + k], f"{name} grad {k}", g_linf, g_l2)
    for k, b in sa.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            _check(b, fx[name + "_buf_" + k], f"{name} {k}", 1e-4, 1e-5)


# ---- cfg3 / cfg4 training steps ---------------------------------------------------------------------------------------
def _cfg_point_clouds(C, B=2, N=4096, seed=11):
    base = S.scene_batch(B, N, seed=seed)
    extra = hash_tensor((B, N, C - 1), f"extra_channels_{C}", seed=seed, scale=1.0, shift=0.1)
    return torch.cat([base[..., :3], extra, base[..., 3:]], -1).contiguous()


def _model(fx_msa, device, C=1, **kw):
    args = dict(src_pos_type="xyz", use_transformer_encoder=True, early_guide=True, check_relation=True)
    args.update(kw)
    model = SpaCapNet(num_class=S.NUM_CLASS, vocabulary=S.make_vocabulary(40), num_heading_bin=1, num_size_cluster=18,
                      mean_size_arr=fx_msa, input_feature_dim=C, num_proposal=64, N=2, h=8, d_model=128, d_ff=128,
                      transformer_dropout=0.0, **args)
    fill_(model, seed=1)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return model.to(device)


CFG_TOL = {"cpu": (2e-4, 5e-5), "cuda:0": (2e-3, 3e-4)}


@pytest.mark.parametrize("tag,C", [("cfg3", 7), ("cfg4", 132)])
@pytest.mark.parametrize("kind", LEGS)
def test_train_step_with_extra_input_channels_matches_reference(kind, tag, C):
    be, device = _backend(kind)
    fx = np.load(os.path.join(G, f"train_step_{tag}.npz"))
    assert int(fx["cfg_C"]) == C
    pc = _cfg_point_clouds(C)
    assert np.array_equal(pc[..., :3].numpy(), fx["xyz"]), "synthetic scene generator changed: regenerate the fixtures"
    lab = S.labels(2, 4096, vocab=40, seed=11)
    lab["center_label"] = torch.from_numpy(fx["label_center_label"])
    lab["ref_center_label"] = torch.from_numpy(fx["label_ref_center_label"])
    d = {"point_clouds": pc.to(device)}
    d.update({k: v.to(device) for k, v in lab.items()})
    with backend.use_backend(be):
        model = _model(fx["mean_size_arr"], device, C=C).train()
        d = model(d)
        d = get_scene_cap_loss(d, use_relation=True, mean_size_arr=fx["mean_size_arr"])
        d["loss"].backward()
    linf, l2 = CFG_TOL[device]
    for k in fx.files:
        if not k.startswith("out_"):
            continue
        name = k[4:]
        flat = name.endswith("__flat7")
        name = name[:-7] if flat else name
        got = _np(d[name])
        got = got.reshape(got.shape[0], -1)[:, ::7] if flat else got
        if got.dtype.kind in "iub":
            assert np.array_equal(got, fx[k]), name
        else:
            _check(got, fx[k], f"{tag} {name}", linf, l2)
    for k in fx.files:
        if k.startswith("loss_"):
            got, want = float(d[k[5:]]), float(fx[k])
            assert abs(got - want) <= 5e-4 * max(1.0, abs(want)) * (1 if device == "cpu" else 4), (k, got, want)
    params = dict(model.named_parameters())
    for k in fx.files:
        if k.startswith("grad_"):
            g = _np(params[k[5:]].grad).reshape(-1)
            g = g[::3] if g.size > 4096 else g
            if device == "cpu":
                _check(g, fx[k], k, 2e-3, 5e-4)
            elif k.startswith("grad_caption."):
                _check(g, fx[k], k, 1e-2, 3e-3)
            # (detector gradients on the GPU leg: see test_detector_gradients_with_frozen_selections below and the
            # fused-SA fixture above, which pin them without the chaotic selection flips of a 30-layer network)


# ---- captioner variants ----------------------------------------------------------------------------------------------
VARIANTS = {
    "late_guide": dict(src_pos_type="xyz", use_transformer_encoder=True, early_guide=False, check_relation=True),
    "no_relation": dict(src_pos_type="xyz", use_transformer_encoder=True, early_guide=True, check_relation=False),
    "no_enc": dict(src_pos_type=None, use_transformer_encoder=False, early_guide=True, check_relation=False),
    "base": dict(src_pos_type=None, use_transformer_encoder=True, early_guide=False, check_relation=False),
}


@pytest.mark.parametrize("tag", list(VARIANTS))
@pytest.mark.parametrize("kind", LEGS)
def test_captioner_variants_match_reference(kind, tag):
    be, device = _backend(kind)
    base = np.load(os.path.join(G, "train_step_cfg1.npz"))
    fx = np.load(os.path.join(G, "captioner_variants.npz"))
    kw = VARIANTS[tag]

    def inputs():
        d = {"point_clouds": torch.from_numpy(base["point_clouds"]).to(device)}
        for k in base.files:
            if k.startswith("label_"):
                d[k[6:]] = torch.from_numpy(base[k]).to(device)
        return d

    with backend.use_backend(be):
        model = _model(base["mean_size_arr"], device, **kw).train()
        d = model(inputs())
        d = get_scene_cap_loss(d, use_relation=kw["check_relation"], mean_size_arr=base["mean_size_arr"])
        d["loss"].backward()
    linf, l2 = CFG_TOL[device]
    assert np.array_equal(_np(d["match_idx"]), fx[tag + "_match_idx"])
    _check(d["lang_cap"], fx[tag + "_lang_cap"], tag + " lang_cap", linf, l2)
    if kw["check_relation"]:
        _check(_np(d["relation_pred"]).reshape(2, -1)[:, ::7], fx[tag + "_relation_pred"], tag + " relation_pred", linf, l2)
    for k in ("loss", "cap_loss", "relation_loss", "det_loss", "cap_acc"):
        got, want = float(d[k]), float(fx[f"{tag}_loss_{k}"])
        assert abs(got - want) <= 5e-4 * max(1.0, abs(want)) * (1 if device == "cpu" else 4), (tag, k, got, want)
    params = dict(model.named_parameters())
    for k in fx.files:
        if k.startswith(tag + "_grad_") and not k.endswith("_grad_absent"):
            name = k[len(tag) + 6:]
            g = _np(params[name].grad).reshape(-1)
            g = g[::3] if g.size > 4096 else g
            if name.startswith("caption."):
                _check(g, fx[k], k, 2e-3 if device == "cpu" else 1e-2, 5e-4 if device == "cpu" else 3e-3)
            elif device == "cpu":
                _check(g, fx[k], k, 2e-3, 5e-4)
    absent = sorted(n for n, p in model.named_parameters() if p.grad is None)
    assert absent == list(fx[tag + "_grad_absent"])
    # greedy decoding (cached incremental path and the reference-style full recomputation)
    with backend.use_backend(be), torch.no_grad():
        model = _model(base["mean_size_arr"], device, **kw).eval()
        e = model(inputs(), is_eval=True)
        caps = _np(e["lang_cap"])
        assert caps.shape == fx[tag + "_eval_lang_cap"].shape
        assert (caps == fx[tag + "_eval_lang_cap"]).mean() > 0.999, tag
        e2 = model.caption.forward_eval(dict(e), use_cache=False)
        assert (_np(e2["lang_cap"]) == fx[tag + "_eval_lang_cap"]).mean() > 0.999, tag
