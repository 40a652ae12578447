"""Every BASELINE.json configuration exercised on the GPU at (or at the per-GPU slice of) its full size.

  cfg2  8 scenes x 40 000 points: FPS 40 000 -> 2 048 and the SA1 ball query, all 8 scenes, bit-exact vs the oracle
        (the per-operator tests use B = 2);
  cfg3 / cfg4  the full-size SA1 first layer with 7 / 132 input channels (``Y = F W1[:, 3:]`` + gather path) against
        a float64 restatement; their model-level golden vectors are in test_golden_r2.py;
  cfg5  80 000 points, 512 proposals, d_model 512 / h 32 with the documented 128 -> 512 token projection: index ops
        bit-exact, attention (h 32, L 512) vs the oracle, and one whole training step vs the CPU checker backend.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from spacap3d_amd.layout import point_major_of

from spacap3d_amd import backend, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def omp_ext():
    from oracle.ext_cpu import OracleExt
    return OracleExt(openmp=True)   # same source as the single-thread oracle, parallel over (scene, centre)


def _centres(xyz, inds):
    return torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()


def test_cfg2_all_eight_scenes_index_ops_bit_exact(hip_ext, omp_ext):
    xyz = S.scene_batch(8, 40000, use_height=False, seed=2024)
    want = omp_ext.furthest_point_sampling(xyz, 2048)
    got = hip_ext.furthest_point_sampling(xyz.to(DEV), 2048).cpu()
    assert torch.equal(got, want), f"first mismatch at {(got != want).nonzero()[0].tolist()}"
    new_xyz = _centres(xyz, want)
    bq_w = omp_ext.ball_query(new_xyz, xyz, 0.2, 64)
    bq_g = hip_ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), 0.2, 64).cpu()
    assert torch.equal(bq_g, bq_w)
    # the rest of the pyramid on the sampled centres (2048 -> 1024 -> 512 -> 256), all scenes
    cur = new_xyz
    for m, r, ns in ((1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16)):
        w = omp_ext.furthest_point_sampling(cur, m)
        g = hip_ext.furthest_point_sampling(cur.to(DEV), m).cpu()
        assert torch.equal(g, w), m
        nxt = _centres(cur, w)
        assert torch.equal(hip_ext.ball_query(nxt.to(DEV), cur.to(DEV), r, ns).cpu(), omp_ext.ball_query(nxt, cur, r, ns)), m
        cur = nxt


def test_cfg5_index_ops_bit_exact(hip_ext, omp_ext):
    xyz = S.scene_batch(2, 80000, use_height=False, seed=55)
    want = omp_ext.furthest_point_sampling(xyz, 2048)
    got = hip_ext.furthest_point_sampling(xyz.to(DEV), 2048).cpu()
    assert torch.equal(got, want), f"first mismatch at {(got != want).nonzero()[0].tolist()}"
    new_xyz = _centres(xyz, want)
    assert torch.equal(hip_ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), 0.2, 64).cpu(), omp_ext.ball_query(new_xyz, xyz, 0.2, 64))
    votes = S.scene_batch(2, 1024, use_height=False, seed=56)          # vote aggregation: 1 024 -> 512 proposals
    w = omp_ext.furthest_point_sampling(votes, 512)
    assert torch.equal(hip_ext.furthest_point_sampling(votes.to(DEV), 512).cpu(), w)
    c = _centres(votes, w)
    assert torch.equal(hip_ext.ball_query(c.to(DEV), votes.to(DEV), 0.3, 16).cpu(), omp_ext.ball_query(c, votes, 0.3, 16))


def test_cfg5_attention_h32_L512():
    from oracle.attention_ref import attention as ref_attention
    q, k, v = (t.view(2, 512, 32, 16).transpose(1, 2) for t in S.attention_inputs(2, 32, 512, 512, 16, seed=7))
    mask = (torch.rand(2, 1, 1, 512, generator=torch.Generator().manual_seed(1)) > 0.3).long()
    mask[..., 0] = 1
    o_ref, p_ref = ref_attention(q, k, v, mask=mask)
    logits_ref = (q @ k.transpose(-2, -1)) / 4.0
    hip = backend.HipBackend()
    o, p = hip.attention(q.to(DEV), k.to(DEV), v.to(DEV), mask=mask.to(DEV))
    assert float((o.cpu() - o_ref).abs().max()) < 1e-4 and float((p.cpu() - p_ref).abs().max()) < 1e-5
    # logits recovered from P relative to key 0 (never masked): north_star's 1e-3 bound on the attention logits
    sel = mask.bool().expand_as(p_ref) & (p_ref > 1e-4)
    lg = torch.log(p.cpu().clamp_min(1e-38))
    d = ((lg - lg[..., :1]) - (logits_ref - logits_ref[..., :1]))[sel]
    assert float(d.abs().max()) < 1e-3


def _fused_node(t):
    """The autograd node of the fused shared-MLP op behind a module output (its saved tensors hold the kernel's own
    pre-activations z1..z3, statistics and arg-max map)."""
    node, seen = t.grad_fn, 0
    while node is not None and "_SAMLP" not in type(node).__name__:
        node, seen = node.next_functions[0][0], seen + 1
        assert seen < 8
    return node


def _sa1_vs_float64(C, seed):
    """SA1 at the cfg3 / cfg4 size (40 000 points, C extra channels, 2 048 x 64 groups, 2 scenes) through the fused op vs
    float64 restatements of group -> [rel/r ; feats] -> conv1x1 -> BN(train) -> ReLU -> ... -> max:
      free    the restatement makes its own ReLU / max-pool selections;
      frozen  the restatement uses the selections the kernels made (ReLU masks recomputed from the kernel's stored
              pre-activations exactly as its backward does, arg-max map as stored): a smooth function of the weights,
              so the comparison is well conditioned.
    Returns (forward error, [free l2 errors of dW1..3], [frozen l2 errors of dW1..3])."""
    from spacap3d_amd import pointnet2_utils as pu
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(seed)
    pc = S.scene_batch(2, 40000, use_color=(C == 7), use_normal=True, use_multiview=(C == 132), seed=seed).to(DEV)
    assert pc.shape[-1] == 3 + C
    xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
    sa = PointnetSAModuleVotes(npoint=2048, radius=0.2, nsample=64, mlp=[C, 64, 64, 128], use_xyz=True,
                               normalize_xyz=True).to(DEV).train()
    new_xyz, out, inds = sa(xyz, feats)
    assert point_major_of(out) is not None, "the fused shared-MLP path did not run"
    node = _fused_node(out)
    saved = node.saved_tensors
    zs, sts, arg = saved[7:10], saved[10:13], saved[14]
    wsum = torch.randn(out.shape, device=DEV)
    (out * wsum).sum().backward()
    got = [l.conv.weight.grad.view(l.conv.out_channels, -1).double() for l in sa.mlp_module.children()]
    with torch.no_grad():
        idx = pu.ball_query(0.2, 64, xyz, new_xyz).long()                        # (B, P, S)
        B, P, Sn = idx.shape
        flat = idx.view(B, -1)
        # the kernels' selections: relu'(bn(z_k)) with the fp32 expression of the kernels, arg-max of the pooled layer
        # (the pooled last layer stores no pre-activation any more, only its value at the arg-max rows, saved[9] of shape
        # (B, P, C3): the frozen restatement gathers the arg-max rows, so that mask is all it needs of the last layer)
        masks = [(((z - st[:, 0]) * st[:, 2] + st[:, 3]) > 0).view(B, P, -1, z.shape[-1]).expand(B, P, Sn, -1)
                 for z, st in zip(zs, sts)]
        argl = arg.view(B, P, 1, -1).long()
    x64, f64 = xyz.double(), feats.double()
    g_xyz = torch.gather(x64, 1, flat.unsqueeze(-1).expand(-1, -1, 3)).view(B, P, Sn, 3)
    rel = (g_xyz - new_xyz.double().unsqueeze(2)) / 0.2
    g_f = torch.gather(f64, 2, flat.unsqueeze(1).expand(-1, C, -1)).view(B, C, P, Sn).permute(0, 2, 3, 1)
    h0 = torch.cat([rel, g_f], -1)                                                # (B, P, S, 3 + C)
    out_errs, grads = [], []
    for frozen in (False, True):
        ws = [l.conv.weight.detach().double().view(l.conv.out_channels, -1).requires_grad_(True) for l in sa.mlp_module.children()]
        h = h0
        for k, (w, l) in enumerate(zip(ws, sa.mlp_module.children())):
            z = h @ w.t()
            mu, var = z.mean((0, 1, 2)), z.var((0, 1, 2), unbiased=False)
            y = (z - mu) / torch.sqrt(var + l.bn.bn.eps) * l.bn.bn.weight.double() + l.bn.bn.bias.double()
            h = y * masks[k].double() if frozen else torch.relu(y)
        ref = (torch.gather(h, 2, argl).squeeze(2) if frozen else h.max(2).values).permute(0, 2, 1)   # (B, 128, P)
        (ref * wsum.double()).sum().backward()
        out_errs.append(float((out.double() - ref).abs().max() / ref.abs().max()))
        grads.append([float((g - w.grad).norm() / w.grad.norm()) for g, w in zip(got, ws)])
        del ws, h, ref
    return max(out_errs), grads[0], grads[1]


@pytest.mark.parametrize("C", [7, 132])
def test_cfg3_cfg4_sa1_full_size_first_layer(C):
    """Three seeds.  The forward agrees with float64 to 2e-5 of scale.  With the kernels' own ReLU / max-pool selections
    frozen into the float64 restatement the weight gradients must agree tightly (the well-conditioned comparison).
    With free selections a candidate pair within fp32 noise of a tie may resolve differently than in float64: one
    flipped arg-max re-routes one of 4 096 summands of a gradient row (~1e-3 of the matrix), and the per-operator torch
    path shows the same spread (tools/lab/sa1_c132_err.py) -- those errors are only bounded at the flip level."""
    res = [_sa1_vs_float64(C, seed) for seed in (C, C + 1000, C + 2000)]
    assert max(r[0] for r in res) < 2e-5, res
    for layer in range(3):
        assert max(r[2][layer] for r in res) < 5e-5, ("frozen selections", layer, [r[2] for r in res])
        assert max(r[1][layer] for r in res) < 2e-2, ("free selections", layer, [r[1] for r in res])


def _cfg5_model(device, layers=2):
    from spacap3d_amd.spacapnet import build_default
    torch.manual_seed(0)
    model = build_default(vocab_size=200, num_proposal=512, N=layers, h=32, d_model=512, d_ff=2048)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model.to(device).train()


def test_cfg5_training_step_matches_the_cpu_checker():
    """One scene of the stress configuration (80 000 points, 512 proposals, d_model 512, h 32, token projection
    128 -> 512) through forward + loss + backward on the HIP path and on the CPU checker backend."""
    from oracle.attention_ref import OracleBackend
    from spacap3d_amd.engine import synthetic_batch
    from spacap3d_amd.loss_helper import get_scene_cap_loss
    data = synthetic_batch(1, 80000, "cpu", seed=5, vocab=200)
    res = {}
    for name, be, dev in (("cpu", OracleBackend(openmp=True), "cpu"), ("hip", backend.HipBackend(), DEV)):
        with backend.use_backend(be):
            model = _cfg5_model(dev)
            assert model.caption.token_proj is not None and model.caption.token_proj.weight.shape == (512, 128)
            inp = {k: v.to(dev) for k, v in data.items()}
            if name == "hip":
                # 512 of 1 024 predicted vote positions by FPS: chaotic in the weights (detector.ProposalModule.forward);
                # the HIP run samples the CPU run's votes so that everything downstream is comparable
                inp["proposal_inds"] = res["cpu"][0]["aggregated_vote_inds"].to(dev)
            d = model(inp)
            d = get_scene_cap_loss(d, use_relation=True, mean_size_arr=S.mean_size_arr().numpy())
            d["loss"].backward()
            res[name] = (d, model)
    c, h = res["cpu"][0], res["hip"][0]
    for k in ("sa1_inds", "sa2_inds", "aggregated_vote_inds", "match_idx", "bbox_mask", "objectness_label"):
        assert torch.equal(c[k].cpu(), h[k].cpu()), k
    assert h["relation_pred"].shape == (1, 512, 512, 9) and h["lang_cap"].shape[:2] == (1, 31)

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

    for k in ("fp2_features", "vote_xyz", "vote_features", "aggregated_vote_features", "lang_cap", "relation_pred", "center"):
        assert rel(h[k], c[k]) < 2e-3, (k, rel(h[k], c[k]))
    for k in ("loss", "vote_loss", "objectness_loss", "box_loss", "cap_loss", "relation_loss"):
        assert abs(float(h[k]) - float(c[k])) <= 2e-3 * max(1.0, abs(float(c[k]))), (k, float(h[k]), float(c[k]))
    pc, ph = dict(res["cpu"][1].named_parameters()), dict(res["hip"][1].named_parameters())
    for n in ("caption.token_proj.weight", "caption.model.generator.proj.weight", "caption.relation_proposal.4.weight",
              "caption.model.encoder.layers.0.self_attn.linears.0.weight"):
        assert rel(ph[n].grad, pc[n].grad) < 2e-2, (n, rel(ph[n].grad, pc[n].grad))


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4", "cfg5"])
def test_bench_runs_every_config(cfg):
    """bench.py --config cfgN end to end (2 scenes per GPU to keep the test short; the default is the config's own
    per-GPU batch)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--batch", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["value"] > 0 and rec["final_loss"] == rec["final_loss"] and cfg in rec["config"]["workload"]
    assert rec["config"]["hip_graph"] is True, "the hipGraph capture must work for every config"


def test_cfg5_wide_relation_head_at_full_size_on_sampled_pair_rows():
    """BASELINE config 5 at its full per-GPU size: 16 scenes x 512 x 512 = 4 194 304 proposal pairs, 512 channels, 32 heads
    (8.6 GB per pair tensor) through linear.RelationWide (models/transformer_captioner.py:319-326, 392-397 at d_model = 512).
    Every stage of the head is LOCAL to a pair row -- hid1 row from P[b, :, i, j] and U[b, j], hid2, pred; backward: dz2, dhid1
    and dP[b, :, i, j] -- so 256 sampled rows are recomputed in float64 from the same inputs and compared: forward 3e-6 of scale,
    dP 3e-5 (a hidden unit within fp32 rounding of zero may gate differently: rows with such a unit are excluded, < 2 %)."""
    from spacap3d_amd.linear import relation_head_wide
    B, H, K, D, C = 16, 32, 512, 16, 512
    g = torch.Generator(device=DEV).manual_seed(5)
    torch.manual_seed(5)
    P = torch.softmax(torch.randn(B, H, K, K, generator=g, device=DEV), -1).requires_grad_(True)
    V = torch.randn(B, H, K, D, generator=g, device=DEV)
    l1, l2, l3 = torch.nn.Linear(H * D, C).to(DEV), torch.nn.Linear(C, C).to(DEV), torch.nn.Linear(C, 9).to(DEV)
    pred = relation_head_wide(P, V, l1, l2, l3)
    assert pred is not None and pred.shape == (B, K, K, 9)
    wsum = torch.randn(B, K, K, 9, generator=g, device=DEV)
    (pred * wsum).sum().backward()
    n = 256
    bs, is_, js = (torch.randint(0, m, (n,), generator=g, device=DEV) for m in (B, K, K))
    W1, b1, W2, b2, W3, b3 = (t.detach().double() for t in (l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias))
    p = P.detach()[bs, :, is_, js].double()                                   # (n, H)
    v = V[bs, :, js, :].double()                                              # (n, H, D)
    feat = (p.unsqueeze(-1) * v).reshape(n, H * D)                            # the pair feature rows
    z1 = feat @ W1.t() + b1
    h1 = z1.clamp_min(0)
    z2 = h1 @ W2.t() + b2
    h2 = z2.clamp_min(0)
    want = h2 @ W3.t() + b3
    got = pred.detach()[bs, is_, js].double()
    assert float((got - want).abs().max() / want.abs().max()) < 3e-6
    gr = wsum[bs, is_, js].double()
    dz2 = (gr @ W3) * (z2 > 0)
    dz1 = (dz2 @ W2) * (z1 > 0)
    dfeat = dz1 @ W1                                                          # (n, H D)
    dp = (dfeat.view(n, H, D) * v).sum(-1)                                    # dP[b, :, i, j]
    near = ((z1.abs() < 2e-7).any(1) | (z2.abs() < 2e-7).any(1))     # (pre-activations here are ~0.03: fp32 noise ~3e-9)
    assert float(near.double().mean()) < 0.02
    gotp = P.grad[bs, :, is_, js].double()
    err = ((gotp - dp).abs().max(1).values / dp.abs().max())[~near]
    assert float(err.max()) < 3e-5, float(err.max())
