"""Round-2 golden vectors, again produced by RUNNING THE REFERENCE'S OWN PYTHON in the build container
(see make_fixtures.py for the four import shims; nothing of the reference is copied or stored).

  sa_modules_fused.npz     the reference's PointnetSAModuleVotes (forward + backward) at the MLP shapes the fused
                           HIP shared-MLP kernels exist for: SA1-like [C,64,64,128] with C = 1 / 7 / 132 extra input
                           channels (BASELINE configs 2 / 3 / 4), SA2-like [128,128,128,256], SA3/4-like
                           [256,128,128,256] and the vote aggregation [256,128,128,128]
  train_step_cfg3.npz      full training step (forward + loss + backward) with input_feature_dim = 7
  train_step_cfg4.npz      ... and 132 (xyz + multiview 128 + normal 3 + height 1), cfg1 sizes
  captioner_variants.npz   the non-default branches of the captioner on the cfg1 inputs: --late_guide (cross-attention
                           over the 1-token memory), --no_relation, --no_enc, and the README's "base model"
                           (late guide, no relation head, sinusoidal source positions); training step + greedy decoding

Inputs that are not stored are regenerated identically by the tests: coordinates / labels from
``spacap3d_amd.synthetic`` (seeded CPU generators), extra feature channels and loss weights from the integer hash of
``detweights``.

Run:  python tests/golden/make_fixtures_r2.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from make_fixtures import import_reference, to_np  # noqa: E402
from detweights import _uniform, fill_  # noqa: E402
from spacap3d_amd import synthetic as S  # noqa: E402


def hash_tensor(shape, key, seed=0, scale=1.0, shift=0.0):
    n = int(np.prod(shape))
    return (torch.from_numpy(_uniform(n, key, seed)).view(*shape) * scale + shift).contiguous()


# name: (N source points, npoint, radius, nsample, mlp, features need a gradient)
SA_CASES = {
    "sa1_c1": (4096, 256, 0.3, 64, [1, 64, 64, 128], False),
    "sa1_c7": (4096, 256, 0.3, 64, [7, 64, 64, 128], True),
    "sa1_c132": (4096, 128, 0.3, 64, [132, 64, 64, 128], False),
    "sa2": (2048, 256, 0.4, 32, [128, 128, 128, 256], True),
    "sa3": (1024, 128, 0.8, 16, [256, 128, 128, 256], True),
    "agg": (1024, 64, 0.3, 16, [256, 128, 128, 128], True),
}


def sa_case_inputs(name):
    N, npoint, radius, ns, mlp, need_grad = SA_CASES[name]
    xyz = S.scene_batch(2, N, use_height=False, seed=100 + len(name))
    feats = hash_tensor((2, mlp[0], N), "feats_" + name, seed=4, scale=2.0)
    wout = hash_tensor((2, mlp[-1], npoint), "wout_" + name, seed=5, scale=2.0)
    return xyz, feats, wout


def make_sa(out_dir):
    import pointnet2_modules as pm
    fx = {}
    for name, (N, npoint, radius, ns, mlp, need_grad) in SA_CASES.items():
        xyz, feats, wout = sa_case_inputs(name)
        sa = pm.PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=ns, mlp=list(mlp), use_xyz=True,
                                      normalize_xyz=True)
        fill_(sa, seed=7)
        sa.train()
        feats.requires_grad_(need_grad)
        new_xyz, new_feats, inds = sa(xyz, feats)
        (new_feats * wout).sum().backward()
        fx[name + "_inds"] = to_np(inds)
        fx[name + "_new_feats__flat3"] = to_np(new_feats).reshape(2, -1)[:, ::3]
        if need_grad:
            fx[name + "_feats_grad__flat17"] = to_np(feats.grad).reshape(2, -1)[:, ::17]
        for k, p in sa.named_parameters():
            g = to_np(p.grad)
            fx[name + "_grad_" + k] = g.reshape(-1)[::3] if g.size > 4096 else g
        for k, b in sa.named_buffers():
            if k.endswith("running_mean") or k.endswith("running_var"):
                fx[name + "_buf_" + k] = to_np(b)
    np.savez_compressed(os.path.join(out_dir, "sa_modules_fused.npz"), **fx)
    print("sa_modules_fused.npz done")


def cfg_point_clouds(C, B=2, N=4096, seed=11):
    """(B, N, 3 + C): xyz and height from the synthetic scene generator, the C - 1 channels between them (colour /
    normal / multiview) from the integer hash."""
    base = S.scene_batch(B, N, seed=seed)            # xyz + height
    extra = hash_tensor((B, N, C - 1), f"extra_channels_{C}", seed=seed, scale=1.0, shift=0.1)
    return torch.cat([base[..., :3], extra, base[..., 3:]], -1).contiguous()


def place_boxes(model, pc, lab, P):
    """As make_fixtures.py: GT boxes on 32 of the proposal positions so that positives exist."""
    with torch.no_grad():
        probe = model({"point_clouds": pc.clone(), **{k: v.clone() for k, v in lab.items()}})
        agg = probe["aggregated_vote_xyz"].clone()
    B = pc.shape[0]
    g = torch.Generator().manual_seed(3)
    n_gt = 32
    pick = torch.stack([torch.randperm(P, generator=g)[:n_gt] for _ in range(B)])
    ctr = torch.gather(agg, 1, pick.unsqueeze(-1).expand(-1, -1, 3)) + 0.05 * torch.randn(B, n_gt, 3, generator=g)
    lab["center_label"][:, :n_gt] = ctr
    lab["ref_center_label"] = ctr[:, 0].clone()
    return lab


GRADS_CFG = ("backbone_net.sa1.mlp_module.layer0.conv.weight", "backbone_net.sa1.mlp_module.layer2.conv.weight",
             "backbone_net.sa1.mlp_module.layer0.bn.bn.weight", "vgen.conv3.weight",
             "caption.model.generator.proj.weight")
OUTS_CFG = ("sa1_inds", "sa2_inds", "sa1_features", "fp2_features", "aggregated_vote_features", "aggregated_vote_inds",
            "objectness_scores", "center", "bbox_mask", "lang_cap", "match_idx", "relation_pred", "object_assignment",
            "objectness_label")
LOSSES = ("loss", "vote_loss", "objectness_loss", "box_loss", "sem_cls_loss", "cap_loss", "relation_loss", "det_loss",
          "cap_acc", "obj_acc")


def zero_dropout(model):
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0


def make_cfg(out_dir, SpaCapNet, ref_loss, DC, C, tag):
    B, N, P, V = 2, 4096, 64, 40
    vocab = S.make_vocabulary(V)
    model = SpaCapNet(num_class=DC.num_class, vocabulary=vocab, num_heading_bin=DC.num_heading_bin,
                      num_size_cluster=DC.num_size_cluster, mean_size_arr=DC.mean_size_arr, input_feature_dim=C,
                      num_proposal=P, transformer_dropout=0.0, src_pos_type="xyz", use_transformer_encoder=True,
                      early_guide=True, check_relation=True, N=2, h=8, d_model=128, d_ff=128)
    fill_(model, seed=1)
    model.train()
    zero_dropout(model)
    pc = cfg_point_clouds(C)
    lab = place_boxes(model, pc, S.labels(B, N, vocab=V, seed=11), P)
    fill_(model, seed=1)   # the probe forward moved the BatchNorm running statistics
    d = {"point_clouds": pc.clone()}
    d.update({k: v.clone() for k, v in lab.items()})
    d = model(d)
    d = ref_loss(d, "cpu", DC, detection=True, caption=True, use_relation=True)
    d["loss"].backward()
    fx = {"cfg_C": C, "mean_size_arr": np.asarray(DC.mean_size_arr), "xyz": to_np(pc[..., :3]),
          "label_center_label": to_np(lab["center_label"]), "label_ref_center_label": to_np(lab["ref_center_label"])}
    for k in OUTS_CFG:
        a = to_np(d[k])
        if a.size > 40000:
            a, k = a.reshape(a.shape[0], -1)[:, ::7], k + "__flat7"
        fx["out_" + k] = a
    for k in LOSSES:
        fx["loss_" + k] = np.float64(float(d[k].detach()) if torch.is_tensor(d[k]) else float(d[k]))
    sd = dict(model.named_parameters())
    for name in GRADS_CFG:
        gr = to_np(sd[name].grad).reshape(-1)
        fx["grad_" + name] = gr[::3] if gr.size > 4096 else gr
    np.savez_compressed(os.path.join(out_dir, f"train_step_{tag}.npz"), **fx)
    print(f"train_step_{tag}.npz: loss", float(d["loss"]))


VARIANTS = {   # scripts/train.py:147-152: flags -> constructor arguments
    "late_guide": dict(src_pos_type="xyz", use_transformer_encoder=True, early_guide=False, check_relation=True),
    "no_relation": dict(src_pos_type="xyz", use_transformer_encoder=True, early_guide=True, check_relation=False),
    # (--no_enc only runs in the reference together with --no_learnt_src_pos: its identity src_embed takes one argument)
    "no_enc": dict(src_pos_type=None, use_transformer_encoder=False, early_guide=True, check_relation=False),
    "base": dict(src_pos_type=None, use_transformer_encoder=True, early_guide=False, check_relation=False),
}
GRADS_VAR = ("caption.model.decoder.layers.0.self_attn.linears.1.weight",
             "caption.model.decoder.layers.1.src_attn.linears.2.weight",
             "caption.model.decoder.layers.1.feed_forward.w_2.weight",
             "caption.model.generator.proj.weight", "proposal.proposal.6.weight")


def make_variants(out_dir, SpaCapNet, ref_loss, DC):
    base = np.load(os.path.join(out_dir, "train_step_cfg1.npz"))
    B, N, P, V = int(base["cfg_B"]), int(base["cfg_N"]), int(base["cfg_P"]), int(base["cfg_V"])
    vocab = S.make_vocabulary(V)
    fx = {}
    for tag, kw in VARIANTS.items():
        model = SpaCapNet(num_class=DC.num_class, vocabulary=vocab, num_heading_bin=DC.num_heading_bin,
                          num_size_cluster=DC.num_size_cluster, mean_size_arr=DC.mean_size_arr, input_feature_dim=1,
                          num_proposal=P, transformer_dropout=0.0, N=2, h=8, d_model=128, d_ff=128, **kw)
        fill_(model, seed=1)
        model.train()
        zero_dropout(model)
        d = {"point_clouds": torch.from_numpy(base["point_clouds"]).clone()}
        for k in base.files:
            if k.startswith("label_"):
                d[k[6:]] = torch.from_numpy(base[k]).clone()
        inputs = {k: v.clone() for k, v in d.items()}
        d = model(d)
        d = ref_loss(d, "cpu", DC, detection=True, caption=True, use_relation=kw["check_relation"])
        d["loss"].backward()
        fx[tag + "_lang_cap"] = to_np(d["lang_cap"])
        fx[tag + "_match_idx"] = to_np(d["match_idx"])
        if kw["check_relation"]:
            fx[tag + "_relation_pred"] = to_np(d["relation_pred"]).reshape(B, -1)[:, ::7]
        for k in ("loss", "cap_loss", "relation_loss", "det_loss", "cap_acc"):
            fx[tag + "_loss_" + k] = np.float64(float(d[k].detach()) if torch.is_tensor(d[k]) else float(d[k]))
        sd = dict(model.named_parameters())
        for name in GRADS_VAR:
            if name in sd and sd[name].grad is not None:
                gr = to_np(sd[name].grad).reshape(-1)
                fx[tag + "_grad_" + name] = gr[::3] if gr.size > 4096 else gr
        fx[tag + "_grad_absent"] = np.array(sorted(n for n, p in model.named_parameters() if p.grad is None))
        # greedy decoding of the same variant
        fill_(model, seed=1)
        model.eval()
        with torch.no_grad():
            e = model({k: v.clone() for k, v in inputs.items()}, is_eval=True)
        fx[tag + "_eval_lang_cap"] = to_np(e["lang_cap"])
        print(f"captioner variant {tag}: loss {float(d['loss']):.5f}  caps {to_np(e['lang_cap'])[0, 0, :6]}")
    np.savez_compressed(os.path.join(out_dir, "captioner_variants.npz"), **fx)
    print("captioner_variants.npz done")


def main():
    SpaCapNet, ref_loss, DCcls, tc, pu = import_reference()
    DC = DCcls()
    make_sa(HERE)
    make_cfg(HERE, SpaCapNet, ref_loss, DC, 7, "cfg3")
    make_cfg(HERE, SpaCapNet, ref_loss, DC, 132, "cfg4")
    make_variants(HERE, SpaCapNet, ref_loss, DC)


if __name__ == "__main__":
    main()
