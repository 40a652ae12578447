"""Generates tests/golden/train_step_cfg1_selections.npz: the DISCRETE SELECTIONS the reference's own run of the cfg1
training step made, wherever a selection is decided by a margin small enough for fp32 re-association to flip it, plus the
gradient of every detector weight of that run.

Why: every ReLU gate and every max-pool arg-max is a step function of the weights.  A HIP run whose pre-activations differ
from the reference's by 1e-6 resolves a handful of near-ties the other way, and each flip re-routes a whole summand of a
weight gradient that sums to ~0 per channel (BatchNorm): the gradients of a FREE run can only be compared at the 1e-2
level (tests/test_golden.py).  With the reference's selections forced into the HIP run the step is a smooth function and
its gradients are held against the reference's at 1e-3 (tests/test_golden.py::test_train_step_gradients_with_the_reference_selections).

What is stored, per BatchNorm -> ReLU layer of the detector (in the reference's module names):
  near_<name>_idx   int32 flat indices, in the reference's (B, C, ...) layout, of the elements with |bn(z)| < tau
  near_<name>_pos   uint8 1 where the reference's bn(z) > 0 (the ReLU passes the gradient), else 0
  tau_<name>        the margin used (grows with depth: the runs' activations drift apart by up to 1e-4 of scale)
per max-pool (the five SA modules):
  pool_<name>_idx   int32 flat indices into (B, C, P) where the best two post-ReLU candidates are closer than tau and > 0
  pool_<name>_arg   uint8 the sample the reference's max_pool2d routed the gradient to (return_indices)
and grad_<parameter name> for every detector parameter (strided to <= 4 096 values as in make_fixtures.py).

The run is the one of make_fixtures.py, fixture 1 (same weights, inputs, labels): the script asserts that its loss and the
gradients stored in train_step_cfg1.npz come out bit-identical before it writes anything.

Only works where /root/reference exists.  Run:  python tests/golden/make_fixtures_selections.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_fixtures as MF  # noqa: E402  (import_reference, to_np, S, fill_)
from make_fixtures import S, fill_, to_np  # noqa: E402

# margin per module family: |bn(z)| below it counts as a near-tie (the BatchNorm output has unit scale)
TAU = {"backbone_net.sa1": 3e-4, "backbone_net.sa2": 5e-4}
TAU_DEFAULT = 2e-3


def tau_of(name):
    for k, v in TAU.items():
        if name.startswith(k):
            return v
    return TAU_DEFAULT


def main():
    SpaCapNet, ref_loss, DCcls, tc, pu = MF.import_reference()
    DC = DCcls()
    old = np.load(os.path.join(HERE, "train_step_cfg1.npz"))
    B, N, P, V = 2, 4096, 64, 40
    cfg = dict(N=2, h=8, d_model=128, d_ff=128)
    torch.manual_seed(0)
    model = SpaCapNet(num_class=DC.num_class, vocabulary=S.make_vocabulary(V), num_heading_bin=DC.num_heading_bin,
                      num_size_cluster=DC.num_size_cluster, mean_size_arr=DC.mean_size_arr, input_feature_dim=1,
                      num_proposal=P, transformer_dropout=0.0, src_pos_type="xyz", use_transformer_encoder=True,
                      early_guide=True, check_relation=True, **cfg)
    fill_(model, seed=1)
    model.train()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    pc = torch.from_numpy(old["point_clouds"])
    lab = {k[6:]: torch.from_numpy(old[k]) for k in old.files if k.startswith("label_")}

    fx = {}
    names = {id(m): n for n, m in model.named_modules()}
    hooks = []

    def bn_hook(mod, inp, out):
        # runs right after the BatchNorm forward, BEFORE the in-place ReLU that follows it in SharedMLP / F.relu
        name = names[id(mod)]
        tau = tau_of(name)
        y = out.detach().reshape(-1)
        idx = torch.nonzero(y.abs() < tau).reshape(-1)
        fx["near_" + name + "_idx"] = idx.to(torch.int32).numpy()
        fx["near_" + name + "_pos"] = (y[idx] > 0).to(torch.uint8).numpy()
        fx["tau_" + name] = np.float64(tau)
        fx["shape_" + name] = np.asarray(out.shape, dtype=np.int64)

    def pool_hook(mod, inp, out):
        # the SharedMLP's output (B, C, P, S), post-ReLU: what PointnetSAModuleVotes.forward max-pools over S
        # (lib/pointnet2/pointnet2_modules.py:256-259)
        name = names[id(mod)]
        tau = tau_of(name)
        y = out.detach()
        _, ind = torch.nn.functional.max_pool2d(y, kernel_size=[1, y.size(3)], return_indices=True)
        arg = (ind.squeeze(-1) % y.size(3)).to(torch.uint8)                 # (B, C, P): sample index inside the group
        top = torch.topk(y, 2, dim=-1).values
        near = ((top[..., 0] - top[..., 1]) < tau) & (top[..., 0] > 0)
        idx = torch.nonzero(near.reshape(-1)).reshape(-1)
        fx["pool_" + name + "_idx"] = idx.to(torch.int32).numpy()
        fx["pool_" + name + "_arg"] = arg.reshape(-1)[idx].numpy()
        fx["poolshape_" + name] = np.asarray(y.shape, dtype=np.int64)

    det_prefixes = ("backbone_net.", "vgen.", "proposal.")
    for n, m in model.named_modules():
        if n.startswith(det_prefixes) and isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            hooks.append(m.register_forward_hook(bn_hook))
        if n.endswith("mlp_module") and n.startswith(det_prefixes):
            hooks.append(m.register_forward_hook(pool_hook))

    d = {"point_clouds": pc.clone()}
    d.update({k: v.clone() for k, v in lab.items()})
    d = model(d)
    d = ref_loss(d, "cpu", DC, detection=True, caption=True, use_relation=True)
    d["loss"].backward()
    for h in hooks:
        h.remove()

    # the same run as train_step_cfg1.npz, bit for bit
    assert np.float64(float(d["loss"])) == old["loss_loss"], (float(d["loss"]), float(old["loss_loss"]))
    sd = dict(model.named_parameters())
    for k in old.files:
        if k.startswith("grad_") and k != "grad_absent":
            gr = to_np(sd[k[5:]].grad).reshape(-1)
            gr = gr[::3] if gr.size > 4096 else gr
            assert np.array_equal(gr, old[k]), k
    assert np.array_equal(to_np(d["aggregated_vote_inds"]), old["out_aggregated_vote_inds"])

    for n, p in model.named_parameters():
        if n.startswith(det_prefixes) and p.grad is not None:
            gr = to_np(p.grad).reshape(-1)
            step = max(1, -(-gr.size // 4096))
            fx["grad_" + n] = gr[::step]
            fx["gradstep_" + n] = np.int64(step)
    fx["aggregated_vote_inds"] = to_np(d["aggregated_vote_inds"])
    np.savez_compressed(os.path.join(HERE, "train_step_cfg1_selections.npz"), **fx)
    n_near = sum(v.size for k, v in fx.items() if k.startswith("near_") and k.endswith("_idx"))
    n_pool = sum(v.size for k, v in fx.items() if k.startswith("pool_") and k.endswith("_idx"))
    print(f"train_step_cfg1_selections.npz: {n_near} near-tie ReLU gates, {n_pool} near-tie pools, "
          f"{sum(1 for k in fx if k.startswith('grad_'))} gradients")
    for k in sorted(fx):
        if k.startswith("near_") and k.endswith("_idx"):
            print(f"  {k[5:-4]:60s} {fx[k].size:7d} of {int(np.prod(fx['shape_' + k[5:-4]])):9d}")


if __name__ == "__main__":
    main()
