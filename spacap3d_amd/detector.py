"""PointNet++ backbone, voting module and proposal module -- counterparts of the reference's
``models/backbone_module.py``, ``models/voting_module.py`` and ``models/proposal_module.py``.

Same constructor arguments, ``data_dict`` keys, tensor shapes and state-dict key layout
(``sa{1..4}.mlp_module.*``, ``fp{1,2}.mlp.*``, ``conv{1,2,3}``, ``bn{1,2}``, ``vote_aggregation.*``,
``proposal.{0,1,3,4,6}.*``) as the reference.  Differences, all below the observable interface:
  * device-agnostic (the reference hard-codes ``.cuda()``, proposal_module.py:100,141,143);
  * the predicted box corners are decoded on the device in float64 instead of the reference's
    GPU -> CPU -> numpy -> GPU round trip every forward (proposal_module.py:81-104); ScanNet boxes are
    axis aligned (heading is identically 0, model_util_scannet.py:136-146), so the corners are
    centre +/- size / 2 in the corner order of utils/box_util.py:360-383.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes
from .layout import ChannelMajorOf, point_major_of


SA_NPOINTS = (2048, 1024, 512, 256)  # backbone_module.py:29,38,47,56


def sampling_pyramid(xyz, npoints=SA_NPOINTS):
    """The four furthest-point-sampling index sets of the backbone, which depend on the input coordinates only
    (SA_{l+1} samples the centres SA_l gathered): inds_l = FPS(xyz_{l-1}, npoint_l), xyz_l = xyz_{l-1}[inds_l].
    Exactly what the four ``PointnetSAModuleVotes.forward`` calls compute (pointnet2_modules.py:237-242) --
    factored out so a trainer can run it ahead of time on a side stream (the module accepts precomputed
    ``inds``, pointnet2_modules.py:214,236-239)."""
    from . import pointnet2_utils as pu
    inds_all = []
    cur = xyz
    for n in npoints:
        inds = pu.furthest_point_sample(cur, n)
        inds_all.append(inds)
        cur = pu.gather_operation(cur.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    return inds_all


SA_RADII = (0.2, 0.4, 0.8, 1.2)       # backbone_module.py:30,39,48,57
SA_NSAMPLES = (64, 32, 16, 16)        # backbone_module.py:31,40,49,58


def _centres(xyz, inds):
    """xyz[b, inds[b, j], :]: one launch on the GPU (no index conversion), torch.gather elsewhere."""
    if xyz.is_cuda and not xyz.requires_grad and inds.dtype == torch.int32:
        from . import ext
        return ext.gather_xyz(xyz.contiguous(), inds.contiguous())
    return torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3))


def geometry_pyramid(xyz, npoints=SA_NPOINTS, radii=SA_RADII, nsamples=SA_NSAMPLES):
    """Everything in the backbone that depends on the input coordinates only, as a flat tuple of 12 (+7 on the GPU) tensors:
    the four sampling index sets (``sampling_pyramid``), the four ball-query groupings of the SA modules
    (pointnet2_modules.py:241-247) and (idx, weight) of the two feature-propagation modules' three-nearest-neighbour
    interpolation (:399-405; fp1: SA3 points from SA4's, fp2: SA2 points from SA3's, backbone_module.py:115-119).
    A trainer runs it for the NEXT batch on a side stream; the modules take the results through their ``inds`` /
    ``idx`` / ``nn`` arguments and compute exactly these values themselves when they get none."""
    from . import pointnet2_utils as pu
    inds_all, idx_all, xyzs = [], [], [xyz]
    cur = xyz
    for n, r, ns in zip(npoints, radii, nsamples):
        inds = pu.furthest_point_sample(cur, n)
        new_xyz = _centres(cur, inds)
        idx_all.append(pu.ball_query(r, ns, cur, new_xyz))
        inds_all.append(inds)
        xyzs.append(new_xyz)
        cur = new_xyz
    fp1 = PointnetFPModule.neighbours(xyzs[3], xyzs[4])
    fp2 = PointnetFPModule.neighbours(xyzs[2], xyzs[3])
    out = tuple(inds_all) + tuple(idx_all) + (fp1[0], fp1[1], fp2[0], fp2[1])
    if xyz.is_cuda:
        # + the inverted indices of the SA2..SA4 groupings (feature-gradient gathers of the fused SA op; SA1's
        # features are an input and get no gradient)
        from .sa_mlp import rows_index
        out = out + tuple(rows_index(idx_all[l], xyzs[l].shape[1]) for l in (1, 2, 3))
        # + the sampled centres themselves (each SA module otherwise gathers them again: an index conversion + a gather)
        out = out + tuple(x.contiguous() for x in xyzs[1:])
    return out


class Pointnet2Backbone(nn.Module):
    """4 set-abstraction + 2 feature-propagation layers (backbone_module.py:28-66)."""

    def __init__(self, input_feature_dim=0):
        super().__init__()
        self.input_feature_dim = input_feature_dim
        self.sa1 = PointnetSAModuleVotes(npoint=2048, radius=0.2, nsample=64,
                                         mlp=[input_feature_dim, 64, 64, 128], use_xyz=True, normalize_xyz=True)
        self.sa2 = PointnetSAModuleVotes(npoint=1024, radius=0.4, nsample=32,
                                         mlp=[128, 128, 128, 256], use_xyz=True, normalize_xyz=True)
        self.sa3 = PointnetSAModuleVotes(npoint=512, radius=0.8, nsample=16,
                                         mlp=[256, 128, 128, 256], use_xyz=True, normalize_xyz=True)
        self.sa4 = PointnetSAModuleVotes(npoint=256, radius=1.2, nsample=16,
                                         mlp=[256, 128, 128, 256], use_xyz=True, normalize_xyz=True)
        self.fp1 = PointnetFPModule(mlp=[256 + 256, 256, 256])
        self.fp2 = PointnetFPModule(mlp=[256 + 256, 256, 256])

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., :3].contiguous()
        if pc.size(-1) <= 3:
            return xyz, None
        if pc.is_cuda and pc.size(-1) > 4 and pc.is_contiguous() and not pc.requires_grad:
            # (B, C, N) as the reference hands it on (models/backbone_module.py:69-73) -- but as the typed transposed VIEW of the
            # cloud's own feature columns: the fused SA op reads the rows in place (sa_mlp._uniform_rows) instead of through a
            # transposed copy here and a second one back to point-major there (2 x 170 MB at 132 channels, 0.85 ms per step)
            from .layout import ChannelMajorOf
            return xyz, ChannelMajorOf.wrap(pc[..., 3:])
        return xyz, pc[..., 3:].transpose(1, 2).contiguous()

    # Furthest-point sampling is a chain of ~4 000 sequentially dependent rounds on B workgroups (one per scene): 2.9 ms +
    # 0.76 + 0.31 + 0.16 at cfg2, while each level's grouping + shared MLP only needs THAT level's indices.  When the
    # caller hands in no precomputed pyramid the four samplings therefore run as one chain on a side stream and each SA
    # module waits for its level's event: SA_l's ball query / GEMMs overlap with the sampling of level l + 1 (the chain,
    # not chain + MLPs, is the forward's critical path).  Values are identical to sampling in line.
    overlap_sampling = True
    _side_streams = {}

    def _sample_ahead(self, xyz):
        from . import pointnet2_utils as pu
        dev = xyz.device
        side = Pointnet2Backbone._side_streams.get(dev)
        if side is None:
            side = Pointnet2Backbone._side_streams[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        levels = []
        # the chain holds one CU per scene while the first SA modules' layer kernels run beside it: they size their
        # persistent grids to the rest of the chip (csrc/sa_mlp.hip: spacap_sa_reserve_cus; cheap, idempotent)
        from ._native import check, lib
        check(lib.spacap_sa_reserve_cus(min(64, int(xyz.shape[0]))), "spacap_sa_reserve_cus")
        with torch.cuda.stream(side), torch.no_grad():
            cur = xyz
            for n in SA_NPOINTS:
                inds = pu.furthest_point_sample(cur, n)
                cur = _centres(cur, inds)
                ev = torch.cuda.Event()
                ev.record(side)
                levels.append((inds, cur, ev))
        return levels

    def forward(self, data_dict):
        xyz, features = self._break_up_pc(data_dict["point_clouds"])
        # optional precomputed sampling pyramid (see sampling_pyramid); None -> each SA module samples itself
        # (4 tensors: sampling_pyramid; 12 / 19: geometry_pyramid, which adds the groupings, the interpolation weights and, on
        # the GPU, the inverted indices of the SA2..SA4 groupings and the four sets of sampled centres)
        pyr = tuple(data_dict.get("fps_pyramid") or ())
        pyr = pyr + (None,) * (19 - len(pyr))
        if pyr[0] is None and xyz.is_cuda and self.overlap_sampling and not xyz.requires_grad:
            levels = self._sample_ahead(xyz)
            data_dict["_sampling_levels"] = levels      # keeps the side stream's tensors alive for the whole step
            cur = torch.cuda.current_stream(xyz.device)
            pyr = list(pyr)
            for l, (inds, new_xyz, ev) in enumerate(levels):
                pyr[l], pyr[15 + l] = inds, new_xyz
            waits = [ev for _, _, ev in levels]
        else:
            waits = [None] * 4
            cur = None
        if waits[0] is not None:
            cur.wait_event(waits[0])
        xyz, features, fps_inds = self.sa1(xyz, features, pyr[0], pyr[4], None, pyr[15])
        data_dict["sa1_inds"], data_dict["sa1_xyz"], data_dict["sa1_features"] = fps_inds, xyz, features
        if waits[1] is not None:
            cur.wait_event(waits[1])
        xyz, features, fps_inds = self.sa2(xyz, features, pyr[1], pyr[5], pyr[12], pyr[16])
        data_dict["sa2_inds"], data_dict["sa2_xyz"], data_dict["sa2_features"] = fps_inds, xyz, features
        if waits[2] is not None:
            cur.wait_event(waits[2])
        xyz, features, fps_inds = self.sa3(xyz, features, pyr[2], pyr[6], pyr[13], pyr[17])
        data_dict["sa3_xyz"], data_dict["sa3_features"] = xyz, features
        if waits[3] is not None:
            cur.wait_event(waits[3])
        xyz, features, fps_inds = self.sa4(xyz, features, pyr[3], pyr[7], pyr[14], pyr[18])
        data_dict["sa4_xyz"], data_dict["sa4_features"] = xyz, features
        features = self.fp1(data_dict["sa3_xyz"], data_dict["sa4_xyz"], data_dict["sa3_features"],
                            data_dict["sa4_features"], nn=(pyr[8], pyr[9]) if pyr[8] is not None else None)
        features = self.fp2(data_dict["sa2_xyz"], data_dict["sa3_xyz"], data_dict["sa2_features"], features,
                            nn=(pyr[10], pyr[11]) if pyr[10] is not None else None)
        data_dict["fp2_features"] = features
        data_dict["fp2_xyz"] = data_dict["sa2_xyz"]
        num_seed = data_dict["fp2_xyz"].shape[1]
        # VoteNet quirk kept on purpose (backbone_module.py:127): the vote loss gathers labels with it.
        data_dict["fp2_inds"] = data_dict["sa1_inds"][:, 0:num_seed]
        return data_dict


def _bn_relu(bn, x, training):
    """relu(bn(x)); in training mode one fused op of the backend (batch statistics, csrc/bn_relu.hip) when it has one."""
    from .backend import ops
    f = getattr(ops(), "bn_relu_train", None) if (training and x.is_cuda) else None
    if f is not None:
        return f(x.contiguous(), bn)
    g = getattr(ops(), "bn_relu_eval", None) if (x.is_cuda and not training) else None   # inference: one launch on the running statistics
    y = g(x, bn) if g is not None else None
    return F.relu(bn(x)) if y is None else y


def _conv(conv, x, training):
    """``conv(x)`` through the backend's 1x1 convolution when it has one: in training mode (slab weight gradient) and in
    the inference forward (no gradient recording: forward kernel only)."""
    from .backend import ops
    f = getattr(ops(), "conv1x1", None) if (training or not torch.is_grad_enabled()) else None
    y = f(x, conv) if f is not None else None
    return conv(x) if y is None else y


def _ops():
    from .backend import ops
    return ops()


class VotingModule(nn.Module):
    """Conv1d 256->256->256->(3+256)*vote_factor with BN+ReLU on the first two, residual votes
    (voting_module.py:28-61)."""

    def __init__(self, vote_factor, seed_feature_dim):
        super().__init__()
        self.vote_factor = vote_factor
        self.in_dim = seed_feature_dim
        self.out_dim = self.in_dim
        self.conv1 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv2 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv3 = nn.Conv1d(self.in_dim, (3 + self.out_dim) * self.vote_factor, 1)
        self.bn1 = nn.BatchNorm1d(self.in_dim)
        self.bn2 = nn.BatchNorm1d(self.in_dim)

    def forward(self, seed_xyz, seed_features):
        B, num_seed = seed_xyz.shape[0], seed_xyz.shape[1]
        num_vote = num_seed * self.vote_factor
        net = _bn_relu(self.bn1, _conv(self.conv1, seed_features, self.training), self.training)
        net = _bn_relu(self.bn2, _conv(self.conv2, net, self.training), self.training)
        net = _conv(self.conv3, net, self.training)
        fused = getattr(_ops(), "vote_assemble", None) if (self.training and net.is_cuda and self.vote_factor == 1) else None
        if fused is not None:
            # seed + offset for coordinates and features, features straight into point-major layout: one launch each way
            vote_xyz, pm = fused(net, seed_xyz, seed_features)
            return vote_xyz, ChannelMajorOf.wrap(pm)
        net = net.transpose(2, 1).view(B, num_seed, self.vote_factor, 3 + self.out_dim)
        vote_xyz = (seed_xyz.unsqueeze(2) + net[:, :, :, 0:3]).contiguous().view(B, num_vote, 3)
        vote_features = seed_features.transpose(2, 1).unsqueeze(2) + net[:, :, :, 3:]
        pm = vote_features.contiguous().view(B, num_vote, self.out_dim)   # (B, num_vote, C): point-major
        if self.training and pm.is_cuda:
            # (B, C, num_vote) as a transposed VIEW carrying the point-major tensor (the fused SA op of the proposal
            # module reads that one): no transposed copy here nor of its gradient
            vote_features = ChannelMajorOf.wrap(pm)
        else:
            vote_features = pm.transpose(2, 1).contiguous()
        return vote_xyz, vote_features


_CORNER_SIGNS = torch.tensor(  # utils/box_util.py:377-379 (l on x, w on y, h on z)
    [[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1], [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]],
    dtype=torch.float64)


class ProposalModule(nn.Module):
    """Vote aggregation SA (npoint=P, r=0.3, ns=16, mlp [256+3,128,128,128]) + Conv1d head
    128->128->128->(2+3+2*NH+4*NS+num_class) (proposal_module.py:34-54) + score decoding (:106-158)."""

    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr, num_proposal, sampling,
                 seed_feat_dim=256, size_decoded=False):
        super().__init__()
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.num_proposal = num_proposal
        self.sampling = sampling
        self.seed_feat_dim = seed_feat_dim
        self.size_decoded = size_decoded
        msa = torch.as_tensor(np.asarray(mean_size_arr), dtype=torch.float64)
        assert msa.shape == (num_size_cluster, 3)
        # non-persistent: not part of the reference's state dict
        self.register_buffer("mean_size_f64", msa, persistent=False)
        self.register_buffer("mean_size_f32", msa.float(), persistent=False)
        self.register_buffer("corner_signs", _CORNER_SIGNS.clone(), persistent=False)
        self.vote_aggregation = PointnetSAModuleVotes(npoint=num_proposal, radius=0.3, nsample=16,
                                                      mlp=[seed_feat_dim, 128, 128, 128], use_xyz=True,
                                                      normalize_xyz=True)
        self.proposal = nn.Sequential(
            nn.Conv1d(128, 128, 1, bias=False), nn.BatchNorm1d(128), nn.ReLU(),
            nn.Conv1d(128, 128, 1, bias=False), nn.BatchNorm1d(128), nn.ReLU(),
            nn.Conv1d(128, 2 + 3 + num_heading_bin * 2 + num_size_cluster * 4 + num_class, 1))

    def forward(self, xyz, features, data_dict):
        # ``proposal_inds`` (B, num_proposal) int32, optional: precomputed vote-sampling indices, handed to the SA module's
        # ``inds`` argument (the reference interface, pointnet2_modules.py:214,236-239).  The default -- furthest-point
        # sampling of the predicted vote positions -- is a discrete, chaotic function of network outputs (a 3e-6 relative
        # weight perturbation changes which votes become proposals); parity / trajectory tests pin it to compare the
        # continuous part of the step.
        xyz, features, fps_inds = self.vote_aggregation(xyz, features, data_dict.get("proposal_inds"))
        data_dict["aggregated_vote_xyz"] = xyz
        pm = point_major_of(features)   # the fused SA op's own (B,K,128) result
        data_dict["aggregated_vote_features"] = pm if pm is not None else features.permute(0, 2, 1).contiguous()
        data_dict["aggregated_vote_inds"] = fps_inds
        net = features.contiguous() if (self.training or not torch.is_grad_enabled()) else features   # (the own 1x1 kernel takes dense tensors)
        layers = list(self.proposal)
        i = 0
        while i < len(layers):
            layer = layers[i]
            if isinstance(layer, nn.Conv1d):
                net = _conv(layer, net, self.training)
            elif isinstance(layer, nn.BatchNorm1d) and i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU):
                net = _bn_relu(layer, net, self.training)   # BatchNorm1d -> ReLU as one op in training mode
                i += 1
            else:
                net = layer(net)
            i += 1
        return self.decode_scores(net, data_dict)

    def decode_pred_box(self, data_dict):
        """(B, P, 8, 3) float64 corners of the arg-max size class box (proposal_module.py:81-104)."""
        center = data_dict["center"].detach().double()
        size_class = torch.argmax(data_dict["size_scores"], -1)
        size_res = torch.gather(data_dict["size_residuals"].detach(), 2,
                                size_class.unsqueeze(-1).unsqueeze(-1).expand(-1, -1, 1, 3)).squeeze(2)
        box_size = self.mean_size_f64[size_class] + size_res.double()  # class2size_batch
        return center.unsqueeze(2) + self.corner_signs * (box_size / 2).unsqueeze(2)

    def decode_scores(self, net, data_dict):
        NH, NS = self.num_heading_bin, self.num_size_cluster
        from .backend import ops
        fused = getattr(ops(), "proposal_decode", None) if net.is_cuda else None
        if fused is not None:
            # transpose, centre add, residual scalings, the four arg-maxes and the float64 box corners in one launch
            nt, center, hres, sres, corners, bmask, sem, scls = fused(net, data_dict["aggregated_vote_xyz"], self.mean_size_f32,
                                                                      self.mean_size_f64, NH, NS)
            d = data_dict
            d["_proposal_net"] = nt
            d["objectness_scores"], d["center"] = nt[:, :, 0:2], center
            d["heading_scores"] = nt[:, :, 5:5 + NH]
            d["heading_residuals_normalized"] = nt[:, :, 5 + NH:5 + NH * 2]
            d["heading_residuals"] = hres
            d["size_scores"] = nt[:, :, 5 + NH * 2:5 + NH * 2 + NS]
            d["size_residuals_normalized"] = nt[:, :, 5 + NH * 2 + NS:5 + NH * 2 + NS * 4].view(nt.shape[0], nt.shape[1], NS, 3)
            d["size_residuals"] = sres
            if self.size_decoded:
                cls = scls.unsqueeze(-1).unsqueeze(-1).expand(-1, -1, 1, 3)
                d["pred_size"] = torch.gather(sres + self.mean_size_f32.unsqueeze(0).unsqueeze(0), 2, cls).squeeze(2)
            d["sem_cls_scores"] = nt[:, :, 5 + NH * 2 + NS * 4:]
            d["bbox_corner"] = corners
            d["bbox_feature"] = d["aggregated_vote_features"]
            d["bbox_mask"], d["bbox_sems"], d["sem_cls"] = bmask, sem, sem
            return d
        nt = net.transpose(2, 1).contiguous()
        B, P = nt.shape[0], nt.shape[1]
        objectness_scores = nt[:, :, 0:2]
        center = data_dict["aggregated_vote_xyz"] + nt[:, :, 2:5]
        heading_scores = nt[:, :, 5:5 + NH]
        heading_residuals_normalized = nt[:, :, 5 + NH:5 + NH * 2]
        size_scores = nt[:, :, 5 + NH * 2:5 + NH * 2 + NS]
        size_residuals_normalized = nt[:, :, 5 + NH * 2 + NS:5 + NH * 2 + NS * 4].view(B, P, NS, 3)
        sem_cls_scores = nt[:, :, 5 + NH * 2 + NS * 4:]
        msa = self.mean_size_f32.unsqueeze(0).unsqueeze(0)
        data_dict["_proposal_net"] = nt   # the raw rows: the fused detection losses read / differentiate them directly
        data_dict["objectness_scores"] = objectness_scores
        data_dict["center"] = center
        data_dict["heading_scores"] = heading_scores
        data_dict["heading_residuals_normalized"] = heading_residuals_normalized
        data_dict["heading_residuals"] = heading_residuals_normalized * (math.pi / NH)
        data_dict["size_scores"] = size_scores
        data_dict["size_residuals_normalized"] = size_residuals_normalized
        data_dict["size_residuals"] = size_residuals_normalized * msa
        if self.size_decoded:
            size_recover = data_dict["size_residuals"] + msa
            cls = torch.argmax(size_scores, -1).unsqueeze(-1).unsqueeze(-1).expand(-1, -1, 1, 3)
            data_dict["pred_size"] = torch.gather(size_recover, 2, cls).squeeze(2)
        data_dict["sem_cls_scores"] = sem_cls_scores
        data_dict["bbox_corner"] = self.decode_pred_box(data_dict)
        data_dict["bbox_feature"] = data_dict["aggregated_vote_features"]
        data_dict["bbox_mask"] = objectness_scores.argmax(-1)
        data_dict["bbox_sems"] = sem_cls_scores.argmax(-1)
        data_dict["sem_cls"] = sem_cls_scores.argmax(-1)
        return data_dict
