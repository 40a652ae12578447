"""Seeded synthetic scenes and labels of the shapes the hot path consumes.

There is no dataset in this environment (ScanNet / ScanRefer are licensed), so benchmarks and parity
tests run on synthetic scans that keep the properties the kernels are sensitive to:

* "room" geometry so that the SA radii (0.2 / 0.4 / 0.8 / 1.2 m) select realistic neighbourhoods:
  70 % of the points on the six faces of a 6 x 6 x 3 m box, 30 % inside 12 random axis-aligned boxes;
* random point order (the dataset subsamples with ``np.random.choice``, lib/dataset.py:335);
* 1 % exact duplicates (sampling with replacement when a scan has fewer than N vertices,
  utils/pc_utils.py:35-36) -- these create exact distance ties for FPS;
* 8 points with |p|^2 <= 1e-3 to exercise the FPS skip rule (sampling_gpu.cu:100-101).

Everything is generated on the CPU from a ``torch.Generator`` so CPU and GPU runs see identical bits.
"""
import math

import torch

MAX_NUM_OBJ = 128  # lib/dataset.py:27
MAX_DES_LEN = 30   # lib/config.py:55
NUM_CLASS = 18
NUM_SIZE_CLUSTER = 18
NUM_HEADING_BIN = 1


def room_xyz(n: int, gen: torch.Generator, duplicates: float = 0.01, near_origin: int = 8) -> torch.Tensor:
    """(n, 3) float32 room-like scan, coordinates in [-3,3] x [-3,3] x [0,3]."""
    n_dup = int(n * duplicates)
    n_base = n - n_dup - near_origin
    n_face = int(n_base * 0.7)
    n_box = n_base - n_face
    ext = torch.tensor([6.0, 6.0, 3.0])
    # faces: pick an axis and a side, uniform on that face
    p = torch.rand(n_face, 3, generator=gen)
    axis = torch.randint(0, 3, (n_face,), generator=gen)
    side = torch.randint(0, 2, (n_face,), generator=gen).float()
    p[torch.arange(n_face), axis] = side
    faces = p * ext
    # furniture: 12 axis-aligned boxes, sizes 0.3 .. 1.5 m, standing on the floor
    nb = 12
    size = 0.3 + 1.2 * torch.rand(nb, 3, generator=gen)
    lo = torch.rand(nb, 3, generator=gen) * (ext - size)
    lo[:, 2] = 0.0
    which = torch.randint(0, nb, (n_box,), generator=gen)
    boxes = lo[which] + torch.rand(n_box, 3, generator=gen) * size[which]
    pts = torch.cat([faces, boxes], 0)
    pts[:, 0] -= 3.0
    pts[:, 1] -= 3.0
    tiny = (torch.rand(near_origin, 3, generator=gen) - 0.5) * 0.03  # |p|^2 <= 6.75e-4
    pts = torch.cat([pts, tiny], 0)
    dup = pts[torch.randint(0, pts.shape[0], (n_dup,), generator=gen)]
    pts = torch.cat([pts, dup], 0)
    perm = torch.randperm(pts.shape[0], generator=gen)
    return pts[perm].contiguous().float()


def uniform_xyz(n: int, gen: torch.Generator) -> torch.Tensor:
    """Pure U[0,1)^3 * (6,6,3) variant for roofline runs."""
    return (torch.rand(n, 3, generator=gen) * torch.tensor([6.0, 6.0, 3.0])).contiguous()


def num_extra_channels(use_color=False, use_normal=False, use_multiview=False, use_height=True) -> int:
    """scripts/train.py:133: input_channels = 128*multiview + 3*normal + 3*color + height."""
    return 128 * int(use_multiview) + 3 * int(use_normal) + 3 * int(use_color) + int(use_height)


def scene_batch(batch: int, n_points: int, use_color: bool = False, use_normal: bool = False,
                use_multiview: bool = False, use_height: bool = True, seed: int = 0) -> torch.Tensor:
    """point_clouds (B, N, 3 + C) in the channel order of lib/dataset.py:309-333:
    xyz, [rgb 3], [normal 3], [multiview 128], [height 1]."""
    gen = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(batch):
        xyz = room_xyz(n_points, gen)
        feats = [xyz]
        if use_color:
            feats.append(torch.rand(n_points, 3, generator=gen) - 0.4)
        if use_normal:
            nrm = torch.randn(n_points, 3, generator=gen)
            feats.append(nrm / nrm.norm(dim=1, keepdim=True).clamp_min(1e-6))
        if use_multiview:
            feats.append(torch.rand(n_points, 128, generator=gen))
        if use_height:
            z = xyz[:, 2]
            floor = torch.quantile(z, 0.0099)  # np.percentile(z, 0.99), lib/dataset.py:331
            feats.append((z - floor).unsqueeze(1))
        out.append(torch.cat(feats, 1))
    return torch.stack(out, 0).contiguous().float()


def mean_size_arr(seed: int = 1234) -> torch.Tensor:
    """Stand-in for data/scannet/meta_data/scannet_reference_means.npz (18 x 3 positive sizes)."""
    gen = torch.Generator().manual_seed(seed)
    return 0.3 + 1.5 * torch.rand(NUM_SIZE_CLUSTER, 3, generator=gen)


def labels(batch: int, n_points: int, vocab: int = 3001, seed: int = 0, n_gt: int = 32) -> dict:
    """Training labels with the keys / shapes / dtypes of lib/dataset.py:498-527."""
    gen = torch.Generator().manual_seed(seed + 7919)
    d = {}
    center = torch.zeros(batch, MAX_NUM_OBJ, 3)
    c = torch.rand(batch, n_gt, 3, generator=gen) * torch.tensor([5.0, 5.0, 2.0]) + torch.tensor([-2.5, -2.5, 0.3])
    center[:, :n_gt] = c
    d["center_label"] = center
    mask = torch.zeros(batch, MAX_NUM_OBJ)
    mask[:, :n_gt] = 1
    d["box_label_mask"] = mask
    d["box_label_mask_int"] = mask.long()
    d["heading_class_label"] = torch.zeros(batch, MAX_NUM_OBJ, dtype=torch.long)
    d["heading_residual_label"] = torch.zeros(batch, MAX_NUM_OBJ)
    scls = torch.zeros(batch, MAX_NUM_OBJ, dtype=torch.long)
    scls[:, :n_gt] = torch.randint(0, NUM_SIZE_CLUSTER, (batch, n_gt), generator=gen)
    d["size_class_label"] = scls
    sres = torch.zeros(batch, MAX_NUM_OBJ, 3)
    sres[:, :n_gt] = (torch.rand(batch, n_gt, 3, generator=gen) - 0.5) * 0.4
    d["size_residual_label"] = sres
    sem = torch.zeros(batch, MAX_NUM_OBJ, dtype=torch.long)
    sem[:, :n_gt] = torch.randint(0, NUM_CLASS, (batch, n_gt), generator=gen)
    d["sem_cls_label"] = sem
    d["vote_label"] = (torch.rand(batch, n_points, 9, generator=gen) - 0.5) * 1.0
    d["vote_label_mask"] = (torch.rand(batch, n_points, generator=gen) < 0.4).long()
    lang = torch.zeros(batch, MAX_DES_LEN + 3, dtype=torch.long)  # sos + <=30 words + eos (+pad): (33,)
    for b in range(batch):
        length = int(torch.randint(8, MAX_DES_LEN + 1, (1,), generator=gen))
        lang[b, 0] = 2  # sos
        lang[b, 1:1 + length] = torch.randint(4, vocab, (length,), generator=gen)
        lang[b, 1 + length] = 3  # eos
    d["lang_ids"] = lang
    d["lang_label"] = lang.clone()
    for ax in "xyz":
        d[f"{ax}_label"] = torch.randint(0, 3, (batch, MAX_NUM_OBJ, MAX_NUM_OBJ), generator=gen)
    pick = torch.randint(0, n_gt, (batch,), generator=gen)
    d["ref_center_label"] = center[torch.arange(batch), pick]
    return d


def anchor_boxes_on_proposals(lab: dict, aggregated_vote_xyz: torch.Tensor, n_gt: int = 32, seed: int = 3) -> dict:
    """Move the first ``n_gt`` ground-truth box centres onto (0.05 m jitter) randomly chosen proposal positions
    ``aggregated_vote_xyz`` (B, P, 3) of a probe forward, and the reference object onto the first of them.  With the
    unrelated random centres of ``labels`` almost no proposal is a positive (objectness_label, lib/loss_helper.py:
    NEAR_THRESHOLD 0.3), so the box / class / relation losses are means over 0-3 items that switch on and off from step
    to step; anchored boxes give every step ~n_gt positives per scene (what the fixtures and trajectory tests use)."""
    agg = aggregated_vote_xyz.detach().cpu()
    B, P, _ = agg.shape
    n_gt = min(n_gt, P)
    g = torch.Generator().manual_seed(seed)
    pick = torch.stack([torch.randperm(P, generator=g)[:n_gt] for _ in range(B)])
    ctr = torch.gather(agg, 1, pick.unsqueeze(-1).expand(-1, -1, 3)) + 0.05 * torch.randn(B, n_gt, 3, generator=g)
    out = dict(lab)
    c = lab["center_label"].detach().cpu().clone()
    c[:, :n_gt] = ctr
    out["center_label"] = c.to(lab["center_label"].device)
    out["ref_center_label"] = ctr[:, 0].clone().to(lab["ref_center_label"].device)
    return out


def make_vocabulary(vocab: int = 3001) -> dict:
    words = ["pad_", "unk", "sos", "eos"] + [f"w{i}" for i in range(vocab - 4)]
    return {"word2idx": {w: i for i, w in enumerate(words)}, "idx2word": {str(i): w for i, w in enumerate(words)}}


def attention_inputs(B: int, h: int, Lq: int, Lk: int, d_k: int, seed: int = 0, scale: float = 1.0):
    gen = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Lq, h * d_k, generator=gen) * scale
    k = torch.randn(B, Lk, h * d_k, generator=gen) * scale
    v = torch.randn(B, Lk, h * d_k, generator=gen)
    return q, k, v


__all__ = ["anchor_boxes_on_proposals", "room_xyz", "uniform_xyz", "scene_batch", "num_extra_channels", "labels", "mean_size_arr", "make_vocabulary",
           "attention_inputs", "MAX_NUM_OBJ", "MAX_DES_LEN", "NUM_CLASS", "NUM_SIZE_CLUSTER", "NUM_HEADING_BIN"]
_ = math
