"""Training loss of the scene captioner -- counterpart of ``lib/loss_helper.py`` (+ ``utils/nn_distance.py``).

Same terms, weights and ``data_dict`` keys as ``get_scene_cap_loss`` (lib/loss_helper.py:291-385):
    loss = 10 * (vote + 0.5 * objectness + box + 0.1 * sem_cls) + caption + 0.1 * relation
Restated so that one training step needs NO host synchronisation:
  * no hard-coded ``.cuda()`` (the reference allocates with it, :94-95,102,158,175,181);
  * ``nn_distance`` broadcasts instead of materialising two ``repeat`` copies (nn_distance.py:49-51);
  * the relation loss (:240-289) selects pairs (i, j) of "positive, GT-valid" proposals with boolean-mask
    indexing, i.e. data-dependent shapes; here the same mean is a dense pair-weighted sum
    sum(W * CE) / sum(W) with W[b,i,j] = sel[b,i] * sel[b,j]  (identical value; when no pair is selected the
    reference yields NaN, this yields 0);
  * caption accuracy / ious use masked sums instead of ``if num_good > 0`` branches (:223-237).
"""
import math

import os
import torch
import torch.nn.functional as F

FAR_THRESHOLD = 0.6
NEAR_THRESHOLD = 0.3
GT_VOTE_FACTOR = 3
OBJECTNESS_CLS_WEIGHTS = [0.2, 0.8]


_CONST = {}


# lab switch (same-box A/B): the fused relation loss on the relation head's stream when the head was forked
REL_LOSS_ON_ITS_STREAM = os.environ.get("SPACAP_REL_LOSS_FORKED", "1") != "0"

def _const(key, device, build):
    """Device-resident constants are built once per device (a host -> device copy is not allowed while a
    hipGraph is being captured; the eager warm-up step populates this cache)."""
    k = (key, str(device))
    t = _CONST.get(k)
    if t is None:
        t = build().to(device)
        _CONST[k] = t
    return t


def huber_loss(error, delta=1.0):
    abs_error = torch.abs(error)
    quadratic = torch.clamp(abs_error, max=delta)
    linear = abs_error - quadratic
    return 0.5 * quadratic ** 2 + delta * linear


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    """pc1 (B,N,C), pc2 (B,M,C) -> dist1 (B,N), idx1 (B,N), dist2 (B,M), idx2 (B,M)  (nn_distance.py:32-62)."""
    diff = pc1.unsqueeze(2) - pc2.unsqueeze(1)  # (B,N,M,C)
    if l1smooth:
        pc_dist = torch.sum(huber_loss(diff, delta), dim=-1)
    elif l1:
        pc_dist = torch.sum(torch.abs(diff), dim=-1)
    else:
        pc_dist = torch.sum(diff ** 2, dim=-1)
    dist1, idx1 = torch.min(pc_dist, dim=2)
    dist2, idx2 = torch.min(pc_dist, dim=1)
    return dist1, idx1, dist2, idx2


def compute_vote_loss(d):
    B, num_seed = d["seed_xyz"].shape[0], d["seed_xyz"].shape[1]
    vote_xyz = d["vote_xyz"]
    seed_inds = d["seed_inds"].long()
    seed_gt_votes_mask = torch.gather(d["vote_label_mask"], 1, seed_inds)
    seed_inds_expand = seed_inds.view(B, num_seed, 1).expand(-1, -1, 3 * GT_VOTE_FACTOR)
    seed_gt_votes = torch.gather(d["vote_label"], 1, seed_inds_expand) + d["seed_xyz"].repeat(1, 1, 3)
    vote_xyz_reshape = vote_xyz.view(B * num_seed, -1, 3)
    seed_gt_votes_reshape = seed_gt_votes.view(B * num_seed, GT_VOTE_FACTOR, 3)
    _, _, dist2, _ = nn_distance(vote_xyz_reshape, seed_gt_votes_reshape, l1=True)
    votes_dist, _ = torch.min(dist2, dim=1)
    votes_dist = votes_dist.view(B, num_seed)
    m = seed_gt_votes_mask.float()
    return torch.sum(votes_dist * m) / (torch.sum(m) + 1e-6)


def compute_objectness_loss(d):
    agg = d["aggregated_vote_xyz"]
    gt_center = d["center_label"][:, :, 0:3]
    dist1, ind1, _, _ = nn_distance(agg, gt_center)
    euclid = torch.sqrt(dist1 + 1e-6)
    near = euclid < NEAR_THRESHOLD
    objectness_label = near.long()
    objectness_mask = (near | (euclid > FAR_THRESHOLD)).float()
    w = _const("objectness_w", agg.device, lambda: torch.tensor(OBJECTNESS_CLS_WEIGHTS, dtype=torch.float32))
    loss = F.cross_entropy(d["objectness_scores"].transpose(2, 1), objectness_label, weight=w, reduction="none")
    loss = torch.sum(loss * objectness_mask) / (torch.sum(objectness_mask) + 1e-6)
    return loss, objectness_label, objectness_mask, ind1


def compute_box_and_sem_cls_loss(d, num_heading_bin, num_size_cluster, mean_size_arr):
    oa = d["object_assignment"]
    pred_center = d["center"]
    gt_center = d["center_label"][:, :, 0:3]
    dist1, _, dist2, _ = nn_distance(pred_center, gt_center)
    box_label_mask = d["box_label_mask"]
    obj = d["objectness_label"].float()
    n_obj = torch.sum(obj) + 1e-6
    center_loss = torch.sum(dist1 * obj) / n_obj + torch.sum(dist2 * box_label_mask) / (torch.sum(box_label_mask) + 1e-6)

    heading_class_label = torch.gather(d["heading_class_label"], 1, oa)
    heading_class_loss = F.cross_entropy(d["heading_scores"].transpose(2, 1), heading_class_label, reduction="none")
    heading_class_loss = torch.sum(heading_class_loss * obj) / n_obj
    heading_residual_label = torch.gather(d["heading_residual_label"], 1, oa)
    heading_residual_normalized_label = heading_residual_label / (math.pi / num_heading_bin)
    heading_one_hot = F.one_hot(heading_class_label, num_heading_bin).float()
    hr = huber_loss(torch.sum(d["heading_residuals_normalized"] * heading_one_hot, -1)
                    - heading_residual_normalized_label, delta=1.0)
    heading_reg_loss = torch.sum(hr * obj) / n_obj

    size_class_label = torch.gather(d["size_class_label"], 1, oa)
    size_class_loss = F.cross_entropy(d["size_scores"].transpose(2, 1), size_class_label, reduction="none")
    size_class_loss = torch.sum(size_class_loss * obj) / n_obj
    size_residual_label = torch.gather(d["size_residual_label"], 1, oa.unsqueeze(-1).expand(-1, -1, 3))
    size_one_hot = F.one_hot(size_class_label, num_size_cluster).float().unsqueeze(-1)  # (B,K,NS,1)
    pred_size_res_norm = torch.sum(d["size_residuals_normalized"] * size_one_hot, 2)
    msa = _const(("msa", id(mean_size_arr)), pred_center.device,
                 lambda: torch.as_tensor(mean_size_arr, dtype=torch.float32).clone()).unsqueeze(0).unsqueeze(0)
    mean_size_label = torch.sum(size_one_hot * msa, 2)
    size_residual_label_normalized = size_residual_label / mean_size_label
    sr = torch.mean(huber_loss(pred_size_res_norm - size_residual_label_normalized, delta=1.0), -1)
    size_reg_loss = torch.sum(sr * obj) / n_obj

    sem_cls_label = torch.gather(d["sem_cls_label"], 1, oa)
    sem_cls_loss = F.cross_entropy(d["sem_cls_scores"].transpose(2, 1), sem_cls_label, reduction="none")
    sem_cls_loss = torch.sum(sem_cls_loss * obj) / n_obj
    return center_loss, heading_class_loss, heading_reg_loss, size_class_loss, size_reg_loss, sem_cls_loss


def compute_cap_loss(d):
    pred_caps = d["lang_cap"]  # (B, num_words, V) log-probs
    num_words, V = pred_caps.size(1), pred_caps.size(2)
    target_caps = d["lang_ids"][:, 1:num_words + 1]
    assert pred_caps.shape[0:2] == target_caps.shape[0:2]
    cap_loss = F.cross_entropy(pred_caps.reshape(-1, V), target_caps.reshape(-1), ignore_index=0, reduction="none")
    good = d["good_bbox_masks"].unsqueeze(1).expand(-1, num_words).reshape(-1).float()
    cap_loss = torch.sum(cap_loss * good) / (torch.sum(good) + 1e-6)
    # accuracy over non-pad words of good boxes (0 when there are none)
    hit = (pred_caps.argmax(-1) == target_caps).reshape(-1).float()
    valid = (target_caps.reshape(-1) != 0).float() * good
    cap_acc = torch.sum(hit * valid) / torch.sum(valid).clamp(min=1.0)
    return cap_loss, cap_acc


def compute_relation_loss(d):
    oa = d["object_assignment"]
    B, K = oa.shape
    M = d["y_label"].shape[1]
    sel = (torch.gather(d["box_label_mask_int"], 1, oa) & d["objectness_label"]).float()  # (B,K)
    W = sel.unsqueeze(2) * sel.unsqueeze(1)                                             # (B,K,K)
    n = W.sum().clamp(min=1.0)
    out = {}
    rows = oa.unsqueeze(-1).expand(-1, -1, M)
    cols = oa.unsqueeze(-2).expand(-1, K, -1)
    for a, sl in (("x", slice(0, 3)), ("y", slice(3, 6)), ("z", slice(6, 9))):
        lab = torch.gather(torch.gather(d[f"{a}_label"], 1, rows), 2, cols)          # (B,K,K)
        pred = d["relation_pred"][..., sl]
        ce = F.cross_entropy(pred.reshape(-1, 3), lab.reshape(-1), reduction="none").view(B, K, K)
        out[f"{a}_loss"] = (ce * W).sum() / n
        out[f"{a}_acc"] = ((pred.argmax(-1) == lab).float() * W).sum() / n
    return out


def _with(d, object_assignment, objectness_label):
    """``d`` plus the two labels the box loss reads (they are written into ``d`` right after)."""
    d["object_assignment"] = object_assignment
    d["objectness_label"] = objectness_label
    return d


def start_detection_losses(d, num_heading_bin=1, num_size_cluster=18, mean_size_arr=None):
    """Vote / objectness / box / class losses (lib/loss_helper.py:291-345): everything that only needs the proposal
    module's outputs.  The engine calls this right after the proposal module; ``get_scene_cap_loss`` picks the result up."""
    from .backend import ops
    fused = getattr(ops(), "detection_losses", None) if d["seed_xyz"].is_cuda else None
    if (fused is not None and "_proposal_net" in d and d["vote_xyz"].shape[1] == d["seed_xyz"].shape[1]
            and d["center_label"].shape[1] <= 256):
        # one autograd op on the HIP library (fused_losses.py): 3 launches forward, 1 backward
        msa = _const(("msa", id(mean_size_arr)), d["seed_xyz"].device,
                     lambda: torch.as_tensor(mean_size_arr, dtype=torch.float32).clone())
        t = fused(d, num_heading_bin, num_size_cluster, msa, NEAR_THRESHOLD, FAR_THRESHOLD, OBJECTNESS_CLS_WEIGHTS)
        d["object_assignment"], d["objectness_label"] = t[4], t[2]
        d["_detection_losses"] = t
        return
    vote_loss = compute_vote_loss(d)
    objectness_loss, objectness_label, objectness_mask, object_assignment = compute_objectness_loss(d)
    center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss, size_reg_loss, sem_cls_loss = \
        compute_box_and_sem_cls_loss(_with(d, object_assignment, objectness_label), num_heading_bin,
                                     num_size_cluster, mean_size_arr)
    box_loss = center_loss + 0.1 * heading_cls_loss + heading_reg_loss + 0.1 * size_cls_loss + size_reg_loss
    d["_detection_losses"] = (vote_loss, objectness_loss, objectness_label, objectness_mask, object_assignment,
                              center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss, size_reg_loss,
                              sem_cls_loss, box_loss)


def get_scene_cap_loss(data_dict, device=None, config=None, detection=True, caption=True, use_relation=False,
                       num_heading_bin=1, num_size_cluster=18, mean_size_arr=None):
    """Mutates and returns ``data_dict`` with every key the reference writes (lib/loss_helper.py:291-385).
    ``config`` may be an object with num_heading_bin / num_size_cluster / mean_size_arr (the reference's DC)."""
    d = data_dict
    if config is not None:
        num_heading_bin, num_size_cluster, mean_size_arr = (config.num_heading_bin, config.num_size_cluster,
                                                            config.mean_size_arr)
    dev = d["seed_xyz"].device
    zero = _const("zero", dev, lambda: torch.zeros(()))

    det_early = "_detection_losses" in d    # computed right after the proposal module (engine.Trainer.loss): before the captioner
    if not det_early:
        start_detection_losses(d, num_heading_bin, num_size_cluster, mean_size_arr)
    (vote_loss, objectness_loss, objectness_label, objectness_mask, object_assignment, center_loss, heading_cls_loss,
     heading_reg_loss, size_cls_loss, size_reg_loss, sem_cls_loss, box_loss) = d.pop("_detection_losses")
    # every derived sum in one matrix-vector product at the end (see below)
    fast = detection and caption and use_relation and vote_loss.is_cuda
    if box_loss is None and not fast:   # the fused detection-loss op leaves the box sum to this function
        box_loss = center_loss + 0.1 * heading_cls_loss + heading_reg_loss + 0.1 * size_cls_loss + size_reg_loss
    if caption:
        pair = d.pop("_cap_loss", None)   # computed with the vocabulary log-softmax by the fused caption head (HIP backend)
        d["cap_loss"], d["cap_acc"] = pair if pair is not None else compute_cap_loss(d)
    else:
        d["cap_loss"], d["cap_acc"], d["pred_ious"] = zero, zero, zero
    total = objectness_label.shape[0] * objectness_label.shape[1]
    d["objectness_label"] = objectness_label
    d["objectness_mask"] = objectness_mask
    d["object_assignment"] = object_assignment
    # every derived scalar (box / det / relation / total loss and the three ratios below) in ONE launch each way when the three
    # fused component ops left their result vectors (fused_losses.LossTail); else the composition that follows
    # (the three private vectors never stay in the returned dict, whichever path runs: they are autograd tensors that would
    # keep the graph alive, and a reused dict must not hand a stale one to a later call)
    det_vec, cap_vec = d.pop("_det_vec", None), d.pop("_cap_vec", None)
    d.pop("_rel_vec", None)
    tail = None
    if fast and det_vec is not None and cap_vec is not None:
        from .backend import ops as _ops
        tail = getattr(_ops(), "loss_tail", None)
    if tail is None:
        d["pos_ratio"] = torch.sum(objectness_label.float()) / float(total)
        d["neg_ratio"] = torch.sum(objectness_mask) / float(total) - d["pos_ratio"]
        d["obj_acc"] = torch.sum((d["bbox_mask"] == objectness_label).float() * objectness_mask) / (
            torch.sum(objectness_mask) + 1e-6)

    rel_stream = d.pop("_rel_stream", None)   # (the relation head ran on a stream of its own: TransformerDecoderModel.fork_relation)
    cur_stream = torch.cuda.current_stream(d["relation_pred"].device) if rel_stream is not None else None
    if use_relation:
        from .backend import ops
        frel = getattr(ops(), "relation_losses", None) if d["relation_pred"].is_cuda else None
        if rel_stream is not None and frel is not None and det_early and REL_LOSS_ON_ITS_STREAM:
            # The relation loss stays on the head's stream: its forward (0.03 ms) then runs beside the decoder instead of after it,
            # and its backward is followed by the head's backward kernel on the same stream without a cross-stream wait.  Everything
            # it reads besides relation_pred (labels, the detection losses' object assignment) exists since before the fork.
            for k in ("object_assignment", "box_label_mask_int", "objectness_label", "x_label", "y_label", "z_label"):
                if torch.is_tensor(d.get(k)) and d[k].is_cuda:
                    d[k].record_stream(rel_stream)
            with torch.cuda.stream(rel_stream):
                rel = frel(d)
            cur_stream.wait_stream(rel_stream)
            d["_rel_vec"].record_stream(cur_stream)
        else:
            if rel_stream is not None:
                cur_stream.wait_stream(rel_stream)
            rel = frel(d) if frel is not None else compute_relation_loss(d)
        d.update(rel)
        if not fast:
            d["relation_loss"] = rel["y_loss"] + rel["z_loss"] + rel["x_loss"]
    else:
        if rel_stream is not None:
            cur_stream.wait_stream(rel_stream)
        for k in ("x_loss", "y_loss", "z_loss", "relation_loss", "x_acc", "y_acc", "z_acc"):
            d[k] = zero

    names = ("vote_loss", "objectness_loss", "center_loss", "heading_cls_loss", "heading_reg_loss",
             "size_cls_loss", "size_reg_loss", "sem_cls_loss", "box_loss")
    vals = (vote_loss, objectness_loss, center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss,
            size_reg_loss, sem_cls_loss, box_loss)
    for k, v in zip(names, vals):
        d[k] = v if detection else zero
    if not detection:
        d["det_loss"] = zero

    rel_vec = d.pop("_rel_vec", None)
    if tail is not None and rel_vec is not None:
        d["loss"], out = tail(det_vec, cap_vec, rel_vec, objectness_label, objectness_mask, d["bbox_mask"])
        d["box_loss"], d["det_loss"], d["relation_loss"] = out[0], out[1], out[2]
        d["pos_ratio"], d["neg_ratio"], d["obj_acc"] = out[4], out[5], out[6]
        return d
    if tail is not None:   # (the relation op did not leave its vector: finish the ratios the composition way)
        d["pos_ratio"] = torch.sum(objectness_label.float()) / float(total)
        d["neg_ratio"] = torch.sum(objectness_mask) / float(total) - d["pos_ratio"]
        d["obj_acc"] = torch.sum((d["bbox_mask"] == objectness_label).float() * objectness_mask) / (
            torch.sum(objectness_mask) + 1e-6)
    if fast:
        # box_loss, det_loss, relation_loss and the total (lib/loss_helper.py:340-383) as ONE 4 x 12 matrix-vector
        # product over the stacked terms instead of ~17 scalar kernels forward and as many backward
        t = torch.stack([d[k] for k in _TERMS])
        box, det, rel_sum, total_loss = (_const("loss_matrix", dev, _loss_matrix) @ t).unbind(0)
        d["box_loss"], d["det_loss"], d["relation_loss"], d["loss"] = box, det, rel_sum, total_loss
        return d
    loss = 0
    if detection:
        d["det_loss"] = d["vote_loss"] + 0.5 * d["objectness_loss"] + d["box_loss"] + 0.1 * d["sem_cls_loss"]
        loss = loss + 10 * d["det_loss"]
    if caption:
        loss = loss + d["cap_loss"]
    if use_relation:
        loss = loss + 0.1 * d["relation_loss"]
    d["loss"] = loss
    return d


_TERMS = ("vote_loss", "objectness_loss", "center_loss", "heading_cls_loss", "heading_reg_loss", "size_cls_loss",
          "size_reg_loss", "sem_cls_loss", "cap_loss", "x_loss", "y_loss", "z_loss")


def _loss_matrix():
    """rows: box_loss (lib/loss_helper.py:340), det_loss (:372), relation_loss (:366), loss (:373-383) over _TERMS."""
    box = torch.tensor([0, 0, 1, 0.1, 1, 0.1, 1, 0, 0, 0, 0, 0], dtype=torch.float64)
    det = box + torch.tensor([1, 0.5, 0, 0, 0, 0, 0, 0.1, 0, 0, 0, 0], dtype=torch.float64)
    rel = torch.tensor([0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1], dtype=torch.float64)
    cap = torch.tensor([0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0], dtype=torch.float64)
    return torch.stack([box, det, rel, 10 * det + cap + 0.1 * rel]).float()
