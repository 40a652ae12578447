"""Vote / objectness / box / class losses as one autograd op on libspacap_hip.so (csrc/losses.hip).

Counterpart of ``compute_vote_loss``, ``compute_objectness_loss`` and ``compute_box_and_sem_cls_loss``
(lib/loss_helper.py:35-197) -- about 300 tiny PyTorch launches forward + backward -- as three launches forward and
one backward.  The unfused composition in ``spacap3d_amd/loss_helper.py`` is the specification; the tests require both
to agree on the eight loss values, the integer labels (bit-exact) and every gradient.
"""
import torch
from torch.autograd import Function

from ._native import check, lib


class DetectionLosses(Function):
    """(net (B,K,CH), center (B,K,3), vote_xyz (B,NSEED,3), <non-differentiable inputs>) ->
    (losses (8,), objectness_label (B,K) i64, objectness_mask (B,K) f32, object_assignment (B,K) i64)."""

    @staticmethod
    def forward(ctx, net, center, vote_xyz, agg_xyz, gt_center, box_mask, heading_cls_label, heading_res_label,
                size_cls_label, size_res_label, sem_cls_label, mean_size, seed_xyz, seed_inds, vote_label, vote_mask,
                NH, NS, near_thr, far_thr, w0, w1):
        dev = net.device
        net, center, vote_xyz = net.contiguous(), center.contiguous(), vote_xyz.contiguous()
        B, K, CH = net.shape
        M, NSEED, N = gt_center.shape[1], seed_xyz.shape[1], vote_label.shape[1]
        NC = CH - 5 - 2 * NH - 4 * NS
        f32 = dict(dtype=torch.float32, device=dev)
        c = lambda t: t.contiguous()
        with torch.cuda.device(dev):
            obj_label = torch.empty(B, K, dtype=torch.int64, device=dev)
            assignment = torch.empty(B, K, dtype=torch.int64, device=dev)
            obj_mask = torch.empty(B, K, **f32)
            dnet_n = torch.empty(B, K, CH, **f32)
            dcen_n = torch.empty(B, K, 6, **f32)
            dvote_n = torch.empty(B, NSEED, 3, **f32)
            part = torch.empty(B * (int(lib.spacap_det_npart()) + 2), **f32)
            losses = torch.empty(8, **f32)
            inv_den = torch.empty(4, **f32)
            args = [c(agg_xyz), c(gt_center[:, :, 0:3]), c(box_mask), c(heading_cls_label), c(heading_res_label),
                    c(size_cls_label), c(size_res_label), c(sem_cls_label), c(mean_size), c(seed_xyz)]
            si, vl, vm = c(seed_inds.to(torch.int32)), c(vote_label), c(vote_mask)
            check(lib.spacap_det_losses_fwd_f32(
                net.data_ptr(), center.data_ptr(), *[a.data_ptr() for a in args], vote_xyz.data_ptr(), si.data_ptr(),
                vl.data_ptr(), vm.data_ptr(), B, K, M, NSEED, N, int(NH), int(NS), int(NC), float(near_thr), float(far_thr),
                float(w0), float(w1), obj_label.data_ptr(), obj_mask.data_ptr(), assignment.data_ptr(), dnet_n.data_ptr(),
                dcen_n.data_ptr(), dvote_n.data_ptr(), part.data_ptr(), losses.data_ptr(), inv_den.data_ptr(),
                torch.cuda.current_stream(dev).cuda_stream), "spacap_det_losses_fwd_f32")
        ctx.save_for_backward(dnet_n, dcen_n, dvote_n, inv_den)
        ctx.meta = (B, K, NSEED, int(NH), int(NS), int(NC))
        ctx.mark_non_differentiable(obj_label, obj_mask, assignment)
        ctx.set_materialize_grads(False)   # no zero tensors for the label outputs' gradients
        return losses, obj_label, obj_mask, assignment

    @staticmethod
    def backward(ctx, g_losses, _a, _b, _c):
        dnet_n, dcen_n, dvote_n, inv_den = ctx.saved_tensors
        B, K, NSEED, NH, NS, NC = ctx.meta
        dev = dnet_n.device
        g = g_losses.contiguous()
        with torch.cuda.device(dev):
            dnet = torch.empty_like(dnet_n)
            dcenter = torch.empty(B, K, 3, dtype=torch.float32, device=dev)
            dvote = torch.empty_like(dvote_n)
            check(lib.spacap_det_losses_bwd_f32(dnet_n.data_ptr(), dcen_n.data_ptr(), dvote_n.data_ptr(), g.data_ptr(),
                                                inv_den.data_ptr(), B, K, NSEED, NH, NS, NC, dnet.data_ptr(),
                                                dcenter.data_ptr(), dvote.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream), "spacap_det_losses_bwd_f32")
        return (dnet, dcenter, dvote) + (None,) * 19


def detection_losses(d, num_heading_bin, num_size_cluster, mean_size_f32, near_thr, far_thr, w):
    """The 12-tuple ``loss_helper.start_detection_losses`` stores (last entry, box_loss: None), from the fused op.  ``d`` must hold the proposal
    head's raw output rows ``_proposal_net`` (B,K,CH) next to the decoded entries."""
    losses, label, mask, oa = DetectionLosses.apply(
        d["_proposal_net"], d["center"], d["vote_xyz"], d["aggregated_vote_xyz"], d["center_label"], d["box_label_mask"],
        d["heading_class_label"], d["heading_residual_label"], d["size_class_label"], d["size_residual_label"],
        d["sem_cls_label"], mean_size_f32, d["seed_xyz"], d["seed_inds"], d["vote_label"], d["vote_label_mask"],
        num_heading_bin, num_size_cluster, near_thr, far_thr, w[0], w[1])
    d["_det_vec"] = losses     # (8,): loss_helper's fused tail reads the vector itself
    vote, objn, center, hcls, hreg, scls, sreg, sem = losses.unbind(0)
    # box_loss (= center + 0.1 * hcls + hreg + 0.1 * scls + sreg) is left to the caller: loss_helper folds it into
    # the one matrix-vector product that also forms det_loss and the total
    return vote, objn, label, mask, oa, center, hcls, hreg, scls, sreg, sem, None


class RelationLoss(Function):
    """relation_pred (B,K,K,9) + labels -> out (7,) = (x, y, z loss, x, y, z accuracy, 1 / #pairs); differentiable in
    the three losses w.r.t. relation_pred."""

    @staticmethod
    def forward(ctx, pred, assignment, box_mask_int, obj_label, xl, yl, zl):
        dev = pred.device
        pred = pred.contiguous()
        B, K = pred.shape[0], pred.shape[1]
        M = box_mask_int.shape[1]
        with torch.cuda.device(dev):
            dnum = torch.empty_like(pred)
            part = torch.empty(int(lib.spacap_rel_loss_nparts(B, K)) * 7, dtype=torch.float32, device=dev)
            out = torch.empty(7, dtype=torch.float32, device=dev)
            check(lib.spacap_rel_loss_fwd_f32(pred.data_ptr(), assignment.contiguous().data_ptr(),
                                              box_mask_int.contiguous().data_ptr(), obj_label.contiguous().data_ptr(),
                                              xl.contiguous().data_ptr(), yl.contiguous().data_ptr(), zl.contiguous().data_ptr(),
                                              B, K, M, dnum.data_ptr(), part.data_ptr(), out.data_ptr(),
                                              torch.cuda.current_stream(dev).cuda_stream), "spacap_rel_loss_fwd_f32")
        ctx.save_for_backward(dnum, out)
        ctx.bk = (B, K)
        return out

    @staticmethod
    def backward(ctx, g):
        dnum, out = ctx.saved_tensors
        B, K = ctx.bk
        g3 = g[0:3].contiguous()
        with torch.cuda.device(dnum.device):
            dpred = torch.empty_like(dnum)
            check(lib.spacap_rel_loss_bwd_f32(dnum.data_ptr(), g3.data_ptr(), out.data_ptr(), B, K, dpred.data_ptr(),
                                              torch.cuda.current_stream(dnum.device).cuda_stream), "spacap_rel_loss_bwd_f32")
        return dpred, None, None, None, None, None, None


def relation_losses(d):
    """{x,y,z}_loss and {x,y,z}_acc as ``loss_helper.compute_relation_loss`` returns them."""
    out = RelationLoss.apply(d["relation_pred"], d["object_assignment"], d["box_label_mask_int"], d["objectness_label"],
                             d["x_label"], d["y_label"], d["z_label"])
    d["_rel_vec"] = out
    return {"x_loss": out[0], "y_loss": out[1], "z_loss": out[2], "x_acc": out[3], "y_acc": out[4], "z_acc": out[5]}


class CaptionHeadLoss(Function):
    """logits (B, W, V) -> (log-probabilities (B, W, V), out (4,) = (cap_loss, cap_acc, 1 / (sum good + 1e-6), sum good)):
    ``Generator``'s log-softmax (models/transformer_captioner.py:93-99) and ``compute_cap_loss`` (lib/loss_helper.py:199-238)
    as two launches forward and one backward (csrc/losses.hip).  The log-probabilities are returned for ``data_dict
    ["lang_cap"]`` and are not differentiable here (the training loss reaches the logits through ``out[0]`` only)."""

    @staticmethod
    def forward(ctx, logits, lang_ids, good):
        if not logits.is_cuda:
            raise RuntimeError("CPU not supported")
        logits = logits.contiguous()
        B, W, V = logits.shape
        ids = lang_ids.contiguous()
        assert ids.dtype == torch.int64 and ids.shape[0] == B and ids.shape[1] >= W + 1
        g8 = good.to(torch.uint8).contiguous()
        dev = logits.device
        with torch.cuda.device(dev):
            logp = torch.empty_like(logits)
            rowstat = torch.empty(B * W, 4, dtype=torch.float32, device=dev)
            out = torch.empty(4, dtype=torch.float32, device=dev)
            # targets = lang_ids[:, 1 : W + 1]: same rows, starting one word later
            check(lib.spacap_cap_loss_fwd_f32(logits.data_ptr(), ids.data_ptr() + 8, g8.data_ptr(), B, W, V, ids.shape[1],
                                              logp.data_ptr(), rowstat.data_ptr(), out.data_ptr(),
                                              torch.cuda.current_stream(dev).cuda_stream), "spacap_cap_loss_fwd_f32")
        ctx.save_for_backward(logp, ids, g8, out)
        ctx.mark_non_differentiable(logp)
        ctx.set_materialize_grads(False)   # no zero tensor for the log-probabilities' gradient
        return logp, out

    @staticmethod
    def backward(ctx, _g_logp, g_out):
        logp, ids, g8, out = ctx.saved_tensors
        B, W, V = logp.shape
        dev = logp.device
        gl = g_out[0:1].contiguous()
        with torch.cuda.device(dev):
            dlogits = torch.empty_like(logp)
            check(lib.spacap_cap_loss_bwd_f32(logp.data_ptr(), ids.data_ptr() + 8, g8.data_ptr(), out.data_ptr(), gl.data_ptr(),
                                              B, W, V, ids.shape[1], dlogits.data_ptr(),
                                              torch.cuda.current_stream(dev).cuda_stream), "spacap_cap_loss_bwd_f32")
        return dlogits, None, None


def caption_head_loss(logits, lang_ids, good):
    """(lang_cap log-probabilities, cap_loss, cap_acc) -- see CaptionHeadLoss."""
    logp, out = CaptionHeadLoss.apply(logits, lang_ids, good)
    return logp, out[0], out[1], out


class ProposalDecode(Function):
    """(net (B,CH,K), agg_xyz (B,K,3), mean_size (NS,3)) -> (nt (B,K,CH), center, heading_residuals, size_residuals, bbox_corner
    f64 (B,K,8,3), bbox_mask, sem_cls, size_cls): decode_scores + decode_pred_box (models/proposal_module.py:81-158) as one
    launch each way (csrc/decode.hip)."""

    @staticmethod
    def forward(ctx, net, agg_xyz, mean_size, mean_size_f64, NH, NS):
        if not net.is_cuda:
            raise RuntimeError("CPU not supported")
        net, agg, msa = net.contiguous(), agg_xyz.contiguous(), mean_size.contiguous()
        B, CH, K = net.shape
        NC = CH - 5 - 2 * NH - 4 * NS
        dev = net.device
        f32 = dict(dtype=torch.float32, device=dev)
        i64 = dict(dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            nt, center = torch.empty(B, K, CH, **f32), torch.empty(B, K, 3, **f32)
            hres, sres = torch.empty(B, K, NH, **f32), torch.empty(B, K, NS, 3, **f32)
            corners = torch.empty(B, K, 8, 3, dtype=torch.float64, device=dev)
            bm, sem, sc = torch.empty(B, K, **i64), torch.empty(B, K, **i64), torch.empty(B, K, **i64)
            check(lib.spacap_proposal_decode_fwd_f32(net.data_ptr(), agg.data_ptr(), msa.data_ptr(),
                                                     mean_size_f64.contiguous().data_ptr() if mean_size_f64 is not None else None, B, K, int(NH), int(NS), int(NC),
                                                     nt.data_ptr(), center.data_ptr(), hres.data_ptr(), sres.data_ptr(),
                                                     corners.data_ptr(), bm.data_ptr(), sem.data_ptr(), sc.data_ptr(),
                                                     torch.cuda.current_stream(dev).cuda_stream), "spacap_proposal_decode_fwd_f32")
        ctx.save_for_backward(msa)
        ctx.meta = (B, K, CH, int(NH), int(NS), int(NC))
        ctx.mark_non_differentiable(corners, bm, sem, sc)
        ctx.set_materialize_grads(False)
        return nt, center, hres, sres, corners, bm, sem, sc

    @staticmethod
    def backward(ctx, g_nt, g_center, g_hres, g_sres, *_):
        (msa,) = ctx.saved_tensors
        B, K, CH, NH, NS, NC = ctx.meta
        dev = msa.device
        c = lambda t: t.contiguous() if t is not None else None
        g_nt, g_center, g_hres, g_sres = c(g_nt), c(g_center), c(g_hres), c(g_sres)
        p = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(dev):
            d_net = torch.empty(B, CH, K, dtype=torch.float32, device=dev)
            check(lib.spacap_proposal_decode_bwd_f32(p(g_nt), p(g_center), p(g_hres), p(g_sres), msa.data_ptr(), B, K, NH, NS, NC,
                                                     d_net.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                  "spacap_proposal_decode_bwd_f32")
        return d_net, g_center, None, None, None, None


def proposal_decode(net, agg_xyz, mean_size, mean_size_f64, NH, NS):
    return ProposalDecode.apply(net, agg_xyz, mean_size, mean_size_f64, NH, NS)


class LossTail(Function):
    """(det (8,), cap (4,), rel (7,) or None, objectness_label, objectness_mask, bbox_mask) -> (loss (), out (8,)) with out =
    (box_loss, det_loss, relation_loss, loss, pos_ratio, neg_ratio, obj_acc, 0): the tail of get_scene_cap_loss
    (lib/loss_helper.py:340-383) as one launch each way.  Only ``loss`` is differentiable (``out`` is for the log entries);
    the three component ops receive dense gradient vectors."""

    @staticmethod
    def forward(ctx, det, cap, rel, obj_label, obj_mask, bbox_mask):
        dev = det.device
        with torch.cuda.device(dev):
            out = torch.empty(8, dtype=torch.float32, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            # the entry point reads raw int64 / f32 / int64 words: convert anything else instead of reading out of bounds
            ol = obj_label.to(torch.int64).contiguous().view(-1)
            om = obj_mask.to(torch.float32).contiguous().view(-1)
            bm = bbox_mask.to(torch.int64).contiguous().view(-1)
            if not (ol.numel() == om.numel() == bm.numel()):
                raise RuntimeError("loss_tail: objectness_label, objectness_mask and bbox_mask must have one entry per proposal")
            check(lib.spacap_loss_tail_fwd_f32(det.contiguous().data_ptr(), cap.contiguous().data_ptr(),
                                               rel.contiguous().data_ptr() if rel is not None else None, ol.data_ptr(),
                                               om.data_ptr(), bm.data_ptr(), ol.numel(), out.data_ptr(), loss.data_ptr(),
                                               torch.cuda.current_stream(dev).cuda_stream), "spacap_loss_tail_fwd_f32")
        ctx.has_rel = rel is not None
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)   # no zero tensor for the log entries' gradient
        return loss, out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        dev = g_loss.device
        with torch.cuda.device(dev):
            gdet = torch.empty(8, dtype=torch.float32, device=dev)
            gcap = torch.empty(4, dtype=torch.float32, device=dev)
            grel = torch.empty(7, dtype=torch.float32, device=dev) if ctx.has_rel else None
            check(lib.spacap_loss_tail_bwd_f32(g_loss.contiguous().data_ptr(), gdet.data_ptr(), gcap.data_ptr(),
                                               grel.data_ptr() if grel is not None else None,
                                               torch.cuda.current_stream(dev).cuda_stream), "spacap_loss_tail_bwd_f32")
        return gdet, gcap, grel, None, None, None


def loss_tail(det, cap, rel, obj_label, obj_mask, bbox_mask):
    return LossTail.apply(det, cap, rel, obj_label, obj_mask, bbox_mask)


class L2NormRows(Function):
    """y = x / |x|_2 over the last dimension (models/SpaCapNet.py:66-67: the vote features, no epsilon)."""

    @staticmethod
    def forward(ctx, x):
        if not x.is_cuda:
            raise RuntimeError("CPU not supported")
        x = x.contiguous()
        D = x.shape[-1]
        rows = x.numel() // D
        with torch.cuda.device(x.device):
            y = torch.empty_like(x)
            inv = torch.empty(rows, dtype=torch.float32, device=x.device)
            check(lib.spacap_l2norm_rows_fwd_f32(x.data_ptr(), rows, D, y.data_ptr(), inv.data_ptr(),
                                                 torch.cuda.current_stream(x.device).cuda_stream), "spacap_l2norm_rows_fwd_f32")
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, g):
        y, inv = ctx.saved_tensors
        g = g.contiguous()
        D = y.shape[-1]
        with torch.cuda.device(y.device):
            dx = torch.empty_like(y)
            check(lib.spacap_l2norm_rows_bwd_f32(g.data_ptr(), y.data_ptr(), inv.data_ptr(), y.numel() // D, D, dx.data_ptr(),
                                                 torch.cuda.current_stream(y.device).cuda_stream), "spacap_l2norm_rows_bwd_f32")
        return dx


def l2norm_rows(x):
    return L2NormRows.apply(x)


class VoteAssemble(Function):
    """vote_xyz, vote_features (point-major) from the voting module's last convolution, one launch each way
    (csrc/decode.hip: vote_assemble_*; models/voting_module.py:49-60 with vote_factor 1)."""

    @staticmethod
    def forward(ctx, net, seed_xyz, seed_features):
        B, CH, N = net.shape
        C = CH - 3
        dev = net.device
        net, seed_xyz, seed_features = net.contiguous(), seed_xyz.contiguous(), seed_features.contiguous()
        with torch.cuda.device(dev):
            vx = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
            vf = torch.empty(B, N, C, dtype=torch.float32, device=dev)
            check(lib.spacap_vote_assemble_fwd_f32(net.data_ptr(), seed_xyz.data_ptr(), seed_features.data_ptr(), B, C, N,
                                                   vx.data_ptr(), vf.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                  "spacap_vote_assemble_fwd_f32")
        ctx.dims = (B, C, N)
        ctx.set_materialize_grads(False)
        return vx, vf

    @staticmethod
    def backward(ctx, g_xyz, g_feat):
        B, C, N = ctx.dims
        ref = g_feat if g_feat is not None else g_xyz
        dev = ref.device
        gx = g_xyz.contiguous() if g_xyz is not None else None
        gf = g_feat.contiguous() if g_feat is not None else None
        with torch.cuda.device(dev):
            d_net = torch.empty(B, 3 + C, N, dtype=torch.float32, device=dev)
            d_seed = torch.empty(B, C, N, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
            check(lib.spacap_vote_assemble_bwd_f32(gx.data_ptr() if gx is not None else None, gf.data_ptr() if gf is not None else None,
                                                   B, C, N, d_net.data_ptr(), d_seed.data_ptr() if d_seed is not None else None,
                                                   torch.cuda.current_stream(dev).cuda_stream), "spacap_vote_assemble_bwd_f32")
        return d_net, (gx if ctx.needs_input_grad[1] else None), d_seed


def vote_assemble(net, seed_xyz, seed_features):
    return VoteAssemble.apply(net, seed_xyz, seed_features)

