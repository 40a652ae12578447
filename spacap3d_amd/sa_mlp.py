"""Training-mode shared MLP of a set-abstraction module on libspacap_hip.so, point-major layout.

One autograd op for what the reference runs per SA module as QueryAndGroup -> SharedMLP -> max_pool2d
(lib/pointnet2/pointnet2_modules.py:241-259; SharedMLP = [Conv2d 1x1 -> BatchNorm2d -> ReLU] x 3,
lib/pointnet2/pytorch_utils.py:11-36).  See ``csrc/sa_mlp.hip`` for the kernels and the data flow.
The grouping indices come from ``ball_query`` as before; eval mode and MLP shapes without kernels keep using
the per-operator path (``QueryAndGroup`` + ``SharedMLP``), which is also HIP.
"""
import os

import torch
from torch.autograd import Function

from .layout import ChannelMajorOf, point_major_of
from ._native import check, lib, sum_slabs


# SA1-shaped modules (no point features, 64 -> 64 -> ...): do not store the first layer's pre-activation (see _SAMLP.forward)
RECOMPUTE_Z1 = True
# ... and take that layer's BatchNorm statistics from the first and second moments of the rows' four inputs (z1 is linear in
# them): 14 sums per row, one thread per row, instead of 2 x 64 sums with 16 threads per row (csrc/sa_mlp.hip:
# sa_l1_moments_kernel; 87 -> ~20 us at SA1).  The statistics differ from the summed-z1 form by rounding only (~1e-7 relative).
L1_MOMENTS = os.environ.get("SPACAP_SA_L1_MOMENTS", "1") not in ("", "0")
# pooled last layer: its WEIGHT gradient from z2 alone (csrc/sa_l3bwd.inc: sa_wgrad_pool_kernel -- the sparse
# term (g d)^T a2 plus the Gram matrix a2^T a2) instead of the dense kernel that streams z3 and z2 and multiplies a [C3 x rows]
# operand with one non-zero per group and channel.  SA1: 92 + 12 us against 249 + 12 us (tools/lab/wgrad_pool_bench.py).  Used
# from POOL_WGRAD_MIN_ROWS rows on (below, the two extra launches of the reduction cost more than the pass saves).
POOL_WGRAD = os.environ.get("SPACAP_SA_POOL_WGRAD", "1") not in ("", "0")
POOL_WGRAD_MIN_ROWS = 131072
# SA1's second layer: weight gradient from the data-gradient kernel's pass (tests / lab: SPACAP_SA_FUSE_L2_WGRAD=0 keeps the two
# kernels apart)
FUSE_L2_WGRAD = os.environ.get("SPACAP_SA_FUSE_L2_WGRAD", "1") not in ("", "0")


def _ptr(t):
    return t.data_ptr() if t is not None else None


_SELFTESTED = set()


def _selftest(dev):
    """Once per process and device, before the first fused SA op: the streaming split-bf16 kernels (csrc/sa_bf3.inc,
    sa_bf3_dgrad.inc) land their prefetched rows in AGPRs that are named inside inline asm and therefore invisible to the
    register allocator.  The build checks the generated assembly for compiler uses of those registers
    (tools/check_landing_regs.py); this is the matching check at RUN time, on the library that was actually loaded: a few
    tiles through the forward layer kernel and the data-gradient kernel against float64 -- a clobbered landing register gives
    errors of order one, the bar is 1e-4 of the result's scale.  Raises instead of training on garbage."""
    key = (dev.type, dev.index)
    if key in _SELFTESTED:
        return
    _SELFTESTED.add(key)
    with torch.cuda.device(dev), torch.no_grad():
        st = torch.cuda.current_stream(dev).cuda_stream
        if torch.cuda.is_current_stream_capturing():
            _SELFTESTED.discard(key)   # (never inside a graph capture: it synchronises)
            return
        g = torch.Generator(device="cpu").manual_seed(1234)
        R, ci, co = 49152 + 96, 128, 128
        zin, W = torch.randn(R, ci, generator=g).to(dev), (0.1 * torch.randn(co, ci, generator=g)).to(dev)
        stats = torch.tensor([0.05, 1.0, 1.1, 0.02], device=dev).repeat(ci, 1).contiguous()
        zout = torch.empty(R, co, dtype=torch.float32, device=dev)
        part = torch.empty(int(lib.spacap_sa_nparts()) * 2 * max(ci, co), dtype=torch.float64, device=dev)
        check(lib.spacap_sa_mid_fwd_f32(zin.data_ptr(), stats.data_ptr(), W.data_ptr(), R, ci, co, zout.data_ptr(), part.data_ptr(), st),
              "spacap_sa_mid_fwd_f32 (self-test)")
        a = ((zin.double() - 0.05) * 1.1 + 0.02).clamp_min(0)
        ref = a @ W.double().t()
        e1 = float((zout.double() - ref).abs().max() / ref.abs().max())
        # data gradient: dy_prev = (dz W) * [relu(bn(z_prev)) > 0], dz = g dy + k0 - k1 z
        dy, zk = torch.randn(R, co, generator=g).to(dev), torch.randn(R, co, generator=g).to(dev)
        coef = torch.tensor([1.05, 0.01, 0.02, 0.0], device=dev).repeat(co, 1).contiguous()
        dyp = torch.empty(R, ci, dtype=torch.float32, device=dev)
        check(lib.spacap_sa_dgrad_f32(dy.data_ptr(), None, 0, zk.data_ptr(), coef.data_ptr(), W.data_ptr(), zin.data_ptr(), stats.data_ptr(),
                                      R, co, ci, dyp.data_ptr(), part.data_ptr(), st), "spacap_sa_dgrad_f32 (self-test)")
        dz = 1.05 * dy.double() + 0.01 - 0.02 * zk.double()
        pre = (zin.double() - 0.05) * 1.1 + 0.02
        ref2 = (dz @ W.double()) * (pre > 0)
        near = pre.abs() < 1e-6
        e2 = float(((dyp.double() - ref2).abs() * (~near)).max() / ref2.abs().max())
    if not (e1 < 1e-4 and e2 < 1e-4):
        raise RuntimeError(f"libspacap_hip.so self-test failed: streaming shared-MLP kernels differ from float64 by {e1:.2e} (forward) / "
                           f"{e2:.2e} (data gradient) of the result's scale -- the landing registers of csrc/sa_bf3*.inc are not safe "
                           f"in this build (tools/check_landing_regs.py); set SPACAP_SA_F32MFMA=1 to run the fp32-MFMA kernels")


class _SAMLP(Function):
    """inputs: xyz (B,Np,3), new_xyz (B,N,3), idx (B,N,S) int32, feat (B,Np) or None [inline 1-channel feature],
    pm (B,Np,Cf) or None [point-major features of the source points: the first layer commutes with the gather, so
    Y = pm W1[:, 3:]^T is formed once per SOURCE point], W1 (C1, 3 + Cf) the FULL first-layer weight (the kernels read
    its first 3 / 4 columns through the row stride), W2 (C2,C1), W3 (C3,C2), gamma/beta x3, then the three BatchNorm
    modules (running statistics are updated in place) and rdiv.  output: (B,N,C3) pooled features.
    One node for the whole module: W1's gradient is assembled once ([xyz columns | feature columns])."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, idx, feat, pm, W1, W2, W3, g1, b1, g2, b2, g3, b3, bns, rdiv, rows_index=None):
        dev = xyz.device
        _selftest(dev)
        B, Np, _ = xyz.shape
        N, S = idx.shape[1], idx.shape[2]
        C1, C2, C3 = W1.shape[0], W2.shape[0], W3.shape[0]
        R, G = B * N * S, B * N
        st = torch.cuda.current_stream(dev).cuda_stream
        f32 = dict(dtype=torch.float32, device=dev)
        xyz, new_xyz, idx = xyz.contiguous(), new_xyz.contiguous(), idx.contiguous()
        W1c, W2c, W3c = W1.contiguous(), W2.contiguous(), W3.contiguous()
        feat = feat.contiguous() if feat is not None else None
        # first layer commuted with the gather: Y = F W1[:, 3:]^T over the SOURCE points (csrc/dense_rows.hip reads the column
        # slice of W1 in place)
        # (rows of Cf contiguous floats at a uniform stride are read in place: the input features of a (B, N, 3 + C) cloud are a
        # column window of its rows -- no transposed / gathered copy of the 170 MB feature block of BASELINE config 4)
        pmc = pm if (pm is None or _uniform_rows(pm)) else pm.contiguous()
        Y = _feature_product(pmc, W1c) if pm is not None else None
        nparts = int(lib.spacap_sa_nparts())
        with torch.cuda.device(dev):
            part = torch.empty(nparts * 2 * max(C1, C2, C3), dtype=torch.float64, device=dev)
            stats = [torch.empty(c, 4, **f32) for c in (C1, C2, C3)]
            # SA1 (no point features, 64 -> 64): z1 never exists in HBM; the statistics pass leaves the rows' four inputs
            # (16 bytes per row) and every later pass rebuilds z1 from them (csrc/sa_mlp.hip: L1In)
            recompute = RECOMPUTE_Z1 and Y is None and C1 == 64 and C2 == 64 and not (xyz.requires_grad or new_xyz.requires_grad)
            has_feat = int(feat is not None)
            z1 = torch.empty(R, 4 if recompute else C1, **f32)
            z2 = torch.empty(R, C2, **f32)
            pool_fused = bool(lib.spacap_sa_mid_fwd_pool_supported(C2, C3, S))
            z3 = torch.empty(R, C3, **f32)
            # (the arg-max rows' pre-activations: the pooled layer's BatchNorm sums in the backward read them instead of
            # gathering 4 bytes per element out of z3)
            zmax = torch.empty(B, N, C3, **f32) if pool_fused else None

            def finalize(k, C, gamma, beta):
                bn = bns[k]
                mom = 0.0 if bn.momentum is None else float(bn.momentum)
                track = bn.track_running_stats and bn.running_mean is not None
                if track:
                    from .fused_bn import bump_counter
                    bump_counter(bn)
                check(lib.spacap_sa_bn_finalize_f32(part.data_ptr(), C, R, float(bn.eps), mom, gamma.data_ptr(),
                                                    beta.data_ptr(), _ptr(bn.running_mean if track else None),
                                                    _ptr(bn.running_var if track else None), stats[k].data_ptr(), st),
                      "spacap_sa_bn_finalize_f32")

            if recompute and L1_MOMENTS:
                mom = torch.empty(nparts * 16, dtype=torch.float64, device=dev)
                check(lib.spacap_sa_l1_moments_f32(_ptr(feat), xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), float(rdiv), B, Np, N, S,
                                                   z1.data_ptr(), mom.data_ptr(), st), "spacap_sa_l1_moments_f32")
                bn0 = bns[0]
                track = bn0.track_running_stats and bn0.running_mean is not None
                if track:
                    from .fused_bn import bump_counter
                    bump_counter(bn0)
                check(lib.spacap_sa_l1_moments_finalize_f32(mom.data_ptr(), W1c.data_ptr(), W1c.shape[1], has_feat, C1, R, float(bn0.eps),
                                                            0.0 if bn0.momentum is None else float(bn0.momentum), g1.data_ptr(),
                                                            b1.data_ptr(), _ptr(bn0.running_mean if track else None),
                                                            _ptr(bn0.running_var if track else None), stats[0].data_ptr(), st),
                      "spacap_sa_l1_moments_finalize_f32")
            elif recompute:
                check(lib.spacap_sa_l1_stats_f32(_ptr(feat), xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), W1c.data_ptr(),
                                                 W1c.shape[1], float(rdiv), B, Np, N, S, C1, z1.data_ptr(), part.data_ptr(), st),
                      "spacap_sa_l1_stats_f32")
                finalize(0, C1, g1, b1)
            if recompute:
                check(lib.spacap_sa_mid_fwd_l1in_f32(z1.data_ptr(), W1c.data_ptr(), W1c.shape[1], has_feat, stats[0].data_ptr(),
                                                     W2c.data_ptr(), R, z2.data_ptr(), part.data_ptr(), st),
                      "spacap_sa_mid_fwd_l1in_f32")
            else:
                check(lib.spacap_sa_l1_fwd_f32(_ptr(Y), _ptr(feat), xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(),
                                               W1c.data_ptr(), W1c.shape[1], float(rdiv), B, Np, N, S, C1, z1.data_ptr(),
                                               part.data_ptr(), st), "spacap_sa_l1_fwd_f32")
                finalize(0, C1, g1, b1)
                check(lib.spacap_sa_mid_fwd_f32(z1.data_ptr(), stats[0].data_ptr(), W2c.data_ptr(), R, C1, C2, z2.data_ptr(),
                                                part.data_ptr(), st), "spacap_sa_mid_fwd_f32")
            finalize(1, C2, g2, b2)
            out = torch.empty(B, N, C3, **f32)
            arg = torch.empty(B, N, C3, dtype=torch.uint8, device=dev)
            if pool_fused:
                # the last layer's kernel leaves the two best pooling candidates per (sub-group, channel); once the layer's
                # statistics are final a G x C3 pass picks the first maximum: z3 is not read again in the forward
                nsub = R // min(S, 32)
                g3c = g3.contiguous()
                cand_v = torch.empty(nsub, C3, 2, **f32)
                cand_i = torch.empty(nsub, C3, 2, dtype=torch.uint8, device=dev)
                check(lib.spacap_sa_mid_fwd_pool_f32(z2.data_ptr(), stats[1].data_ptr(), W3c.data_ptr(), g3c.data_ptr(), R, C2, C3,
                                                     S, _ptr(z3), part.data_ptr(), cand_v.data_ptr(), cand_i.data_ptr(), st),
                      "spacap_sa_mid_fwd_pool_f32")
                finalize(2, C3, g3, b3)
                check(lib.spacap_sa_pool_finalize_f32(cand_v.data_ptr(), cand_i.data_ptr(), stats[2].data_ptr(), g3c.data_ptr(),
                                                      G, S, C3, out.data_ptr(), arg.data_ptr(), _ptr(zmax), st),
                      "spacap_sa_pool_finalize_f32")
            else:
                check(lib.spacap_sa_mid_fwd_f32(z2.data_ptr(), stats[1].data_ptr(), W3c.data_ptr(), R, C2, C3, z3.data_ptr(),
                                                part.data_ptr(), st), "spacap_sa_mid_fwd_f32")
                finalize(2, C3, g3, b3)
                check(lib.spacap_sa_pool_fwd_f32(z3.data_ptr(), stats[2].data_ptr(), G, S, C3, out.data_ptr(), arg.data_ptr(),
                                                 st), "spacap_sa_pool_fwd_f32")
        from . import selections
        if selections.HOOK is not None:    # tests only: see selections.py
            selections.visit("sa", gammas=[g1, g2, g3], zs=[None if recompute else z1, z2, z3], stats=stats, arg=arg, out=out,
                             zmax=zmax, dims=(B, N, S))
        ctx.save_for_backward(xyz, new_xyz, idx, feat, W1c, W2c, W3c, z1, z2, z3, stats[0], stats[1], stats[2], out, arg, zmax)
        ctx.pm = pmc                  # (a tensor input: kept outside save_for_backward only to keep the saved tuple's layout)
        ctx.rdiv = float(rdiv)
        ctx.rows_index = rows_index   # prebuilt inverted index of idx (rows_index(idx, Np)), or None
        ctx.has_Y = Y is not None
        ctx.recompute = recompute     # z1 in the saved tuple is then rel4 (R, 4)
        ctx.need_xyz = xyz.requires_grad or new_xyz.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        xyz, new_xyz, idx, feat, W1, W2, W3, z1, z2, z3, st1, st2, st3, out, arg, zmax = ctx.saved_tensors
        dev = xyz.device
        B, Np, _ = xyz.shape
        N, S = idx.shape[1], idx.shape[2]
        C1, C2, C3 = W1.shape[0], W2.shape[0], W3.shape[0]
        R, G = B * N * S, B * N
        st = torch.cuda.current_stream(dev).cuda_stream
        f32 = dict(dtype=torch.float32, device=dev)
        dout = dout.contiguous()
        nparts = int(lib.spacap_sa_nparts())
        with torch.cuda.device(dev):
            part = torch.empty(nparts * 2 * max(C1, C2, C3), dtype=torch.float64, device=dev)
            coef = [torch.empty(c, 4, **f32) for c in (C1, C2, C3)]
            dg = [torch.empty(c, **f32) for c in (C1, C2, C3)]
            db = [torch.empty(c, **f32) for c in (C1, C2, C3)]

            def finalize(k, C, stats):
                check(lib.spacap_sa_bwd_finalize_f32(part.data_ptr(), C, R, stats.data_ptr(), coef[k].data_ptr(),
                                                     dg[k].data_ptr(), db[k].data_ptr(), st), "spacap_sa_bwd_finalize_f32")

            # pooled layer: masked gradient + BN sums over the arg-max rows
            dym = torch.empty(G, C3, **f32)
            check(lib.spacap_sa_pool_bwd_f32(dout.data_ptr(), out.data_ptr(), arg.data_ptr(), z3.data_ptr(), _ptr(zmax), st3.data_ptr(), G, S, C3,
                                             dym.data_ptr(), part.data_ptr(), st), "spacap_sa_pool_bwd_f32")
            finalize(2, C3, st3)
            dy2 = torch.empty(R, C2, **f32)
            # layer 3: weight gradient, then data gradient (its epilogue produces layer 2's BN sums)
            if POOL_WGRAD and R >= POOL_WGRAD_MIN_ROWS and lib.spacap_sa_wgrad_pool_supported(C2, C3, S):
                # from z2 alone: dW3 = (g d)^T a2 + k0 (x) colsum a2 - diag(k1) W3 a2^T a2
                npw, nfl = int(lib.spacap_sa_wgrad_pool_parts(R, C2, C3, S)), int(lib.spacap_sa_l3bwd_part_floats(C2, C3))
                pw = torch.empty(npw, nfl, **f32)
                check(lib.spacap_sa_wgrad_pool_f32(dym.data_ptr(), arg.data_ptr(), S, coef[2].data_ptr(), z2.data_ptr(),
                                                   st2.data_ptr(), R, C3, C2, pw.data_ptr(), st), "spacap_sa_wgrad_pool_f32")
                sums = torch.empty(nfl, dtype=torch.float64, device=dev)
                dW3 = torch.empty(C3, C2, **f32)
                check(lib.spacap_sa_l3bwd_dw_f32(pw.data_ptr(), npw, coef[2].data_ptr(), W3.data_ptr(), C3, C2, sums.data_ptr(),
                                                 dW3.data_ptr(), st), "spacap_sa_l3bwd_dw_f32")
            else:
                pw = torch.empty(int(lib.spacap_sa_wgrad_slabs(R, C3, C2, 1)), C3, C2, **f32)
                check(lib.spacap_sa_wgrad_f32(dym.data_ptr(), arg.data_ptr(), S, z3.data_ptr(), coef[2].data_ptr(),
                                              z2.data_ptr(), st2.data_ptr(), R, C3, C2, pw.data_ptr(), st), "spacap_sa_wgrad_f32")
                dW3 = sum_slabs(pw, deferrable=True)
            check(lib.spacap_sa_dgrad_f32(dym.data_ptr(), arg.data_ptr(), S, z3.data_ptr(), coef[2].data_ptr(),
                                          W3.data_ptr(), z2.data_ptr(), st2.data_ptr(), R, C3, C2, dy2.data_ptr(),
                                          part.data_ptr(), st), "spacap_sa_dgrad_f32")
            finalize(1, C2, st2)
            # layer 2
            has_feat = int(feat is not None)
            fuse_l1 = (not ctx.has_Y) and (not ctx.need_xyz) and C1 == 64 and C2 == 64
            # SA1 with the rebuilt first layer: the layer's weight gradient rides the data-gradient kernel (one read of dy2 / z2
            # instead of two, one launch fewer: csrc/sa_mlp.hip, sa_dgrad_kernel<.., WG>)
            fuse_w2 = fuse_l1 and ctx.recompute and FUSE_L2_WGRAD
            if fuse_w2:
                pw = torch.empty(int(lib.spacap_sa_dgrad_wgrad_l1in_slabs(R)), C2, C1, **f32)
            else:
                pw = torch.empty(int(lib.spacap_sa_wgrad_slabs(R, C2, C1, 0)), C2, C1, **f32)
                if ctx.recompute:
                    check(lib.spacap_sa_wgrad_l1in_f32(dy2.data_ptr(), z2.data_ptr(), coef[1].data_ptr(), z1.data_ptr(), W1.data_ptr(),
                                                       W1.shape[1], has_feat, st1.data_ptr(), R, pw.data_ptr(), st),
                          "spacap_sa_wgrad_l1in_f32")
                else:
                    check(lib.spacap_sa_wgrad_f32(dy2.data_ptr(), None, 0, z2.data_ptr(), coef[1].data_ptr(), z1.data_ptr(),
                                                  st1.data_ptr(), R, C2, C1, pw.data_ptr(), st), "spacap_sa_wgrad_f32")
                dW2 = sum_slabs(pw, deferrable=True)
            if fuse_l1:
                # SA1: the first layer's weight gradient comes out of this kernel's epilogue as three sums
                # (csrc/sa_mlp.hip, L1Args); dy1 is never written and the first-layer backward pass is skipped
                pl1 = torch.empty(nparts, C1 * 8 + 4, **f32)
                if fuse_w2:
                    check(lib.spacap_sa_dgrad_wgrad_l1in_f32(dy2.data_ptr(), z2.data_ptr(), coef[1].data_ptr(), W2.data_ptr(),
                                                             z1.data_ptr(), W1.data_ptr(), W1.shape[1], has_feat, st1.data_ptr(), B, N, S,
                                                             part.data_ptr(), pl1.data_ptr(), pw.data_ptr(), st),
                          "spacap_sa_dgrad_wgrad_l1in_f32")
                    dW2 = sum_slabs(pw, deferrable=True)
                elif ctx.recompute:
                    check(lib.spacap_sa_dgrad_l1in_f32(dy2.data_ptr(), z2.data_ptr(), coef[1].data_ptr(), W2.data_ptr(),
                                                       z1.data_ptr(), W1.data_ptr(), W1.shape[1], has_feat, st1.data_ptr(), B, N, S,
                                                       part.data_ptr(), pl1.data_ptr(), st), "spacap_sa_dgrad_l1in_f32")
                else:
                    check(lib.spacap_sa_dgrad_l1_f32(dy2.data_ptr(), z2.data_ptr(), coef[1].data_ptr(), W2.data_ptr(),
                                                     z1.data_ptr(), st1.data_ptr(), _ptr(feat), xyz.data_ptr(),
                                                     new_xyz.data_ptr(), idx.data_ptr(), ctx.rdiv, B, Np, N, S, C2, C1,
                                                     part.data_ptr(), pl1.data_ptr(), st), "spacap_sa_dgrad_l1_f32")
                del dy2
                finalize(0, C1, st1)
                dW1 = torch.empty(C1, W1.shape[1], **f32)     # = g S1 + k0 S2 - k1 S3 from the kernel's three sums, one launch
                check(lib.spacap_sa_l1_dw_f32(pl1.data_ptr(), nparts, coef[0].data_ptr(), C1, W1.shape[1], dW1.data_ptr(), st),
                      "spacap_sa_l1_dw_f32")
                return (None, None, None, None, None, dW1, dW2, dW3, dg[0], db[0], dg[1], db[1], dg[2], db[2], None, None, None)
            dy1 = torch.empty(R, C1, **f32)
            check(lib.spacap_sa_dgrad_f32(dy2.data_ptr(), None, 0, z2.data_ptr(), coef[1].data_ptr(), W2.data_ptr(),
                                          z1.data_ptr(), st1.data_ptr(), R, C2, C1, dy1.data_ptr(), part.data_ptr(), st),
                  "spacap_sa_dgrad_f32")
            del dy2
            finalize(0, C1, st1)
            # layer 1: dz1 in place, dW1 partials, optional gradient of the relative coordinates
            pw1 = torch.empty(nparts, C1, 4, **f32)
            drel = torch.empty(R, 3, **f32) if ctx.need_xyz else None
            check(lib.spacap_sa_l1_bwd_f32(dy1.data_ptr(), z1.data_ptr(), coef[0].data_ptr(), _ptr(feat), xyz.data_ptr(),
                                           new_xyz.data_ptr(), idx.data_ptr(), W1.data_ptr(), W1.shape[1], ctx.rdiv, B, Np,
                                           N, S, C1, pw1.data_ptr(), _ptr(drel), int(ctx.has_Y), st),
                  "spacap_sa_l1_bwd_f32")
            dY = ws = None
            nbytes = int(lib.spacap_sa_rows_scatter_workspace_bytes(B, Np, N * S))
            ri = ctx.rows_index
            if ri is not None and not (ri.numel() == nbytes and ri.device == dev):
                ri = None
            if ctx.has_Y:
                dY = torch.empty(B, Np, C1, **f32)
                if ri is not None:
                    # the inverted index came with the geometry pyramid: only the gather is left
                    check(lib.spacap_sa_rows_gather_f32(dy1.data_ptr(), B, Np, N * S, C1, ri.data_ptr(), dY.data_ptr(), st),
                          "spacap_sa_rows_gather_f32")
                else:
                    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                    check(lib.spacap_sa_rows_scatter_f32(dy1.data_ptr(), idx.data_ptr(), B, Np, N * S, C1, dY.data_ptr(),
                                                         ws.data_ptr(), st), "spacap_sa_rows_scatter_f32")
            dpm = None
            if ctx.has_Y:
                # gradient of Y = pm W1[:, 3:]^T: d pm = dY W1[:, 3:]; the weight part reduces over all B*Np source points
                pm = ctx.pm
                Cf = W1.shape[1] - 3
                g2, x2 = dY.view(-1, C1), _rows2d(pm)
                if ctx.needs_input_grad[4]:
                    from .linear import dense_product
                    dpm = dense_product(g2, W1, False, col0=3).view_as(pm)    # dY W1[:, 3:]
                # [rel columns | feature columns] from the two sets of partial results in one launch (the values of the two
                # slab sums + the concatenation).  Feature partials: the Linear layers' slab kernel for multiples of 128
                # channels (SA2 - SA4), else the tall-and-narrow kernel (SA1 with 7 / 132 feature channels: cfg3, cfg4)
                nslab = int(lib.spacap_linear_wgrad_slabs(B * Np, C1, Cf))
                if nslab:
                    x2c = x2.contiguous()
                    pf = torch.empty(nslab, C1 * Cf, **f32)
                    check(lib.spacap_linear_wgrad_f32(g2.data_ptr(), x2c.data_ptr(), B * Np, C1, Cf, 0, pf.data_ptr(), st),
                          "spacap_linear_wgrad_f32")
                else:   # (reads the rows at their own stride)
                    nslab = int(lib.spacap_dense_wgrad_tall_slabs(B * Np, C1, Cf))
                    pf = torch.empty(nslab, C1 * Cf, **f32)
                    check(lib.spacap_dense_wgrad_tall_f32(g2.data_ptr(), C1, x2.data_ptr(), x2.stride(0), B * Np, C1, Cf, nslab,
                                                          pf.data_ptr(), st), "spacap_dense_wgrad_tall_f32")
                dW1 = torch.empty(C1, 3 + Cf, **f32)
                check(lib.spacap_sa_dw1_assemble_f32(pw1.data_ptr(), nparts, pf.data_ptr(), nslab, C1, Cf, dW1.data_ptr(), st),
                      "spacap_sa_dw1_assemble_f32")
            else:
                dW1 = sum_slabs(pw1)[:, :W1.shape[1]].contiguous()    # (C1, 4): rel x, y, z, inline feature
            dxyz = dnew = None
            if drel is not None:
                # gradient of rel = (xyz[idx] - new_xyz) / r to both sources, one launch, fixed summation orders
                want_x, want_n = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
                if want_x and ri is None and ws is None:
                    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                    check(lib.spacap_sa_rows_index_f32(idx.data_ptr(), B, Np, N * S, ws.data_ptr(), st), "spacap_sa_rows_index_f32")
                index = ri if ri is not None else ws
                dxyz = torch.empty(B, Np, 3, **f32) if want_x else None
                dnew = torch.empty(B, N, 3, **f32) if want_n else None
                if want_x or want_n:
                    check(lib.spacap_sa_drel_sums_f32(drel.data_ptr(), B, Np, N, S, _ptr(index) if want_x else None, _ptr(dxyz), _ptr(dnew),
                                                      st), "spacap_sa_drel_sums_f32")
        return (dxyz, dnew, None, None, dpm, dW1, dW2, dW3, dg[0], db[0], dg[1], db[1], dg[2], db[2], None, None, None)


def _uniform_rows(pm):
    """(B, Np, Cf) whose B * Np rows of Cf contiguous floats lie at ONE uniform stride (dense, or a column window of wider rows)."""
    return pm.stride(2) == 1 and pm.stride(0) == pm.shape[1] * pm.stride(1) and pm.data_ptr() % 4 == 0


def _rows2d(pm):
    """The (B * Np, Cf) view of a ``_uniform_rows`` tensor (row stride pm.stride(1))."""
    B, Np, Cf = pm.shape
    return pm.as_strided((B * Np, Cf), (pm.stride(1), 1), pm.storage_offset())


def _feature_product(pm, W1c):
    """Y (B, Np, C1) = pm (B, Np, Cf) W1[:, 3:]^T."""
    from .linear import dense_product
    B, Np, Cf = pm.shape
    return dense_product(_rows2d(pm), W1c, True, col0=3).view(B, Np, W1c.shape[0])


def supported(mlp_module, nsample):
    layers = list(mlp_module.children())
    if len(layers) != 3 or not all(getattr(l, "has_bn", False) for l in layers):
        return False
    if any(l.bn.bn.momentum is None for l in layers):
        return False   # cumulative-average running statistics need the batch count on the host: per-operator path
    c = [l.conv.out_channels for l in layers]
    return bool(lib.spacap_sa_mlp_supported(*c)) and 1 <= nsample <= 255


def rows_index(idx, Np):
    """Inverted index of a grouping idx (B,N,S) over Np source points (which grouped rows reference each point, in
    ascending row order) as an opaque uint8 tensor: what the feature-gradient gather of the fused SA op needs.  It
    depends on idx only, so a trainer can build it ahead of the step (detector.geometry_pyramid)."""
    B, N, S = idx.shape
    with torch.cuda.device(idx.device):
        ws = torch.empty(int(lib.spacap_sa_rows_scatter_workspace_bytes(B, Np, N * S)), dtype=torch.uint8, device=idx.device)
        check(lib.spacap_sa_rows_index_f32(idx.data_ptr(), B, Np, N * S, ws.data_ptr(),
                                           torch.cuda.current_stream(idx.device).cuda_stream), "spacap_sa_rows_index_f32")
    return ws


def sa_mlp_train(xyz, new_xyz, features, idx, mlp_module, rdiv, use_xyz=True, rows_idx=None):
    """xyz (B,Np,3), new_xyz (B,N,3), features (B,Cf,Np) or None, idx (B,N,S) -> (B,C3,N) [a ``layout.ChannelMajorOf``: the transposed view of the point-major (B,N,C3) result].  ``mlp_module``: the SharedMLP whose parameters / BatchNorm statistics are used and
    updated.  Returns None when this MLP has no fused kernels (the caller then uses the per-operator path)."""
    if not use_xyz or not xyz.is_cuda or not supported(mlp_module, idx.shape[2]):
        return None
    l1, l2, l3 = list(mlp_module.children())
    W1 = l1.conv.weight.view(l1.conv.out_channels, -1)
    Cf = W1.shape[1] - 3
    feat = pm = None
    if features is not None:
        assert features.shape[1] == Cf
        if Cf == 1 and not features.requires_grad:
            feat = features.reshape(features.shape[0], -1)                   # inline: no (B,Np,C1) product needed
        else:
            # the first layer commutes with the gather: multiply once per source point (inside the op)
            pm = point_major_of(features)   # (B,Np,Cf) form of a previous module's output, if any
            pm = pm if pm is not None else features.transpose(1, 2)
    else:
        assert Cf == 0
    bns = [l.bn.bn for l in (l1, l2, l3)]
    out = _SAMLP.apply(xyz, new_xyz, idx, feat, pm, W1, l2.conv.weight.view(l2.conv.out_channels, -1),
                       l3.conv.weight.view(l3.conv.out_channels, -1), bns[0].weight, bns[0].bias, bns[1].weight,
                       bns[1].bias, bns[2].weight, bns[2].bias, bns, rdiv, rows_idx)
    # (B,C3,N) as a transposed VIEW of the point-major result, typed (layout.ChannelMajorOf): the next SA module and the
    # proposal head ask for the point-major tensor itself, so no transposed copy is made unless a consumer needs one
    return ChannelMajorOf.wrap(out)


def _running_stats_rows(bn):
    """stats rows (mean, 1/std, gamma/std, beta) of a BatchNorm layer in inference mode: running statistics."""
    istd = torch.rsqrt(bn.running_var + bn.eps)
    return torch.stack([bn.running_mean, istd, bn.weight * istd, bn.bias], 1).contiguous()


@torch.no_grad()
def sa_mlp_eval(xyz, new_xyz, features, idx, mlp_module, rdiv, use_xyz=True):
    """Inference-mode counterpart of ``sa_mlp_train``: the same fused point-major kernels with the BatchNorm layers
    folded to their running statistics (what ``model.eval()`` computes per operator: QueryAndGroup -> [Conv2d 1x1 ->
    BatchNorm2d(running stats) -> ReLU] x 3 -> max_pool2d, lib/pointnet2/pointnet2_modules.py:241-259) -- no grouped
    (B, C+3, P, S) tensor, no statistics passes.  Returns (B, C3, N) or None when the shape has no fused kernels."""
    if not use_xyz or not xyz.is_cuda or not supported(mlp_module, idx.shape[2]):
        return None
    l1, l2, l3 = list(mlp_module.children())
    bns = [l.bn.bn for l in (l1, l2, l3)]
    if any(not b.track_running_stats or b.running_mean is None for b in bns):
        return None
    dev = xyz.device
    B, Np, _ = xyz.shape
    N, S = idx.shape[1], idx.shape[2]
    W1 = l1.conv.weight.view(l1.conv.out_channels, -1)
    W2 = l2.conv.weight.view(l2.conv.out_channels, -1).contiguous()
    W3 = l3.conv.weight.view(l3.conv.out_channels, -1).contiguous()
    C1, C2, C3 = W1.shape[0], W2.shape[0], W3.shape[0]
    Cf = W1.shape[1] - 3
    feat = Y = None
    if features is not None:
        assert features.shape[1] == Cf
        if Cf == 1:
            feat, W1a = features.reshape(B, -1).contiguous(), W1.contiguous()
        else:
            pm = point_major_of(features)
            pm = pm if pm is not None else features.transpose(1, 2)
            Y = _feature_product(pm if _uniform_rows(pm) else pm.contiguous(), W1.contiguous())
            W1a = W1[:, :3].contiguous()
    else:
        W1a = W1.contiguous()
    R, G = B * N * S, B * N
    st = torch.cuda.current_stream(dev).cuda_stream
    f32 = dict(dtype=torch.float32, device=dev)
    xyz, new_xyz, idx = xyz.contiguous(), new_xyz.contiguous(), idx.contiguous()
    with torch.cuda.device(dev):
        part = torch.empty(int(lib.spacap_sa_nparts()) * 2 * max(C1, C2, C3), dtype=torch.float64, device=dev)   # (ignored)
        stats = [_running_stats_rows(b) for b in bns]
        z1 = torch.empty(R, C1, **f32)
        check(lib.spacap_sa_l1_fwd_f32(_ptr(Y), _ptr(feat), xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), W1a.data_ptr(),
                                       W1a.shape[1], float(rdiv), B, Np, N, S, C1, z1.data_ptr(), part.data_ptr(), st),
              "spacap_sa_l1_fwd_f32")
        z2 = torch.empty(R, C2, **f32)
        check(lib.spacap_sa_mid_fwd_f32(z1.data_ptr(), stats[0].data_ptr(), W2.data_ptr(), R, C1, C2, z2.data_ptr(),
                                        part.data_ptr(), st), "spacap_sa_mid_fwd_f32")
        del z1
        z3 = torch.empty(R, C3, **f32)
        check(lib.spacap_sa_mid_fwd_f32(z2.data_ptr(), stats[1].data_ptr(), W3.data_ptr(), R, C2, C3, z3.data_ptr(),
                                        part.data_ptr(), st), "spacap_sa_mid_fwd_f32")
        del z2
        out = torch.empty(B, N, C3, **f32)
        arg = torch.empty(B, N, C3, dtype=torch.uint8, device=dev)
        check(lib.spacap_sa_pool_fwd_f32(z3.data_ptr(), stats[2].data_ptr(), G, S, C3, out.data_ptr(), arg.data_ptr(), st),
              "spacap_sa_pool_fwd_f32")
    return ChannelMajorOf.wrap(out)
