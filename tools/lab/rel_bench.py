"""Relation head at the benchmark shape (B = 8, K = 256: 524 288 proposal pairs): the fused kernels (csrc/relation_fused.hip)
beside the composed path (relation_layer1 + relation_tail).  HIP events around ITERS launches.
    python tools/lab/rel_bench.py [ITERS]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spacap3d_amd import attention, linear  # noqa: E402
from spacap3d_amd._native import check, lib  # noqa: E402

DEV = torch.device("cuda:0")
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def t(fn, what, flops=None, bytes_=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / ITERS
    extra = (f"  {flops / us * 1e-6:6.1f} TFLOP/s" if flops else "") + (f"  {bytes_ / us * 1e-6:5.2f} TB/s" if bytes_ else "")
    print(f"{what:52s} {us:8.1f} us{extra}", flush=True)


def main():
    if len(sys.argv) > 2:
        check(lib.spacap_sa_reserve_cus(int(sys.argv[2])), "reserve")   # as beside the sampling chain
    B, K, H, D = 8, 256, 8, 16
    r = lambda *s: torch.randn(*s, device=DEV)
    P, U = torch.softmax(r(B, H, K, K), -1), r(B, K, H, 128) * 0.3
    b1, W2, b2, W3, b3 = r(128) * 0.1, r(128, 128) * 0.1, r(128) * 0.1, r(9, 128) * 0.1, r(9)
    R = B * K * K
    hid2, pred, dpred = torch.empty(R, 128, device=DEV), torch.empty(R, 9, device=DEV), r(R, 9)
    st = torch.cuda.current_stream().cuda_stream
    ffwd = R * (2.0 * 8 * 128 + 2.0 * 128 * 128 + 2.0 * 128 * 9)
    t(lambda: check(lib.spacap_relation_fused_fwd_f32(P.data_ptr(), U.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3.data_ptr(),
                                                      b3.data_ptr(), B, K, hid2.data_ptr(), pred.data_ptr(), st), "f"),
      "fused forward", ffwd, R * (128 + 9 + 8) * 4.0)
    nparts = int(lib.spacap_relation_fused_nparts(B, K))
    zs = int(lib.spacap_relation_fused_zsplit(B, K, nparts))
    dP, dU = torch.empty_like(P), torch.empty(zs, B, K, H, 128, device=DEV)
    part = torch.empty(nparts, int(lib.spacap_relation_fused_part_floats()), device=DEV)
    t(lambda: check(lib.spacap_relation_fused_bwd_f32(dpred.data_ptr(), hid2.data_ptr(), P.data_ptr(), U.data_ptr(), b1.data_ptr(),
                                                      W2.data_ptr(), W3.data_ptr(), B, K, nparts, zs, dP.data_ptr(), dU.data_ptr(),
                                                      part.data_ptr(), st), "b"),
      "fused backward", R * (2.0 * 8 * 128 * 3 + 4.0 * 128 * 128 + 4.0 * 128 * 9), R * (128 + 9 + 16) * 4.0)
    V = r(B, H, K, D)
    lins = [torch.nn.Linear(128, 128).to(DEV), torch.nn.Linear(128, 128).to(DEV), torch.nn.Linear(128, 9).to(DEV)]
    Pg = P.clone().requires_grad_(True)

    def composed():
        hid = attention.relation_layer1(Pg, V, lins[0].weight, lins[0].bias)
        return linear.relation_tail(hid, lins[1], lins[2])

    def fused():
        return linear.relation_head(Pg, V, *lins)

    for name, fn in (("composed", composed), ("fused", fused)):
        t(fn, f"{name}: forward (autograd on)")
        w = r(B, K, K, 9)

        def both():
            (fn() * w).sum().backward()
        t(both, f"{name}: forward + backward (incl. the test's own loss)")


if __name__ == "__main__":
    main()
