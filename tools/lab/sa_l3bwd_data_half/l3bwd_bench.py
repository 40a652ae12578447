"""Lab: the pooled last layer's backward with and without its stored pre-activation, at the step's shapes.
old = spacap_sa_wgrad_f32 + spacap_sa_dgrad_f32 (both read z3), new = spacap_sa_l3bwd_prep / _f32 / _dw (one pass over z2);
and the forward layer kernel with / without storing z3."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spacap3d_amd._native import check, lib
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def stats(C):
    s = torch.empty(C, 4, device=dev)
    s[:, 0] = 0.05 * torch.randn(C, device=dev); s[:, 1] = 1 + 0.1 * torch.rand(C, device=dev)
    s[:, 2] = s[:, 1] * (1 + 0.1 * torch.rand(C, device=dev)); s[:, 3] = 0.1 * torch.randn(C, device=dev)
    return s


B = 8
for label, R, c2, c3, S in (("SA1", B * 2048 * 64, 64, 128, 64), ("SA2", B * 1024 * 32, 128, 256, 32), ("SA3", B * 512 * 16, 128, 256, 16),
                            ("SA4", B * 256 * 16, 128, 256, 16), ("vote-agg", B * 256 * 16, 128, 128, 16)):
    G = R // S
    dym, arg = torch.randn(G, c3, device=dev), torch.randint(0, S, (G, c3), dtype=torch.uint8, device=dev)
    z2, z3 = torch.randn(R, c2, device=dev), torch.randn(R, c3, device=dev)
    W3, coef, st2 = 0.1 * torch.randn(c3, c2, device=dev), stats(c3), stats(c2)
    coef[:, 1:3] *= 0.02
    part = torch.empty(int(lib.spacap_sa_nparts()) * 2 * max(c2, c3), dtype=torch.float64, device=dev)
    dy2 = torch.empty(R, c2, device=dev)
    pw_old = torch.empty(int(lib.spacap_sa_wgrad_slabs(R, c3, c2, 1)), c3, c2, device=dev)
    t_w = timeit(lambda: check(lib.spacap_sa_wgrad_f32(dym.data_ptr(), arg.data_ptr(), S, z3.data_ptr(), coef.data_ptr(), z2.data_ptr(),
                                                       st2.data_ptr(), R, c3, c2, pw_old.data_ptr(), st), "w"))
    t_d = timeit(lambda: check(lib.spacap_sa_dgrad_f32(dym.data_ptr(), arg.data_ptr(), S, z3.data_ptr(), coef.data_ptr(), W3.data_ptr(),
                                                       z2.data_ptr(), st2.data_ptr(), R, c3, c2, dy2.data_ptr(), part.data_ptr(), st), "d"))
    mneg, vrow = torch.empty(c2, c2, device=dev), torch.empty(c2, device=dev)
    npw, nfl = int(lib.spacap_sa_l3bwd_parts(R, c2, c3)), int(lib.spacap_sa_l3bwd_part_floats(c2, c3))
    pw, sums, dW3 = torch.empty(npw, nfl, device=dev), torch.empty(nfl, dtype=torch.float64, device=dev), torch.empty(c3, c2, device=dev)
    t_p = timeit(lambda: check(lib.spacap_sa_l3bwd_prep_f32(coef.data_ptr(), W3.data_ptr(), c3, c2, mneg.data_ptr(), vrow.data_ptr(), st), "p"))
    t_n = timeit(lambda: check(lib.spacap_sa_l3bwd_f32(dym.data_ptr(), arg.data_ptr(), S, coef.data_ptr(), W3.data_ptr(), mneg.data_ptr(),
                                                       vrow.data_ptr(), z2.data_ptr(), st2.data_ptr(), R, c3, c2, dy2.data_ptr(),
                                                       part.data_ptr(), pw.data_ptr(), st), "n"))
    t_s = timeit(lambda: check(lib.spacap_sa_l3bwd_dw_f32(pw.data_ptr(), npw, coef.data_ptr(), W3.data_ptr(), c3, c2, sums.data_ptr(),
                                                          dW3.data_ptr(), st), "s"))
    t_os = timeit(lambda: pw_old.sum(0))
    gf = 2.0 * R * (2 * c2 * c2) / 1e9
    print(f"{label:9s} R={R:8d} {c2}->{c3} S={S}: old wgrad {t_w:6.1f} + dgrad {t_d:6.1f} (+ slab sum {t_os:5.1f}) = {t_w + t_d:6.1f} us | "
          f"new prep {t_p:5.1f} + pass {t_n:6.1f} ({gf / t_n * 1e3:5.1f} TF/s, {4.0 * R * c2 * 2 / t_n * 1e-3:6.0f} GB/s) + dw {t_s:5.1f} "
          f"= {t_p + t_n + t_s:6.1f} us  [{npw} partials]")
    if lib.spacap_sa_mid_fwd_pool_supported(c2, c3, S):
        g3 = torch.ones(c3, device=dev)
        nsub = R // min(S, 32)
        cv, ci = torch.empty(nsub, c3, 2, device=dev), torch.empty(nsub, c3, 2, dtype=torch.uint8, device=dev)
        f = lambda zo: check(lib.spacap_sa_mid_fwd_pool_f32(z2.data_ptr(), st2.data_ptr(), W3.data_ptr(), g3.data_ptr(), R, c2, c3, S, zo,
                                                            part.data_ptr(), cv.data_ptr(), ci.data_ptr(), st), "f")
        print(f"          forward layer 3: storing z3 {timeit(lambda: f(z3.data_ptr())):6.1f} us, not storing {timeit(lambda: f(None)):6.1f} us")
    del dym, arg, z2, z3, dy2, pw_old, pw
    torch.cuda.empty_cache()
