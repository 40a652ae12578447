"""Lab: the pooled layer's weight gradient, dense kernel (reads z3 and z2) against sa_wgrad_pool_kernel (reads z2 only), at the
step's shapes; each with its reduction of the workgroups' partials."""
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
for label, R, c2, c3, S in (("SA1", B * 2048 * 64, 64, 128, 64), ("SA2", B * 1024 * 32, 128, 256, 32), ("small", B * 256 * 32, 128, 128, 32)):
    G = R // S
    dym, arg = torch.randn(G, c3, device=dev), torch.randint(0, S, (G, c3), dtype=torch.uint8, device=dev)
    z2, z3 = torch.randn(R, c2, device=dev), torch.randn(R, c3, device=dev)
    W3, coef, st2 = 0.1 * torch.randn(c3, c2, device=dev), stats(c3), stats(c2)
    coef[:, 1:3] *= 0.02
    pw_old = torch.empty(int(lib.spacap_sa_wgrad_slabs(R, c3, c2, 1)), c3, c2, device=dev)
    t_w = timeit(lambda: check(lib.spacap_sa_wgrad_f32(dym.data_ptr(), arg.data_ptr(), S, z3.data_ptr(), coef.data_ptr(), z2.data_ptr(),
                                                       st2.data_ptr(), R, c3, c2, pw_old.data_ptr(), st), "w"))
    t_os = timeit(lambda: pw_old.sum(0))
    npw, nfl = int(lib.spacap_sa_wgrad_pool_parts(R, c2, c3, S)), int(lib.spacap_sa_l3bwd_part_floats(c2, c3))
    pw, sums, dW3 = torch.empty(npw, nfl, device=dev), torch.empty(nfl, dtype=torch.float64, device=dev), torch.empty(c3, c2, device=dev)
    t_n = timeit(lambda: check(lib.spacap_sa_wgrad_pool_f32(dym.data_ptr(), arg.data_ptr(), S, coef.data_ptr(), z2.data_ptr(), st2.data_ptr(),
                                                            R, c3, c2, pw.data_ptr(), st), "n"))
    t_s = timeit(lambda: check(lib.spacap_sa_l3bwd_dw_f32(pw.data_ptr(), npw, coef.data_ptr(), W3.data_ptr(), c3, c2, sums.data_ptr(),
                                                          dW3.data_ptr(), st), "s"))
    gf = 2.0 * R * c2 * c2 / 1e9
    print(f"{label:6s} R={R:8d} {c2}->{c3} S={S}: dense {t_w:6.1f} (+ slab sum {t_os:5.1f}, {len(pw_old)} slabs) | from z2 {t_n:6.1f} us "
          f"({gf / t_n * 1e3:5.1f} TF/s Gram, {4.0 * R * c2 / t_n * 1e-3:6.0f} GB/s) + dw {t_s:5.1f}  [{npw} partials]", flush=True)
