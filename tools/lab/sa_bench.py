"""Per-kernel timings of the fused SA shared-MLP entry points at the cfg2 shapes (B=8)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spacap3d_amd._native import lib, check
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream

def timeit(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def bench(name, B, Np, N, S, C1, C2, C3):
    R, G = B * N * S, B * N
    f32 = dict(dtype=torch.float32, device=dev)
    z1, z2, z3 = torch.randn(R, C1, **f32), torch.randn(R, C2, **f32), torch.randn(R, C3, **f32)
    W2, W3 = torch.randn(C2, C1, **f32) * 0.1, torch.randn(C3, C2, **f32) * 0.1
    def stats(C):
        s = torch.zeros(C, 4, **f32); s[:, 1] = 1; s[:, 2] = 1; return s
    s1, s2, s3 = stats(C1), stats(C2), stats(C3)
    coef = [torch.rand(c, 4, **f32) for c in (C1, C2, C3)]
    part = torch.empty(int(lib.spacap_sa_nparts()) * 2 * max(C1, C2, C3), dtype=torch.float64, device=dev)
    out = torch.empty(G, C3, **f32); arg = torch.empty(G, C3, dtype=torch.uint8, device=dev)
    dym = torch.randn(G, C3, **f32); dy2 = torch.randn(R, C2, **f32); dy1 = torch.empty(R, C1, **f32)
    g = torch.ones(C3, **f32)
    res = {}
    res["mid_fwd L2"] = timeit(lambda: check(lib.spacap_sa_mid_fwd_f32(z1.data_ptr(), s1.data_ptr(), W2.data_ptr(), R, C1, C2, z2.data_ptr(), part.data_ptr(), st), "x"))
    res["mid_fwd L3"] = timeit(lambda: check(lib.spacap_sa_mid_fwd_f32(z2.data_ptr(), s2.data_ptr(), W3.data_ptr(), R, C2, C3, z3.data_ptr(), part.data_ptr(), st), "x"))
    z3.normal_(); z2.normal_()
    res["finalize"] = timeit(lambda: check(lib.spacap_sa_bn_finalize_f32(part.data_ptr(), C3, R, 1e-5, 0.1, g.data_ptr(), g.data_ptr(), None, None, s3.data_ptr(), st), "x"))
    s3 = stats(C3)
    res["pool_fwd"] = timeit(lambda: check(lib.spacap_sa_pool_fwd_f32(z3.data_ptr(), s3.data_ptr(), G, S, C3, out.data_ptr(), arg.data_ptr(), st), "x"))
    res["pool_bwd"] = timeit(lambda: check(lib.spacap_sa_pool_bwd_f32(dym.data_ptr(), out.data_ptr(), arg.data_ptr(), z3.data_ptr(), None, s3.data_ptr(), G, S, C3, dym.data_ptr(), part.data_ptr(), st), "x"))
    pw = torch.empty(int(lib.spacap_sa_wgrad_slabs(R, C3, C2, 1)), C3, C2, **f32)
    res["wgrad L3"] = timeit(lambda: check(lib.spacap_sa_wgrad_f32(dym.data_ptr(), arg.data_ptr(), S, z3.data_ptr(), coef[2].data_ptr(), z2.data_ptr(), s2.data_ptr(), R, C3, C2, pw.data_ptr(), st), "x"))
    res["wgrad L3 sum"] = timeit(lambda: pw.sum(0))
    res["dgrad L3"] = timeit(lambda: check(lib.spacap_sa_dgrad_f32(dym.data_ptr(), arg.data_ptr(), S, z3.data_ptr(), coef[2].data_ptr(), W3.data_ptr(), z2.data_ptr(), s2.data_ptr(), R, C3, C2, dy2.data_ptr(), part.data_ptr(), st), "x"))
    pw = torch.empty(int(lib.spacap_sa_wgrad_slabs(R, C2, C1, 0)), C2, C1, **f32)
    res["wgrad L2"] = timeit(lambda: check(lib.spacap_sa_wgrad_f32(dy2.data_ptr(), None, 0, z2.data_ptr(), coef[1].data_ptr(), z1.data_ptr(), s1.data_ptr(), R, C2, C1, pw.data_ptr(), st), "x"))
    res["dgrad L2"] = timeit(lambda: check(lib.spacap_sa_dgrad_f32(dy2.data_ptr(), None, 0, z2.data_ptr(), coef[1].data_ptr(), W2.data_ptr(), z1.data_ptr(), s1.data_ptr(), R, C2, C1, dy1.data_ptr(), part.data_ptr(), st), "x"))
    gf = lambda ci, co: 2.0 * R * ci * co / 1e9
    print(f"{name}: R={R} C=({C1},{C2},{C3})")
    for k, v in res.items():
        extra = ""
        if "L3" in k and "sum" not in k: extra = f"  {gf(C2, C3) / v * 1e3:7.1f} TFLOP/s"
        if "L2" in k: extra = f"  {gf(C1, C2) / v * 1e3:7.1f} TFLOP/s"
        print(f"   {k:14s} {v:8.1f} us{extra}")

bench("SA1", 8, 40000, 2048, 64, 64, 64, 128)
bench("SA2", 8, 2048, 1024, 32, 128, 128, 256)
bench("SA3", 8, 1024, 512, 16, 128, 128, 256)
