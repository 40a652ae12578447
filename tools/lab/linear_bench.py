"""Row-panel kernel (spacap_linear_rows_f32) against torch's BLAS GEMM at the Transformer's projection shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
import spacap3d_amd  # noqa
from spacap3d_amd.linear import rows_product

dev = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for R in (256, 2048):
    for K, CO, trans in ((128, 384, 1), (128, 128, 1), (128, 2048, 1), (384, 128, 0), (128, 128, 0), (128, 2048, 0), (512, 128, 0), (256, 128, 1)):
        a = torch.randn(R, K, device=dev)
        W = torch.randn(CO, K, device=dev) * 0.1 if trans else torch.randn(K, CO, device=dev) * 0.1
        b = torch.randn(CO, device=dev) if trans else None
        ref = (F.linear(a.double(), W.double(), b.double()) if trans else a.double() @ W.double())
        got = rows_product(a, W, b, trans)
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        t_blas = timeit((lambda: F.linear(a, W, b)) if trans else (lambda: a @ W))
        t_rows = timeit(lambda: rows_product(a, W, b, trans))
        print(f"R={R:5d} K={K:4d} CO={CO:5d} {'x W^T' if trans else 'g W  '}  blas {t_blas:7.1f} us   rows {t_rows:7.1f} us   err {err:.1e}")
