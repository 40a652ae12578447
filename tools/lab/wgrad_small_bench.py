"""Lab: spacap_dense_wgrad_small_f32 at the vocabulary projection's shape, with / without two-level rows, by M."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spacap3d_amd._native import check, lib
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream


def timeit(fn, iters=20, warm=3):
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


B, T, D = 8, 32, 128
for V in (3001, 3008, 1024, 256):
    L = T - 1
    g = torch.randn(B * L, V, device=dev)
    n = torch.randn(B, T, D, device=dev)
    dW, db = torch.empty(V, D, device=dev), torch.empty(V, device=dev)
    R = B * L
    t2 = timeit(lambda: check(lib.spacap_dense_wgrad_small_f32(g.data_ptr(), V, n.data_ptr(), D, L, T * D, 1, R, V, D, dW.data_ptr(), db.data_ptr(), st), "a"))
    x2 = n[:, 1:, :].contiguous().view(R, D)
    t1 = timeit(lambda: check(lib.spacap_dense_wgrad_small_f32(g.data_ptr(), V, x2.data_ptr(), D, 0, 0, 0, R, V, D, dW.data_ptr(), db.data_ptr(), st), "b"))
    t0 = timeit(lambda: check(lib.spacap_dense_wgrad_small_f32(g.data_ptr(), V, x2.data_ptr(), D, 0, 0, 0, R, V, D, dW.data_ptr(), None, st), "c"))
    tt = timeit(lambda: torch.mm(g.t(), x2))
    print(f"V={V:5d} R={R}: two-level rows {t2:6.1f} us, plain rows {t1:6.1f} us, no bias sum {t0:6.1f} us, torch.mm {tt:6.1f} us", flush=True)
