import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spacap3d_amd._native import lib, check
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
def timeit(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for R in (2048, 256):
    g = torch.randn(R, 128, device=dev); W = torch.randn(128, 2048, device=dev); y = torch.relu(torch.randn(R, 2048, device=dev)); dx = torch.empty_like(y)
    t1 = timeit(lambda: check(lib.spacap_linear_dgrad_mask_f32(g.data_ptr(), W.data_ptr(), y.data_ptr(), 1.1, R, 128, 2048, dx.data_ptr(), st), "x"))
    t2 = timeit(lambda: g @ W)
    t3 = timeit(lambda: torch.where(y > 0, (g @ W) * 1.1, 0.0))
    print(R, f"fused {t1:.1f} us   g@W {t2:.1f} us   g@W + mask (torch) {t3:.1f} us")
