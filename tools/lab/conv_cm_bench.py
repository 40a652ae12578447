"""Lab: the channel-major 1x1 convolution kernel (csrc/conv1x1.hip) beside the library's convolution at the step's shapes."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spacap3d_amd._native import check, lib  # noqa: E402

DEV = torch.device("cuda:0")


def t(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for B, CI, CO, N in ((8, 256, 256, 1024), (8, 256, 259, 1024), (8, 768, 256, 512), (8, 256, 256, 512), (8, 512, 256, 1024),
                     (8, 128, 128, 256), (8, 128, 97, 256), (8, 3, 128, 256)):
    x, W, b = torch.randn(B, CI, N, device=DEV), torch.randn(CO, CI, 1, device=DEV) * 0.1, torch.randn(CO, device=DEV)
    g = torch.randn(B, CO, N, device=DEV)
    y, dx = torch.empty(B, CO, N, device=DEV), torch.empty(B, CI, N, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    f_lib = t(lambda: F.conv1d(x, W, b))
    f_own = t(lambda: check(lib.spacap_conv1x1_cm_f32(0, W.data_ptr(), x.data_ptr(), b.data_ptr(), B, CI, CO, N, y.data_ptr(), st), "f"))
    d_lib = t(lambda: torch.ops.aten.convolution_backward(g, x, W, None, [1], [0], [1], False, [0], 1, [True, False, False]))
    d_own = t(lambda: check(lib.spacap_conv1x1_cm_f32(1, W.data_ptr(), g.data_ptr(), None, B, CI, CO, N, dx.data_ptr(), st), "d"))
    fl = 2.0 * B * CI * CO * N
    print(f"B={B} {CI:4d}->{CO:4d} N={N:5d}: forward library {f_lib:6.1f} us, own {f_own:6.1f} us ({fl / f_own * 1e-6:5.1f} TFLOP/s); "
          f"input gradient library {d_lib:6.1f} us, own {d_own:6.1f} us", flush=True)
