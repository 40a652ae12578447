"""linear_wgrad_kernel + sum_slabs at the Transformer's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import spacap3d_amd  # noqa
from spacap3d_amd._native import lib, check, sum_slabs
dev = torch.device("cuda:0")
for R, CK, CP in ((2048, 2048, 128), (2048, 128, 2048), (2048, 384, 128), (2048, 128, 128), (256, 2048, 128), (256, 128, 128), (16384, 128, 128)):
    g = torch.randn(R, CK, device=dev); x = torch.randn(R, CP, device=dev)
    ns = int(lib.spacap_linear_wgrad_slabs(R, CK, CP))
    part = torch.empty(ns, CK * CP + CK, device=dev)
    def run():
        check(lib.spacap_linear_wgrad_f32(g.data_ptr(), x.data_ptr(), R, CK, CP, 1, part.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "w")
    def run2():
        run(); return sum_slabs(part)
    for f, name in ((run, "wgrad"), (run2, "wgrad+sum")):
        for _ in range(3): f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20): f()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        print(f"R={R} CK={CK} CP={CP} slabs={ns} {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", end="   ")
    s = run2()
    ref = g.double().t() @ x.double()
    print(f"err {float((s[:CK*CP].view(CK, CP).double() - ref).abs().max() / ref.abs().max()):.1e}")
