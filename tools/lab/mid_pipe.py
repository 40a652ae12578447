"""sa_mid_fwd: phase-alternating kernel vs the software-pipelined variant (SPACAP_SA_PIPE=1): checksum + time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spacap3d_amd._native import lib, check
dev = torch.device("cuda:0")
torch.manual_seed(0)
for R, Cin, Cout in ((262144, 128, 128), (262144, 128, 256), (65536, 128, 256), (32768, 128, 128)):
    z = torch.randn(R, Cin, device=dev)
    st_ = torch.zeros(Cin, 4, device=dev); st_[:, 0] = 0.1; st_[:, 2] = 1.3; st_[:, 3] = 0.2
    W = torch.randn(Cout, Cin, device=dev) * 0.1
    out = torch.empty(R, Cout, device=dev)
    part = torch.empty(int(lib.spacap_sa_nparts()) * 2 * Cout, dtype=torch.float64, device=dev)
    def run():
        check(lib.spacap_sa_mid_fwd_f32(z.data_ptr(), st_.data_ptr(), W.data_ptr(), R, Cin, Cout, out.data_ptr(), part.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "mid")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = torch.relu((z.double() - 0.1) * 1.3 + 0.2) @ W.double().t()
    err = float((out.double() - ref).abs().max())
    psum = part.view(-1, 2, Cout).sum(0)
    perr = float((psum[0] - ref.sum(0)).abs().max() / ref.sum(0).abs().max())
    print(f"R={R} {Cin}->{Cout}: {us:7.1f} us  {2.0 * R * Cin * Cout / us / 1e6:6.1f} TFLOP/s   max err {err:.2e}  stats err {perr:.1e}  checksum {float(out.double().sum()):.6f}")
