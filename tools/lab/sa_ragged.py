import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spacap3d_amd._native import lib, check
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
f32 = dict(dtype=torch.float32, device=dev)
for R in (65536, 1025*64, 9664, 9696, 100, 64):
    for C1, C2 in ((64, 64), (64, 128), (128, 128), (128, 256)):
        z1 = torch.randn(R, C1, **f32); W = torch.randn(C2, C1, **f32) * 0.1
        s = torch.zeros(C1, 4, **f32); s[:, 1] = 1; s[:, 2] = 1
        part = torch.empty(int(lib.spacap_sa_nparts()) * 2 * C2, dtype=torch.float64, device=dev)
        z2 = torch.empty(R + 64, C2, **f32).fill_(-7)
        check(lib.spacap_sa_mid_fwd_f32(z1.data_ptr(), s.data_ptr(), W.data_ptr(), R, C1, C2, z2.data_ptr(), part.data_ptr(), st), "x")
        torch.cuda.synchronize()
        want = torch.relu(z1) @ W.t()
        e = (z2[:R] - want).abs().max(1).values
        bad = (e > 1e-4).nonzero().flatten()
        print(R, C1, C2, "max err", e.max().item(), "bad rows", bad.numel(), (bad[:4].tolist(), bad[-4:].tolist()) if bad.numel() else "", flush=True)
