import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spacap3d_amd._native import check, lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, Np, N, S, C1, feat_on, Cf) in ((8, 40000, 2048, 64, 64, True, 0), (8, 2048, 1024, 32, 128, False, 128), (8, 1024, 512, 16, 128, False, 256), (8, 512, 256, 16, 128, False, 256), (8, 1024, 256, 16, 128, False, 256)):
    xyz = torch.rand(B, Np, 3, device=dev); new_xyz = xyz[:, :N].contiguous()
    idx = torch.randint(0, Np, (B, N, S), dtype=torch.int32, device=dev)
    feat = torch.randn(B, Np, device=dev) if feat_on else None
    Y = torch.randn(B, Np, C1, device=dev) if Cf else None
    W1 = torch.randn(C1, 3 + (1 if feat_on else 0) + Cf, device=dev)
    R = B * N * S
    z1 = torch.empty(R, C1, device=dev); part = torch.empty(int(lib.spacap_sa_nparts()) * 2 * C1, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        check(lib.spacap_sa_l1_fwd_f32(Y.data_ptr() if Y is not None else None, feat.data_ptr() if feat is not None else None, xyz.data_ptr(),
                                       new_xyz.data_ptr(), idx.data_ptr(), W1.data_ptr(), W1.shape[1], 0.2, B, Np, N, S, C1, z1.data_ptr(), part.data_ptr(), st), "l1")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print((B, Np, N, S, C1), "%.1f us" % (e0.elapsed_time(e1) * 50), "checksum", z1.double().sum().item(), part.sum().item(), flush=True)
