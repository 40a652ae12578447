"""Lab: a few launches of sa_wgrad_pool_kernel at the SA1 / SA2 shapes, for rocprofv3 --pmc passes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spacap3d_amd._native import check, lib
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
B = 8
for label, R, c2, c3, S in (("SA1", B * 2048 * 64, 64, 128, 64), ("SA2", B * 1024 * 32, 128, 256, 32)):
    G = R // S
    dym, arg = torch.randn(G, c3, device=dev), torch.randint(0, S, (G, c3), dtype=torch.uint8, device=dev)
    z2 = torch.randn(R, c2, device=dev)
    coef, st2 = torch.rand(c3, 4, device=dev), torch.rand(c2, 4, device=dev)
    npw, nfl = int(lib.spacap_sa_wgrad_pool_parts(R, c2, c3, S)), int(lib.spacap_sa_l3bwd_part_floats(c2, c3))
    pw = torch.empty(npw, nfl, device=dev)
    for _ in range(3):
        check(lib.spacap_sa_wgrad_pool_f32(dym.data_ptr(), arg.data_ptr(), S, coef.data_ptr(), z2.data_ptr(), st2.data_ptr(), R, c3, c2,
                                           pw.data_ptr(), st), "n")
    torch.cuda.synchronize()
