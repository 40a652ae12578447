import sys
sys.path.insert(0, "/root/repo")
import torch
from spacap3d_amd._native import check, lib, sum_slabs
dev = "cuda:0"
for (B, CO, CI, N) in ((2, 128, 131, 4096), (2, 128, 128, 4096), (2, 256, 128, 4096), (8, 256, 256, 1024), (8, 259, 256, 1024), (8, 128, 512, 512)):
    g = torch.randn(B, CO, N, device=dev); x = torch.relu(torch.randn(B, CI, N, device=dev))
    ns = int(lib.spacap_conv1x1_wgrad_slabs(B, CO, CI, N))
    part = torch.empty(ns, CO * CI, device=dev)
    check(lib.spacap_conv1x1_wgrad_f32(g.data_ptr(), x.data_ptr(), B, CO, CI, N, part.data_ptr(), torch.cuda.current_stream().cuda_stream), "w")
    got = part.sum(0).view(CO, CI).double()
    want = torch.einsum("bon,bin->oi", g.double(), x.double())
    e = (got - want).abs().max() / want.abs().max()
    print((B, CO, CI, N), "slabs", ns, "rel err", float(e), flush=True)
