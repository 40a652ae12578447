import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spacap3d_amd._native import check, lib
dev = "cuda:0"
for (R, ci, co, S) in ((64 * 512, 64, 128, 64), (64 * 1024, 64, 128, 64), (64 * 1100, 64, 128, 64), (64 * 2048, 64, 128, 64), (64 * 4096, 64, 128, 64),
                       (32 * 2048, 128, 256, 32), (32 * 4096, 128, 256, 32), (32 * 8192, 128, 256, 32), (32 * 8192, 128, 128, 32), (16 * 16384, 128, 256, 16)):
    torch.manual_seed(1)
    zin = torch.randn(R, ci, device=dev)
    st_in = torch.tensor([0.0, 1.0, 1.0, 0.0], device=dev).repeat(ci, 1).contiguous()
    W = 0.1 * torch.randn(co, ci, device=dev)
    gamma = torch.ones(co, device=dev)
    zout = torch.full((R, co), float("nan"), device=dev)
    nparts = int(lib.spacap_sa_nparts())
    part = torch.empty(nparts * 2 * co, dtype=torch.float64, device=dev)
    nsub = R // min(S, 32)
    cv = torch.empty(nsub, co, 2, device=dev); ci_ = torch.empty(nsub, co, 2, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    check(lib.spacap_sa_mid_fwd_pool_f32(zin.data_ptr(), st_in.data_ptr(), W.data_ptr(), gamma.data_ptr(), R, ci, co, S, zout.data_ptr(),
                                         part.data_ptr(), cv.data_ptr(), ci_.data_ptr(), s), "x")
    torch.cuda.synchronize()
    ref = torch.relu(zin).double() @ W.double().t()
    e = (zout.double() - ref).abs()
    bad_rows = (e.max(1).values > 1e-4).nonzero().flatten()
    print((R, ci, co, S), "tiles/wave", R / 32 / (8 * 256 // (co // 128)), "max err", e.max().item(), "bad rows", bad_rows.numel(),
          "first bad tiles", sorted(set((bad_rows[:2000] // 32).tolist()))[:12], flush=True)
    if bad_rows.numel():
        GW = 8 * 256 // (co // 128)
        t = GW  # first bad tile
        a = zout[t * 32:(t + 1) * 32].double(); r2 = ref[t * 32:(t + 1) * 32]; r1 = ref[(t - GW) * 32:(t - GW + 1) * 32]
        print("   tile", t, "err vs own ref", (a - r2).abs().max().item(), "| vs own + first tile", (a - r2 - r1).abs().max().item(),
              "| vs first tile's ref", (a - r1).abs().max().item(), flush=True)
        refs = ref.view(-1, 32, co)
        for tt in (GW, GW + 1, GW + 5):
            a = zout[tt * 32:(tt + 1) * 32].double()
            e_all = (refs - a.unsqueeze(0)).abs().amax(dim=(1, 2))
            print("   tile", tt, "best matching ref tile", int(e_all.argmin()), "err", float(e_all.min()), flush=True)
            # half-tile / per-slice mixtures: compare with a GEMM of mixed inputs
            zi = torch.relu(zin).double().view(-1, 32, ci)
            for other in (tt - GW, tt + GW if tt + GW < refs.shape[0] else tt):
                for split in (32,):
                    mix = torch.cat([zi[tt][:, :split], zi[other][:, split:]], dim=1) @ W.double().t()
                    mix2 = torch.cat([zi[other][:, :split], zi[tt][:, split:]], dim=1) @ W.double().t()
                    print("      slices [own | tile %d]: err %.3g   [tile %d | own]: err %.3g" % (other, (a - mix).abs().max(), other, (a - mix2).abs().max()), flush=True)
        break
