"""Lab: fp32-MFMA vs streaming split-bf16 data-gradient kernels: error against float64 and time (one process per variant)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "run":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch, kernel_cases as KC
    dev = torch.device("cuda:0")
    tag = "fp32-mfma" if os.environ.get("SPACAP_SA_DGRAD_F32") else "split-bf16"
    nparts = int(KC.lib.spacap_sa_nparts())
    for R, ck, cp, pooled, S in ((262144, 256, 128, True, 32), (262144, 128, 128, False, 32), (1048576, 128, 64, True, 64), (65536, 256, 128, True, 16),
                                 (32768 + 16, 256, 128, True, 16), (1000 * 32 + 7, 128, 128, False, 1), (77 * 48, 128, 64, True, 48), (64, 256, 128, True, 64)):
        torch.manual_seed(1)
        c = KC.sa_dgrad(R, ck, cp, pooled, S, dev, "lab")
        dy, arg, zk, zp, coef, stp, W, dyp, part = c["keep"]
        dyp.fill_(float("nan")); part.fill_(float("nan"))
        c["run"](); torch.cuda.synchronize()
        n = min(R, 8192)
        sel = torch.cat([torch.arange(n // 2, device=dev), torch.arange(R - n // 2, R, device=dev)]) if R > n else torch.arange(R, device=dev)
        if pooled:
            grp, sidx = sel // S, sel % S
            d = torch.where(arg[grp].long() == sidx.unsqueeze(1), dy[grp], torch.zeros((), device=dev))
        else:
            d = dy[sel]
        dz = (coef[:, 0] * d + coef[:, 1] - coef[:, 2] * zk[sel]).double()       # the kernel's own fp32 dz, then exact
        da = dz @ W.double()
        pre = (zp[sel] - stp[:, 0]) * stp[:, 2] + stp[:, 3]
        ref = torch.where(pre > 0, da, torch.zeros((), dtype=torch.float64, device=dev))
        got = dyp[sel].double()
        near = pre.abs() < 1e-5                                                   # the mask may flip within rounding of 0
        err = ((got - ref).abs() * (~near)).max().item() / ref.abs().max().item()
        p = part[:nparts * 2 * cp].view(nparts, 2, cp).sum(0)
        s_ref = dyp.double().sum(0)
        q_ref = (dyp.double() * ((zp - stp[:, 0]) * stp[:, 1]).double()).sum(0)
        es = ((p[0] - s_ref).abs().max() / s_ref.abs().max()).item()
        eq = ((p[1] - q_ref).abs().max() / q_ref.abs().max()).item()
        nan = bool(torch.isnan(dyp).any())
        us = KC.time_case(c)
        print(f"{tag} {ck:3d}->{cp:3d} R={R:8d} {'pooled S=%d' % S if pooled else 'dense':12s}: {us:7.1f} us {c['flops'] / us * 1e-6:6.1f} TF/s {c['bytes'] / us * 1e-3:7.1f} GB/s | err {err:.2e} sum {es:.1e} sq {eq:.1e} nan {nan}", flush=True)
        del c
else:
    for f32 in (True, False):
        env = dict(os.environ)
        if f32: env["SPACAP_SA_DGRAD_F32"] = "1"
        subprocess.run([sys.executable, __file__, "run"], env=env)
