"""Lab: fp32-MFMA vs split-bf16 shared-MLP forward kernels: error against float64 and time, one process per variant
(the switches are read once).  SPACAP_SA_BF16X3 = unset (fp32 MFMA) / 1 (LDS-staged activations) / 2 (streaming kernel)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "run":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch, kernel_cases as KC
    dev = torch.device("cuda:0")
    tag = f"STREAM={os.environ.get('SPACAP_SA_STREAM', '-')} BF16X3={os.environ.get('SPACAP_SA_BF16X3', '-')} INTER={os.environ.get('SPACAP_SA_INTER', '-')} LAB={os.environ.get('SPACAP_SA_LAB', '-')}"
    nparts = int(KC.lib.spacap_sa_nparts())
    shapes = ((262144, 128, 256), (262144, 128, 128), (65536, 128, 256), (1048576, 64, 64), (1048576, 64, 128), (1000 * 33 + 7, 128, 256), (32768, 128, 128), (77, 64, 128))
    if os.environ.get("SPACAP_SA_LAB"):
        shapes = [x for x in shapes if x[2] % 128 == 0][:4]
    if os.environ.get("BF3_SMALL"):
        shapes = [x for x in shapes if x[0] <= 131072] + [(131072, 128, 256), (524288, 64, 128)]
    if os.environ.get("BF3_SHAPES"):
        shapes = list(shapes)[:int(os.environ["BF3_SHAPES"])]
    for R, ci, co in shapes:
        torch.manual_seed(0)
        c = KC.sa_mid_fwd(R, ci, co, dev, "lab")
        zin, st, W, zout, part = c["keep"]
        zout.fill_(float("nan")); part.fill_(float("nan"))
        c["run"](); torch.cuda.synchronize()
        n = min(R, 8192)
        sel = torch.cat([torch.arange(n // 2, device=dev), torch.arange(R - n // 2, R, device=dev)])
        a = torch.relu((zin[sel].double() - st[:, 0].double()) * st[:, 2].double() + st[:, 3].double())
        a32 = torch.relu((zin[sel] - st[:, 0]) * st[:, 2] + st[:, 3]).double()   # the kernel's own fp32 activation
        ref = a32 @ W.double().t()
        err = (zout[sel].double() - ref).abs().max().item() / ref.abs().max().item()
        p = part.view(nparts, 2, -1)[:, :, :co].sum(0)
        s_ref, q_ref = zout.double().sum(0), (zout.double() ** 2).sum(0)
        es = ((p[0] - s_ref).abs().max() / s_ref.abs().max()).item()
        eq = ((p[1] - q_ref).abs().max() / q_ref.abs().max()).item()
        nan = bool(torch.isnan(zout).any())
        us = KC.time_case(c)
        print(f"{tag} {ci:3d}->{co:3d} R={R:8d}: {us:7.1f} us {c['flops'] / us * 1e-6:6.1f} TF/s {c['bytes'] / us * 1e-3:7.1f} GB/s | err {err:.2e} sum {es:.1e} sq {eq:.1e} nan {nan}", flush=True)
        del c
else:
    for v, inter, lab in (("0", "", ""), ("1", "", ""), ("2", "", ""), ("2", "", "1"), ("2", "", "2")):
        env = dict(os.environ)
        if v == "S":
            env["SPACAP_SA_STREAM"] = "1"; v = ""
        if lab: env["SPACAP_SA_LAB"] = lab
        if v: env["SPACAP_SA_BF16X3"] = v
        if inter: env["SPACAP_SA_INTER"] = inter
        subprocess.run([sys.executable, __file__, "run"], env=env)
