"""Lab: the SA module forward / backward with and without pooling fused into the last layer kernel (child processes)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "run":
    sys.path.insert(0, ROOT)
    import torch
    from spacap3d_amd.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(0)
    dev = "cuda:0"
    outs = []
    for (Np, N, S, Cf, mlp, radius) in ((1024, 256, 64, 1, [1, 64, 64, 128], 0.2), (256, 128, 32, 128, [128, 128, 128, 256], 0.4),
                                        (128, 64, 16, 256, [256, 128, 128, 256], 0.8), (40000, 2048, 64, 1, [1, 64, 64, 128], 0.2)):
        sa = PointnetSAModuleVotes(npoint=N, radius=radius, nsample=S, mlp=list(mlp), use_xyz=True, normalize_xyz=True).to(dev).train()
        xyz = torch.rand(2, Np, 3, device=dev)
        feats = torch.randn(2, Cf, Np, device=dev).requires_grad_(True)
        new_xyz, f, inds = sa(xyz, feats)
        f.square().sum().backward()
        outs += [f.detach().cpu(), feats.grad.cpu()] + [p.grad.cpu() for p in sa.parameters()]
    torch.save(outs, sys.argv[2])
else:
    a, b = "/tmp/pf_a.pt", "/tmp/pf_b.pt"
    subprocess.run([sys.executable, __file__, "run", a], check=True)
    subprocess.run([sys.executable, __file__, "run", b], env=dict(os.environ, SPACAP_SA_NO_POOL_FUSION="1"), check=True)
    import torch
    c = "/tmp/pf_c.pt"
    subprocess.run([sys.executable, __file__, "run", c], check=True)
    d = "/tmp/pf_d.pt"
    subprocess.run([sys.executable, __file__, "run", d], env=dict(os.environ, SPACAP_SA_NO_POOL_FUSION="1"), check=True)
    A, B, C, D = torch.load(a), torch.load(b), torch.load(c), torch.load(d)
    print("fused vs fused:", [torch.equal(x, y) for x, y in zip(A, C)].count(False), "of", len(A), "differ")
    print("plain vs plain:", [torch.equal(x, y) for x, y in zip(B, D)].count(False), "of", len(B), "differ")
    for i, (x, y) in enumerate(zip(A, B)):
        d = (x - y).abs().max().item()
        if i % 11 == 0: print(i, tuple(x.shape), "max abs diff", d, "equal" if torch.equal(x, y) else "DIFFERENT", flush=True)
