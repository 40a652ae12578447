"""Lab: the step's persistent kernels beside each kind of side-stream neighbour (real sampling kernels of every level, the SA1 ball
query, fat / thin sleeping workgroups), with the forward grids sized as in the step (RESERVE, default 8)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, kernel_cases as KC
from spacap3d_amd import pointnet2_utils as pu
probe = ctypes.CDLL(os.path.join(ROOT, "tools", "lab", "libcumask_probe.so"))
probe.probe_spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda:0")
KC.check(KC.lib.spacap_sa_reserve_cus(int(os.environ.get("RESERVE", "8"))), "reserve")
B, R2, R1 = 8, 8 * 1024 * 32, 8 * 2048 * 64
side = torch.cuda.Stream(device=dev)
fps1 = KC.fps(B, 40000, 2048, dev); fps2 = KC.fps(B, 2048, 1024, dev); fps3 = KC.fps(B, 1024, 512, dev)
from spacap3d_amd import synthetic as S
xyz = S.scene_batch(B, 40000, use_height=False, seed=1000).to(dev)
ctr = xyz[:, :2048].contiguous()
def bq():
    for _ in range(8): pu.ball_query(0.2, 64, xyz, ctr)
def spin(g, t, lds): return lambda: probe.probe_spin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), g, t, lds, 6000)
def rep(fn, n):
    def f():
        for _ in range(n): fn()
    return f
NB = (("alone", None), ("FPS1 40000->2048", fps1["run"]), ("FPS2 2048->1024 x5", rep(fps2["run"], 5)), ("FPS3 1024->512 x10", rep(fps3["run"], 10)),
      ("SA1 ball query x8", bq), ("8 fat sleeping (1024 thr + 100 KB)", spin(8, 1024, 100 * 1024)), ("8 thin sleeping (64 thr)", spin(8, 64, 0)))
def timed(case, nb, iters):
    for _ in range(3): case["run"]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if nb is not None:
        with torch.cuda.stream(side):
            nb()
    e0.record()
    for _ in range(iters): case["run"]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for make, iters in ((lambda: KC.rel_fused(B, 256, 1, dev), 4), (lambda: KC.rel_fused(B, 256, 0, dev), 8),
                    (lambda: KC.sa_mid_fwd_pool(R1, 64, 128, 64, dev, "SA1 L3", True), 8), (lambda: KC.sa_mid_fwd_pool(R2, 128, 256, 32, dev, "SA2 L3", True), 10),
                    (lambda: KC.sa_mid_fwd_l1in(R1, dev, "SA1 L2"), 10), (lambda: KC.sa_dgrad(R2, 256, 128, True, 32, dev, "SA2 L3"), 10),
                    (lambda: KC.sa_wgrad_pool(R2, 128, 256, 32, dev, "SA2 L3"), 10), (lambda: KC.tf_ffn(B * 256, 2048, 0, dev), 40),
                    (lambda: KC.mha_fwd(B, 8, 256, 16, dev, True), 60)):
    c = make()
    base = timed(c, None, iters)
    out = []
    for name, nb in NB:
        t = timed(c, nb, iters)
        out.append(f"{name}: {t:6.1f}" + ("" if nb is None else f" ({t / base:.2f}x)"))
    print(f"{c['name'][:50]:50s} | " + " | ".join(out), flush=True)
    del c
