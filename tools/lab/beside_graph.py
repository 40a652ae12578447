"""Lab: rel_fused_bwd / SA1 L3 forward as a small captured graph, replayed beside the sampling pyramid (as a graph replay, as eager
launches, FPS1 only): does the step's 1.8x reproduce outside the step?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, kernel_cases as KC
from spacap3d_amd.detector import geometry_pyramid
from spacap3d_amd import synthetic as S
dev = torch.device("cuda:0")
KC.check(KC.lib.spacap_sa_reserve_cus(8), "reserve")
B, R2, R1 = 8, 8 * 1024 * 32, 8 * 2048 * 64
main, side = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
xyz = S.scene_batch(B, 40000, use_height=False, seed=1000).to(dev)
with torch.cuda.stream(side), torch.no_grad():
    for _ in range(2): geometry_pyramid(xyz)
torch.cuda.synchronize()
gs = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(gs, stream=side):
    pyr = geometry_pyramid(xyz)
torch.cuda.synchronize()
fps1 = KC.fps(B, 40000, 2048, dev)
def side_graph():
    with torch.cuda.stream(side): gs.replay()
def side_eager():
    with torch.cuda.stream(side), torch.no_grad(): geometry_pyramid(xyz)
def side_fps1():
    with torch.cuda.stream(side): fps1["run"]()
for make, reps, delay_us in ((lambda: KC.rel_fused(B, 256, 1, dev), 2, (0, 1000, 2000, 3000, 3500, 4000, 4500)),
                             (lambda: KC.sa_mid_fwd_pool(R1, 64, 128, 64, dev, "SA1 L3", True), 4, (0, 1000, 3000, 4000)),
                             (lambda: KC.sa_mid_fwd_pool(R2, 128, 256, 32, dev, "SA2 L3", True), 4, (0, 1000, 3000, 4000))):
    c = make()
    with torch.cuda.stream(main):
        for _ in range(3): c["run"]()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        for _ in range(reps): c["run"]()
    torch.cuda.synchronize()
    def timed(nb, delay):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if nb is not None: nb()
        with torch.cuda.stream(main):
            if delay: KC.check(KC.lib.spacap_stream_delay(delay, main.cuda_stream), "delay")
            e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    print(c["name"][:60])
    for name, nb in (("alone", None), ("pyramid graph", side_graph), ("pyramid eager", side_eager), ("FPS1 eager", side_fps1)):
        row = []
        for d in delay_us:
            ts = sorted(timed(nb, d) for _ in range(3))
            row.append(f"+{d} us: {ts[1]:6.1f}")
        print(f"   {name:14s} " + " | ".join(row), flush=True)
    del c, g
