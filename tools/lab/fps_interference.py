"""Lab: how much slower do the step's kernels run while the SA1 furthest-point sampling (8 workgroups, 2.9 ms) runs beside them?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, kernel_cases as KC
dev = torch.device("cuda:0")
if os.environ.get("RESERVE"):
    KC.check(KC.lib.spacap_sa_reserve_cus(int(os.environ["RESERVE"])), "reserve")
B, R2, R1 = 8, 8 * 1024 * 32, 8 * 2048 * 64
fps = KC.fps(B, 40000, 2048, dev)
side = torch.cuda.Stream(device=dev)
def timed(case, with_fps, iters):
    for _ in range(3): case["run"]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if with_fps:
        with torch.cuda.stream(side):
            fps["run"]()      # NB: kernel_cases launches on the CURRENT stream
    e0.record()
    for _ in range(iters): case["run"]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for make, iters in ((lambda: KC.sa_mid_fwd(R2, 128, 256, dev, "SA2 L3"), 16), (lambda: KC.sa_mid_fwd(R1, 64, 64, dev, "SA1 L2"), 16),
                    (lambda: KC.sa_dgrad(R2, 256, 128, True, 32, dev, "SA2 L3"), 10), (lambda: KC.sa_wgrad(R2, 256, 128, True, 32, dev, "SA2 L3"), 10),
                    (lambda: KC.rel_tail_fwd(B * 256 * 256, dev), 10), (lambda: KC.mha_fwd(B, 8, 256, 16, dev, True), 60)):
    c = make()
    a = timed(c, False, iters); b = timed(c, True, iters); a2 = timed(c, False, iters)
    print(f"{c['name']:60s} alone {a:7.1f} / {a2:7.1f} us   beside FPS {b:7.1f} us  ({b / min(a, a2):.2f}x)", flush=True)
    del c
