"""Lab: fps_small_kernel (one barrier per round, centre from LDS) by workgroup shape at the step's small-N samplings, against the
round-5 kernels (SPACAP_FPS_LEGACY=1 in a child process gives their indices and times) -- indices must be identical."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from spacap3d_amd._native import check, lib
from spacap3d_amd import synthetic as S
dev = torch.device("cuda:0")
B = 8
full = S.scene_batch(B, 40000, use_height=False, seed=1000).to(dev)
def timed(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
legacy = os.environ.get("SPACAP_FPS_LEGACY") == "1"
lib.spacap_lab_fps_small.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for N, m in ((2048, 1024), (1024, 512), (512, 256), (1024, 256), (256, 64), (8192, 2048), (4096, 1024), (100, 30), (777, 333)):
    xyz = full[:, :N].contiguous()
    if N == 1024 and m == 256:   # votes: clustered points, duplicates
        xyz = (xyz * 0.3).round(decimals=1).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(max(int(lib.spacap_fps_workspace_bytes(B, N)), 16), dtype=torch.uint8, device=dev)
    ref = torch.empty(B, m, dtype=torch.int32, device=dev)
    run = lambda: check(lib.spacap_fps_f32(xyz.data_ptr(), B, N, m, ws.data_ptr(), ref.data_ptr(), st), "fps")
    t = timed(run)
    print(f"N={N:5d} m={m:4d} {'legacy' if legacy else 'default'} dispatch: {t:7.1f} us = {t / (m - 1):.3f} us/round  checksum {int(ref.long().sum())}", flush=True)
    if legacy:
        torch.save(ref.cpu(), f"/tmp/fps_ref_{N}_{m}.pt")
        continue
    if os.path.exists(f"/tmp/fps_ref_{N}_{m}.pt"):
        old = torch.load(f"/tmp/fps_ref_{N}_{m}.pt")
        print("      identical to the round-5 kernels:", bool((old == ref.cpu()).all()))
    for block, tpl in ((64, 1), (64, 2), (64, 4), (64, 8), (64, 16), (128, 2), (128, 4), (128, 8), (256, 1), (256, 2), (256, 4), (256, 8), (512, 1), (512, 2), (512, 4), (1024, 1), (1024, 2), (1024, 4), (1024, 8)):
        if block * tpl < N or block * tpl >= 4 * N and block > 64:
            continue
        idx = torch.empty(B, m, dtype=torch.int32, device=dev)
        runv = lambda: check(lib.spacap_lab_fps_small(xyz.data_ptr(), B, N, m, block, tpl, idx.data_ptr(), st), "lab")
        tv = timed(runv)
        print(f"      {block:4d} x {tpl:2d}: {tv:7.1f} us = {tv / (m - 1):.3f} us/round  same indices: {bool((idx == ref).all())}", flush=True)
