"""Lab: per-phase cycle stamps of the streaming split-bf16 kernel (SPACAP_SA_BF16X3=2 SPACAP_SA_LAB=9)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, kernel_cases as KC
dev = torch.device("cuda:0")
R, ci, co = [int(x) for x in sys.argv[1:4]]
c = KC.sa_mid_fwd(R, ci, co, dev, "lab")
for _ in range(3):
    c["run"]()
torch.cuda.synchronize()
buf = np.zeros((4, 8, 32, 5), dtype=np.uint64)
f = ctypes.CDLL(KC.lib._name).spacap_lab_bf3s_trace
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
io = np.zeros((4, 8, 4), dtype=np.uint64)
assert f(buf.ctypes.data, io.ctypes.data) == 0
us = KC.time_case(c)
io = io.astype(np.int64)
for wg in range(4):
    e0, e1 = io[wg, :, 0].min(), io[wg, :, 1].max()
    first = buf[wg, :, 0, 0].astype(np.int64).min()
    r0, r1 = io[wg, :, 2].min(), io[wg, :, 3].max()
    print(f"workgroup {wg}: s_memrealtime span {r1 - r0} ticks (100 MHz -> {(r1 - r0) / 100:.1f} us) -> s_memtime rate {(e1 - e0) / max(r1 - r0, 1) * 100:.0f} MHz")
    print(f"workgroup {wg}: entry -> exit {e1 - e0} ticks; entry -> first phase {first - e0} ticks; event-timed launch {us:.1f} us -> {(e1 - e0) / us:.0f} ticks/us if the workgroup spans the launch")
t = buf.astype(np.int64)
t0 = t[:, :, 0, 0].min()
NH = ci // 32
np.set_printoptions(linewidth=200)
for wg in (0,):
    for w in (0, 5):
        print(f"--- workgroup {wg} wave {w}: per phase [start, wait, split+fetch, mfma, stores] cycles (s_memtime ticks)")
        for p in range(min(32, (R // 32 // (8 * (256 // (co // 128)))) * NH)):
            s = t[wg, w, p]
            if s[0] == 0: break
            print(f"  p{p:2d} start {s[0]-t0:8d} | wait {s[1]-s[0]:6d} split {s[2]-s[1]:6d} mfma {s[3]-s[2]:6d} stores {s[4]-s[3]:6d} | total {s[4]-s[0]:6d}")
