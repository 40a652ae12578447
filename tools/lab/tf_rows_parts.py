"""Lab: where the row kernel's time goes at the encoder's largest launch (2 048 rows): by number of partial products added and
with / without the second product."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from spacap3d_amd._native import TfRowsArgs, check, lib
import kernel_cases as KC
dev = torch.device("cuda:0")
for R in (2048, 256):
    for nparts in (16, 8, 4, 1):
        for n2 in (384, 0):
            c = KC.tf_rows(R, 128 * nparts, dev)
            a = c["keep"][-1]
            a.n2 = n2
            us = KC.time_case(c, iters=50)
            print(f"R={R:5d} nparts={nparts:2d} n2={n2:3d}: {us:6.1f} us", flush=True)
            del c
