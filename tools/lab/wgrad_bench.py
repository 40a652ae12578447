"""Lab: the shared-MLP weight-gradient kernel at the step's shapes (default = split-bf16 for 128-multiples; run with
SPACAP_SA_F32MFMA=1 for the fp32-MFMA kernel).    python tools/lab/wgrad_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import kernel_cases as KC  # noqa: E402

dev = torch.device("cuda:0")
R1, R2 = 8 * 2048 * 64, 8 * 1024 * 32
for a in ((R1, 128, 64, True, 64, "SA1 layer 3"), (R2, 256, 128, True, 32, "SA2 layer 3"), (R2, 128, 128, False, 32, "SA2 layer 2"),
          (8 * 512 * 16, 256, 128, True, 16, "SA3 layer 3"), (8 * 512 * 16, 128, 128, False, 16, "SA3 layer 2")):
    c = KC.sa_wgrad(*a[:5], dev, a[5])
    us = KC.time_case(c)
    print(f"{c['name']:60s} {us:8.1f} us  {c['flops'] / us * 1e-6:6.1f} TFLOP/s  {c['bytes'] / us * 1e-6:5.2f} TB/s", flush=True)
    del c
