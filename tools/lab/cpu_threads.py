"""Lab: CPU-baseline training step time vs torch thread count (the host of the GPU box has 128 hardware threads)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from oracle.attention_ref import OracleBackend
cfg = bench.CFG["cfg2"]
for threads in (int(a) for a in sys.argv[1:]):
    ts = bench._cpu_steps(cfg, OracleBackend(openmp=True), threads, 2, 2)
    print(f"threads {threads:3d}: B=2 step {sum(ts) / len(ts):.2f} s -> {2 / (sum(ts) / len(ts)):.3f} scenes/s", flush=True)
