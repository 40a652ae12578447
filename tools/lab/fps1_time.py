import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, kernel_cases as KC
dev = torch.device("cuda:0")
for N, m in ((40000, 2048), (80000, 2048)):
    c = KC.fps(8, N, m, dev)
    print(c["name"], "%.1f us" % KC.time_case(c, iters=5, warm=1), flush=True)
