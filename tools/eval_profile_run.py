"""The inference forward (detector + encoder once + greedy decoding of B*K captions) N times between two marker kernels
(`delay_kernel`, spacap_stream_delay) -- run under rocprofv3 by tools/prof_eval.sh; tools/prof_between.py cuts the window."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spacap3d_amd._native import check, lib  # noqa: E402
from spacap3d_amd.engine import synthetic_batch  # noqa: E402
from spacap3d_amd.spacapnet import build_default  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = build_default().to(dev).eval()
data = synthetic_batch(8, 40000, dev, seed=0)
st = torch.cuda.current_stream(dev).cuda_stream
with torch.no_grad():
    for _ in range(2):
        model(dict(data), is_eval=True)
    torch.cuda.synchronize()
    check(lib.spacap_stream_delay(1, st), "marker")
    t0 = time.perf_counter()
    for _ in range(N):
        d = model(dict(data), is_eval=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    check(lib.spacap_stream_delay(1, st), "marker")
    torch.cuda.synchronize()
print(f"eval forward: {dt / N * 1e3:.2f} ms per forward (8 scenes, 2048 captions x 31 words, host-timed under the profiler), "
      f"caps[0,0,:6]={d['lang_cap'][0, 0, :6].tolist()}")
