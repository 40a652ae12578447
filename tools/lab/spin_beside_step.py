"""Lab: what does a kernel running on ANOTHER queue cost the replayed training step, by the kind of neighbour (no memory traffic at all:
sleeping workgroups)?  Separates "the sampling chain takes CUs / L2" from "a second active queue slows the step's dispatches"."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
probe = ctypes.CDLL(os.path.join(ROOT, "tools", "lab", "libcumask_probe.so"))
probe.probe_spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_default(input_feature_dim=1, num_proposal=256).to(dev).train()
trainer = Trainer(model, S.mean_size_arr().numpy(), use_relation=True)
data = synthetic_batch(8, 40000, dev, seed=1000)
trainer.step(data, next_data=data)
assert trainer.enable_graph(data), trainer.graph_error
def timed(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("pipelined step: %.3f ms" % timed(lambda: trainer.step(data, next_data=data)), flush=True)
trainer.prefetch(data); torch.cuda.synchronize()
saved = data["_fps_prefetch"]
side = torch.cuda.Stream(device=dev)
def with_side(launch):
    def f():
        data["_fps_prefetch"] = saved
        if launch is not None:
            side.wait_stream(torch.cuda.current_stream())
            launch(ctypes.c_void_p(side.cuda_stream))
        trainer.step(data, next_data=None)
    return f
print("no side work: %.3f ms" % timed(with_side(None)), flush=True)
for name, fn in (
    ("8 x 1024 sleeping 5.5 ms", lambda s: probe.probe_spin(s, 8, 1024, 0, 5500)),
    ("8 x 1024 + 64 KB LDS sleeping 5.5 ms", lambda s: probe.probe_spin(s, 8, 1024, 65536, 5500)),
    ("8 x 64 sleeping 5.5 ms", lambda s: probe.probe_spin(s, 8, 64, 0, 5500)),
    ("1 x 64 sleeping 5.5 ms", lambda s: probe.probe_spin(s, 1, 64, 0, 5500)),
    ("25 kernels of 8 x 1024 sleeping 220 us", lambda s: [probe.probe_spin(s, 8, 1024, 0, 220) for _ in range(25)]),
    ("250 kernels of 8 x 64 sleeping 20 us", lambda s: [probe.probe_spin(s, 8, 64, 0, 20) for _ in range(250)]),
    ("1 x 64 sleeping 2 ms", lambda s: probe.probe_spin(s, 1, 64, 0, 2000)),
):
    print("%-44s %.3f ms" % (name, timed(with_side(fn))), flush=True)
