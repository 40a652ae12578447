"""Lab: is it the sampling kernel's work or merely the presence of a long-running kernel on another queue that slows the others?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, kernel_cases as KC
from torch.utils.cpp_extension import load_inline
dev = torch.device("cuda:0")
src = r'''
#include <hip/hip_runtime.h>
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
__global__ void spin_kernel(long ticks, int mode, float* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float a = threadIdx.x;
  __shared__ float s[4096];
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) {
    if (mode == 0) __builtin_amdgcn_s_sleep(32);
    else if (mode == 1) { for (int i = 0; i < 64; ++i) a = a * 1.0001f + 0.5f; }
    else { for (int i = 0; i < 64; ++i) { s[(threadIdx.x * 17 + i) & 4095] = a; a += s[(threadIdx.x + i * 33) & 4095]; } }
  }
  if (a == 12345.f) sink[0] = a;
}
void spin(int blocks, int threads, long us, int mode, torch::Tensor sink) {
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), 0, c10::hip::getCurrentHIPStream(), us * 100, mode, sink.data_ptr<float>());
}
'''
m = load_inline("spin_ext", cpp_sources="void spin(int blocks, int threads, long us, int mode, torch::Tensor sink);", cuda_sources=src,
                functions=["spin"], with_cuda=True, extra_cuda_cflags=["--offload-arch=gfx950"], verbose=False)
sink = torch.zeros(1, device=dev)
side = torch.cuda.Stream(device=dev)
R1, R2 = 8 * 2048 * 64, 8 * 1024 * 32
def timed(case, beside, iters):
    for _ in range(3): case["run"]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if beside is not None:
        with torch.cuda.stream(side):
            m.spin(beside[0], beside[1], 4000, beside[2], sink)
    e0.record()
    for _ in range(iters): case["run"]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for make, iters in ((lambda: KC.sa_mid_fwd(R1, 64, 64, dev, "SA1 L2"), 16), (lambda: KC.sa_wgrad(R2, 256, 128, True, 32, dev, "SA2 L3"), 10),
                    (lambda: KC.mha_fwd(8, 8, 256, 16, dev, True), 60)):
    c = make()
    base = timed(c, None, iters)
    out = [f"alone {base:7.1f}"]
    for name, cfg in (("8x1024 sleeping", (8, 1024, 0)), ("8x1024 VALU", (8, 1024, 1)), ("8x1024 LDS", (8, 1024, 2)), ("8x64 VALU", (8, 64, 1)), ("1x64 sleeping", (1, 64, 0))):
        out.append(f"{name} {timed(c, cfg, iters) / base:.2f}x")
    print(f"{c['name'][:44]:44s} " + " | ".join(out), flush=True)
    del c
