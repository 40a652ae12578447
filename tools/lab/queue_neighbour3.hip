// Lab: a hipGraph chain that ALTERNATES different kernels (code objects, LDS sizes, workgroup sizes, register counts) beside a resident
// neighbour on another queue: does the per-boundary cost grow?  (queue_neighbour.hip: identical kernels, +0.05 us per boundary.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin_kernel(uint64_t ticks) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
template <int V>
__global__ void k_variant(float *p, int us) {
  extern __shared__ float sm[];
  float acc[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = p[(threadIdx.x + i * 64) & 1023];
  if (us > 0) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)us * 100) __builtin_amdgcn_s_sleep(2);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) s += acc[i];
  if (threadIdx.x == 0) p[1024 + blockIdx.x] = s;
}
int main() {
  float *buf; CK(hipMalloc(&buf, 1 << 22)); CK(hipMemset(buf, 0, 1 << 22));
  hipStream_t s, side; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  CK(hipFuncSetAttribute((const void *)k_variant<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)k_variant<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)k_variant<200>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int K = 300;
  for (int mode = 0; mode < 4; ++mode) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < K; ++k) {
      const int us = mode >= 2 ? 3 : 0;
      if (mode == 0 || mode == 2) hipLaunchKernelGGL(k_variant<4>, dim3(256), dim3(256), 0, s, buf, us);
      else switch (k % 3) {
        case 0: hipLaunchKernelGGL(k_variant<4>, dim3(128), dim3(256), 0, s, buf, us); break;
        case 1: hipLaunchKernelGGL(k_variant<64>, dim3(256), dim3(512), 100 * 1024, s, buf, us); break;
        default: hipLaunchKernelGGL(k_variant<200>, dim3(1024), dim3(256), 16 * 1024, s, buf, us); break;
      }
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%s:", mode == 0 ? "same kernel, empty body   " : mode == 1 ? "three kernels, empty body  " : mode == 2 ? "same kernel, 3 us body    " : "three kernels, 3 us body   ");
    struct Nb { const char *name; int grid, threads; };
    const Nb nbs[] = {{"alone", 0, 0}, {"8x64 sleeping", 8, 64}, {"8x1024 sleeping", 8, 1024}, {"1x64 sleeping", 1, 64}, {"alone", 0, 0}};
    for (const Nb &n : nbs) {
      CK(hipDeviceSynchronize());
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      if (n.grid) hipLaunchKernelGGL(spin_kernel, dim3(n.grid), dim3(n.threads), 0, side, (uint64_t)100000 * 100);
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf(" | %s %.2f us/kernel", n.name, ms * 1e3 / 10 / K);
      CK(hipDeviceSynchronize());
    }
    printf("\n"); fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
