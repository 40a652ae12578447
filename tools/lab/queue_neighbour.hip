// Lab: what does a long-running kernel on ANOTHER queue cost a hipGraph chain of short dependent kernels?
//   hipcc --offload-arch=gfx950 -O2 tools/lab/queue_neighbour.hip -o tools/lab/queue_neighbour && ./tools/lab/queue_neighbour
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin_kernel(uint64_t ticks) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
// body: ~`us` microseconds of sleeping per workgroup, then one store
__global__ void tiny_kernel(float *p, int us) {
  extern __shared__ float sm[];
  if (us > 0) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)us * 100) __builtin_amdgcn_s_sleep(2);
  }
  if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
}

int main() {
  float *buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
  hipStream_t s, side; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t side_lo, side_hi; CK(hipStreamCreateWithPriority(&side_lo, hipStreamNonBlocking, lo)); CK(hipStreamCreateWithPriority(&side_hi, hipStreamNonBlocking, hi));
  printf("priority range: least %d greatest %d\n", lo, hi);
  const int K = 300;
  struct Case { const char *name; int grid, threads, lds, us; };
  const Case cases[] = {{"256 x 256, no LDS, empty body", 256, 256, 0, 0}, {"256 x 256, empty, 100 KB LDS", 256, 256, 100 * 1024, 0},
                        {"1024 x 256, empty", 1024, 256, 0, 0}, {"128 x 256, 5 us body", 128, 256, 0, 5}, {"256 x 512 100 KB LDS, 10 us body", 256, 512, 100 * 1024, 10},
                        {"248 x 512 100 KB LDS, 10 us body", 248, 512, 100 * 1024, 10}};
  CK(hipFuncSetAttribute((const void *)tiny_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (const Case &c : cases) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny_kernel, dim3(c.grid), dim3(c.threads), c.lds, s, buf, c.us);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Nb { const char *name; hipStream_t st; int grid, threads, us, count; };
    const Nb nbs[] = {{"alone", nullptr, 0, 0, 0, 0}, {"8x64 sleeping", side, 8, 64, 100000, 1}, {"1x64 sleeping", side, 1, 64, 100000, 1},
                      {"8x1024 sleeping", side, 8, 1024, 100000, 1}, {"8x64 sleeping, low-priority stream", side_lo, 8, 64, 100000, 1},
                      {"8x64 sleeping, high-priority stream", side_hi, 8, 64, 100000, 1}, {"64x64 sleeping", side, 64, 64, 100000, 1}};
    printf("%-40s", c.name);
    for (const Nb &n : nbs) {
      CK(hipDeviceSynchronize());
      for (int w = 0; w < 2; ++w) CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      if (n.st) hipLaunchKernelGGL(spin_kernel, dim3(n.grid), dim3(n.threads), 0, n.st, (uint64_t)n.us * 100);
      const int reps = 10;
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf(" | %s %.2f us/kernel", n.name, ms * 1e3 / reps / K);
      CK(hipDeviceSynchronize());
    }
    printf("\n"); fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
