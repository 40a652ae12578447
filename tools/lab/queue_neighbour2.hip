// Lab: a chain of whole-CU kernels (one 100 KB-LDS workgroup per CU, G workgroups, 10 us body) beside 8 FAT resident workgroups
// (1024 threads + 100 KB LDS each: nothing of the chain can share their CUs).  Which G avoids a second round?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin_kernel(uint64_t ticks, uint32_t *where) {
  extern __shared__ float sm[];
  if (threadIdx.x == 0 && where) { where[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20); where[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4); }
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void body_kernel(float *p, int us, uint32_t *where) {
  extern __shared__ float sm[];
  if (threadIdx.x == 0 && where) { where[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20); where[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4); }
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)us * 100) __builtin_amdgcn_s_sleep(2);
  if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
}
int main() {
  float *buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
  uint32_t *wh; CK(hipMalloc(&wh, 1 << 16)); uint32_t *wh2; CK(hipMalloc(&wh2, 1 << 16));
  hipStream_t s, side; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  CK(hipFuncSetAttribute((const void *)body_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int K = 100, LDS = 100 * 1024;
  const int grids[] = {256, 248, 240, 232, 224, 216, 208, 192};
  for (int fat = 0; fat < 3; ++fat) {
    printf(fat == 0 ? "alone:\n" : fat == 1 ? "beside 8 fat resident workgroups (one per XCD by round robin):\n" : "beside 16 fat resident workgroups:\n");
    for (int G : grids) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < K; ++k) hipLaunchKernelGGL(body_kernel, dim3(G), dim3(512), LDS, s, buf, 10, (uint32_t *)nullptr);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipDeviceSynchronize());
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      if (fat) hipLaunchKernelGGL(spin_kernel, dim3(8 * fat), dim3(1024), LDS, side, (uint64_t)60000 * 100, wh);
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  grid %3d: %.2f us/kernel\n", G, ms * 1e3 / 5 / K); fflush(stdout);
      CK(hipDeviceSynchronize());
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    if (fat) {
      uint32_t h[64]; CK(hipMemcpy(h, wh, sizeof(uint32_t) * 2 * 8 * fat, hipMemcpyDeviceToHost));
      printf("  resident workgroups sat on (xcc, se, cu):");
      for (int i = 0; i < 8 * fat; ++i) printf(" (%u,%u,%u)", h[2 * i] & 15, (h[2 * i + 1] >> 13) & 7, (h[2 * i + 1] >> 8) & 15);
      printf("\n");
    }
  }
  // where does a 248-grid land beside the 8 fat ones?  count workgroups per (xcc, se)
  hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(1024), LDS, side, (uint64_t)3000 * 100, wh);
  hipLaunchKernelGGL(body_kernel, dim3(248), dim3(512), LDS, s, buf, 50, wh2);
  CK(hipDeviceSynchronize());
  uint32_t h2[2 * 248]; CK(hipMemcpy(h2, wh2, sizeof(h2), hipMemcpyDeviceToHost));
  int cnt[8][8] = {};
  for (int i = 0; i < 248; ++i) cnt[h2[2 * i] & 15][(h2[2 * i + 1] >> 13) & 7]++;
  printf("248-grid beside 8 fat: workgroups per (xcc: se0 se1 se2 se3):");
  for (int x = 0; x < 8; ++x) printf("  %d: %d %d %d %d", x, cnt[x][0], cnt[x][1], cnt[x][2], cnt[x][3]);
  printf("\n");
  return 0;
}
