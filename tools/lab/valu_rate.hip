// VALU issue-rate probe: one workgroup on one CU, T threads, independent fp32 ops.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ void k(float *out, int iters, float s) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
      if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 2) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
      if (KIND == 3) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 4) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += a[i];
  out[threadIdx.x + blockIdx.x * blockDim.x] = r;
}
template <int KIND>
void run(const char *name, int T, int blocks) {
  float *out; hipMalloc(&out, 1 << 22);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(T), 0, 0, out, iters, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(T), 0, 0, out, iters, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = (double)iters * 16 * (T / 64) / 4.0;  // wave-instructions per SIMD (one block per CU)
  printf("%-14s T=%4d blocks=%3d  %8.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, T, blocks, ms,
         ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  hipFree(out);
}
int main() {
  for (int T : {256, 512, 1024}) {
    run<0>("v_fma_f32", T, 1); run<1>("v_mul_f32", T, 1); run<2>("v_subrev sgpr", T, 1); run<3>("v_min_f32", T, 1); run<4>("v_max_i32", T, 1);
  }
  run<0>("v_fma_f32", 1024, 8); run<0>("v_fma_f32", 1024, 256);
  return 0;
}
