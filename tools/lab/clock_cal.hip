// Lab: what does s_memtime count, and what is the shader clock under load?
//   kernel A: every wave runs N x 16 x "s_nop 15" (16 cycles each at the shader clock) -- light load
//   kernel B: every wave runs N dependent-free bf16 MFMAs on 4 accumulators (32 cycles each, matrix pipe saturated) -- heavy load
// Reports s_memtime ticks, s_memrealtime ticks and the HIP-event time of each launch.
//   hipcc --offload-arch=gfx950 -O3 -o clock_cal tools/lab/clock_cal.hip && ./clock_cal
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void nops(int n, unsigned long long *out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                 "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0, out[1] = r1 - r0;
}
__global__ __launch_bounds__(256) void mfmas(int n, unsigned long long *out, float *sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x + i), b[i] = (__bf16)(float)(i + 1);
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0, out[1] = r1 - r0;
  if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[0] = 1.f;
}
template <int KIND>
__global__ __launch_bounds__(256) void mfmas32(int n, unsigned long long *out, float *sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float a = (float)threadIdx.x, b = 1.5f;
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {}, d4 = {}, d5 = {}, d6 = {}, d7 = {};
  for (int i = 0; i < n; ++i) {
    if (KIND == 0) {   // 32x32x2 f32, 4 accumulators
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    } else {           // 16x16x4 f32, 8 accumulators
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d3, 0, 0, 0);
      d4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d4, 0, 0, 0);
      d5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d5, 0, 0, 0);
      d6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d6, 0, 0, 0);
      d7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d7, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
  if (c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[1] + d2[2] + d3[3] + d4[0] + d5[0] + d6[0] + d7[0] == 12345.f) sink[0] = 1.f;
}
template <class F> float timed(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
  unsigned long long *out, h[2]; float *sink;
  hipMalloc(&out, 16); hipMalloc(&sink, 4);
  for (int rep = 0; rep < 2; ++rep) {
    const int n = 20000;
    float us = timed([&] { hipLaunchKernelGGL(nops, dim3(256), dim3(64), 0, 0, n, out); });
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("nops  (1 wave/CU): %d x 256 cycles = %.0f cycles | s_memtime %llu s_memrealtime %llu | %.1f us -> %.0f cycles/us, %.1f memtime ticks/us\n",
           n, n * 256.0, h[0], h[1], us, n * 256.0 / us, h[0] / us);
    const int m = 40000;
    us = timed([&] { hipLaunchKernelGGL(mfmas, dim3(256 * 2), dim3(256), 0, 0, m, out, sink); });
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("mfmas (8 waves/CU, 2 per SIMD): %d x 4 x 32 cycles x 2 waves = %.0f cycles | s_memtime %llu s_memrealtime %llu | %.1f us -> %.0f cycles/us, %.1f memtime ticks/us\n",
           m, m * 4 * 32.0 * 2, h[0], h[1], us, m * 4 * 32.0 * 2 / us, h[0] / us);
    for (int wpc = 1; wpc <= 2; ++wpc) {
      const int k = 20000;
      us = timed([&] { hipLaunchKernelGGL(mfmas32<0>, dim3(256 * wpc), dim3(256), 0, 0, k, out, sink); });
      hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
      printf("fp32 32x32x2 (%d waves/SIMD): %.1f us, %llu ticks -> clock %.0f MHz, %.1f ticks per MFMA per SIMD, %.1f TFLOP/s\n", wpc, us,
             h[0], h[0] / us, (double)h[0] / (k * 4.0 * wpc), 256.0 * wpc * 4 * k * 4.0 * 4096 / us * 1e-6);
      us = timed([&] { hipLaunchKernelGGL(mfmas32<1>, dim3(256 * wpc), dim3(256), 0, 0, k, out, sink); });
      hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
      printf("fp32 16x16x4 (%d waves/SIMD): %.1f us, %llu ticks -> clock %.0f MHz, %.1f ticks per MFMA per SIMD, %.1f TFLOP/s\n", wpc, us,
             h[0], h[0] / us, (double)h[0] / (k * 8.0 * wpc), 256.0 * wpc * 4 * k * 8.0 * 2048 / us * 1e-6);
    }
  }
  return 0;
}
