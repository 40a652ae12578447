// Rate of the fp32 MFMAs from registers vs number of independent accumulators, waves per SIMD and instruction shape.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate2 tools/lab/mfma_rate2.hip && /tmp/mfma_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));
template <int NACC, int SHAPE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  float s = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int u = 0; u < 16; ++u) acc[i][u] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < NACC; ++i)
      for (int u = 0; u < 16; ++u) s += acc[i][u];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int SHAPE>
void run(int wg_per_cu) {
  const int grid = 256 * wg_per_cu, iters = 4000;
  float *out; hipMalloc(&out, grid * 256 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, SHAPE>), dim3(grid), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, SHAPE>), dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * 4 * iters * NACC * (SHAPE == 16 ? 2048.0 : 4096.0);
  printf("%dx%d NACC=%2d, %d waves/SIMD: %7.2f ms, %6.1f TFLOP/s\n", SHAPE, SHAPE, NACC, wg_per_cu, ms, flop / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1, 16>(1); run<2, 16>(1); run<4, 16>(1); run<8, 16>(1); run<16, 16>(1); run<32, 16>(1);
  run<1, 16>(2); run<2, 16>(2); run<4, 16>(2); run<8, 16>(2); run<16, 16>(2);
  run<2, 16>(4); run<8, 16>(4);
  run<1, 32>(1); run<2, 32>(1); run<4, 32>(1); run<8, 32>(1);
  run<1, 32>(2); run<2, 32>(2); run<4, 32>(2);
  return 0;
}
