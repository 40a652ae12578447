// Peak rate of v_mfma_f32_16x16x4_f32 from registers only: calibrates what "MFMA bound" means on this part.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/lab/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = float __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wg_per_cu) {
  const int grid = 256 * wg_per_cu, iters = 4000;
  float *out; hipMalloc(&out, grid * 256 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * 4 * iters * NACC * 2048.0;
  printf("NACC=%d, %d WG/CU: %.2f ms, %.1f TFLOP/s\n", NACC, wg_per_cu, ms, flop / ms / 1e9);
  hipFree(out);
}
int main() {
  run<8>(1); run<8>(2); run<4>(2); run<16>(1); run<8>(4);
  return 0;
}
