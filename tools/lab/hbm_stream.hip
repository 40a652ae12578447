// Lab: what one pass of 16-byte reads over a large fp32 array reaches on this chip, by launch shape.
//   plain: grid-stride, every thread sums its float4s (UNR independent loads in flight)
//   tiles: workgroups of NT threads own 16 KB tiles t, t + grid, ...; the rows of the next NS tiles sit in registers,
//          each tile goes through LDS with one barrier (the skeleton of the streaming shared-MLP kernels)
// build: hipcc --offload-arch=gfx950 -O3 tools/lab/hbm_stream.hip -o tools/lab/hbm_stream ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = float __attribute__((ext_vector_type(4)));

template <int UNR>
__global__ __launch_bounds__(256) void plain(const f32x4 *__restrict__ x, long n4, float *out) {
  f32x4 acc = {0, 0, 0, 0};
  long i = (long)blockIdx.x * 256 * UNR + threadIdx.x;
  const long step = (long)gridDim.x * 256 * UNR;
  for (; i + (UNR - 1) * 256 < n4; i += step) {
    f32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = x[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

template <int NT, int NS>
__global__ __launch_bounds__(NT) void tiles(const f32x4 *__restrict__ x, long ntiles, float *out) {
  constexpr int T4 = 1024, NV = T4 / NT;   // 16 KB tiles
  __shared__ f32x4 s[2][T4];
  f32x4 p[NS][NV];
  long t = blockIdx.x;
  const long g = gridDim.x;
  const int tid = threadIdx.x;
  auto request = [&](f32x4(&q)[NV], long tt) {
#pragma unroll
    for (int i = 0; i < NV; ++i) q[i] = x[tt * T4 + i * NT + tid];
  };
#pragma unroll
  for (int k = 0; k < NS; ++k)
    if (t + k * g < ntiles) request(p[k], t + k * g);
  f32x4 acc = {0, 0, 0, 0};
  int buf = 0;
  for (bool more = t < ntiles; more;) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
#pragma unroll
      for (int i = 0; i < NV; ++i) s[buf][i * NT + tid] = p[k][i];
      if (t + NS * g < ntiles) request(p[k], t + NS * g);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NV; ++i) acc += s[buf][(i * NT + tid + 64) % T4];
      buf ^= 1;
      t += g;
      if (t >= ntiles) {
        more = false;
        break;
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

template <typename F>
float time_us(F f, int iters = 10) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  f(), f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main() {
  for (long mb : {134L, 268L, 805L}) {
    const long n4 = mb * 1000 * 1000 / 16 / 1024 * 1024;
    f32x4 *x;
    float *out;
    hipMalloc(&x, n4 * 16), hipMalloc(&out, 4);
    hipMemset(x, 0, n4 * 16);
    printf("== %ld MB\n", mb);
    for (int wgs : {256, 512, 1024, 2048, 4096}) {
      float t4 = time_us([&] { hipLaunchKernelGGL(plain<4>, dim3(wgs), dim3(256), 0, 0, x, n4, out); });
      float t8 = time_us([&] { hipLaunchKernelGGL(plain<8>, dim3(wgs), dim3(256), 0, 0, x, n4, out); });
      printf("plain  %5d wgs: unr4 %7.1f us %5.2f TB/s | unr8 %7.1f us %5.2f TB/s\n", wgs, t4, n4 * 16 / t4 * 1e-6, t8, n4 * 16 / t8 * 1e-6);
    }
    const long nt = n4 / 1024;
    for (int wgs : {256, 512, 768, 1024}) {
      float a = time_us([&] { hipLaunchKernelGGL((tiles<256, 2>), dim3(wgs), dim3(256), 0, 0, x, nt, out); });
      float b = time_us([&] { hipLaunchKernelGGL((tiles<256, 4>), dim3(wgs), dim3(256), 0, 0, x, nt, out); });
      float c = time_us([&] { hipLaunchKernelGGL((tiles<512, 2>), dim3(wgs), dim3(512), 0, 0, x, nt, out); });
      float d = time_us([&] { hipLaunchKernelGGL((tiles<512, 4>), dim3(wgs), dim3(512), 0, 0, x, nt, out); });
      printf("tiles  %5d wgs: 256thr ns2 %6.1f us %5.2f TB/s, ns4 %6.1f us %5.2f | 512thr ns2 %6.1f us %5.2f, ns4 %6.1f us %5.2f TB/s\n", wgs, a,
             n4 * 16 / a * 1e-6, b, n4 * 16 / b * 1e-6, c, n4 * 16 / c * 1e-6, d, n4 * 16 / d * 1e-6);
    }
    hipFree(x), hipFree(out);
  }
  return 0;
}
