// Standalone ablation harness for the large-scene FPS kernel (not part of the product build).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fps_lab tools/lab/fps_lab.hip && ./fps_lab
// Variants are compile-time flags so that the cost of each phase of a round can be read off by difference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include "../../spacap3d_amd/csrc/common.hpp"
namespace spacap { void set_error(const char*, ...) {} }
#pragma clang fp contract(off)
using namespace spacap;
using f32x4 = float __attribute__((ext_vector_type(4)));
using u32x4 = unsigned int __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned fps_key(int k, int lg) {
  const unsigned low = (unsigned)k & ((1u << lg) - 1u);
  const unsigned rev = lg ? (__brev(low) >> (32 - lg)) : 0u;
  return (rev << 20) | ((unsigned)k >> lg);
}
__device__ __forceinline__ int fps_unkey(unsigned key, int lg) {
  const unsigned rev = key >> 20;
  const unsigned low = lg ? (__brev(rev) >> (32 - lg)) : 0u;
  return (int)(((key & 0xFFFFFu) << lg) | low);
}
__device__ __forceinline__ f32x4 sqdist4(f32x4 x, f32x4 y, f32x4 z, float x1, float y1, float z1) {
  const f32x4 dx = x - x1, dy = y - y1, dz = z - z1;
  return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ float vmin(float a, float b) {
#ifdef ASM_MIN
  float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#else
  return fminf(a, b);
#endif
}
__device__ __forceinline__ int update4(f32x4 &t, f32x4 d, int lmax) {
  t.x = vmin(d.x, t.x); t.y = vmin(d.y, t.y); t.z = vmin(d.z, t.z); t.w = vmin(d.w, t.w);
  lmax = max(lmax, max(__float_as_int(t.x), __float_as_int(t.y)));
  return max(lmax, max(__float_as_int(t.z), __float_as_int(t.w)));
}

// FLAGS bit0: skip compute of REG groups, bit1: skip LDS groups, bit2: skip STREAM groups,
//       bit3: skip index search (use fake key), bit4: skip centre load (fixed centre), bit5: no barriers/reduction
template <int G_REG, int G_LDS, int G_STR, int FLAGS>
__global__ __launch_bounds__(1024) void fps_hybrid_kernel(const float *__restrict__ xyz_all,
                                                          float *__restrict__ ws_all, int N, int m, int lg,
                                                          int32_t *__restrict__ idx_all) {
  constexpr int BLOCK = 1024, NW = 16, G = G_REG + G_LDS + G_STR;
  constexpr int NPAD = G * 4 * BLOCK;
  __shared__ __attribute__((aligned(16))) f32x4 s_pts[(G_LDS > 0 ? G_LDS : 1) * 3 * BLOCK];
  __shared__ int s_wmax[16];
  __shared__ unsigned s_key[2];
  const float *__restrict__ xyz = xyz_all + (size_t)blockIdx.x * N * 3;
  float *__restrict__ planes = ws_all + (size_t)blockIdx.x * 3 * NPAD;
  int32_t *__restrict__ idxs = idx_all + (size_t)blockIdx.x * m;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  f32x4 t[G];
  f32x4 rx[G_REG > 0 ? G_REG : 1], ry[G_REG > 0 ? G_REG : 1], rz[G_REG > 0 ? G_REG : 1];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int k0 = (g * BLOCK + tid) * 4;
    f32x4 x, y, z, tt;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u;
      const int kk = k < N ? k : N - 1;
      const float px = xyz[kk * 3 + 0], py = xyz[kk * 3 + 1], pz = xyz[kk * 3 + 2];
      const float mag = (px * px) + (py * py) + (pz * pz);
      const bool skip = (k >= N) || ((double)mag <= 1e-3);
      x[u] = px; y[u] = py; z[u] = pz;
      tt[u] = skip ? -1.0f : 1e10f;
    }
    t[g] = tt;
    if (g < G_REG) { rx[g] = x; ry[g] = y; rz[g] = z; }
    else if (g < G_REG + G_LDS) {
      const int l = g - G_REG;
      s_pts[(l * 3 + 0) * BLOCK + tid] = x; s_pts[(l * 3 + 1) * BLOCK + tid] = y; s_pts[(l * 3 + 2) * BLOCK + tid] = z;
    } else {
      *reinterpret_cast<f32x4 *>(planes + 0 * NPAD + k0) = x;
      *reinterpret_cast<f32x4 *>(planes + 1 * NPAD + k0) = y;
      *reinterpret_cast<f32x4 *>(planes + 2 * NPAD + k0) = z;
    }
  }
  if (tid == 0) { idxs[0] = 0; s_key[0] = 0xFFFFFFFFu; s_key[1] = 0xFFFFFFFFu; }
  __threadfence_block();
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)planes, 0, 3 * NPAD * 4, 0x00020000);
  const int voff = tid * 16;
  int old = 0;
  for (int j = 1; j < m; ++j) {
    float x1, y1, z1;
    if (FLAGS & 16) { x1 = 0.1f * j; y1 = 0.2f; z1 = 0.3f; }
    else { x1 = xyz[old * 3 + 0]; y1 = xyz[old * 3 + 1]; z1 = xyz[old * 3 + 2]; }
    int lmax = __float_as_int(-1.0f);
    f32x4 sx[2], sy[2], sz[2];
    if (G_STR > 0 && !(FLAGS & 4)) {
      const int soff = G_REG + G_LDS;
      sx[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (0 * NPAD + soff * 4 * BLOCK) * 4, 0));
      sy[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (1 * NPAD + soff * 4 * BLOCK) * 4, 0));
      sz[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (2 * NPAD + soff * 4 * BLOCK) * 4, 0));
    }
    if (!(FLAGS & 1)) {
#pragma unroll
      for (int g = 0; g < G_REG; ++g) lmax = update4(t[g], sqdist4(rx[g], ry[g], rz[g], x1, y1, z1), lmax);
    }
    if (!(FLAGS & 2)) {
#pragma unroll
      for (int l = 0; l < G_LDS; ++l) {
        const f32x4 x = s_pts[(l * 3 + 0) * BLOCK + tid];
        const f32x4 y = s_pts[(l * 3 + 1) * BLOCK + tid];
        const f32x4 z = s_pts[(l * 3 + 2) * BLOCK + tid];
        lmax = update4(t[G_REG + l], sqdist4(x, y, z, x1, y1, z1), lmax);
      }
    }
    if (!(FLAGS & 4)) {
#pragma unroll
      for (int q = 0; q < G_STR; ++q) {
        if (q + 1 < G_STR) {
          const int gq = G_REG + G_LDS + q + 1;
          sx[(q + 1) & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (0 * NPAD + gq * 4 * BLOCK) * 4, 0));
          sy[(q + 1) & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (1 * NPAD + gq * 4 * BLOCK) * 4, 0));
          sz[(q + 1) & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (2 * NPAD + gq * 4 * BLOCK) * 4, 0));
        }
        asm volatile("" ::: "memory");
        lmax = update4(t[G_REG + G_LDS + q], sqdist4(sx[q & 1], sy[q & 1], sz[q & 1], x1, y1, z1), lmax);
      }
    }
    if (FLAGS & 32) { old = (lmax & 1023) % N; if (tid == 0) idxs[j] = old; old = __builtin_amdgcn_readfirstlane(old); continue; }
    const int wmax = wave_max_i32(lmax);
    if (lane == 0) s_wmax[wid] = wmax;
    __syncthreads();
    int M = s_wmax[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = max(M, s_wmax[w]);
    if (wmax == M && M >= 0) {
      unsigned key = 0xFFFFFFFFu;
      if (FLAGS & 8) { key = (unsigned)((wid * 64 + lane) * 4) ; key = fps_key(key % N, lg); }
      else {
        int lgv = lg;
        asm volatile("" : "+s"(lgv));
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned ki = fps_key((g * BLOCK + tid) * 4 + u, lgv);
            key = (__float_as_int(t[g][u]) == M) ? min(key, ki) : key;
          }
        }
      }
      key = wave_min_u32(key);
      if (lane == 0) atomicMin(&s_key[j & 1], key);
    }
    __syncthreads();
    const unsigned key = s_key[j & 1];
    if (tid == 0) s_key[(j + 1) & 1] = 0xFFFFFFFFu;
    old = (M < 0) ? 0 : fps_unkey(key, lg);
    old = __builtin_amdgcn_readfirstlane(old);
    if (tid == 0) idxs[j] = old;
  }
}

template <int GR, int GL, int GS, int FLAGS>
float run(const float *xyz, float *ws, int B, int N, int m, int32_t *idx, const char *name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((fps_hybrid_kernel<GR, GL, GS, FLAGS>), dim3(B), dim3(1024), 0, 0, xyz, ws, N, m, 9, idx);
  hipEventRecord(e0);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((fps_hybrid_kernel<GR, GL, GS, FLAGS>), dim3(B), dim3(1024), 0, 0, xyz, ws, N, m, 9, idx);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("%-44s G=(%d,%d,%d) %8.3f ms  %6.3f us/round\n", name, GR, GL, GS, ms, ms * 1e3 / (m - 1));
  return ms;
}

int main() {
  const int B = 8, N = 40000, m = 2048;
  std::vector<float> h((size_t)B * N * 3);
  srand(1);
  for (auto &v : h) v = (float)rand() / RAND_MAX * 6.f - 3.f;
  float *xyz, *ws; int32_t *idx;
  hipMalloc(&xyz, h.size() * 4); hipMalloc(&ws, (size_t)B * 3 * 40960 * 4); hipMalloc(&idx, (size_t)B * m * 4);
  hipMemcpy(xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<3, 3, 4, 0>(xyz, ws, B, N, m, idx, "full");
  run<3, 3, 4, 32>(xyz, ws, B, N, m, idx, "compute only");
  run<3, 3, 4, 32 + 4 + 2>(xyz, ws, B, N, m, idx, "compute only, reg groups only");
  run<3, 3, 4, 32 + 1 + 2>(xyz, ws, B, N, m, idx, "compute only, stream only");
  run<3, 3, 4, 32 + 1 + 4>(xyz, ws, B, N, m, idx, "compute only, LDS only");
  return 0;
}
