// Furthest point sampling for gfx950 (MI355X).
//
// Replaces the reference kernel lib/pointnet2/_ext_src/src/sampling_gpu.cu:69-173 (host wrapper
// src/sampling.cpp:66-87).  Results are bit-identical to that kernel's, including
//   * the |p|^2 <= 1e-3 skip (sampling_gpu.cu:100-101; the literal is a double),
//   * the un-contracted fp32 distance, left to right (:103-104)  [this file is built with
//     -ffp-contract=off and carries the pragma below],
//   * the tie-break of its 512-slot shared-memory tree, in which the lower slot wins at every level
//     (:59-65, :115-168): among equal maxima the winner minimises
//     (bitreverse_{log2 bs}(k mod bs), k div bs) with bs = opt_n_threads(N).
//
// Design (not a translation): the reference streams xyz and `temp` from global memory every round and
// spends nine __syncthreads() per round on the tree.  Here one workgroup owns a scene, every lane
// keeps the running minimum distance of its points in VGPRs for the whole kernel (and their
// coordinates too when they fit), the arg-max is a DPP wave reduction on the value only, and the
// index is recovered by a second, rare pass that only the wave(s) holding the maximum execute.
// Two barriers per round.  Skipped / padded slots carry temp = -1: fminf(d, -1) = -1 keeps them
// out of every arg-max exactly as the reference's `continue` does.
#include "common.hpp"

#pragma clang fp contract(off)

namespace {

using namespace spacap;

__device__ __forceinline__ float sqdist(float x2, float y2, float z2, float x1, float y1, float z1) {
  const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
  return dx * dx + dy * dy + dz * dz;
}

// key(k) orders candidates the way the reference's tree does; smaller key wins.
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

template <int BLOCK, int TPL>
__global__ __launch_bounds__(BLOCK) void fps_kernel(const float *__restrict__ xyz_all, int N, int m,
                                                    int lg, int32_t *__restrict__ idx_all) {
  constexpr int NW = BLOCK / 64;
  __shared__ int s_wmax[16];
  __shared__ unsigned s_key[2];

  const float *__restrict__ xyz = xyz_all + (size_t)blockIdx.x * N * 3;
  int32_t *__restrict__ idxs = idx_all + (size_t)blockIdx.x * m;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  float t[TPL];
  float px[TPL], py[TPL], pz[TPL];
#pragma unroll
  for (int i = 0; i < TPL; ++i) {
    const int k = tid + i * BLOCK;
    const int kk = k < N ? k : N - 1;
    const float x = xyz[kk * 3 + 0], y = xyz[kk * 3 + 1], z = xyz[kk * 3 + 2];
    const float mag = (x * x) + (y * y) + (z * z);
    const bool skip = (k >= N) || ((double)mag <= 1e-3);
    t[i] = skip ? -1.0f : 1e10f;
    px[i] = x; py[i] = y; pz[i] = z;
  }
  if (tid == 0) {
    idxs[0] = 0;
    s_key[0] = 0xFFFFFFFFu;
    s_key[1] = 0xFFFFFFFFu;
  }
  __syncthreads();

  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
    int lmax = __float_as_int(-1.0f);
#pragma unroll
    for (int i = 0; i < TPL; ++i) {
      const float d = sqdist(px[i], py[i], pz[i], x1, y1, z1);
      t[i] = fminf(d, t[i]);
      // values are -1 or >= +0 and never NaN, so signed-integer order == float order
      lmax = max(lmax, __float_as_int(t[i]));
    }
    const int wmax = wave_max_i32(lmax);
    if (NW > 1) {
      if (lane == 0) s_wmax[wid] = wmax;
      __syncthreads();
    }
    int M = wmax;
    if (NW > 1) {
      M = s_wmax[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) M = max(M, s_wmax[w]);
    }
    if (wmax == M && M >= 0) {  // wave-uniform: only waves that hold the maximum look for its index
      unsigned key = 0xFFFFFFFFu;
      int lgv = lg;
      asm volatile("" : "+s"(lgv));  // keep the key arithmetic inside this rare branch (no hoist, no spill)
#pragma unroll
      for (int i = 0; i < TPL; ++i) {
        const unsigned ki = fps_key(tid + i * BLOCK, lgv);
        key = (__float_as_int(t[i]) == M) ? min(key, ki) : key;
      }
      key = wave_min_u32(key);
      if (lane == 0) atomicMin(&s_key[j & 1], key);
    }
    __syncthreads();
    const unsigned key = s_key[j & 1];
    if (tid == 0) s_key[(j + 1) & 1] = 0xFFFFFFFFu;
    old = (M < 0) ? 0 : fps_unkey(key, lg);  // every point skipped: the reference returns index 0
    old = __builtin_amdgcn_readfirstlane(old);
    if (tid == 0) idxs[j] = old;
  }
}

// Any N: temp lives in the caller's workspace (as in the reference), same round structure.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fps_generic_kernel(const float *__restrict__ xyz_all,
                                                            float *__restrict__ temp_all, int N, int m,
                                                            int lg, int32_t *__restrict__ idx_all) {
  constexpr int NW = BLOCK / 64;
  __shared__ int s_wmax[16];
  __shared__ unsigned s_key[2];
  const float *__restrict__ xyz = xyz_all + (size_t)blockIdx.x * N * 3;
  float *__restrict__ temp = temp_all + (size_t)blockIdx.x * N;
  int32_t *__restrict__ idxs = idx_all + (size_t)blockIdx.x * m;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  for (int k = tid; k < N; k += BLOCK) {
    const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
    const float mag = (x * x) + (y * y) + (z * z);
    temp[k] = ((double)mag <= 1e-3) ? -1.0f : 1e10f;
  }
  if (tid == 0) {
    idxs[0] = 0;
    s_key[0] = 0xFFFFFFFFu;
    s_key[1] = 0xFFFFFFFFu;
  }
  __syncthreads();
  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
    int lmax = __float_as_int(-1.0f);
    for (int k = tid; k < N; k += BLOCK) {
      const float d = sqdist(xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2], x1, y1, z1);
      const float d2 = fminf(d, temp[k]);
      temp[k] = d2;
      lmax = max(lmax, __float_as_int(d2));
    }
    const int wmax = wave_max_i32(lmax);
    if (lane == 0) s_wmax[wid] = wmax;
    __syncthreads();
    int M = s_wmax[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = max(M, s_wmax[w]);
    if (wmax == M && M >= 0) {
      unsigned key = 0xFFFFFFFFu;
      for (int k = tid; k < N; k += BLOCK)
        key = (__float_as_int(temp[k]) == M) ? min(key, fps_key(k, lg)) : key;
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

// ---------------------------------------------------------------------------------------------
// Large scenes (20 480 < N <= 40 960): one 1024-thread workgroup per scene cannot hold 40 points x
// (x, y, z, temp) per lane in its 128 VGPRs, and re-reading xyz from L2 every round costs ~3 us/round
// (480 KB at ~64 B/clk/CU).  So the lane's points are split three ways, in groups of four
// consecutive points (one 16-byte access per coordinate plane):
//   G_REG groups  coordinates in VGPRs for the whole kernel,
//   G_LDS groups  coordinates in LDS ([group][plane][lane] float4: conflict-free ds_read_b128),
//   G_STR groups  coordinates re-read every round from a structure-of-arrays copy of the scene in the
//                 caller's workspace (coalesced buffer_load_dwordx4, one group prefetched ahead),
// while every temp stays in VGPRs.  Point k belongs to group g = k / 4096, lane (k / 4) % 1024.
using f32x4 = float __attribute__((ext_vector_type(4)));
using u32x4 = unsigned int __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 sqdist4(f32x4 x, f32x4 y, f32x4 z, float x1, float y1, float z1) {
  const f32x4 dx = x - x1, dy = y - y1, dz = z - z1;
  return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ int update4(f32x4 &t, f32x4 d, int lmax) {
  t.x = fminf(d.x, t.x); t.y = fminf(d.y, t.y); t.z = fminf(d.z, t.z); t.w = fminf(d.w, t.w);
  lmax = max(lmax, max(__float_as_int(t.x), __float_as_int(t.y)));
  return max(lmax, max(__float_as_int(t.z), __float_as_int(t.w)));
}

template <int G_REG, int G_LDS, int G_STR>
__global__ __launch_bounds__(1024) void fps_hybrid_kernel(const float *__restrict__ xyz_all,
                                                          float *__restrict__ ws_all, int N, int m, int lg,
                                                          int32_t *__restrict__ idx_all) {
  constexpr int BLOCK = 1024, NW = 16, G = G_REG + G_LDS + G_STR;
  constexpr int NPAD = G * 4 * BLOCK;
  __shared__ __attribute__((aligned(16))) f32x4 s_pts[(G_LDS > 0 ? G_LDS : 1) * 3 * BLOCK];
  __shared__ int s_wmax[16];
  __shared__ unsigned s_key[2];

  const float *__restrict__ xyz = xyz_all + (size_t)blockIdx.x * N * 3;
  float *__restrict__ planes = ws_all + (size_t)blockIdx.x * 3 * NPAD;  // x[NPAD] y[NPAD] z[NPAD]
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
    if (g < G_REG) {
      rx[g] = x; ry[g] = y; rz[g] = z;
    } else if (g < G_REG + G_LDS) {
      const int l = g - G_REG;
      s_pts[(l * 3 + 0) * BLOCK + tid] = x;
      s_pts[(l * 3 + 1) * BLOCK + tid] = y;
      s_pts[(l * 3 + 2) * BLOCK + tid] = z;
    } else {
      *reinterpret_cast<f32x4 *>(planes + 0 * NPAD + k0) = x;
      *reinterpret_cast<f32x4 *>(planes + 1 * NPAD + k0) = y;
      *reinterpret_cast<f32x4 *>(planes + 2 * NPAD + k0) = z;
    }
  }
  if (tid == 0) {
    idxs[0] = 0;
    s_key[0] = 0xFFFFFFFFu;
    s_key[1] = 0xFFFFFFFFu;
  }
  __threadfence_block();
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)planes, 0, 3 * NPAD * 4, 0x00020000);
  const int voff = tid * 16;

  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
    int lmax = __float_as_int(-1.0f);

    f32x4 sx[2], sy[2], sz[2];
    if (G_STR > 0) {  // first streamed group goes out before any arithmetic
      const int soff = G_REG + G_LDS;
      sx[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (0 * NPAD + soff * 4 * BLOCK) * 4, 0));
      sy[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (1 * NPAD + soff * 4 * BLOCK) * 4, 0));
      sz[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (2 * NPAD + soff * 4 * BLOCK) * 4, 0));
    }
#pragma unroll
    for (int g = 0; g < G_REG; ++g) lmax = update4(t[g], sqdist4(rx[g], ry[g], rz[g], x1, y1, z1), lmax);
#pragma unroll
    for (int l = 0; l < G_LDS; ++l) {
      const f32x4 x = s_pts[(l * 3 + 0) * BLOCK + tid];
      const f32x4 y = s_pts[(l * 3 + 1) * BLOCK + tid];
      const f32x4 z = s_pts[(l * 3 + 2) * BLOCK + tid];
      lmax = update4(t[G_REG + l], sqdist4(x, y, z, x1, y1, z1), lmax);
    }
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

    const int wmax = wave_max_i32(lmax);
    if (lane == 0) s_wmax[wid] = wmax;
    __syncthreads();
    int M = s_wmax[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = max(M, s_wmax[w]);
    if (wmax == M && M >= 0) {
      unsigned key = 0xFFFFFFFFu;
      int lgv = lg;
      asm volatile("" : "+s"(lgv));  // keep the key arithmetic inside this rare branch (no hoist, no spill)
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned ki = fps_key((g * BLOCK + tid) * 4 + u, lgv);
          key = (__float_as_int(t[g][u]) == M) ? min(key, ki) : key;
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

template <int BLOCK, int TPL>
void launch_fps(const float *xyz, int B, int N, int m, int lg, int32_t *idx, hipStream_t s) {
  hipLaunchKernelGGL((fps_kernel<BLOCK, TPL>), dim3(B), dim3(BLOCK), 0, s, xyz, N, m, lg, idx);
}

template <int G_REG, int G_LDS, int G_STR>
void launch_fps_hybrid(const float *xyz, float *ws, int B, int N, int m, int lg, int32_t *idx, hipStream_t s) {
  hipLaunchKernelGGL((fps_hybrid_kernel<G_REG, G_LDS, G_STR>), dim3(B), dim3(1024), 0, s, xyz, ws, N, m, lg, idx);
}

}  // namespace

// Workspace: the structure-of-arrays copy used by the hybrid kernel (3 planes of 40 960 floats per
// scene) or the reference-style temp array (N floats per scene) of the generic kernel.
extern "C" size_t spacap_fps_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  const size_t per_scene = (size_t)(N > 3 * 40960 ? N : 3 * 40960) * sizeof(float);
  return (size_t)B * per_scene;
}

extern "C" int spacap_fps_f32(const float *xyz, int B, int N, int m, void *workspace, int32_t *idx,
                              spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 0 && N >= 1 && m >= 0, "spacap_fps_f32: bad sizes B=%d N=%d m=%d", B, N, m);
  if (B == 0 || m == 0) return SPACAP_OK;
  SPACAP_REQUIRE(xyz && idx, "spacap_fps_f32: null pointer");
  SPACAP_REQUIRE(N < (1 << 28), "spacap_fps_f32: N=%d too large", N);
  hipStream_t s = spacap::as_stream(stream);
  const int bs = spacap_opt_n_threads(N);
  int lg = 0;
  while ((1 << lg) < bs) ++lg;

#define FPS_CASE(BLOCK, TPL)                                   \
  if (N <= (BLOCK) * (TPL)) {                                  \
    launch_fps<BLOCK, TPL>(xyz, B, N, m, lg, idx, s);          \
    SPACAP_CHECK_LAUNCH("spacap_fps_f32");                     \
    return SPACAP_OK;                                          \
  }
  FPS_CASE(64, 1)
  FPS_CASE(64, 2)
  FPS_CASE(64, 4)
  FPS_CASE(64, 8)
  FPS_CASE(256, 4)
  FPS_CASE(256, 8)
  FPS_CASE(1024, 4)
  FPS_CASE(1024, 8)
  FPS_CASE(1024, 16)
  FPS_CASE(1024, 20)
#undef FPS_CASE
  SPACAP_REQUIRE(workspace, "spacap_fps_f32: workspace required for N=%d", N);
  float *ws = reinterpret_cast<float *>(workspace);
#define FPS_HYB(GR, GL, GS)                                                \
  if (N <= 4096 * ((GR) + (GL) + (GS))) {                                  \
    launch_fps_hybrid<GR, GL, GS>(xyz, ws, B, N, m, lg, idx, s);           \
    SPACAP_CHECK_LAUNCH("spacap_fps_f32(hybrid)");                         \
    return SPACAP_OK;                                                      \
  }
  FPS_HYB(3, 3, 0)
  FPS_HYB(3, 3, 2)
  FPS_HYB(3, 3, 4)
#undef FPS_HYB
  hipLaunchKernelGGL((fps_generic_kernel<1024>), dim3(B), dim3(1024), 0, s, xyz, ws, N, m, lg, idx);
  SPACAP_CHECK_LAUNCH("spacap_fps_f32(generic)");
  return SPACAP_OK;
}
