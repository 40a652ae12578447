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
// keeps the running minimum distance of its points in VGPRs for the whole kernel, the arg-max is a DPP
// wave reduction on the value only, and the index is recovered by a second, rare pass that only the
// wave(s) holding the maximum execute.  Two barriers per round.  Skipped / padded slots carry
// temp = -1: fminf(d, -1) = -1 keeps them out of every arg-max exactly as the reference's `continue`.
//   N <= 8 192   : coordinates in VGPRs too (fps_kernel)
//   N <= 81 920  : Morton-bucketed scene, whole buckets skipped when provably unaffected (fps_bucket.inc)
//   larger       : reference-style streaming with temp in the workspace (fps_generic_kernel)
#include "common.hpp"

#pragma clang fp contract(off)

namespace {

using namespace spacap;
struct __attribute__((aligned(16))) f32x4s { float x, y, z, w; };

// v_min_f32 without the canonicalising v_max hipcc emits in front of fminf (operands are never sNaN here);
// in IEEE mode it returns the non-NaN operand, i.e. the reference's fminf (sampling_gpu.cu:106).
__device__ __forceinline__ float vmin_f32(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

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
      t[i] = vmin_f32(d, t[i]);
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

// 512 < N <= 1 024 (the vote-aggregation sampling, 1 024 votes -> 256 proposals, sits on the step's critical path):
// ONE wavefront per scene, 16 points per lane in registers, all coordinates also in LDS so that the new centre is an
// LDS broadcast read instead of a dependent global load, the tie-break keys precomputed per slot, no barriers and
// no atomics: 0.92 -> ~0.4 us per round.  Same selection rule as fps_kernel (first maximum in the reference's tree
// order through fps_key).
template <int TPL>
__global__ __launch_bounds__(64) void fps_wave_kernel(const float *__restrict__ xyz_all, int N, int m, int lg,
                                                      int32_t *__restrict__ idx_all) {
  __shared__ float s_xyz[64 * TPL * 3];
  const float *__restrict__ xyz = xyz_all + (size_t)blockIdx.x * N * 3;
  int32_t *__restrict__ idxs = idx_all + (size_t)blockIdx.x * m;
  const int lane = threadIdx.x;
  for (int i = lane; i < N * 3; i += 64) s_xyz[i] = xyz[i];
  float t[TPL], px[TPL], py[TPL], pz[TPL];
  unsigned keys[TPL];
#pragma unroll
  for (int i = 0; i < TPL; ++i) {
    const int k = lane + i * 64;
    const int kk = k < N ? k : N - 1;
    const float x = xyz[kk * 3 + 0], y = xyz[kk * 3 + 1], z = xyz[kk * 3 + 2];
    const float mag = (x * x) + (y * y) + (z * z);
    const bool skip = (k >= N) || ((double)mag <= 1e-3);
    t[i] = skip ? -1.0f : 1e10f;
    px[i] = x; py[i] = y; pz[i] = z;
    keys[i] = fps_key(k, lg);
  }
  if (lane == 0) idxs[0] = 0;
  __syncthreads();
  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = s_xyz[old * 3 + 0], y1 = s_xyz[old * 3 + 1], z1 = s_xyz[old * 3 + 2];
    int lmax = __float_as_int(-1.0f);
#pragma unroll
    for (int i = 0; i < TPL; ++i) {
      const float d = sqdist(px[i], py[i], pz[i], x1, y1, z1);
      t[i] = vmin_f32(d, t[i]);
      lmax = max(lmax, __float_as_int(t[i]));
    }
    const int M = wave_max_i32(lmax);
    unsigned key = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < TPL; ++i) key = (__float_as_int(t[i]) == M) ? min(key, keys[i]) : key;
    key = wave_min_u32(key);
    old = (M < 0) ? 0 : fps_unkey(key, lg);  // every point skipped: the reference returns index 0
    old = __builtin_amdgcn_readfirstlane(old);
    if (lane == 0) idxs[j] = old;
  }
}

// N <= 8 192, the shape of every level below the first (2 048 -> 1 024 -> 512 -> 256) and of the vote-aggregation sampling
// (1 024 votes -> 256 proposals, on the step's critical path): ONE barrier per round and no dependent global load.
//   * all coordinates also live in LDS as float4: the new centre is one broadcast ds_read_b128 (fps_kernel: three dependent
//     global loads, ~0.2 us of every round);
//   * every wave ALWAYS computes its (maximum, winning key) pair -- 3 more VALU per point, but no second, conditional pass,
//     no LDS atomic and no second barrier; the pairs meet in an LDS record double-buffered by round parity, and every wave
//     reduces the NW records itself (two DPP reductions).
// Same selection rule as fps_kernel: first maximum in the reference's tree order through fps_key; M < 0 (every point
// skipped) gives index 0.  Rounds: 0.75 -> 0.57 us at N = 2 048, 0.62 -> 0.50 us at N = 1 024, 0.39 at 512, 0.32 at 256: a
// single wave costs 0.28 us + 17 ns per point slot and round (tools/lab/fps_small_bench.py).
template <int BLOCK, int TPL>
__global__ __launch_bounds__(BLOCK) void fps_small_kernel(const float *__restrict__ xyz_all, int N, int m, int lg,
                                                          int32_t *__restrict__ idx_all) {
  constexpr int NW = BLOCK / 64;
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // [N] float4 coordinates
  __shared__ __attribute__((aligned(8))) int s_rec[2][NW > 1 ? NW : 1][2];   // per round parity, per wave: {max temp bits, key}
  f32x4s *s_xyz = reinterpret_cast<f32x4s *>(s_dyn);
  const float *__restrict__ xyz = xyz_all + (size_t)blockIdx.x * N * 3;
  int32_t *__restrict__ idxs = idx_all + (size_t)blockIdx.x * m;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float t[TPL], px[TPL], py[TPL], pz[TPL];
  unsigned keys[TPL];
#pragma unroll
  for (int i = 0; i < TPL; ++i) {
    const int k = tid + i * BLOCK;
    const int kk = k < N ? k : N - 1;
    const float x = xyz[kk * 3 + 0], y = xyz[kk * 3 + 1], z = xyz[kk * 3 + 2];
    const float mag = (x * x) + (y * y) + (z * z);
    const bool skip = (k >= N) || ((double)mag <= 1e-3);
    t[i] = skip ? -1.0f : 1e10f;
    px[i] = x; py[i] = y; pz[i] = z;
    keys[i] = fps_key(k, lg);
    if (k < N) s_xyz[k] = f32x4s{x, y, z, 0.f};
  }
  if (tid == 0) idxs[0] = 0;
  __syncthreads();
  int old = 0;
  for (int j = 1; j < m; ++j) {
    const f32x4s c = s_xyz[old];   // wave-uniform address: one broadcast read
    int lmax = __float_as_int(-1.0f);
#pragma unroll
    for (int i = 0; i < TPL; ++i) {
      const float d = sqdist(px[i], py[i], pz[i], c.x, c.y, c.z);
      t[i] = vmin_f32(d, t[i]);
      lmax = max(lmax, __float_as_int(t[i]));   // values are -1 or >= +0, never NaN: signed-integer order == float order
    }
    const int wmax = wave_max_i32_fast(lmax);
    unsigned key = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < TPL; ++i) key = min(key, (__float_as_int(t[i]) == wmax) ? keys[i] : 0xFFFFFFFFu);
    key = wave_min_u32_fast(key);
    int bw = wmax;
    unsigned bk = key;
    if (NW > 1) {
      if (lane == 0) *reinterpret_cast<int2 *>(&s_rec[j & 1][wid][0]) = make_int2(wmax, (int)key);
      __syncthreads();
      if (NW <= 8) {   // every lane reads all NW records (broadcast reads, issued together) and reduces them itself
        int rw[NW];
        unsigned rk[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const int2 r = *reinterpret_cast<const int2 *>(&s_rec[j & 1][w][0]);
          rw[w] = r.x; rk[w] = (unsigned)r.y;
        }
        bw = rw[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) bw = max(bw, rw[w]);
        bk = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < NW; ++w) bk = min(bk, rw[w] == bw ? rk[w] : 0xFFFFFFFFu);
      } else {
        int rw = (int)0x80000000;
        unsigned rk = 0xFFFFFFFFu;
        if (lane < NW) { rw = s_rec[j & 1][lane][0]; rk = (unsigned)s_rec[j & 1][lane][1]; }
        bw = wave_max_i32_fast(rw);
        bk = wave_min_u32_fast(rw == bw ? rk : 0xFFFFFFFFu);
      }
    }
    old = (bw < 0) ? 0 : fps_unkey(bk, lg);   // every point skipped: the reference returns index 0
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
      const float d2 = vmin_f32(d, temp[k]);
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

using f32x4 = float __attribute__((ext_vector_type(4)));

// Large scenes (8 192 < N <= 81 920): spatially bucketed kernel with exact pruning.
#include "fps_bucket.inc"

template <int BLOCK, int TPL>
void launch_fps(const float *xyz, int B, int N, int m, int lg, int32_t *idx, hipStream_t s) {
  hipLaunchKernelGGL((fps_kernel<BLOCK, TPL>), dim3(B), dim3(BLOCK), 0, s, xyz, N, m, lg, idx);
}

template <int BLOCK, int TPL>
hipError_t launch_fps_small(const float *xyz, int B, int N, int m, int lg, int32_t *idx, hipStream_t s) {
  const int lds = N * 16;
  static unsigned long long lds_ok = 0;
  if (lds > 48 * 1024) {
    const hipError_t e = spacap::allow_dynamic_lds(reinterpret_cast<const void *>(&fps_small_kernel<BLOCK, TPL>), 128 * 1024, lds_ok);
    if (e != hipSuccess) return e;
  }
  // (giving these workgroups a CU of their own by asking for the CU's whole LDS: slower for the side chain's levels beside the
  // step, 7.29 vs 7.22 ms; no effect for the vote-aggregation sampling on the step's own stream, nor for the bucketed kernel)
  hipLaunchKernelGGL((fps_small_kernel<BLOCK, TPL>), dim3(B), dim3(BLOCK), lds, s, xyz, N, m, lg, idx);
  return hipSuccess;
}

}  // namespace

// Workspace: the bucket-ordered structure-of-arrays copy of the bucketed kernel (3 planes of ceil(N/4096)*4096
// floats per scene) or the reference-style temp array (N floats per scene) of the generic kernel.
extern "C" size_t spacap_fps_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  const size_t planes = N <= 81920 ? fps_bucket_workspace_floats(N) : 0;
  const size_t per_scene = (planes > (size_t)N ? planes : (size_t)N) * sizeof(float);
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
  // SPACAP_FPS_LEGACY=1 (tests / lab): the round-5 kernels (fps_kernel, fps_wave_kernel) instead of fps_small_kernel
  static const bool legacy = getenv("SPACAP_FPS_LEGACY") != nullptr && atoi(getenv("SPACAP_FPS_LEGACY")) != 0;
  if (!legacy && N <= 2048) {
#define FPS_SMALL(BLOCK, TPL)                                                                          \
  if (N <= (BLOCK) * (TPL)) {                                                                          \
    SPACAP_CHECK_HIP((launch_fps_small<BLOCK, TPL>(xyz, B, N, m, lg, idx, s)), "spacap_fps_f32(small)"); \
    SPACAP_CHECK_LAUNCH("spacap_fps_f32(small)");                                                      \
    return SPACAP_OK;                                                                                  \
  }
    // (measured per shape, tools/lab/fps_small_bench.py; beyond 2 048 points the two-barrier kernel with its rare second pass
    // is the faster one: 16 waves pay 3 VALU per point for keys that one wave needs)
    FPS_SMALL(64, 1)
    FPS_SMALL(64, 2)
    FPS_SMALL(64, 4)
    FPS_SMALL(64, 8)
    FPS_SMALL(256, 4)
    FPS_SMALL(512, 4)
#undef FPS_SMALL
  }
  FPS_CASE(64, 1)
  FPS_CASE(64, 2)
  FPS_CASE(64, 4)
  FPS_CASE(64, 8)
  if (N <= 1024) {
    hipLaunchKernelGGL((fps_wave_kernel<16>), dim3(B), dim3(64), 0, s, xyz, N, m, lg, idx);
    SPACAP_CHECK_LAUNCH("spacap_fps_f32(wave)");
    return SPACAP_OK;
  }
  FPS_CASE(256, 4)
  FPS_CASE(256, 8)
  FPS_CASE(1024, 4)
  FPS_CASE(1024, 8)
#undef FPS_CASE
  SPACAP_REQUIRE(workspace, "spacap_fps_f32: workspace required for N=%d", N);
  float *ws = reinterpret_cast<float *>(workspace);
  if (N <= 81920) {  // bucketed kernel: index map as u16 in LDS up to 65 535 points, in the workspace beyond
    launch_fps_bucket(xyz, ws, B, N, m, lg, idx, s);
    SPACAP_CHECK_LAUNCH("spacap_fps_f32(bucket)");
    return SPACAP_OK;
  }
  hipLaunchKernelGGL((fps_generic_kernel<1024>), dim3(B), dim3(1024), 0, s, xyz, ws, N, m, lg, idx);
  SPACAP_CHECK_LAUNCH("spacap_fps_f32(generic)");
  return SPACAP_OK;
}

// LAB (tools/lab/fps_small_bench.py; not declared in the public header): fps_small_kernel at a chosen workgroup shape
extern "C" int spacap_lab_fps_small(const float *xyz, int B, int N, int m, int block, int tpl, int32_t *idx, spacap_stream_t stream) {
  SPACAP_REQUIRE(xyz && idx && B >= 1 && N >= 1 && m >= 1 && N <= 8192 && N <= block * tpl, "spacap_lab_fps_small: bad arguments");
  hipStream_t s = spacap::as_stream(stream);
  const int bs = spacap_opt_n_threads(N);
  int lg = 0;
  while ((1 << lg) < bs) ++lg;
#define V(BL, TP)                                                                                             \
  if (block == BL && tpl == TP) {                                                                             \
    SPACAP_CHECK_HIP((launch_fps_small<BL, TP>(xyz, B, N, m, lg, idx, s)), "spacap_lab_fps_small");           \
    SPACAP_CHECK_LAUNCH("spacap_lab_fps_small");                                                              \
    return SPACAP_OK;                                                                                         \
  }
  V(64, 1) V(64, 2) V(64, 4) V(64, 8) V(64, 16) V(128, 2) V(128, 4) V(128, 8) V(256, 1) V(256, 2) V(256, 4) V(256, 8) V(512, 1) V(512, 2)
  V(512, 4) V(1024, 1) V(1024, 2) V(1024, 4) V(1024, 8)
#undef V
  SPACAP_REQUIRE(false, "spacap_lab_fps_small: (%d, %d) not instantiated", block, tpl);
}
