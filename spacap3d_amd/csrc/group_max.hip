// Max over the sample dimension of a grouped tensor, and its gradient, for gfx950 (MI355X).
//
// Replaces `F.max_pool2d(new_features, kernel_size=[1, nsample])` of the set-abstraction module
// (lib/pointnet2/pointnet2_modules.py:256-259) and its autograd backward.  Semantics of PyTorch's
// max_pool2d: the first maximum wins (strict `>` scan in sample order), a NaN wins over everything; the
// gradient goes to that one sample.
//
// PyTorch's NCHW pooling kernel walks each (1 x S) window with one thread (1.9 ms for the SA1 tensor,
// 0.28 TB/s).  Here S/4 lanes share a row, each loads one float4 (a wave reads 1 KiB contiguous per
// instruction), the row maximum is a shuffle reduction over those lanes, and four rows-groups are in
// flight per lane.  The backward writes the (rows x S) gradient with 16-byte stores.
#include <math.h>

#include "common.hpp"

namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
  // candidate (v, i) replaces (bv, bi): larger value, NaN beats non-NaN, ties / NaN-NaN keep the lower index
  const bool vn = v != v, bn = bv != bv;
  if (vn != bn) return vn;
  if (vn) return i < bi;
  return v > bv || (v == bv && i < bi);
}

template <int S>
__global__ __launch_bounds__(256) void group_max_kernel(const float *__restrict__ x, long rows,
                                                        float *__restrict__ out, uint8_t *__restrict__ arg) {
  constexpr int LPR = S / 4;        // lanes per row
  constexpr int RPW = 64 / LPR;     // rows per wave per step
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR, rin = lane / LPR;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  for (long r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
    const long row = r0 + rin;
    const bool ok = row < rows;
    const f32x4 v = ok ? *reinterpret_cast<const f32x4 *>(x + row * S + sub * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    float bv = v.x;
    int bi = sub * 4;
#pragma unroll
    for (int u = 1; u < 4; ++u)
      if (better(v[u], sub * 4 + u, bv, bi)) { bv = v[u]; bi = sub * 4 + u; }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (ok && sub == 0) {
      out[row] = bv;
      arg[row] = (uint8_t)bi;
    }
  }
}

__global__ __launch_bounds__(256) void group_max_generic_kernel(const float *__restrict__ x, long rows, int S,
                                                                float *__restrict__ out,
                                                                uint8_t *__restrict__ arg) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float *p = x + row * S;
  float bv = p[0];
  int bi = 0;
  for (int s = 1; s < S; ++s)
    if (better(p[s], s, bv, bi)) { bv = p[s]; bi = s; }
  out[row] = bv;
  arg[row] = (uint8_t)bi;
}

// grad_in[row, s] = (s == arg[row]) ? grad_out[row] : 0
template <int S>
__global__ __launch_bounds__(256) void group_max_grad_kernel(const float *__restrict__ go,
                                                             const uint8_t *__restrict__ arg, long rows,
                                                             float *__restrict__ gi) {
  constexpr int LPR = S / 4;
  const long total = rows * LPR;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long row = e / LPR;
    const int sub = (int)(e % LPR);
    const int a = arg[row] - sub * 4;
    const float g = go[row];
    f32x4 v = {a == 0 ? g : 0.f, a == 1 ? g : 0.f, a == 2 ? g : 0.f, a == 3 ? g : 0.f};
    *reinterpret_cast<f32x4 *>(gi + e * 4) = v;
  }
}

__global__ __launch_bounds__(256) void group_max_grad_generic_kernel(const float *__restrict__ go,
                                                                     const uint8_t *__restrict__ arg, long rows,
                                                                     int S, float *__restrict__ gi) {
  const long total = rows * S;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long row = e / S;
    gi[e] = ((int)(e % S) == arg[row]) ? go[row] : 0.f;
  }
}

}  // namespace

extern "C" int spacap_group_max_f32(const float *x, long rows, int S, float *out, uint8_t *arg,
                                    spacap_stream_t stream) {
  SPACAP_REQUIRE(rows >= 0 && S >= 1 && S <= 256, "spacap_group_max_f32: bad sizes rows=%ld S=%d", rows, S);
  if (rows == 0) return SPACAP_OK;
  SPACAP_REQUIRE(x && out && arg, "spacap_group_max_f32: null pointer");
  hipStream_t s = spacap::as_stream(stream);
  const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
#define GM_CASE(SV)                                                                                   \
  if (S == SV && aligned) {                                                                           \
    const long rpb = 4 * (64 / (SV / 4)) * 4; /* rows per block per step x 4 steps */                 \
    const unsigned grid = (unsigned)((rows + rpb - 1) / rpb < 1 ? 1 : ((rows + rpb - 1) / rpb > 65535 * 16 ? 65535 * 16 : (rows + rpb - 1) / rpb)); \
    hipLaunchKernelGGL((group_max_kernel<SV>), dim3(grid), dim3(256), 0, s, x, rows, out, arg);       \
    SPACAP_CHECK_LAUNCH("spacap_group_max_f32");                                                      \
    return SPACAP_OK;                                                                                 \
  }
  GM_CASE(16) GM_CASE(32) GM_CASE(64) GM_CASE(128)
#undef GM_CASE
  hipLaunchKernelGGL(group_max_generic_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, x, rows, S, out, arg);
  SPACAP_CHECK_LAUNCH("spacap_group_max_f32(generic)");
  return SPACAP_OK;
}

extern "C" int spacap_group_max_grad_f32(const float *grad_out, const uint8_t *arg, long rows, int S,
                                         float *grad_in, spacap_stream_t stream) {
  SPACAP_REQUIRE(rows >= 0 && S >= 1 && S <= 256, "spacap_group_max_grad_f32: bad sizes rows=%ld S=%d", rows, S);
  if (rows == 0) return SPACAP_OK;
  SPACAP_REQUIRE(grad_out && arg && grad_in, "spacap_group_max_grad_f32: null pointer");
  hipStream_t s = spacap::as_stream(stream);
  const bool aligned = (reinterpret_cast<uintptr_t>(grad_in) & 15) == 0;
  const unsigned grid = 256 * 16;
#define GMG_CASE(SV)                                                                                       \
  if (S == SV && aligned) {                                                                                \
    hipLaunchKernelGGL((group_max_grad_kernel<SV>), dim3(grid), dim3(256), 0, s, grad_out, arg, rows, grad_in); \
    SPACAP_CHECK_LAUNCH("spacap_group_max_grad_f32");                                                      \
    return SPACAP_OK;                                                                                      \
  }
  GMG_CASE(16) GMG_CASE(32) GMG_CASE(64) GMG_CASE(128)
#undef GMG_CASE
  hipLaunchKernelGGL(group_max_grad_generic_kernel, dim3(grid), dim3(256), 0, s, grad_out, arg, rows, S, grad_in);
  SPACAP_CHECK_LAUNCH("spacap_group_max_grad_f32(generic)");
  return SPACAP_OK;
}
