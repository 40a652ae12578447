// Ball query for gfx950 (MI355X).
//
// Replaces lib/pointnet2/_ext_src/src/ball_query_gpu.cu:9-44 (host: src/ball_query.cpp:8-32).
// Output contract (bit-exact): for every centre the FIRST `nsample` points in ascending index with
// d2 < radius^2 (fp32, un-contracted, terms in the order of ball_query_gpu.cu:31-32), padded with the
// first hit; all-zero rows when no point is inside.
//
// Design: the reference gives each thread a centre and lets it walk all N points alone (one block per
// scene, uncoalesced broadcast reads).  Here a wavefront owns CPW centres whose coordinates sit in
// SGPRs; its 64 lanes test 64 consecutive points per step (one coalesced read of the tile serves all
// CPW centres), hits are compacted in index order with a ballot + prefix popcount, and the scan
// stops as soon as every centre of the wave is full.  Grid = B * m / (4 * CPW) workgroups, so SA1
// (B=8, m=2048) fills the chip instead of eight CUs.
#include "common.hpp"

#pragma clang fp contract(off)

namespace {

template <int CPW>
__global__ __launch_bounds__(256) void ball_query_kernel(const float *__restrict__ new_xyz_all,
                                                         const float *__restrict__ xyz_all, int N, int m,
                                                         float radius2, int nsample,
                                                         int32_t *__restrict__ idx_all) {
  const int b = blockIdx.y;
  const float *__restrict__ xyz = xyz_all + (size_t)b * N * 3;
  const float *__restrict__ ctr = new_xyz_all + (size_t)b * m * 3;
  int32_t *__restrict__ idx = idx_all + (size_t)b * m * nsample;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int c0 = wave * CPW;
  if (c0 >= m) return;

  float cx[CPW], cy[CPW], cz[CPW];
  int cnt[CPW], first[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int j = min(c0 + c, m - 1);
    cx[c] = ctr[j * 3 + 0];
    cy[c] = ctr[j * 3 + 1];
    cz[c] = ctr[j * 3 + 2];
    cnt[c] = (c0 + c < m) ? 0 : nsample;  // centres past the end count as full
    first[c] = 0;
  }

  for (int k0 = 0; k0 < N; k0 += 64) {
    const int k = k0 + lane;
    const bool valid = k < N;
    const int kk = valid ? k : N - 1;
    const float x = xyz[kk * 3 + 0], y = xyz[kk * 3 + 1], z = xyz[kk * 3 + 2];
    bool all_full = true;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      if (cnt[c] < nsample) {  // wave-uniform
        const float d2 = (cx[c] - x) * (cx[c] - x) + (cy[c] - y) * (cy[c] - y) + (cz[c] - z) * (cz[c] - z);
        const bool hit = valid && (d2 < radius2);
        const unsigned long long mask = __ballot(hit);
        if (mask) {
          const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
          const int pos = cnt[c] + below;
          if (hit && pos < nsample) idx[(size_t)(c0 + c) * nsample + pos] = k;
          if (cnt[c] == 0) first[c] = k0 + __builtin_ctzll(mask);
          cnt[c] += __builtin_popcountll(mask);
        }
        all_full = all_full && (cnt[c] >= nsample);
      }
    }
    if (all_full) break;
  }

  // pad with the first hit (or zeros when the ball is empty): ball_query_gpu.cu:34-38, ball_query.cpp:19-21
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    if (c0 + c < m && cnt[c] < nsample) {
      const int fill = cnt[c] > 0 ? first[c] : 0;
      for (int l = cnt[c] + lane; l < nsample; l += 64) idx[(size_t)(c0 + c) * nsample + l] = fill;
    }
  }
}

}  // namespace

extern "C" int spacap_ball_query_f32(const float *new_xyz, const float *xyz, int B, int N, int m,
                                     float radius, int nsample, int32_t *idx, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 0 && N >= 0 && m >= 0 && nsample >= 0, "spacap_ball_query_f32: bad sizes");
  if (B == 0 || m == 0 || nsample == 0) return SPACAP_OK;
  SPACAP_REQUIRE(new_xyz && idx, "spacap_ball_query_f32: null pointer");
  hipStream_t s = spacap::as_stream(stream);
  if (N == 0) {  // nothing to scan: the reference leaves its zero-initialised output untouched
    SPACAP_CHECK_HIP(hipMemsetAsync(idx, 0, sizeof(int32_t) * (size_t)B * m * nsample, s),
                     "spacap_ball_query_f32(memset)");
    return SPACAP_OK;
  }
  SPACAP_REQUIRE(xyz, "spacap_ball_query_f32: null xyz");
  const float radius2 = radius * radius;  // fp32 product, ball_query_gpu.cu:22
  if ((long)B * m >= 8192) {
    constexpr int CPW = 4;
    dim3 grid((m + 4 * CPW - 1) / (4 * CPW), B);
    hipLaunchKernelGGL((ball_query_kernel<CPW>), grid, dim3(256), 0, s, new_xyz, xyz, N, m, radius2, nsample, idx);
  } else {
    constexpr int CPW = 1;
    dim3 grid((m + 4 * CPW - 1) / (4 * CPW), B);
    hipLaunchKernelGGL((ball_query_kernel<CPW>), grid, dim3(256), 0, s, new_xyz, xyz, N, m, radius2, nsample, idx);
  }
  SPACAP_CHECK_LAUNCH("spacap_ball_query_f32");
  return SPACAP_OK;
}
