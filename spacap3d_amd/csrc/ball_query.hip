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


// =============================================================================================================
// Cell-grid ball query: the same output as above (bit-exact), without testing every point against every centre.
//
// SA1 tests 8 x 2 048 centres against 40 000 points each (655 M distance tests, 0.35 ms of VALU work) although a
// 0.2 m ball holds ~100 of them.  Here the points of a scene are binned into cells at least `radius` wide
// (counting sort: count, scan, scatter -- the order inside a cell is arbitrary), a wavefront owns one centre,
// tests only the points of the 3 x 3 x 3 neighbouring cells (9 contiguous runs of the sorted array, the x index
// being the fastest) with the reference's own distance expression, marks the hits in an N-bit map in LDS, and
// reads the first `nsample` set bits back in ascending order -- which is exactly "the first nsample points in index
// order inside the ball" (ball_query_gpu.cu:24-41), whatever order the cells were filled in.
//
// Why a point inside the ball cannot be missed: cells are h >= 1.001 * radius wide per axis and the cell index
// floor((p - origin) * (1 / h)) is monotonic in p, so |p - q| < radius moves the (real-valued) index by less than
// 0.9991 plus ~1e-5 of rounding: the integer cells differ by at most one.  Clamping the largest index to the
// last cell keeps that property.
namespace {
constexpr int BQ_DIM_CAP = 64, BQ_DIMZ_CAP = 16;                  // <= 65 536 cells per scene
constexpr int BQ_NCELL_MAX = BQ_DIM_CAP * BQ_DIM_CAP * BQ_DIMZ_CAP;
struct BqGrid {   // per scene, 16 floats
  float ox, oy, oz, ihx, ihy, ihz;
  int nx, ny, nz, ncell, pad[6];
};

__device__ __forceinline__ int bq_cell1(float p, float o, float ih) { return (int)floorf((p - o) * ih); }

// one workgroup per scene: bounding box -> grid parameters; clears the scene's cell counters
__global__ __launch_bounds__(1024) void bq_bounds_kernel(const float *__restrict__ xyz_all, int N, float radius,
                                                         BqGrid *__restrict__ grids, int *__restrict__ cnt_all) {
  __shared__ float s_lo[3][16], s_hi[3][16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float *xyz = xyz_all + (size_t)b * N * 3;
  float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int i = tid; i < N; i += 1024)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = xyz[(size_t)i * 3 + a];
      lo[a] = fminf(lo[a], v), hi[a] = fmaxf(hi[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)), hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
    if (lane == 0) s_lo[a][w] = lo[a], s_hi[a][w] = hi[a];
  }
  __syncthreads();
  __shared__ int s_ncell;
  if (tid == 0) {
    BqGrid g;
    float o[3], ih[3];
    int n[3];
    const int cap[3] = {BQ_DIM_CAP, BQ_DIM_CAP, BQ_DIMZ_CAP};
    for (int a = 0; a < 3; ++a) {
      float l = s_lo[a][0], h = s_hi[a][0];
      for (int k = 1; k < 16; ++k) l = fminf(l, s_lo[a][k]), h = fmaxf(h, s_hi[a][k]);
      const float ext = fmaxf(h - l, 0.f);
      const float cell = fmaxf(radius * 1.001f, ext / (float)(cap[a] - 1));
      o[a] = l, ih[a] = 1.0f / cell;
      n[a] = min(cap[a], max(1, (int)floorf(ext * ih[a]) + 1));
    }
    g.ox = o[0], g.oy = o[1], g.oz = o[2], g.ihx = ih[0], g.ihy = ih[1], g.ihz = ih[2];
    g.nx = n[0], g.ny = n[1], g.nz = n[2], g.ncell = n[0] * n[1] * n[2];
    for (int k = 0; k < 6; ++k) g.pad[k] = 0;
    grids[b] = g;
    s_ncell = g.ncell;
  }
  __syncthreads();
  int *cnt = cnt_all + (size_t)b * (BQ_NCELL_MAX + 1);
  for (int i = tid; i <= s_ncell; i += 1024) cnt[i] = 0;
}

__device__ __forceinline__ int bq_cell_of(const BqGrid &g, float x, float y, float z) {
  const int cx = min(g.nx - 1, max(0, bq_cell1(x, g.ox, g.ihx)));
  const int cy = min(g.ny - 1, max(0, bq_cell1(y, g.oy, g.ihy)));
  const int cz = min(g.nz - 1, max(0, bq_cell1(z, g.oz, g.ihz)));
  return (cz * g.ny + cy) * g.nx + cx;
}

__global__ __launch_bounds__(256) void bq_count_kernel(const float *__restrict__ xyz_all, int N, const BqGrid *__restrict__ grids,
                                                       int *__restrict__ cnt_all, int *__restrict__ cell_all) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const BqGrid g = grids[b];
  const float *p = xyz_all + ((size_t)b * N + i) * 3;
  const int c = bq_cell_of(g, p[0], p[1], p[2]);
  cell_all[(size_t)b * N + i] = c;
  atomicAdd(&cnt_all[(size_t)b * (BQ_NCELL_MAX + 1) + c], 1);
}

// one workgroup per scene: cnt[0..ncell) -> exclusive starts in start[0..ncell], cursor = copy of the starts
__global__ __launch_bounds__(1024) void bq_scan_kernel(const BqGrid *__restrict__ grids, const int *__restrict__ cnt_all,
                                                       int *__restrict__ start_all, int *__restrict__ cursor_all) {
  __shared__ int s_part[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int ncell = grids[b].ncell;
  const int *cnt = cnt_all + (size_t)b * (BQ_NCELL_MAX + 1);
  int *start = start_all + (size_t)b * (BQ_NCELL_MAX + 1), *cursor = cursor_all + (size_t)b * (BQ_NCELL_MAX + 1);
  const int per = (ncell + 1023) / 1024, i0 = tid * per, i1 = min(ncell, i0 + per);
  int sum = 0;
  for (int i = i0; i < i1; ++i) sum += cnt[i];
  s_part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {   // inclusive scan of the 1 024 partial sums
    const int v = tid >= o ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - sum;
  for (int i = i0; i < i1; ++i) {
    start[i] = run, cursor[i] = run;
    run += cnt[i];
  }
  if (tid == 1023) start[ncell] = s_part[1023];
}

__global__ __launch_bounds__(256) void bq_scatter_kernel(const float *__restrict__ xyz_all, int N, const int *__restrict__ cell_all,
                                                         int *__restrict__ cursor_all, float4 *__restrict__ sorted_all) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float *p = xyz_all + ((size_t)b * N + i) * 3;
  const int c = cell_all[(size_t)b * N + i];
  const int pos = atomicAdd(&cursor_all[(size_t)b * (BQ_NCELL_MAX + 1) + c], 1);
  sorted_all[(size_t)b * N + pos] = make_float4(p[0], p[1], p[2], __int_as_float(i));
}

// one wavefront per centre; dynamic LDS: 4 waves x WPL*64 words of hit bitmap
__global__ __launch_bounds__(256) void bq_query_kernel(const float *__restrict__ new_xyz_all, const float4 *__restrict__ sorted_all,
                                                       const BqGrid *__restrict__ grids, const int *__restrict__ start_all, int N,
                                                       int m, float radius2, int nsample, int WPL, int32_t *__restrict__ idx_all) {
  extern __shared__ unsigned s_bits_all[];
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + w;
  if (j >= m) return;
  unsigned *bits = s_bits_all + (size_t)w * WPL * 64;
  const BqGrid g = grids[b];
  const float4 *sorted = sorted_all + (size_t)b * N;
  const int *start = start_all + (size_t)b * (BQ_NCELL_MAX + 1);
  const float *q = new_xyz_all + ((size_t)b * m + j) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  for (int i = lane; i < WPL * 64; i += 64) bits[i] = 0u;
  // lanes 0..8: the candidate run of one (dy, dz) row of cells
  const int cx = bq_cell1(qx, g.ox, g.ihx), cy = bq_cell1(qy, g.oy, g.ihy), cz = bq_cell1(qz, g.oz, g.ihz);
  int rs = 0, re = 0;
  if (lane < 9) {
    const int yy = cy + lane % 3 - 1, zz = cz + lane / 3 - 1;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz && x0 <= x1) {
      const int row = (zz * g.ny + yy) * g.nx;
      rs = start[row + x0], re = start[row + x1 + 1];
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll 1
  for (int r = 0; r < 9; ++r) {
    const int s0 = __shfl(rs, r), e0 = __shfl(re, r);
    for (int t = s0 + lane; t < e0; t += 64) {
      const float4 p = sorted[t];
      const float d2 = (qx - p.x) * (qx - p.x) + (qy - p.y) * (qy - p.y) + (qz - p.z) * (qz - p.z);
      if (d2 < radius2) {
        const int k = __float_as_int(p.w);
        atomicOr(&bits[k >> 5], 1u << (k & 31));
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  // ordered read-back: lane L owns words [L*WPL, (L+1)*WPL)
  int count = 0;
  for (int i = 0; i < WPL; ++i) count += __builtin_popcount(bits[lane * WPL + i]);
  int incl = count;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  const int total = __shfl(incl, 63);
  int pos = incl - count;
  int32_t *out = idx_all + ((size_t)b * m + j) * nsample;
  int first = 0x7fffffff;
  if (count > 0) {
    for (int i = 0; i < WPL; ++i) {
      unsigned v = bits[lane * WPL + i];
      while (v) {
        const int k = (lane * WPL + i) * 32 + __builtin_ctz(v);
        v &= v - 1;
        if (pos == 0) first = k;
        if (pos < nsample) out[pos] = k;
        ++pos;
      }
      if (pos >= nsample) break;
    }
  }
  if (total < nsample) {   // pad with the first hit, zeros when the ball is empty
    int f = first;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) f = min(f, __shfl_xor(f, o));
    const int fill = total > 0 ? f : 0;
    for (int l = total + lane; l < nsample; l += 64) out[l] = fill;
  }
}
}  // namespace

extern "C" size_t spacap_ball_query_grid_workspace_bytes(int B, int N) {
  if (B < 1 || N < 1) return 0;
  const size_t per = 64 + 3 * (size_t)(BQ_NCELL_MAX + 1) * 4 + (size_t)N * 4 + 256 + (size_t)N * 16 + 256;
  return (size_t)B * per + 1024;
}

extern "C" int spacap_ball_query_grid_f32(const float *new_xyz, const float *xyz, int B, int N, int m, float radius, int nsample,
                                          int32_t *idx, void *workspace, size_t workspace_bytes, spacap_stream_t stream) {
  const char *what = "spacap_ball_query_grid_f32";
  SPACAP_REQUIRE(B >= 1 && N >= 1 && m >= 1 && nsample >= 1 && N <= (1 << 20) && radius > 0.f, "%s: bad sizes", what);
  SPACAP_REQUIRE(new_xyz && xyz && idx && workspace, "%s: null pointer", what);
  SPACAP_REQUIRE(workspace_bytes >= spacap_ball_query_grid_workspace_bytes(B, N), "%s: workspace too small", what);
  hipStream_t s = spacap::as_stream(stream);
  auto align = [](char *p) { return (char *)(((uintptr_t)p + 255) & ~(uintptr_t)255); };
  char *p = align((char *)workspace);
  BqGrid *grids = (BqGrid *)p;                 p = align(p + (size_t)B * sizeof(BqGrid));
  int *cnt = (int *)p;                         p = align(p + (size_t)B * (BQ_NCELL_MAX + 1) * 4);
  int *start = (int *)p;                       p = align(p + (size_t)B * (BQ_NCELL_MAX + 1) * 4);
  int *cursor = (int *)p;                      p = align(p + (size_t)B * (BQ_NCELL_MAX + 1) * 4);
  int *cell = (int *)p;                        p = align(p + (size_t)B * N * 4);
  float4 *sorted = (float4 *)p;
  const float radius2 = radius * radius;   // fp32 product, ball_query_gpu.cu:22
  const dim3 pgrid((N + 255) / 256, B);
  hipLaunchKernelGGL(bq_bounds_kernel, dim3(B), dim3(1024), 0, s, xyz, N, radius, grids, cnt);
  hipLaunchKernelGGL(bq_count_kernel, pgrid, dim3(256), 0, s, xyz, N, grids, cnt, cell);
  hipLaunchKernelGGL(bq_scan_kernel, dim3(B), dim3(1024), 0, s, grids, cnt, start, cursor);
  hipLaunchKernelGGL(bq_scatter_kernel, pgrid, dim3(256), 0, s, xyz, N, cell, cursor, sorted);
  int WPL = ((N + 31) / 32 + 63) / 64;
  WPL |= 1;   // odd stride: the per-lane word runs fall on different LDS banks
  const size_t lds = (size_t)4 * WPL * 64 * sizeof(unsigned);
  SPACAP_REQUIRE(lds <= 64 * 1024, "%s: N=%d too large for the hit bitmap", what, N);
  hipLaunchKernelGGL(bq_query_kernel, dim3((m + 3) / 4, B), dim3(256), lds, s, new_xyz, sorted, grids, start, N, m, radius2,
                     nsample, WPL, idx);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
