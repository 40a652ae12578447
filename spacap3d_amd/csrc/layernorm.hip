// The Transformer's LayerNorm, forward and backward, for gfx950 (MI355X).
//
// Replaces the reference's hand-written module (models/transformer_captioner.py:102-113)
//     mean = x.mean(-1); std = x.std(-1)            # UNBIASED std (n - 1)
//     y = a_2 * (x - mean) / (std + eps) + b_2      # eps added to std, not to the variance
// which PyTorch runs as ~8 elementwise / reduction kernels forward and ~20 backward, 26 times per training
// step (13 encoder + 13 decoder instances) on 1 MB tensors -- about 700 launches of pure latency.
// Here: one launch forward (a wavefront per row, shuffle reductions, row statistics kept for backward),
// two launches backward (dx + per-block partial sums of da / db, then a fixed-order reduction of the
// partials: no atomics, bitwise reproducible).
#include <math.h>

#include "common.hpp"

namespace {

constexpr int ROWS_PER_BLOCK = 32;  // 4 waves x 8 rows: one block's partial da / db covers 32 rows

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ a,
                                                            const float *__restrict__ b, long rows, int D, float eps,
                                                            float *__restrict__ y, float *__restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (long row = wave * 8; row < min(rows, wave * 8 + 8); ++row) {
    const float *xr = x + row * D;
    float s = 0.f;
    for (int j = lane; j < D; j += 64) s += xr[j];
    const float mu = spacap::wave_sum_f32(s) / (float)D;
    float q = 0.f;
    for (int j = lane; j < D; j += 64) { const float c = xr[j] - mu; q += c * c; }
    const float var = spacap::wave_sum_f32(q) / (float)(D - 1);
    const float r = 1.0f / (sqrtf(var) + eps);
    float *yr = y + row * D;
    for (int j = lane; j < D; j += 64) yr[j] = a[j] * ((xr[j] - mu) * r) + b[j];
    if (lane == 0) { stats[row * 2] = mu; stats[row * 2 + 1] = r; }
  }
}

// dx for 32 rows per block + that block's partial sums of da (= sum dy * xhat) and db (= sum dy)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ a,
                                                            const float *__restrict__ stats,
                                                            const float *__restrict__ dy, long rows, int D, float eps,
                                                            float *__restrict__ dx, float *__restrict__ part,
                                                            const float *__restrict__ addend) {
  extern __shared__ float s_part[];  // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long wave = (long)blockIdx.x * 4 + wid;
  float *pa = s_part + (size_t)wid * 2 * D, *pb = pa + D;
  for (int j = lane; j < D; j += 64) { pa[j] = 0.f; pb[j] = 0.f; }
  for (long row = wave * 8; row < min(rows, wave * 8 + 8); ++row) {
    const float *xr = x + row * D, *gr = dy + row * D;
    const float mu = stats[row * 2], r = stats[row * 2 + 1];
    float s1 = 0.f, s2 = 0.f;  // sum dxhat, sum dxhat * xc
    for (int j = lane; j < D; j += 64) {
      const float g = gr[j], xc = xr[j] - mu, dxh = g * a[j];
      s1 += dxh;
      s2 += dxh * xc;
      pa[j] += g * (xc * r);
      pb[j] += g;
    }
    s1 = spacap::wave_sum_f32(s1);
    s2 = spacap::wave_sum_f32(s2);
    const float sd = 1.0f / r - eps;                       // the unbiased std
    const float c2 = -(s2 * r * r) / (sd * (float)(D - 1));  // 0/0 = NaN on a constant row, as autograd gives
    const float c1 = r * s1 / (float)D;
    float *dr = dx + row * D;
    for (int j = lane; j < D; j += 64)
      dr[j] = r * (gr[j] * a[j]) + c2 * (xr[j] - mu) - c1 + (addend ? addend[row * D + j] : 0.f);
  }
  __syncthreads();
  float *out = part + (size_t)blockIdx.x * 2 * D;
  for (int j = threadIdx.x; j < 2 * D; j += 256)
    out[j] = s_part[j] + s_part[2 * D + j] + s_part[4 * D + j] + s_part[6 * D + j];
}

__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float *__restrict__ part, int nblocks, int D,
                                                                   float *__restrict__ da, float *__restrict__ db) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= 2 * D) return;
  float s = 0.f;
  for (int p = 0; p < nblocks; ++p) s += part[(size_t)p * 2 * D + j];
  if (j < D) da[j] = s; else db[j - D] = s;
}

// ---- register-resident variants for D <= 512 (the model: D = 128) ---------------------------------------------
// One wavefront per row with the row held in registers (NV values per lane): one pass over x, no dependent
// row-after-row chain, and rows / 4 workgroups instead of rows / 32 (2 048 rows: 512 workgroups on 256 CUs).
// Same arithmetic and summation order per row as the generic kernels above.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_reg_kernel(const float *__restrict__ x, const float *__restrict__ a,
                                                                const float *__restrict__ b, long rows, int D, float eps,
                                                                float *__restrict__ y, float *__restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *xr = x + row * D;
  float v[NV], s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    v[i] = j < D ? xr[j] : 0.f;
    if (j < D) s += v[i];
  }
  const float mu = spacap::wave_sum_f32(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < D) { const float c = v[i] - mu; q += c * c; }
  const float var = spacap::wave_sum_f32(q) / (float)(D - 1);
  const float r = 1.0f / (sqrtf(var) + eps);
  float *yr = y + row * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    if (j < D) yr[j] = a[j] * ((v[i] - mu) * r) + b[j];
  }
  if (lane == 0) { stats[row * 2] = mu; stats[row * 2 + 1] = r; }
}

constexpr int BWD_RW = 2;  // rows per wave of the register backward kernel (8 rows per workgroup partial)

template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_reg_kernel(const float *__restrict__ x, const float *__restrict__ a,
                                                                const float *__restrict__ stats,
                                                                const float *__restrict__ dy, long rows, int D, float eps,
                                                                float *__restrict__ dx, float *__restrict__ part,
                                                                const float *__restrict__ addend) {
  __shared__ float s_part[4][2][NV * 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long row0 = ((long)blockIdx.x * 4 + wid) * BWD_RW;
  float av[NV], pa[NV], pb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    av[i] = j < D ? a[j] : 0.f;
    pa[i] = pb[i] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < BWD_RW; ++k) {
    const long row = row0 + k;
    if (row >= rows) break;
    const float *xr = x + row * D, *gr = dy + row * D;
    const float mu = stats[row * 2], r = stats[row * 2 + 1];
    float xc[NV], g[NV], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      const bool in = j < D;
      g[i] = in ? gr[j] : 0.f;
      xc[i] = in ? xr[j] - mu : 0.f;
      const float dxh = g[i] * av[i];
      if (in) { s1 += dxh; s2 += dxh * xc[i]; }
      pa[i] += g[i] * (xc[i] * r);
      pb[i] += g[i];
    }
    s1 = spacap::wave_sum_f32(s1);
    s2 = spacap::wave_sum_f32(s2);
    const float sd = 1.0f / r - eps;
    const float c2 = -(s2 * r * r) / (sd * (float)(D - 1));
    const float c1 = r * s1 / (float)D;
    float *dr = dx + row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < D) dr[j] = r * (g[i] * av[i]) + c2 * xc[i] - c1 + (addend ? addend[row * D + j] : 0.f);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    s_part[wid][0][lane + 64 * i] = pa[i];
    s_part[wid][1][lane + 64 * i] = pb[i];
  }
  __syncthreads();
  float *out = part + (size_t)blockIdx.x * 2 * D;
  for (int e = threadIdx.x; e < 2 * D; e += 256) {
    const int k = e / D, j = e - k * D;
    out[e] = s_part[0][k][j] + s_part[1][k][j] + s_part[2][k][j] + s_part[3][k][j];
  }
}

// fixed-order sum of the workgroup partials: 64 columns x 4 slabs of partial rows per workgroup
__global__ __launch_bounds__(256) void layernorm_bwd_reduce4_kernel(const float *__restrict__ part, int nblocks, int D,
                                                                    float *__restrict__ da, float *__restrict__ db) {
  __shared__ float s[4][64];
  const int c = threadIdx.x & 63, slab = threadIdx.x >> 6, j = blockIdx.x * 64 + c;
  float a = 0.f;
  if (j < 2 * D) {
    const int per = (nblocks + 3) / 4, p0 = slab * per, p1 = min(nblocks, p0 + per);
#pragma unroll 8
    for (int p = p0; p < p1; ++p) a += part[(size_t)p * 2 * D + j];
  }
  s[slab][c] = a;
  __syncthreads();
  if (slab == 0 && j < 2 * D) {
    const float t = (s[0][c] + s[1][c]) + (s[2][c] + s[3][c]);
    if (j < D) da[j] = t; else db[j - D] = t;
  }
}

}  // namespace

extern "C" int spacap_layernorm_fwd_f32(const float *x, const float *a, const float *b, long rows, int D, float eps,
                                        float *y, float *stats, spacap_stream_t stream) {
  SPACAP_REQUIRE(rows >= 0 && D >= 2 && D <= 8192, "spacap_layernorm_fwd_f32: bad sizes rows=%ld D=%d", rows, D);
  if (rows == 0) return SPACAP_OK;
  SPACAP_REQUIRE(x && a && b && y && stats, "spacap_layernorm_fwd_f32: null pointer");
  hipStream_t s = spacap::as_stream(stream);
  const unsigned g4 = (unsigned)((rows + 3) / 4);
  if (D <= 128)
    hipLaunchKernelGGL((layernorm_fwd_reg_kernel<2>), dim3(g4), dim3(256), 0, s, x, a, b, rows, D, eps, y, stats);
  else if (D <= 256)
    hipLaunchKernelGGL((layernorm_fwd_reg_kernel<4>), dim3(g4), dim3(256), 0, s, x, a, b, rows, D, eps, y, stats);
  else if (D <= 512)
    hipLaunchKernelGGL((layernorm_fwd_reg_kernel<8>), dim3(g4), dim3(256), 0, s, x, a, b, rows, D, eps, y, stats);
  else {
    const unsigned grid = (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid), dim3(256), 0, s, x, a, b, rows, D, eps, y, stats);
  }
  SPACAP_CHECK_LAUNCH("spacap_layernorm_fwd_f32");
  return SPACAP_OK;
}

extern "C" size_t spacap_layernorm_bwd_workspace_bytes(long rows, int D) {
  if (rows <= 0 || D <= 0) return 0;
  const long per = D <= 512 ? 4 * BWD_RW : ROWS_PER_BLOCK;  // rows per workgroup partial
  return (size_t)((rows + per - 1) / per) * 2 * D * sizeof(float);
}

// dx = (LayerNorm backward of dy) + addend: the pre-norm residual connection x + f(norm(x)) sends the gradient of its
// output to x twice (directly and through the norm); with `addend` = the direct part the sum needs no extra pass.
extern "C" int spacap_layernorm_bwd_add_f32(const float *x, const float *a, const float *stats, const float *dy,
                                            const float *addend, long rows, int D, float eps, float *dx, float *da, float *db,
                                            void *workspace, spacap_stream_t stream);

extern "C" int spacap_layernorm_bwd_f32(const float *x, const float *a, const float *stats, const float *dy, long rows,
                                        int D, float eps, float *dx, float *da, float *db, void *workspace,
                                        spacap_stream_t stream) {
  return spacap_layernorm_bwd_add_f32(x, a, stats, dy, nullptr, rows, D, eps, dx, da, db, workspace, stream);
}

extern "C" int spacap_layernorm_bwd_add_f32(const float *x, const float *a, const float *stats, const float *dy,
                                            const float *addend, long rows, int D, float eps, float *dx, float *da, float *db,
                                            void *workspace, spacap_stream_t stream) {
  SPACAP_REQUIRE(rows >= 0 && D >= 2 && D <= 2048, "spacap_layernorm_bwd_f32: bad sizes rows=%ld D=%d", rows, D);
  // da == db == null: the caller reduces the workgroup partials in `workspace` ([blocks][2*D], blocks =
  // spacap_layernorm_bwd_workspace_bytes / (8*D)) itself -- e.g. batched with other slab sums (spacap_sum_slabs_*)
  SPACAP_REQUIRE((da && db) || (!da && !db), "spacap_layernorm_bwd_f32: da / db must both be given or both be null");
  const bool reduce = da != nullptr;
  hipStream_t s = spacap::as_stream(stream);
  if (rows == 0) {
    if (reduce) {
      SPACAP_CHECK_HIP(hipMemsetAsync(da, 0, sizeof(float) * D, s), "spacap_layernorm_bwd_f32");
      SPACAP_CHECK_HIP(hipMemsetAsync(db, 0, sizeof(float) * D, s), "spacap_layernorm_bwd_f32");
    }
    return SPACAP_OK;
  }
  SPACAP_REQUIRE(x && a && stats && dy && dx && workspace, "spacap_layernorm_bwd_f32: null pointer");
  float *part = reinterpret_cast<float *>(workspace);
  if (D <= 512) {
    const unsigned grid = (unsigned)((rows + 4 * BWD_RW - 1) / (4 * BWD_RW));
    if (D <= 128)
      hipLaunchKernelGGL((layernorm_bwd_reg_kernel<2>), dim3(grid), dim3(256), 0, s, x, a, stats, dy, rows, D, eps, dx, part, addend);
    else if (D <= 256)
      hipLaunchKernelGGL((layernorm_bwd_reg_kernel<4>), dim3(grid), dim3(256), 0, s, x, a, stats, dy, rows, D, eps, dx, part, addend);
    else
      hipLaunchKernelGGL((layernorm_bwd_reg_kernel<8>), dim3(grid), dim3(256), 0, s, x, a, stats, dy, rows, D, eps, dx, part, addend);
    if (reduce)
      hipLaunchKernelGGL(layernorm_bwd_reduce4_kernel, dim3((2 * D + 63) / 64), dim3(256), 0, s, part, (int)grid, D, da, db);
  } else {
    const unsigned grid = (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(grid), dim3(256), sizeof(float) * 8 * D, s, x, a, stats, dy, rows, D,
                       eps, dx, part, addend);
    if (reduce)
      hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((2 * D + 255) / 256), dim3(256), 0, s, part, (int)grid, D, da, db);
  }
  SPACAP_CHECK_LAUNCH("spacap_layernorm_bwd_f32");
  return SPACAP_OK;
}
