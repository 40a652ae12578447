// three_nn / three_interpolate (+grad) for gfx950 (MI355X).
//
// Replaces lib/pointnet2/_ext_src/src/interpolate_gpu.cu:9-154 (host: src/interpolate.cpp:14-99).
//  * three_nn: indices bit-exact -- sequential scan over the known points in index order with the
//    reference's strict `<` cascade (interpolate_gpu.cu:34-49), un-contracted fp32 distance (:32).
//    The reference holds its bests in doubles initialised to 1e40 but only ever compares them with,
//    and assigns them from, fp32 values; fp32 bests initialised to +inf take the same branches and
//    convert to the same outputs (1e40 -> +inf in fp32).
//  * three_interpolate: p[i1]*w1 + p[i2]*w2 + p[i3]*w3, left to right, un-contracted (:98-99).
//  * three_interpolate_grad: the reference does three float atomic adds per element (:139-141); here a gather
//    in the oracle's (n, k) order (three_interpolate_grad_gather_kernel below).
//
// Design: one lane per unknown point with the known point of each step held in SGPRs (its index is
// wave-uniform, so the loads are scalar and broadcast for free); 64-thread workgroups so that the
// (B, n/64) grid spreads over the chip.  Interpolation uses lane = unknown index (coalesced weight /
// index / output traffic) and a channel slab per workgroup.
#include <math.h>

#include "common.hpp"

#pragma clang fp contract(off)

namespace {

// WEIGHTS: instead of the squared distances, the normalised inverse distances of the feature-propagation modules
// (pointnet2_modules.py:399-405: dist = sqrt(dist2), r = 1 / (dist + 1e-8), w = r / (r0 + r1 + r2)), each operation rounded
// as the reference's separate tensor operations round it.
template <bool WEIGHTS>
__global__ __launch_bounds__(64) void three_nn_kernel(const float *__restrict__ unknown_all,
                                                      const float *__restrict__ known_all, int n, int m,
                                                      float *__restrict__ dist2_all,
                                                      int32_t *__restrict__ idx_all) {
  // The known points go through LDS in chunks of KT (every lane reads the same point: a broadcast read instead of three
  // dependent global loads per candidate, which made the m = 512 search of the second feature-propagation level 119 us on the
  // side stream) and the three best are kept by selects: the reference's strict `<` cascade, without its divergent branches.
  constexpr int KT = 512;
  __shared__ float s_k[KT * 3];
  const int b = blockIdx.y;
  const int j = blockIdx.x * 64 + threadIdx.x;
  const float *__restrict__ known = known_all + (size_t)b * m * 3;
  const int jj = j < n ? j : n - 1;
  const float *__restrict__ u = unknown_all + ((size_t)b * n + jj) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  float best1 = INFINITY, best2 = INFINITY, best3 = INFINITY;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k0 = 0; k0 < m; k0 += KT) {
    const int kn = min(KT, m - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < kn * 3; i += 64) s_k[i] = known[(size_t)k0 * 3 + i];
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < kn; ++k) {
      const float x = s_k[k * 3 + 0], y = s_k[k * 3 + 1], z = s_k[k * 3 + 2];
      const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
      const int kk = k0 + k;
      const bool l1 = d < best1, l2 = d < best2, l3 = d < best3;
      // (l1 implies l2 implies l3: best1 <= best2 <= best3)
      best3 = l2 ? best2 : (l3 ? d : best3);
      besti3 = l2 ? besti2 : (l3 ? kk : besti3);
      best2 = l1 ? best1 : (l2 ? d : best2);
      besti2 = l1 ? besti1 : (l2 ? kk : besti2);
      best1 = l1 ? d : best1;
      besti1 = l1 ? kk : besti1;
    }
  }
  if (j < n) {
    float *__restrict__ dd = dist2_all + ((size_t)b * n + j) * 3;
    int32_t *__restrict__ ii = idx_all + ((size_t)b * n + j) * 3;
    if (WEIGHTS) {
      const float r1 = 1.0f / (sqrtf(best1) + 1e-8f), r2 = 1.0f / (sqrtf(best2) + 1e-8f), r3 = 1.0f / (sqrtf(best3) + 1e-8f);
      const float norm = (r1 + r3) + r2;   // the order torch.sum(dim=2) of a (B, n, 3) tensor adds in (four strided accumulators)
      dd[0] = r1 / norm; dd[1] = r2 / norm; dd[2] = r3 / norm;
    } else {
      dd[0] = best1; dd[1] = best2; dd[2] = best3;
    }
    ii[0] = besti1; ii[1] = besti2; ii[2] = besti3;
  }
}

// out[b, j, :] = xyz[b, idx[b, j], :] (the sampled centres of a set-abstraction level)
__global__ __launch_bounds__(256) void gather_xyz_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ idx, int N, int m,
                                                         long total, float *__restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long b = i / m;
  const float *p = xyz + ((size_t)b * N + idx[i]) * 3;
  out[i * 3 + 0] = p[0], out[i * 3 + 1] = p[1], out[i * 3 + 2] = p[2];
}

constexpr int CHUNK = 8;

__global__ __launch_bounds__(256) void three_interpolate_kernel(const float *__restrict__ points,
                                                                const int32_t *__restrict__ idx,
                                                                const float *__restrict__ weight, int C,
                                                                int m, int n, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const size_t r = ((size_t)b * n + j) * 3;
  const float w1 = weight[r + 0], w2 = weight[r + 1], w3 = weight[r + 2];
  const int i1 = idx[r + 0], i2 = idx[r + 1], i3 = idx[r + 2];
  const int c0 = blockIdx.y * CHUNK, c1 = min(c0 + CHUNK, C);
  const float *__restrict__ p = points + ((size_t)b * C + c0) * m;
  float *__restrict__ o = out + ((size_t)b * C + c0) * n + j;
  for (int c = c0; c < c1; ++c, p += m, o += n) *o = p[i1] * w1 + p[i2] * w2 + p[i3] * w3;
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(const float *__restrict__ grad_out,
                                                                     const int32_t *__restrict__ idx,
                                                                     const float *__restrict__ weight,
                                                                     int C, int n, int m,
                                                                     float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const size_t r = ((size_t)b * n + j) * 3;
  const float w1 = weight[r + 0], w2 = weight[r + 1], w3 = weight[r + 2];
  const int i1 = idx[r + 0], i2 = idx[r + 1], i3 = idx[r + 2];
  const int c0 = blockIdx.y * CHUNK, c1 = min(c0 + CHUNK, C);
  float *__restrict__ g = grad_points + ((size_t)b * C + c0) * m;
  const float *__restrict__ go = grad_out + ((size_t)b * C + c0) * n + j;
  for (int c = c0; c < c1; ++c, g += m, go += n) {
    const float v = *go;
    atomicAdd(g + i1, v * w1);
    atomicAdd(g + i2, v * w2);
    atomicAdd(g + i3, v * w3);
  }
}

// Gather form of the same gradient: the (unknown point, neighbour slot) pairs that reference each known point are
// collected per workgroup (16 known points), sorted into ascending (n, k) order and summed by one lane per channel:
// no float atomics (scattered 4-byte float atomics: 115 us for the FP2 gradient), every output written exactly once,
// and the same summation order as the CPU oracle's loop (bit-identical, unlike the reference's atomics).
constexpr int IG_JT = 16, IG_CAP = 96;

__global__ __launch_bounds__(256) void three_interpolate_grad_gather_kernel(const float *__restrict__ grad_out,
                                                                            const int32_t *__restrict__ idx,
                                                                            const float *__restrict__ weight, int C,
                                                                            int n, int m,
                                                                            float *__restrict__ grad_points) {
  __shared__ int s_cnt[IG_JT];
  __shared__ int s_e[IG_JT][IG_CAP];
  const int b = blockIdx.y, j0 = blockIdx.x * IG_JT, tid = threadIdx.x;
  const int32_t *__restrict__ ib = idx + (size_t)b * n * 3;
  const float *__restrict__ wb = weight + (size_t)b * n * 3;
  if (tid < IG_JT) s_cnt[tid] = 0;
  __syncthreads();
  for (int e = tid; e < 3 * n; e += 256) {
    const int jj = ib[e] - j0;
    if (jj >= 0 && jj < IG_JT) {
      const int pos = atomicAdd(&s_cnt[jj], 1);
      if (pos < IG_CAP) s_e[jj][pos] = e;
    }
  }
  __syncthreads();
  if (tid < IG_JT) {  // ascending (n, k): insertion sort of a short list
    const int cnt = min(s_cnt[tid], IG_CAP);
    for (int a = 1; a < cnt; ++a) {
      const int v = s_e[tid][a];
      int q = a - 1;
      while (q >= 0 && s_e[tid][q] > v) { s_e[tid][q + 1] = s_e[tid][q]; --q; }
      s_e[tid][q + 1] = v;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const float *__restrict__ go = grad_out + ((size_t)b * C + c) * n;
    float *__restrict__ gp = grad_points + ((size_t)b * C + c) * m;
    for (int jj = 0; jj < IG_JT && j0 + jj < m; ++jj) {
      const int cnt = s_cnt[jj];
      float acc = 0.f;
      if (cnt <= IG_CAP) {
        for (int p = 0; p < cnt; ++p) {
          const int e = s_e[jj][p];
          acc += go[e / 3] * wb[e];
        }
      } else {  // more references than the list holds: walk every pair in order
        for (int e = 0; e < 3 * n; ++e)
          if (ib[e] == j0 + jj) acc += go[e / 3] * wb[e];
      }
      gp[j0 + jj] = acc;
    }
  }
}

// The same gather on POINT-MAJOR gradients: grad_pm [B, n, C] -> grad_points_pm [B, m, C].  With channel-major
// operands every list entry is one 4-byte element of a 4 KB row, so a workgroup drags all C rows of its scene through
// the cache (85 us for the FP2 gradient); point-major, an entry is a contiguous C-float row read by consecutive
// lanes.  Same lists, same ascending (n, k) summation order, same values.
__global__ __launch_bounds__(256) void three_interpolate_grad_pm_kernel(const float *__restrict__ grad_pm,
                                                                        const int32_t *__restrict__ idx,
                                                                        const float *__restrict__ weight, int C, int n,
                                                                        int m, float *__restrict__ grad_points_pm) {
  __shared__ int s_cnt[IG_JT];
  __shared__ int s_raw[IG_JT][IG_CAP];
  __shared__ int s_e[IG_JT][IG_CAP];
  __shared__ float s_w[IG_JT][IG_CAP];
  const int b = blockIdx.y, j0 = blockIdx.x * IG_JT, tid = threadIdx.x;
  const int32_t *__restrict__ ib = idx + (size_t)b * n * 3;
  const float *__restrict__ wb = weight + (size_t)b * n * 3;
  if (tid < IG_JT) s_cnt[tid] = 0;
  __syncthreads();
  for (int e = tid; e < 3 * n; e += 256) {
    const int jj = ib[e] - j0;
    if (jj >= 0 && jj < IG_JT) {
      const int pos = atomicAdd(&s_cnt[jj], 1);
      if (pos < IG_CAP) s_raw[jj][pos] = e;
    }
  }
  __syncthreads();
  // ascending (n, k) by rank: every entry counts the smaller ones of its list (all threads, no serial sort)
  for (int t = tid; t < IG_JT * IG_CAP; t += 256) {
    const int jj = t / IG_CAP, p = t - jj * IG_CAP, cnt = min(s_cnt[jj], IG_CAP);
    if (p < cnt) {
      const int v = s_raw[jj][p];
      int rank = 0;
      for (int q = 0; q < cnt; ++q) rank += s_raw[jj][q] < v;
      s_e[jj][rank] = v;
      s_w[jj][rank] = wb[v];
    }
  }
  __syncthreads();
  const float *__restrict__ go = grad_pm + (size_t)b * n * C;
  float *__restrict__ gp = grad_points_pm + (size_t)b * m * C;
  if ((C & 3) == 0 && ((reinterpret_cast<uintptr_t>(grad_pm) | reinterpret_cast<uintptr_t>(grad_points_pm)) & 15) == 0) {
    // thread = (known point, four channels): 16-byte pieces of the rows, the points of the tile side by side (the first version
    // walked the tile's 16 points one after the other with one channel per thread: 96 dependent 4-byte loads per thread)
    const int C4 = C >> 2;
    for (int t = tid; t < IG_JT * C4; t += 256) {
      const int jj = t / C4, c4 = t - jj * C4;
      if (j0 + jj >= m) continue;
      const int cnt = s_cnt[jj];
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cnt <= IG_CAP) {
        for (int p = 0; p < cnt; ++p) {
          const float4 g = *reinterpret_cast<const float4 *>(go + (size_t)(s_e[jj][p] / 3) * C + 4 * c4);
          const float w = s_w[jj][p];
          acc.x += g.x * w, acc.y += g.y * w, acc.z += g.z * w, acc.w += g.w * w;
        }
      } else {  // more references than the list holds: walk every pair in order
        for (int e = 0; e < 3 * n; ++e)
          if (ib[e] == j0 + jj) {
            const float4 g = *reinterpret_cast<const float4 *>(go + (size_t)(e / 3) * C + 4 * c4);
            const float w = wb[e];
            acc.x += g.x * w, acc.y += g.y * w, acc.z += g.z * w, acc.w += g.w * w;
          }
      }
      *reinterpret_cast<float4 *>(gp + (size_t)(j0 + jj) * C + 4 * c4) = acc;
    }
    return;
  }
  for (int c = tid; c < C; c += 256) {
    for (int jj = 0; jj < IG_JT && j0 + jj < m; ++jj) {
      const int cnt = s_cnt[jj];
      float acc = 0.f;
      if (cnt <= IG_CAP) {
        for (int p = 0; p < cnt; ++p) acc += go[(size_t)(s_e[jj][p] / 3) * C + c] * s_w[jj][p];
      } else {  // more references than the list holds: walk every pair in order
        for (int e = 0; e < 3 * n; ++e)
          if (ib[e] == j0 + jj) acc += go[(size_t)(e / 3) * C + c] * wb[e];
      }
      gp[(size_t)(j0 + jj) * C + c] = acc;
    }
  }
}

}  // namespace

// grad_pm f32 [B,n,C] (point-major gradient of the interpolated features), idx i32 [B,n,3], weight f32 [B,n,3]
// -> grad_points_pm f32 [B,m,C]; every element written.
extern "C" int spacap_three_interpolate_grad_pm_f32(const float *grad_pm, const int32_t *idx, const float *weight, int B,
                                                    int C, int n, int m, float *grad_points_pm, spacap_stream_t stream) {
  const char *what = "spacap_three_interpolate_grad_pm_f32";
  SPACAP_REQUIRE(B >= 0 && C >= 0 && m >= 0 && n >= 0 && B <= 65535, "%s: bad sizes", what);
  if (B == 0 || C == 0 || m == 0) return SPACAP_OK;
  SPACAP_REQUIRE(grad_points_pm, "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
  if (n == 0) {
    SPACAP_CHECK_HIP(hipMemsetAsync(grad_points_pm, 0, sizeof(float) * (size_t)B * C * m, s), what);
    return SPACAP_OK;
  }
  SPACAP_REQUIRE(grad_pm && idx && weight, "%s: null pointer", what);
  hipLaunchKernelGGL(three_interpolate_grad_pm_kernel, dim3((m + IG_JT - 1) / IG_JT, B), dim3(256), 0, s, grad_pm, idx, weight, C,
                     n, m, grad_points_pm);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

static int three_nn_launch(const char *what, bool weights, const float *unknown, const float *known, int B, int n, int m, float *out,
                           int32_t *idx, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 0 && n >= 0 && m >= 0, "%s: bad sizes", what);
  if (B == 0 || n == 0) return SPACAP_OK;
  SPACAP_REQUIRE(unknown && out && idx && (known || m == 0), "%s: null pointer", what);
  SPACAP_REQUIRE(B <= 65535, "%s: B out of range", what);
  dim3 grid((n + 63) / 64, B);
  if (weights) hipLaunchKernelGGL(three_nn_kernel<true>, grid, dim3(64), 0, spacap::as_stream(stream), unknown, known, n, m, out, idx);
  else hipLaunchKernelGGL(three_nn_kernel<false>, grid, dim3(64), 0, spacap::as_stream(stream), unknown, known, n, m, out, idx);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_three_nn_f32(const float *unknown, const float *known, int B, int n, int m,
                                   float *dist2, int32_t *idx, spacap_stream_t stream) {
  return three_nn_launch("spacap_three_nn_f32", false, unknown, known, B, n, m, dist2, idx, stream);
}

extern "C" int spacap_three_nn_weights_f32(const float *unknown, const float *known, int B, int n, int m,
                                           float *weight, int32_t *idx, spacap_stream_t stream) {
  return three_nn_launch("spacap_three_nn_weights_f32", true, unknown, known, B, n, m, weight, idx, stream);
}

extern "C" int spacap_gather_xyz_f32(const float *xyz, const int32_t *idx, int B, int N, int m, float *out, spacap_stream_t stream) {
  const char *what = "spacap_gather_xyz_f32";
  SPACAP_REQUIRE(B >= 0 && N >= 0 && m >= 0, "%s: bad sizes", what);
  const long total = (long)B * m;
  if (total == 0) return SPACAP_OK;
  SPACAP_REQUIRE(xyz && idx && out && N > 0, "%s: null pointer", what);
  hipLaunchKernelGGL(gather_xyz_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, spacap::as_stream(stream), xyz, idx, N, m,
                     total, out);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_three_interpolate_f32(const float *points, const int32_t *idx, const float *weight,
                                            int B, int C, int m, int n, float *out,
                                            spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 0 && C >= 0 && m >= 0 && n >= 0, "spacap_three_interpolate_f32: bad sizes");
  if (B == 0 || C == 0 || n == 0) return SPACAP_OK;
  SPACAP_REQUIRE(points && idx && weight && out, "spacap_three_interpolate_f32: null pointer");
  SPACAP_REQUIRE(B <= 65535, "spacap_three_interpolate_f32: B out of range");
  dim3 grid((n + 255) / 256, (C + CHUNK - 1) / CHUNK, B);
  hipLaunchKernelGGL(three_interpolate_kernel, grid, dim3(256), 0, spacap::as_stream(stream), points, idx,
                     weight, C, m, n, out);
  SPACAP_CHECK_LAUNCH("spacap_three_interpolate_f32");
  return SPACAP_OK;
}

extern "C" int spacap_three_interpolate_grad_f32(const float *grad_out, const int32_t *idx,
                                                 const float *weight, int B, int C, int n, int m,
                                                 float *grad_points, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 0 && C >= 0 && m >= 0 && n >= 0, "spacap_three_interpolate_grad_f32: bad sizes");
  if (B == 0 || C == 0 || m == 0) return SPACAP_OK;
  SPACAP_REQUIRE(grad_points, "spacap_three_interpolate_grad_f32: null pointer");
  hipStream_t s = spacap::as_stream(stream);
  if (n == 0) {
    SPACAP_CHECK_HIP(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)B * C * m, s),
                     "spacap_three_interpolate_grad_f32(memset)");
    return SPACAP_OK;
  }
  SPACAP_REQUIRE(grad_out && idx && weight, "spacap_three_interpolate_grad_f32: null pointer");
  SPACAP_REQUIRE(B <= 65535, "spacap_three_interpolate_grad_f32: B out of range");
  hipLaunchKernelGGL(three_interpolate_grad_gather_kernel, dim3((m + IG_JT - 1) / IG_JT, B), dim3(256), 0, s, grad_out, idx,
                     weight, C, n, m, grad_points);
  SPACAP_CHECK_LAUNCH("spacap_three_interpolate_grad_f32");
  return SPACAP_OK;
}

// ===========================================================================================================
// The input of a feature-propagation module's shared MLP in one launch each way (lib/pointnet2/pointnet2_modules.py:406-412:
//   interpolated = three_interpolate(known_feats, idx, weight);  new_features = torch.cat([interpolated, unknow_feats], dim=1)).
// cat[b][c][j] (channel-major, what the 1x1 convolution reads), c < K1: sum_i w[b][j][i] * known[b][idx[b][j][i]][c] with the
// known features either point-major (B, m, K1) -- an SA module's own output: three coalesced row reads per point -- or channel-
// major (B, K1, m) -- a previous FP module's output --; c >= K1: the skip features, point-major (B, n, K2) (an SA module's output).
// Replaces a transposed copy + three_interpolate + cat.  32 points x 32 channels per workgroup through an LDS tile: reads run
// along the channels of point-major rows, writes along the points of channel-major rows.  The three products are added left to
// right like three_interpolate_kernel (same values).  Backward: the gradient of cat is split into its two halves, both written
// point-major (what three_interpolate_grad_pm and the SA modules' backward read): replaces two narrowed, transposed copies.
namespace {
__global__ __launch_bounds__(256) void fp_concat_fwd_kernel(const float *__restrict__ known, int known_pm, const int32_t *__restrict__ idx,
                                                            const float *__restrict__ weight, const float *__restrict__ skip, int K1,
                                                            int K2, int m, int n, float *__restrict__ out) {
  __shared__ float s[32][33];
  const int b = blockIdx.z, j0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tid = threadIdx.x;
  const int jl = tid >> 3, cq = (tid & 7) * 4;       // load map: point jl, channels c0 + cq .. + 3
  const int j = j0 + jl;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (j < n) {
    if (c0 < K1) {
      const size_t r = ((size_t)b * n + j) * 3;
      const float w1 = weight[r], w2 = weight[r + 1], w3 = weight[r + 2];
      const int i1 = idx[r], i2 = idx[r + 1], i3 = idx[r + 2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + cq + u;
        if (c < K1) {
          if (known_pm) {
            const float *p = known + (size_t)b * m * K1 + c;
            v[u] = p[(size_t)i1 * K1] * w1 + p[(size_t)i2 * K1] * w2 + p[(size_t)i3 * K1] * w3;
          } else {
            const float *p = known + ((size_t)b * K1 + c) * m;
            v[u] = p[i1] * w1 + p[i2] * w2 + p[i3] * w3;
          }
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 - K1 + cq + u;
        if (c < K2) v[u] = skip[((size_t)b * n + j) * K2 + c];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) s[jl][cq + u] = v[u];
  __syncthreads();
  const int cl = tid >> 3, jq = (tid & 7) * 4;        // store map: channel cl, points j0 + jq .. + 3
  const int c = c0 + cl;
  if (c < K1 + K2 && (c0 >= K1 || c < K1)) {
    float *o = out + ((size_t)b * (K1 + K2) + c) * n + j0 + jq;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (j0 + jq + u < n) o[u] = s[jq + u][cl];
  }
}

// g (B, K1 + K2, n) channel-major -> g1 (B, n, K1), g2 (B, n, K2) point-major
__global__ __launch_bounds__(256) void fp_concat_bwd_kernel(const float *__restrict__ g, int K1, int K2, int n, float *__restrict__ g1,
                                                            float *__restrict__ g2) {
  __shared__ float s[32][33];
  const int b = blockIdx.z, j0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tid = threadIdx.x;
  const int cl = tid >> 3, jq = (tid & 7) * 4;        // load map: channel cl, points j0 + jq .. + 3
  const int c = c0 + cl;
  const bool cok = c < K1 + K2 && (c0 >= K1 || c < K1);
#pragma unroll
  for (int u = 0; u < 4; ++u) s[jq + u][cl] = (cok && j0 + jq + u < n) ? g[((size_t)b * (K1 + K2) + c) * n + j0 + jq + u] : 0.f;
  __syncthreads();
  const int jl = tid >> 3, cq = (tid & 7) * 4;        // store map: point jl, channels c0 + cq .. + 3
  const int j = j0 + jl;
  if (j >= n) return;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int cc = c0 + cq + u;
    if (c0 < K1) {
      if (cc < K1) g1[((size_t)b * n + j) * K1 + cc] = s[jl][cq + u];
    } else if (cc - K1 < K2) {
      g2[((size_t)b * n + j) * K2 + cc - K1] = s[jl][cq + u];
    }
  }
}
}  // namespace

/* cat f32 [B, K1+K2, n] = [three_interpolate(known, idx, weight) ; skip^T]: known f32 [B,m,K1] (known_pm != 0) or [B,K1,m], idx
   i32 [B,n,3], weight f32 [B,n,3], skip f32 [B,n,K2] point-major.  K1 a multiple of 32 (a tile never straddles the two halves). */
extern "C" int spacap_fp_concat_fwd_f32(const float *known, int known_pm, const int32_t *idx, const float *weight, const float *skip,
                                        int B, int K1, int K2, int m, int n, float *out, spacap_stream_t stream) {
  const char *what = "spacap_fp_concat_fwd_f32";
  SPACAP_REQUIRE(B >= 0 && K1 >= 32 && K1 % 32 == 0 && K2 >= 1 && m >= 1 && n >= 1 && B <= 65535, "%s: bad sizes", what);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(known && idx && weight && skip && out, "%s: null pointer", what);
  hipLaunchKernelGGL(fp_concat_fwd_kernel, dim3((n + 31) / 32, (K1 + K2 + 31) / 32, B), dim3(256), 0, spacap::as_stream(stream), known,
                     known_pm, idx, weight, skip, K1, K2, m, n, out);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

/* the gradient of that concatenation split into its halves, both point-major: g f32 [B,K1+K2,n] -> g1 f32 [B,n,K1], g2 f32 [B,n,K2] */
extern "C" int spacap_fp_concat_bwd_f32(const float *g, int B, int K1, int K2, int n, float *g1, float *g2, spacap_stream_t stream) {
  const char *what = "spacap_fp_concat_bwd_f32";
  SPACAP_REQUIRE(B >= 0 && K1 >= 32 && K1 % 32 == 0 && K2 >= 1 && n >= 1 && B <= 65535, "%s: bad sizes", what);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(g && g1 && g2, "%s: null pointer", what);
  hipLaunchKernelGGL(fp_concat_bwd_kernel, dim3((n + 31) / 32, (K1 + K2 + 31) / 32, B), dim3(256), 0, spacap::as_stream(stream), g, K1, K2, n,
                     g1, g2);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
