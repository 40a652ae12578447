// Train-mode BatchNorm + ReLU (+ max over the samples of a group) for the shared MLPs, gfx950 (MI355X).
//
// Replaces, inside `SharedMLP` (lib/pointnet2/pytorch_utils.py:11-36: Conv2d(1x1, bias=False) -> BatchNorm2d ->
// ReLU(inplace)) and the pooling after it (lib/pointnet2/pointnet2_modules.py:253-259), the chain
//     BatchNorm2d (batch statistics)  ->  ReLU  [ ->  F.max_pool2d(kernel=[1, nsample]) ]
// and its backward.  On the SA1 tensors (B=8: 268 / 268 / 537 MB per layer) PyTorch runs that as MIOpen BN
// (2 passes), an in-place clamp (read + write), and the pooling read; backward adds threshold_backward and a
// two-pass BN backward over dense gradients.  Here:
//   forward   stats (1 read, fp64 accumulation)  +  apply: y = relu((z - mean) * invstd * gamma + beta) in one
//             read + write -- or, for the last layer, apply + max over the S samples of each group in one read
//             (the normalised tensor is never written);
//   backward  sums (dbeta = sum dy, dgamma = sum dy * xhat)  +  dz = gamma * invstd * (dy - mean(dy) - xhat *
//             mean(dy * xhat)) with dy = dA * [y > 0] formed on the fly; in the pooled variant dy is non-zero at
//             one sample per group, so the sums are taken from the pooled gradient (S times fewer elements) and the
//             dense pass reads only z.
// Numerics: PyTorch's formula (biased variance for normalisation, unbiased for running_var, eps inside the
// sqrt); the statistics are accumulated in fp64, which is at least as accurate as the library's fp32 Welford.
// Everything is deterministic (fixed partial-sum order, no atomics).
#include <math.h>

#include <atomic>

#include "common.hpp"

namespace {
// element index of the float4 group i of channel c: (i / per_b) rows of C * L + (i % per_b) * 4.  i and per_b fit 32 bits on every
// call this repo makes; a 64-bit division (~200 instructions on this chip) per 16 bytes was a third of these kernels' time.
__device__ __forceinline__ size_t group_off(long i, long per_b, int C, int c, long L) {
  long b, l4;
  if (((unsigned long long)i | (unsigned long long)per_b) >> 32) {
    b = i / per_b, l4 = i - b * per_b;
  } else {
    const unsigned q = (unsigned)i / (unsigned)per_b;
    b = q, l4 = (unsigned)i - q * (unsigned)per_b;
  }
  return ((size_t)b * C + c) * L + (size_t)l4 * 4;
}

using f32x4 = float __attribute__((ext_vector_type(4)));
constexpr int STAT_THREADS = 256;

__device__ __forceinline__ double block_sum_f64(double v, double *s_buf) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_buf[wid] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < STAT_THREADS / 64; ++w) r += s_buf[w];
  return r;
}

// ---- forward statistics: partial (sum, sum of squares) of channel c over slice `split` of the B*L elements
__global__ __launch_bounds__(STAT_THREADS) void bn_stats_partial_kernel(const float *__restrict__ z, int B, int C,
                                                                       long L, int nsplit,
                                                                       double *__restrict__ part) {
  __shared__ double s_buf[STAT_THREADS / 64];
  const int c = blockIdx.x, sp = blockIdx.y;
  double s = 0.0, q = 0.0;
  const long per_b = (L + 3) / 4;  // float4 groups per (b, c) row when L % 4 == 0
  if ((L & 3) == 0) {
    const long total = (long)B * per_b;
    for (long i = (long)sp * STAT_THREADS + threadIdx.x; i < total; i += (long)nsplit * STAT_THREADS) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(z + group_off(i, per_b, C, c, L));
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    const long total = (long)B * L;
    for (long i = (long)sp * STAT_THREADS + threadIdx.x; i < total; i += (long)nsplit * STAT_THREADS) {
      const float v = z[((size_t)(i / L) * C + c) * L + i % L];
      s += v;
      q += (double)v * v;
    }
  }
  s = block_sum_f64(s, s_buf);
  q = block_sum_f64(q, s_buf);
  if (threadIdx.x == 0) {
    part[((size_t)c * nsplit + sp) * 2 + 0] = s;
    part[((size_t)c * nsplit + sp) * 2 + 1] = q;
  }
}

__global__ void bn_stats_final_kernel(const double *__restrict__ part, int C, int nsplit, double M, float eps,
                                      float momentum, float *__restrict__ running_mean,
                                      float *__restrict__ running_var, float *__restrict__ stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int p = 0; p < nsplit; ++p) {
    s += part[((size_t)c * nsplit + p) * 2 + 0];
    q += part[((size_t)c * nsplit + p) * 2 + 1];
  }
  const double mean = s / M;
  double var = q / M - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[c * 2 + 0] = (float)mean;
  stats[c * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

// ---- forward apply: out = relu((z - mean) * (invstd * gamma) + beta), rows = (b, c)
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const float *__restrict__ z, const float *__restrict__ stats,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, int C, long L,
                                                            float *__restrict__ out) {
  const int row = blockIdx.y, c = row % C;
  const float mean = stats[c * 2], sc = stats[c * 2 + 1] * gamma[c], sh = beta[c];
  const float *zr = z + (size_t)row * L;
  float *outr = out + (size_t)row * L;
  if ((L & 3) == 0) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < L; i += (long)gridDim.x * 1024) {
      f32x4 v = *reinterpret_cast<const f32x4 *>(zr + i);
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = fmaxf((v[u] - mean) * sc + sh, 0.f);
      *reinterpret_cast<f32x4 *>(outr + i) = v;
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long)gridDim.x * 256)
      outr[i] = fmaxf((zr[i] - mean) * sc + sh, 0.f);
  }
}

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// ---- forward apply + max over the S samples of each (b, c, p) row; S in {16, 32, 64, 128}
template <int S>
__global__ __launch_bounds__(256) void bn_relu_max_kernel(const float *__restrict__ z, const float *__restrict__ stats,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, int C, int P, long rows,
                                                          float *__restrict__ out, uint8_t *__restrict__ arg) {
  constexpr int LPR = S / 4, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane % LPR, rin = lane / LPR;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  for (long r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
    const long row = r0 + rin;
    const bool ok = row < rows;
    const int c = ok ? (int)((row / P) % C) : 0;
    const float mean = stats[c * 2], sc = stats[c * 2 + 1] * gamma[c], sh = beta[c];
    f32x4 v = ok ? *reinterpret_cast<const f32x4 *>(z + row * S + sub * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = fmaxf((v[u] - mean) * sc + sh, 0.f);
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
    if (ok && sub == 0) { out[row] = bv; arg[row] = (uint8_t)bi; }
  }
}

// ---- backward sums, dense upstream gradient: s1 = sum dy, s2 = sum dy * xhat, dy = dA * [y > 0]
__global__ __launch_bounds__(STAT_THREADS) void bn_bwd_partial_kernel(const float *__restrict__ z,
                                                                     const float *__restrict__ dA,
                                                                     const float *__restrict__ stats,
                                                                     const float *__restrict__ gamma,
                                                                     const float *__restrict__ beta, int B, int C,
                                                                     long L, int nsplit, double *__restrict__ part) {
  __shared__ double s_buf[STAT_THREADS / 64];
  const int c = blockIdx.x, sp = blockIdx.y;
  const float mean = stats[c * 2], r = stats[c * 2 + 1], g = gamma[c], bt = beta[c];
  double s1 = 0.0, s2 = 0.0;
  if ((L & 3) == 0) {
    const long per_b = L / 4, total = (long)B * per_b;
    for (long i = (long)sp * STAT_THREADS + threadIdx.x; i < total; i += (long)nsplit * STAT_THREADS) {
      const size_t off = group_off(i, per_b, C, c, L);
      const f32x4 v = *reinterpret_cast<const f32x4 *>(z + off);
      const f32x4 d = *reinterpret_cast<const f32x4 *>(dA + off);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float xh = (v[u] - mean) * r;
        const float dy = (xh * g + bt > 0.f) ? d[u] : 0.f;
        s1 += dy;
        s2 += (double)dy * xh;
      }
    }
  } else {
    const long total = (long)B * L;
    for (long i = (long)sp * STAT_THREADS + threadIdx.x; i < total; i += (long)nsplit * STAT_THREADS) {
      const size_t off = ((size_t)(i / L) * C + c) * L + i % L;
      const float xh = (z[off] - mean) * r;
      const float dy = (xh * g + bt > 0.f) ? dA[off] : 0.f;
      s1 += dy;
      s2 += (double)dy * xh;
    }
  }
  s1 = block_sum_f64(s1, s_buf);
  s2 = block_sum_f64(s2, s_buf);
  if (threadIdx.x == 0) {
    part[((size_t)c * nsplit + sp) * 2 + 0] = s1;
    part[((size_t)c * nsplit + sp) * 2 + 1] = s2;
  }
}

// ---- backward sums, pooled upstream gradient: only sample arg[row] of each row carries gradient
__global__ __launch_bounds__(STAT_THREADS) void bn_bwd_pooled_partial_kernel(
    const float *__restrict__ z, const float *__restrict__ dP, const uint8_t *__restrict__ arg,
    const float *__restrict__ stats, const float *__restrict__ gamma, const float *__restrict__ beta, int B, int C,
    int P, int S, int nsplit, double *__restrict__ part) {
  __shared__ double s_buf[STAT_THREADS / 64];
  const int c = blockIdx.x, sp = blockIdx.y;
  const float mean = stats[c * 2], r = stats[c * 2 + 1], g = gamma[c], bt = beta[c];
  double s1 = 0.0, s2 = 0.0;
  const long total = (long)B * P;
  for (long i = (long)sp * STAT_THREADS + threadIdx.x; i < total; i += (long)nsplit * STAT_THREADS) {
    const size_t row = ((size_t)(i / P) * C + c) * P + i % P;
    const float xh = (z[row * S + arg[row]] - mean) * r;
    const float dy = (xh * g + bt > 0.f) ? dP[row] : 0.f;
    s1 += dy;
    s2 += (double)dy * xh;
  }
  s1 = block_sum_f64(s1, s_buf);
  s2 = block_sum_f64(s2, s_buf);
  if (threadIdx.x == 0) {
    part[((size_t)c * nsplit + sp) * 2 + 0] = s1;
    part[((size_t)c * nsplit + sp) * 2 + 1] = s2;
  }
}

// dgamma = s2, dbeta = s1, coef[c] = (s1 / M, s2 / M)
__global__ void bn_bwd_final_kernel(const double *__restrict__ part, int C, int nsplit, double M,
                                    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int p = 0; p < nsplit; ++p) {
    s1 += part[((size_t)c * nsplit + p) * 2 + 0];
    s2 += part[((size_t)c * nsplit + p) * 2 + 1];
  }
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  coef[c * 2 + 0] = (float)(s1 / M);
  coef[c * 2 + 1] = (float)(s2 / M);
}

// ---- backward dense pass: dz = gamma * invstd * (dy - k1 - xhat * k2)
template <bool POOLED>
__global__ __launch_bounds__(256) void bn_bwd_dz_kernel(const float *__restrict__ z, const float *__restrict__ up,
                                                        const uint8_t *__restrict__ arg,
                                                        const float *__restrict__ stats,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta,
                                                        const float *__restrict__ coef, int C, long L, int S,
                                                        float *__restrict__ dz) {
  const int row = blockIdx.y, c = row % C;  // row = (b, c)
  const float mean = stats[c * 2], r = stats[c * 2 + 1], g = gamma[c], bt = beta[c];
  const float k1 = coef[c * 2], k2 = coef[c * 2 + 1], gr = g * r;
  const float *zr = z + (size_t)row * L;
  float *dzr = dz + (size_t)row * L;
  // POOLED: `up` is dP (B, C, P) and sample arg[...] of each group of S carries it; else `up` is dA (B, C, L)
  if ((L & 3) == 0 && (!POOLED || (S & 3) == 0)) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < L; i += (long)gridDim.x * 1024) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(zr + i);
      f32x4 d;
      if (POOLED) {
        const size_t grp = (size_t)row * (L / S) + i / S;
        const int a = (int)arg[grp] - (int)(i % S);
        const float p = up[grp];
        d = (f32x4){a == 0 ? p : 0.f, a == 1 ? p : 0.f, a == 2 ? p : 0.f, a == 3 ? p : 0.f};
      } else {
        d = *reinterpret_cast<const f32x4 *>(up + (size_t)row * L + i);
      }
      f32x4 o;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float xh = (v[u] - mean) * r;
        const float dy = (xh * g + bt > 0.f) ? d[u] : 0.f;
        o[u] = gr * (dy - k1 - xh * k2);
      }
      *reinterpret_cast<f32x4 *>(dzr + i) = o;
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long)gridDim.x * 256) {
      float d;
      if (POOLED) {
        const size_t grp = (size_t)row * (L / S) + i / S;
        d = ((int)arg[grp] == (int)(i % S)) ? up[grp] : 0.f;
      } else {
        d = up[(size_t)row * L + i];
      }
      const float xh = (zr[i] - mean) * r;
      const float dy = (xh * g + bt > 0.f) ? d : 0.f;
      dzr[i] = gr * (dy - k1 - xh * k2);
    }
  }
}

// ---- small tensors: statistics + apply (forward) and sums + dz (backward) of one channel in ONE launch --------------------
// The three-launch forms above exist to spread a channel's reduction over several workgroups; with <= 32 768 elements per
// channel and >= 64 channels (vote module, proposal head, position nets: 2 048 .. 8 192 elements) one workgroup per channel
// fills the chip by itself and the two extra launches + the round trip through the partial-sum workspace only cost time
// (9 + 9 such layers per step).  Same arithmetic, expression for expression, as the kernels above; the sums are fp64.
__global__ __launch_bounds__(STAT_THREADS) void bn_relu_train_small_kernel(
    const float *__restrict__ z, int B, int C, long L, double M, float eps, float momentum, float *__restrict__ running_mean,
    float *__restrict__ running_var, const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ stats,
    float *__restrict__ out) {
  __shared__ double s_buf[STAT_THREADS / 64];
  const int c = blockIdx.x;
  const long per_b = L / 4, total = (long)B * per_b;
  double s = 0.0, q = 0.0;
  for (long i = threadIdx.x; i < total; i += STAT_THREADS) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(z + group_off(i, per_b, C, c, L));
    s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  s = block_sum_f64(s, s_buf);
  q = block_sum_f64(q, s_buf);
  const double mean_d = s / M;
  double var = q / M - mean_d * mean_d;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mean_d, istd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    stats[c * 2 + 0] = mean;
    stats[c * 2 + 1] = istd;
    if (running_mean) {
      const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean_d);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
  }
  const float sc = istd * gamma[c], sh = beta[c];
  for (long i = threadIdx.x; i < total; i += STAT_THREADS) {
    const size_t off = group_off(i, per_b, C, c, L);
    f32x4 v = *reinterpret_cast<const f32x4 *>(z + off);
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = fmaxf((v[u] - mean) * sc + sh, 0.f);
    *reinterpret_cast<f32x4 *>(out + off) = v;
  }
}

__global__ __launch_bounds__(STAT_THREADS) void bn_relu_bwd_small_kernel(
    const float *__restrict__ z, const float *__restrict__ dA, const float *__restrict__ stats, const float *__restrict__ gamma,
    const float *__restrict__ beta, int B, int C, long L, double M, float *__restrict__ dz, float *__restrict__ dgamma,
    float *__restrict__ dbeta) {
  __shared__ double s_buf[STAT_THREADS / 64];
  const int c = blockIdx.x;
  const float mean = stats[c * 2], r = stats[c * 2 + 1], g = gamma[c], bt = beta[c];
  const long per_b = L / 4, total = (long)B * per_b;
  double s1 = 0.0, s2 = 0.0;
  for (long i = threadIdx.x; i < total; i += STAT_THREADS) {
    const size_t off = group_off(i, per_b, C, c, L);
    const f32x4 v = *reinterpret_cast<const f32x4 *>(z + off);
    const f32x4 d = *reinterpret_cast<const f32x4 *>(dA + off);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float xh = (v[u] - mean) * r;
      const float dy = (xh * g + bt > 0.f) ? d[u] : 0.f;
      s1 += dy;
      s2 += (double)dy * xh;
    }
  }
  s1 = block_sum_f64(s1, s_buf);
  s2 = block_sum_f64(s2, s_buf);
  if (threadIdx.x == 0) dbeta[c] = (float)s1, dgamma[c] = (float)s2;
  const float k1 = (float)(s1 / M), k2 = (float)(s2 / M), gr = g * r;
  for (long i = threadIdx.x; i < total; i += STAT_THREADS) {
    const size_t off = group_off(i, per_b, C, c, L);
    const f32x4 v = *reinterpret_cast<const f32x4 *>(z + off);
    const f32x4 d = *reinterpret_cast<const f32x4 *>(dA + off);
    f32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float xh = (v[u] - mean) * r;
      const float dy = (xh * g + bt > 0.f) ? d[u] : 0.f;
      o[u] = gr * (dy - k1 - xh * k2);
    }
    *reinterpret_cast<f32x4 *>(dz + off) = o;
  }
}

static std::atomic<int> g_single_launch{1};   // spacap_bn_set_single_launch (tests compare the two forms bit for bit)
inline bool bn_small(int B, int C, long L) {
  return g_single_launch.load(std::memory_order_relaxed) != 0 && (L & 3) == 0 && C >= 64 && (double)B * (double)L <= 32768.0;
}

int pick_split(int C, double elems_per_channel) {
  int want = (int)(4096 / (C > 0 ? C : 1));
  if (want < 1) want = 1;
  const int cap = (int)(elems_per_channel / (STAT_THREADS * 8)) + 1;
  if (want > cap) want = cap;
  if (want > 256) want = 256;
  return want;
}

unsigned grid_x(long L) {
  long g = (L + 4095) / 4096;
  if (g < 1) g = 1;
  if (g > 64) g = 64;
  return (unsigned)g;
}

}  // namespace

// workspace: fp64 partials [C][256][2] + fp32 coefficients [C][2]
extern "C" size_t spacap_bn_workspace_bytes(int C) {
  if (C <= 0) return 0;
  return (size_t)C * 256 * 2 * sizeof(double) + (size_t)C * 2 * sizeof(float) + 64;
}

extern "C" int spacap_bn_stats_f32(const float *z, int B, int C, long L, float eps, float momentum,
                                   float *running_mean, float *running_var, float *stats, void *workspace,
                                   spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 1 && C >= 1 && L >= 1, "spacap_bn_stats_f32: bad sizes B=%d C=%d L=%ld", B, C, L);
  SPACAP_REQUIRE(z && stats && workspace, "spacap_bn_stats_f32: null pointer");
  SPACAP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "spacap_bn_stats_f32: running stats");
  SPACAP_REQUIRE(C <= 65535, "spacap_bn_stats_f32: C out of range");
  hipStream_t s = spacap::as_stream(stream);
  double *part = reinterpret_cast<double *>(workspace);
  const double M = (double)B * (double)L;
  const int nsplit = pick_split(C, M);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, nsplit), dim3(STAT_THREADS), 0, s, z, B, C, L, nsplit, part);
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + 63) / 64), dim3(64), 0, s, part, C, nsplit, M, eps, momentum,
                     running_mean, running_var, stats);
  SPACAP_CHECK_LAUNCH("spacap_bn_stats_f32");
  return SPACAP_OK;
}

/* statistics + apply in one call (one launch for small tensors, else spacap_bn_stats_f32 + spacap_bn_relu_apply_f32) */
extern "C" int spacap_bn_relu_train_f32(const float *z, int B, int C, long L, float eps, float momentum, float *running_mean,
                                        float *running_var, const float *gamma, const float *beta, float *stats, float *out,
                                        void *workspace, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 1 && C >= 1 && L >= 1 && (long)B * C <= 65535, "spacap_bn_relu_train_f32: bad sizes B=%d C=%d L=%ld", B, C, L);
  SPACAP_REQUIRE(z && gamma && beta && stats && out && workspace, "spacap_bn_relu_train_f32: null pointer");
  SPACAP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "spacap_bn_relu_train_f32: running stats");
  if (bn_small(B, C, L)) {
    hipLaunchKernelGGL(bn_relu_train_small_kernel, dim3(C), dim3(STAT_THREADS), 0, spacap::as_stream(stream), z, B, C, L,
                       (double)B * (double)L, eps, momentum, running_mean, running_var, gamma, beta, stats, out);
    SPACAP_CHECK_LAUNCH("spacap_bn_relu_train_f32");
    return SPACAP_OK;
  }
  const int rc = spacap_bn_stats_f32(z, B, C, L, eps, momentum, running_mean, running_var, stats, workspace, stream);
  if (rc != SPACAP_OK) return rc;
  return spacap_bn_relu_apply_f32(z, stats, gamma, beta, B, C, L, out, stream);
}

extern "C" int spacap_bn_relu_apply_f32(const float *z, const float *stats, const float *gamma, const float *beta,
                                        int B, int C, long L, float *out, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 1 && C >= 1 && L >= 1 && (long)B * C <= 65535, "spacap_bn_relu_apply_f32: bad sizes");
  SPACAP_REQUIRE(z && stats && gamma && beta && out, "spacap_bn_relu_apply_f32: null pointer");
  hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(grid_x(L), B * C), dim3(256), 0, spacap::as_stream(stream), z, stats,
                     gamma, beta, C, L, out);
  SPACAP_CHECK_LAUNCH("spacap_bn_relu_apply_f32");
  return SPACAP_OK;
}

extern "C" int spacap_bn_relu_max_f32(const float *z, const float *stats, const float *gamma, const float *beta,
                                      int B, int C, int P, int S, float *out, uint8_t *arg,
                                      spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 1 && C >= 1 && P >= 1, "spacap_bn_relu_max_f32: bad sizes");
  SPACAP_REQUIRE(S == 16 || S == 32 || S == 64 || S == 128, "spacap_bn_relu_max_f32: S=%d unsupported", S);
  SPACAP_REQUIRE(z && stats && gamma && beta && out && arg, "spacap_bn_relu_max_f32: null pointer");
  const long rows = (long)B * C * P;
  hipStream_t s = spacap::as_stream(stream);
  long g = (rows * (S / 4) + 1023) / 1024 / 4;
  if (g < 1) g = 1;
  if (g > 65535 * 8) g = 65535 * 8;
#define BRM_CASE(SV) \
  if (S == SV) hipLaunchKernelGGL((bn_relu_max_kernel<SV>), dim3((unsigned)g), dim3(256), 0, s, z, stats, gamma, beta, C, P, rows, out, arg);
  BRM_CASE(16) BRM_CASE(32) BRM_CASE(64) BRM_CASE(128)
#undef BRM_CASE
  SPACAP_CHECK_LAUNCH("spacap_bn_relu_max_f32");
  return SPACAP_OK;
}

extern "C" int spacap_bn_relu_bwd_f32(const float *z, const float *stats, const float *gamma, const float *beta,
                                      const float *dA, int B, int C, long L, float *dz, float *dgamma, float *dbeta,
                                      void *workspace, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 1 && C >= 1 && L >= 1 && (long)B * C <= 65535, "spacap_bn_relu_bwd_f32: bad sizes");
  SPACAP_REQUIRE(z && stats && gamma && beta && dA && dz && dgamma && dbeta && workspace,
                 "spacap_bn_relu_bwd_f32: null pointer");
  hipStream_t s = spacap::as_stream(stream);
  if (bn_small(B, C, L)) {
    hipLaunchKernelGGL(bn_relu_bwd_small_kernel, dim3(C), dim3(STAT_THREADS), 0, s, z, dA, stats, gamma, beta, B, C, L,
                       (double)B * (double)L, dz, dgamma, dbeta);
    SPACAP_CHECK_LAUNCH("spacap_bn_relu_bwd_f32");
    return SPACAP_OK;
  }
  double *part = reinterpret_cast<double *>(workspace);
  float *coef = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + (size_t)C * 256 * 2 * sizeof(double));
  const double M = (double)B * (double)L;
  const int nsplit = pick_split(C, M);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(C, nsplit), dim3(STAT_THREADS), 0, s, z, dA, stats, gamma, beta, B, C,
                     L, nsplit, part);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 63) / 64), dim3(64), 0, s, part, C, nsplit, M, dgamma, dbeta, coef);
  hipLaunchKernelGGL((bn_bwd_dz_kernel<false>), dim3(grid_x(L), B * C), dim3(256), 0, s, z, dA, (const uint8_t *)nullptr,
                     stats, gamma, beta, coef, C, L, 1, dz);
  SPACAP_CHECK_LAUNCH("spacap_bn_relu_bwd_f32");
  return SPACAP_OK;
}

extern "C" int spacap_bn_relu_max_bwd_f32(const float *z, const float *stats, const float *gamma, const float *beta,
                                          const float *dP, const uint8_t *arg, int B, int C, int P, int S, float *dz,
                                          float *dgamma, float *dbeta, void *workspace, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 1 && C >= 1 && P >= 1 && S >= 1 && S <= 256 && (long)B * C <= 65535,
                 "spacap_bn_relu_max_bwd_f32: bad sizes");
  SPACAP_REQUIRE(z && stats && gamma && beta && dP && arg && dz && dgamma && dbeta && workspace,
                 "spacap_bn_relu_max_bwd_f32: null pointer");
  hipStream_t s = spacap::as_stream(stream);
  double *part = reinterpret_cast<double *>(workspace);
  float *coef = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + (size_t)C * 256 * 2 * sizeof(double));
  const long L = (long)P * S;
  const double M = (double)B * (double)L;
  const int nsplit = pick_split(C, (double)B * P);
  hipLaunchKernelGGL(bn_bwd_pooled_partial_kernel, dim3(C, nsplit), dim3(STAT_THREADS), 0, s, z, dP, arg, stats, gamma,
                     beta, B, C, P, S, nsplit, part);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 63) / 64), dim3(64), 0, s, part, C, nsplit, M, dgamma, dbeta, coef);
  hipLaunchKernelGGL((bn_bwd_dz_kernel<true>), dim3(grid_x(L), B * C), dim3(256), 0, s, z, dP, arg, stats, gamma, beta,
                     coef, C, L, S, dz);
  SPACAP_CHECK_LAUNCH("spacap_bn_relu_max_bwd_f32");
  return SPACAP_OK;
}

/* 1 (default): tensors with <= 32 768 elements per channel take the one-launch kernels; 0: always the partial / final / apply
   form (same fp32 expressions and fp64 sums: tests compare the two bit for bit). */
extern "C" int spacap_bn_set_single_launch(int enabled) {
  g_single_launch.store(enabled ? 1 : 0, std::memory_order_relaxed);
  return SPACAP_OK;
}
