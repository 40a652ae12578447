// Shared MLP of a set-abstraction module in point-major ("rows") layout, training mode, for gfx950 (MI355X).
//
// Replaces, for the training step, the chain the reference runs per SA module
// (lib/pointnet2/pointnet2_modules.py:241-259: QueryAndGroup -> SharedMLP -> max_pool2d, with SharedMLP =
// [Conv2d 1x1 -> BatchNorm2d -> ReLU] x 3, lib/pointnet2/pytorch_utils.py:11-36) on (B, C, npoint, nsample)
// tensors.  Same mathematics (fp32, batch statistics over all B*npoint*nsample positions, biased variance,
// eps inside the square root, first-maximum pooling), different data flow:
//
//   * rows r = (b, centre, sample) are the slow index, channels the fast one: every pass over an activation is one
//     contiguous stream, the 1x1 convolutions are row-major GEMMs on the matrix cores (v_mfma_f32_16x16x4_f32,
//     fp32 in / fp32 accumulate), and no NCHW <-> NHWC transposes exist;
//   * the first layer commutes with the grouping gather:  W1 [rel_xyz ; f(idx)] = Wx rel_xyz + (Wf f)(idx), so
//     Wf f is computed once per SOURCE point (8..16x fewer rows than grouped positions) and layer 1 is a gather
//     of that product plus a 3-term update (sa_l1_fwd_kernel);
//   * only the pre-activation z_k of every layer is stored; BatchNorm + ReLU of layer k are applied while the
//     tile is staged into LDS for layer k+1 (and again in the backward), and each layer's batch statistics are
//     accumulated in the epilogue of the GEMM that produces it -- one write and one read per activation in the
//     forward pass instead of write + 3 reads + write;
//   * the backward of max-pool -> ReLU -> BN is evaluated on the fly from (masked pooled gradient, argmax, z_k)
//     while staging the GEMM operands: dz_k = g dy + k0 - k1 z_k with per-channel constants (sa_bwd_finalize).
//
// Weights stay in registers for the whole kernel (each wave owns 16*NT output channels); the activation tile
// (64 rows) goes through LDS with a 4-word row padding, which makes the MFMA operand reads conflict-free.
// All reductions (statistics, weight gradients) are two-stage with a fixed order: no float atomics.
#include <hipcub/hipcub.hpp>

#include <type_traits>

#include "common.hpp"
#include <atomic>

namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int TM = 64;      // rows per tile of the forward / data-gradient GEMMs
constexpr int TW = 32;      // rows per tile of the weight-gradient GEMM
constexpr int NPART = 1024; // partial-sum rows (= persistent workgroups) of every statistics reduction

// stats row of a layer: {mean, 1/sqrt(var+eps), gamma/sqrt(var+eps), beta}
// coef  row of a layer (backward): {g, k0, k1, -}:  dz = g*dy + k0 - k1*z

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

// r / d for row indices: a 64-bit division is ~200 instructions on this chip and the first-layer passes paid two of them per
// row and thread; row counts fit 32 bits on the model's path (one 32-bit division, ~30 instructions), the general case stays
__device__ __forceinline__ long row_div(long r, long d) {
  if (((unsigned long long)r | (unsigned long long)d) >> 32) return r / d;
  return (long)((unsigned)r / (unsigned)d);
}

// ---- layer 1 forward --------------------------------------------------------------------------------------
// z1[r, :] = Y[b, idx[r], :] + W1[:, 0:3] rel(r) + W1[:, 3] feat[b, idx[r]]      (Y and feat optional)
// rel(r) = (xyz[b, idx[r]] - new_xyz[b, n]) / rdiv
// First-layer pre-activation of one grouped row for four channels: z = wx rx + wy ry + wz rz (+ wf f).  ONE definition for the
// statistics pass and for every pass that rebuilds z1 from the row's four inputs instead of reading it back (L1In): the values
// must agree bit for bit.  This file is compiled with -ffp-contract=fast, so the rounding is whatever fused form the compiler
// picks for this expression; it is kept as the plain vector expression the original first-layer kernel had (the values the
// golden fixtures were recorded against -- spelling it out as separately rounded operations, or as an explicit fma chain,
// both changed them), and tests/test_sa_mlp_gpu.py::test_first_layer_rebuilt_instead_of_stored fails if any of the kernels
// that inline it should ever contract it differently.
__device__ __forceinline__ f32x4 l1_row(f32x4 wx, f32x4 wy, f32x4 wz, f32x4 wf, f32x4 in, bool has_feat) {
  f32x4 z = wx * in[0] + wy * in[1] + wz * in[2];
  if (has_feat) z += wf * in[3];
  return z;
}
// z_prev = the first layer's pre-activation rebuilt from rel4 [R][4] (relative x, y, z, inline feature) and W1 [C][ldw]
struct L1In {
  const float *W1;
  int ldw, has_feat;
};
__device__ __forceinline__ void l1_weights(const L1In &li, int c0, f32x4 &wx, f32x4 &wy, f32x4 &wz, f32x4 &wf) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float *w = li.W1 + (size_t)(c0 + u) * li.ldw;
    wx[u] = w[0], wy[u] = w[1], wz[u] = w[2];
    wf[u] = li.has_feat ? w[3] : 0.f;
  }
}

template <int C1, int UR = 4>
__global__ __launch_bounds__(256) void sa_l1_fwd_kernel(const float *__restrict__ Y, const float *__restrict__ feat,
                                                        const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                        const int32_t *__restrict__ idx, const float *__restrict__ W1,
                                                        int ldw, float rdiv, int Np, int N, int S, long R,
                                                        float *__restrict__ z1, double *__restrict__ part,
                                                        float *__restrict__ rel4 = nullptr) {
  constexpr int C4 = C1 / 4, RP = 256 / C4;
  __shared__ float s_red[2][RP][C1];
  const int tid = threadIdx.x, c4 = tid % C4, rs = tid / C4;
  f32x4 wx, wy, wz, wf = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float *w = W1 + (size_t)(c4 * 4 + u) * ldw;
    wx[u] = w[0], wy[u] = w[1], wz[u] = w[2];
    if (feat) wf[u] = w[3];
  }
  f32x4 sum = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
  const long NS = (long)N * S;
  // The row's inputs hang off a dependent chain (idx -> point -> coordinates): four rows per iteration keep four chains in
  // flight (one row at a time the kernel sat at the chain's latency, 119 us for SA1 against an HBM time of 55 us; 8 or 16 rows
  // measured slower at every SA shape: 45.7 -> 65.6 us at SA2, tools/lab/l1_fwd_time.py).  The
  // statistics are accumulated in the same row order as before.
  const long G = (long)gridDim.x * RP;
  long r = (long)blockIdx.x * RP + rs;
  auto row_in = [&](long rr, int &p, long &b) {
    b = row_div(rr, NS);
    p = idx[rr];
  };
  auto row_z = [&](long rr, int p, long b) {
    const long g = row_div(rr, S);
    const float *q = xyz + ((size_t)b * Np + p) * 3, *c = new_xyz + (size_t)g * 3;
    const f32x4 in = {(q[0] - c[0]) / rdiv, (q[1] - c[1]) / rdiv, (q[2] - c[2]) / rdiv, feat ? feat[(size_t)b * Np + p] : 0.f};
    if (rel4 && c4 == 0) st4(rel4 + (size_t)rr * 4, in);   // the row's four inputs: what the later passes rebuild z1 from
    f32x4 z = l1_row(wx, wy, wz, wf, in, feat != nullptr);
    if (Y) z += ld4(Y + ((size_t)b * Np + p) * C1 + c4 * 4);
    return z;
  };
  for (; r + (UR - 1) * G < R; r += UR * G) {
    int p[UR];
    long b[UR];
#pragma unroll
    for (int k = 0; k < UR; ++k) row_in(r + k * G, p[k], b[k]);
    f32x4 z[UR];
#pragma unroll
    for (int k = 0; k < UR; ++k) z[k] = row_z(r + k * G, p[k], b[k]);
#pragma unroll
    for (int k = 0; k < UR; ++k) {
      if (z1) st4(z1 + (size_t)(r + k * G) * C1 + c4 * 4, z[k]);
      sum += z[k];
      sq += z[k] * z[k];
    }
  }
  for (; r < R; r += G) {
    int p;
    long b;
    row_in(r, p, b);
    const f32x4 z = row_z(r, p, b);
    if (z1) st4(z1 + (size_t)r * C1 + c4 * 4, z);
    sum += z;
    sq += z * z;
  }
  st4(&s_red[0][rs][c4 * 4], sum);
  st4(&s_red[1][rs][c4 * 4], sq);
  __syncthreads();
  if (tid < 2 * C1) {
    const int k = tid / C1, c = tid % C1;
    double a = 0.0;
    for (int i = 0; i < RP; ++i) a += (double)s_red[k][i][c];
    part[((size_t)blockIdx.x * 2 + k) * C1 + c] = a;
  }
}

// ---- first layer without point features (SA1): BatchNorm statistics in closed form -------------------------------------------
// z1[r][c] = W1[c] . in[r] is LINEAR in the row's four inputs in = (rel x, rel y, rel z, inline feature), so the layer's batch
// statistics need only the first and second moments of `in` over all rows:
//     sum_r z1[r][c] = W1[c] . S,   sum_r z1[r][c]^2 = W1[c]^T Mom W1[c],   S = sum_r in[r] (4),  Mom = sum_r in[r] in[r]^T (10 distinct)
// 14 sums per row instead of 2 x 64 -- and one THREAD per row instead of 16 (the 64-channel form repeated the row's dependent
// chain idx -> point -> coordinates and its three divisions in the 16 threads that shared a row: 87 us at SA1).  The pass still
// leaves rel4 [R][4], from which every later pass rebuilds z1.  Reference: lib/pointnet2/pointnet2_utils.py:350-355 (grouping),
// lib/pointnet2/pytorch_utils.py:11-36 (Conv2d -> BatchNorm2d: statistics over all rows).
// part [gridDim.x][16] doubles: S0..S3 | M00 M01 M02 M03 M11 M12 M13 M22 M23 M33 | 0 0
__global__ __launch_bounds__(256) void sa_l1_moments_kernel(const float *__restrict__ feat, const float *__restrict__ xyz,
                                                            const float *__restrict__ new_xyz, const int32_t *__restrict__ idx,
                                                            float rdiv, int Np, int N, int S, long R, float *__restrict__ rel4,
                                                            double *__restrict__ part) {
  __shared__ float s_m[14][256 + 1];
  const int tid = threadIdx.x;
  const long NS = (long)N * S, G = (long)gridDim.x * 256;
  float a[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) a[i] = 0.f;
  auto row = [&](long r, int p) {
    const long b = row_div(r, NS), g = row_div(r, S);
    const float *q = xyz + ((size_t)b * Np + p) * 3, *c = new_xyz + (size_t)g * 3;
    const f32x4 in = {(q[0] - c[0]) / rdiv, (q[1] - c[1]) / rdiv, (q[2] - c[2]) / rdiv, feat ? feat[(size_t)b * Np + p] : 0.f};
    st4(rel4 + (size_t)r * 4, in);
    return in;
  };
  auto add = [&](f32x4 in) {
    a[0] += in[0], a[1] += in[1], a[2] += in[2], a[3] += in[3];
    a[4] += in[0] * in[0], a[5] += in[0] * in[1], a[6] += in[0] * in[2], a[7] += in[0] * in[3];
    a[8] += in[1] * in[1], a[9] += in[1] * in[2], a[10] += in[1] * in[3];
    a[11] += in[2] * in[2], a[12] += in[2] * in[3], a[13] += in[3] * in[3];
  };
  constexpr int UR = 4;   // four rows' chains (idx -> point -> coordinates) in flight per thread
  long r = (long)blockIdx.x * 256 + tid;
  for (; r + (UR - 1) * G < R; r += UR * G) {
    int p[UR];
#pragma unroll
    for (int k = 0; k < UR; ++k) p[k] = idx[r + k * G];
    f32x4 in[UR];
#pragma unroll
    for (int k = 0; k < UR; ++k) in[k] = row(r + k * G, p[k]);
#pragma unroll
    for (int k = 0; k < UR; ++k) add(in[k]);
  }
  for (; r < R; r += G) add(row(r, idx[r]));
#pragma unroll
  for (int i = 0; i < 14; ++i) s_m[i][tid] = a[i];
  __syncthreads();
  if (tid < 16) {
    double v = 0.0;
    if (tid < 14)
      for (int t = 0; t < 256; ++t) v += (double)s_m[tid][t];
    part[(size_t)blockIdx.x * 16 + tid] = v;
  }
}

// moments [nparts][16] -> stats [C1][4] of the first layer (mean, 1 / std, gamma / std, beta) + running statistics (torch semantics)
__global__ __launch_bounds__(1024) void sa_l1_moments_finalize_kernel(const double *__restrict__ part, int nparts, const float *__restrict__ W1,
                                                                     int ldw, int has_feat, int C1, double M, float eps, float momentum,
                                                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                     float *__restrict__ running_mean, float *__restrict__ running_var,
                                                                     float *__restrict__ stats) {
  __shared__ double s_mom[16];
  const int tid = threadIdx.x, m = tid >> 6, lane = tid & 63;
  if (m < 14) {   // wave m adds moment m over the partial rows: lane l takes rows l, l + 64, .. in order, then a fixed tree
    double v = 0.0;
    for (int p = lane; p < nparts; p += 64) v += part[(size_t)p * 16 + m];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) s_mom[m] = v;
  }
  __syncthreads();
  if (tid < C1) {
    const float *wr = W1 + (size_t)tid * ldw;
    const double w[4] = {wr[0], wr[1], wr[2], has_feat ? wr[3] : 0.0};
    const double *Sm = s_mom, *Q = s_mom + 4;   // Q: 00 01 02 03 11 12 13 22 23 33
    const double sm = w[0] * Sm[0] + w[1] * Sm[1] + w[2] * Sm[2] + w[3] * Sm[3];
    const double q = w[0] * w[0] * Q[0] + w[1] * w[1] * Q[4] + w[2] * w[2] * Q[7] + w[3] * w[3] * Q[9] +
                     2.0 * (w[0] * w[1] * Q[1] + w[0] * w[2] * Q[2] + w[0] * w[3] * Q[3] + w[1] * w[2] * Q[5] + w[1] * w[3] * Q[6] +
                            w[2] * w[3] * Q[8]);
    const double mean = sm / M;
    double var = q / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    stats[tid * 4 + 0] = (float)mean;
    stats[tid * 4 + 1] = istd;
    stats[tid * 4 + 2] = gamma[tid] * istd;
    stats[tid * 4 + 3] = beta[tid];
    if (running_mean) {
      const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
      running_mean[tid] = (float)((1.0 - momentum) * running_mean[tid] + momentum * mean);
      running_var[tid] = (float)((1.0 - momentum) * running_var[tid] + momentum * unbiased);
    }
  }
}

// ---- statistics finalisation --------------------------------------------------------------------------------
// part [NPART][2][C] (sum, sum of squares) -> stats [C][4]; optional running-statistics update (torch semantics)
__global__ __launch_bounds__(1024) void sa_bn_finalize_kernel(const double *__restrict__ part, int nparts, int C,
                                                             double M, float eps, float momentum,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float *__restrict__ running_mean,
                                                             float *__restrict__ running_var, float *__restrict__ stats) {
  // workgroup = 8 channels x {sum, sq} x 64 slabs of partial rows
  __shared__ double s[64][16];
  const int tid = threadIdx.x, col = tid & 15, slab = tid >> 4;
  const int k = col >> 3, c = blockIdx.x * 8 + (col & 7);
  double a = 0.0;
  if (c < C) {
#pragma unroll 16
    for (int p = slab; p < nparts; p += 64) a += part[((size_t)p * 2 + k) * C + c];
  }
  s[slab][col] = a;
  __syncthreads();
  if (tid < 8 && c < C) {
    double sm = 0.0, q = 0.0;
    for (int i = 0; i < 64; ++i) sm += s[i][tid], q += s[i][tid + 8];
    const double mean = sm / M;
    double var = q / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    stats[c * 4 + 0] = (float)mean;
    stats[c * 4 + 1] = istd;
    stats[c * 4 + 2] = gamma[c] * istd;
    stats[c * 4 + 3] = beta[c];
    if (running_mean) {
      const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
  }
}

// part [NPART][2][C] (sum dy, sum dy*xhat) -> coef [C][4], dgamma, dbeta
__global__ __launch_bounds__(1024) void sa_bwd_finalize_kernel(const double *__restrict__ part, int nparts, int C,
                                                              double M, const float *__restrict__ stats,
                                                              float *__restrict__ coef, float *__restrict__ dgamma,
                                                              float *__restrict__ dbeta) {
  __shared__ double s[64][16];
  const int tid = threadIdx.x, col = tid & 15, slab = tid >> 4;
  const int k = col >> 3, c = blockIdx.x * 8 + (col & 7);
  double a = 0.0;
  if (c < C) {
#pragma unroll 16
    for (int p = slab; p < nparts; p += 64) a += part[((size_t)p * 2 + k) * C + c];
  }
  s[slab][col] = a;
  __syncthreads();
  if (tid < 8 && c < C) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < 64; ++i) s1 += s[i][tid], s2 += s[i][tid + 8];
    const float mean = stats[c * 4], istd = stats[c * 4 + 1], g = stats[c * 4 + 2];
    const float a1 = (float)(s1 / M), b1 = (float)(s2 / M);
    const float k1 = g * b1 * istd;
    coef[c * 4 + 0] = g;
    coef[c * 4 + 1] = k1 * mean - g * a1;
    coef[c * 4 + 2] = k1;
    coef[c * 4 + 3] = 0.f;
    dgamma[c] = (float)s2;
    dbeta[c] = (float)s1;
  }
}

// ---- middle layers forward: zout = relu(bn(zin)) W^T, statistics of zout -------------------------------------
// Per 64-row tile: [prefetched registers -> BN+ReLU -> LDS] | sync | issue the next tile's loads | MFMA |
// accumulators -> LDS (transposed staging) | sync | full-row 16-byte stores.  The loads of tile t+1 and the stores
// of tile t are in flight while the matrix cores work on tile t.
// TAIL = true turns the same pipeline into the relation head's last two layers
// (models/transformer_captioner.py:319-326, 392-397: Linear(128,128) -> ReLU -> Linear(128,9) on B*K*K pair rows):
// rows are staged as they are (the first layer's ReLU output), the epilogue adds the bias and applies the ReLU, the
// tile is stored (the backward needs it) and multiplied by the 9 x 128 output weights (padded to one 16-row MFMA
// operand) while it is still in LDS.  No statistics.
struct TailArgs {
  const float *bias;   // [Cout]
  const float *W3;     // [NO3][Cout]
  const float *b3;     // [NO3]
  float *pred;         // [R][NO3]
  int NO3;             // <= 16
};
template <int CIN, int NT, bool TAIL = false, int TMT = 64>
__global__ __launch_bounds__(256) void sa_mid_fwd_kernel(const float *__restrict__ zin, const float *__restrict__ st_in,
                                                         const float *__restrict__ W, int Cout, long R,
                                                         float *__restrict__ zout, double *__restrict__ part, TailArgs ta,
                                                         L1In li = L1In{nullptr, 0, 0}) {
  // LD = CIN + 8 (== 8 mod 64 words) together with the K order below makes every ds_read_b128 of the B operand
  // conflict-free: MFMA step s of lane group lg (= lane / 16) uses channel kperm(s, lg); the four steps 4q..4q+3 of
  // a lane are 4 consecutive words, lg 0/1 (and 2/3) interleave in 4-word chunks, lg 0,1 own the first half of the
  // row and lg 2,3 the second (ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...:
  // each holds all 16 rows with lg in {0,1} or {2,3}, and 8*row + 4*(lg&1) tiles the 64 banks exactly once).
  constexpr int LD = CIN + 8, KS = CIN / 4, KQ = KS / 4, C4 = CIN / 4, NV = TMT * C4 / 256, RSTEP = 256 / C4;
  constexpr int COB = 64 * NT, LDO = COB + 4, O4 = COB / 4, NO = TMT * O4 / 256, OSTEP = 256 / O4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_a = smem;             // [TMT][LD]   activations (MFMA B operand)
  float *s_o = smem + TMT * LD;   // [TMT][LDO]  output tile, row-major
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int cbb = blockIdx.y * COB, wc = w * 16 * NT, cb = cbb + wc;
  float wf[NT][KS];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int q = 0; q < KQ; ++q) {   // steps 4q .. 4q+3 of a lane are 4 consecutive channels: one 16-byte load
      const f32x4 w4 = ld4(W + (size_t)(cb + 16 * j + l15) * CIN + (lg >> 1) * (CIN / 2) + q * 8 + (lg & 1) * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) wf[j][q * 4 + u] = w4[u];
    }
  const int c4 = tid % C4, r0 = tid / C4, o4 = tid % O4, or0 = tid / O4;
  f32x4 mean = {0.f, 0.f, 0.f, 0.f}, sc = mean, be = mean;
  if (!TAIL) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float *s = st_in + (size_t)(c4 * 4 + u) * 4;
      mean[u] = s[0], sc[u] = s[2], be[u] = s[3];
    }
  }
  float wf3[TAIL ? COB / 4 : 1];
  f32x4 bv[NT];
  if (TAIL) {
#pragma unroll
    for (int ks = 0; ks < COB / 4; ++ks) wf3[TAIL ? ks : 0] = l15 < ta.NO3 ? ta.W3[(size_t)l15 * Cout + ks * 4 + lg] : 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = ld4(ta.bias + cb + 16 * j + 4 * lg);
  }
  f32x4 ssum[NT], ssq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) ssum[j] = ssq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long ntiles = (R + TMT - 1) / TMT;
  f32x4 pre[NV];
  // Prefetch with hand-issued loads: the compiler's wait-count insertion would drain the previous tile's stores
  // too (vmcnt is one in-order counter for loads and stores on gfx9); here the wait before staging is
  // vmcnt(#stores of one tile), which leaves those stores in flight.  Row index clamped: rows past the end are
  // zeroed when staged.
  auto fetch = [&](long t) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      long grow = t * TMT + r0 + i * RSTEP;
      grow = grow < R ? grow : R - 1;
      const float *src = li.W1 ? zin + (size_t)grow * 4 : zin + (size_t)grow * CIN + c4 * 4;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(pre[i]) : "v"(src) : "memory");
    }
  };
  f32x4 l1x = {0.f, 0.f, 0.f, 0.f}, l1y = l1x, l1z = l1x, l1f = l1x;
  if (li.W1) l1_weights(li, c4 * 4, l1x, l1y, l1z, l1f);
  auto wait_prefetch = [&](bool stores_pending) {
    if (stores_pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NO) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(pre[i]));  // uses of pre[] stay below the wait
  };
  // one tile; FULL = all TMT rows exist (no bounds checks: the stores are then straight-line code too)
  auto tile = [&](long t, auto full, bool stores_pending) {
    constexpr bool FULL = decltype(full)::value;
    const long row0 = t * TMT;
    {
      wait_prefetch(stores_pending);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int row = r0 + i * RSTEP;
        f32x4 a;
        const f32x4 zv = li.W1 ? l1_row(l1x, l1y, l1z, l1f, pre[i], li.has_feat != 0) : pre[i];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = TAIL ? zv[u] : fmaxf((zv[u] - mean[u]) * sc[u] + be[u], 0.f);
        if (!FULL && row0 + row >= R) a = f32x4{0.f, 0.f, 0.f, 0.f};
        st4(&s_a[row * LD + c4 * 4], a);
      }
      __syncthreads();
      if (FULL) fetch(t + gridDim.x);  // (the ragged tile is the last one: nothing to prefetch)
    }
    f32x4 acc[TMT / 16][NT];
#pragma unroll
    for (int mt = 0; mt < TMT / 16; ++mt)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // B operand by ds_read_b128, software-pipelined in chunks of (16 rows x half of K): two register buffers; the
      // reads of chunk c + 2 are issued right after the MFMAs of chunk c, i.e. one chunk of MFMA time (16 * NT
      // instructions) before their first use.  sched_barriers pin that order (the scheduler otherwise sinks every
      // read to just before its use, exposing the LDS latency once per chunk).
      constexpr int KH = KQ / 2 > 0 ? KQ / 2 : 1, NCH = (TMT / 16) * (KQ / KH);
      const float *bsrc = s_a + l15 * LD + (lg >> 1) * (CIN / 2) + (lg & 1) * 4;
      f32x4 bq[2][KH];
      auto bload = [&](int c) {
        const int mt = c / (KQ / KH), h = c % (KQ / KH);
#pragma unroll
        for (int q = 0; q < KH; ++q) bq[c & 1][q] = ld4(bsrc + mt * 16 * LD + (h * KH + q) * 8);
      };
      bload(0);
      if (NCH > 1) bload(1);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int mt = c / (KQ / KH), h = c % (KQ / KH);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < KH; ++q)
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[mt][j] = MFMA16(wf[j][(h * KH + q) * 4 + u], bq[c & 1][q][u], acc[mt][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < NCH) bload(c + 2);
      }
    }
#pragma unroll
    for (int mt = 0; mt < TMT / 16; ++mt)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (TAIL) {
          f32x4 v = acc[mt][j] + bv[j];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
          st4(&s_o[(mt * 16 + l15) * LDO + wc + 16 * j + 4 * lg], v);
        } else {
          st4(&s_o[(mt * 16 + l15) * LDO + wc + 16 * j + 4 * lg], acc[mt][j]);
          if (FULL || row0 + mt * 16 + l15 < R) {
            ssum[j] += acc[mt][j];
            ssq[j] += acc[mt][j] * acc[mt][j];
          }
        }
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NO; ++i) {
      const int row = or0 + i * OSTEP;
      if (FULL || row0 + row < R) st4(zout + (size_t)(row0 + row) * Cout + cbb + o4 * 4, ld4(&s_o[row * LDO + o4 * 4]));
    }
    if (TAIL) {   // wave w: rows 16 w .. 16 w + 15 of the tile times the padded output weights
      f32x4 a3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < COB / 4; ++ks) a3 = MFMA16(wf3[TAIL ? ks : 0], s_o[(w * 16 + l15) * LDO + ks * 4 + lg], a3);
      const long row = row0 + w * 16 + l15;
      if (FULL || row < R) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (4 * lg + u < ta.NO3) ta.pred[(size_t)row * ta.NO3 + 4 * lg + u] = a3[u] + ta.b3[4 * lg + u];
      }
    }
  };
  const long nfull = R / TMT;
  bool pending = false;
  if ((long)blockIdx.x < nfull) fetch(blockIdx.x);
  for (long t = blockIdx.x; t < nfull; t += gridDim.x) {
    tile(t, std::true_type{}, pending);
    pending = true;
  }
  // the last prefetch is unused, but its destination registers must stay reserved until it has landed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NV; ++i) asm volatile("" ::"v"(pre[i]));
  if (nfull < ntiles && (long)blockIdx.x == nfull % gridDim.x) {  // ragged last tile
    fetch(nfull);
    tile(nfull, std::false_type{}, false);
  }
  if (TAIL) return;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float a = ssum[j][u], q = ssq[j][u];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) a += __shfl_xor(a, o), q += __shfl_xor(q, o);
      if (l15 == 0) {
        const int c = cb + 16 * j + 4 * lg + u;
        part[((size_t)blockIdx.x * 2 + 0) * Cout + c] = (double)a;
        part[((size_t)blockIdx.x * 2 + 1) * Cout + c] = (double)q;
        for (int pr = blockIdx.x + gridDim.x; pr < NPART; pr += gridDim.x)  // partial rows without a workgroup
          part[((size_t)pr * 2 + 0) * Cout + c] = 0.0, part[((size_t)pr * 2 + 1) * Cout + c] = 0.0;
      }
    }
}

// ---- the same layer on the bf16 matrix cores with fp32-equivalent accuracy ("bf16 x 3") ---------------------------------
// fp32 MFMA runs at 1/16 of the bf16 MFMA rate on gfx950, which makes these 128-wide layers matrix-core bound.  Every fp32
// operand is split exactly into three bf16 pieces, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2):
// 3 x 8 = 24 significant bits), and the product a*w is evaluated as the six bf16 products with weight >= 2^-16
//     a1 w1 + a1 w2 + a2 w1 + a2 w2 + a1 w3 + a3 w1        (dropped: a2 w3, a3 w2, a3 w3 <= 2^-24 |a w|)
// each exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16: 6/16 of the fp32-MFMA time for the same fp32-level result.
// The kernels are in sa_bf3.inc (forward layers) and sa_bf3_dgrad.inc (data gradient).  Earlier variants of this layer (a
// 32x32x2 fp32-MFMA kernel, an LDS-staged split-bf16 kernel, a streaming fp32 kernel, timing builds) are in the
// history (round 2, `git log -- tools/lab/sa_variants`), not in the tree.
using f32x16 = float __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ void split3(float v, __bf16 &h, __bf16 &m, __bf16 &l) {
  h = (__bf16)v;
  const float r = v - (float)h;
  m = (__bf16)r;
  l = (__bf16)(r - (float)m);
}

#include "sa_bf3.inc"
#include "sa_bf3_dgrad.inc"
#include "sa_l3bwd.inc"

// ---- pooling forward: out[g, c] = max_s relu(bn(z[g*S+s, c])), first maximum ------------------------------------
__global__ __launch_bounds__(256) void sa_pool_fwd_kernel(const float *__restrict__ z, const float *__restrict__ st,
                                                          long G, int S, int C, float *__restrict__ out,
                                                          uint8_t *__restrict__ arg) {
  const int C4 = C / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= G * C4) return;
  const long g = i / C4;
  const int c4 = (int)(i % C4);
  f32x4 mean, sc, be;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float *s = st + (size_t)(c4 * 4 + u) * 4;
    mean[u] = s[0], sc[u] = s[2], be[u] = s[3];
  }
  f32x4 best = {-1.f, -1.f, -1.f, -1.f};
  int bi[4] = {0, 0, 0, 0};
  const float *p = z + ((size_t)g * S) * C + c4 * 4;
#pragma unroll 4
  for (int s = 0; s < S; ++s, p += C) {
    const f32x4 v = ld4(p);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float a = fmaxf((v[u] - mean[u]) * sc[u] + be[u], 0.f);
      if (a > best[u]) best[u] = a, bi[u] = s;
    }
  }
  st4(out + (size_t)g * C + c4 * 4, best);
  *reinterpret_cast<uchar4 *>(arg + (size_t)g * C + c4 * 4) =
      make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
}

// ---- pooling backward, pass 1: masked gradient dym = (out > 0) ? dout : 0 and the BN sums over the arg-max rows
__global__ __launch_bounds__(256) void sa_pool_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ out,
                                                          const uint8_t *__restrict__ arg, const float *__restrict__ z,
                                                          const float *__restrict__ zmax, const float *__restrict__ st, long G,
                                                          int S, int C, float *__restrict__ dym, double *__restrict__ part) {
  __shared__ float s_red[2][256];
  const int tid = threadIdx.x;
  const int c = tid % C;           // C in {64, 128, 256}: 256 / C groups per pass
  const int gs = tid / C, GP = 256 / C;
  const float mean = st[c * 4], istd = st[c * 4 + 1];
  // zmax [G][C] (the forward's pooling pass kept the arg-max rows' pre-activations) replaces a 4-byte gather per element out of
  // z [G S][C]; it equals z at the arg-max row wherever a gradient passes, except in a channel whose BatchNorm weight is exactly 0
  // (every row ties): such a channel reads z when z is given
  const bool from_max = zmax && (st[c * 4 + 2] != 0.f || !z);
  float s1 = 0.f, s2 = 0.f;
  for (long g = (long)blockIdx.x * GP + gs; g < G; g += (long)gridDim.x * GP) {
    const size_t o = (size_t)g * C + c;
    const float dy = out[o] > 0.f ? dout[o] : 0.f;
    dym[o] = dy;
    const float zz = from_max ? zmax[o] : z[((size_t)g * S + arg[o]) * C + c];
    s1 += dy;
    s2 += dy * ((zz - mean) * istd);
  }
  s_red[0][tid] = s1, s_red[1][tid] = s2;
  __syncthreads();
  if (tid < C) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < GP; ++i) a += (double)s_red[0][i * C + tid], b += (double)s_red[1][i * C + tid];
    part[((size_t)blockIdx.x * 2 + 0) * C + tid] = a;
    part[((size_t)blockIdx.x * 2 + 1) * C + tid] = b;
  }
}

// dz of the current tile element group (4 channels) from the dense or the pooled gradient source
template <bool POOLED>
__device__ __forceinline__ f32x4 load_dz(const float *__restrict__ dy, const uint8_t *__restrict__ arg, int S,
                                         const float *__restrict__ zk, long grow, int CK, int c0, f32x4 g, f32x4 k0,
                                         f32x4 k1) {
  const f32x4 z = ld4(zk + (size_t)grow * CK + c0);
  f32x4 d;
  if (POOLED) {
    // S is a power of two on the model's path (64 / 32 / 16): shift + mask instead of a 64-bit division per element
    const int lgS = (S & (S - 1)) == 0 ? __builtin_ctz((unsigned)S) : -1;
    const long grp = lgS >= 0 ? (grow >> lgS) : grow / S;
    const int s = (int)(grow - grp * S);
    const f32x4 dm = ld4(dy + (size_t)grp * CK + c0);
    const uchar4 a = *reinterpret_cast<const uchar4 *>(arg + (size_t)grp * CK + c0);
    d[0] = a.x == s ? dm[0] : 0.f;
    d[1] = a.y == s ? dm[1] : 0.f;
    d[2] = a.z == s ? dm[2] : 0.f;
    d[3] = a.w == s ? dm[3] : 0.f;
  } else {
    d = ld4(dy + (size_t)grow * CK + c0);
  }
  return g * d + k0 - k1 * z;
}

// ---- data gradient: dy_prev = (dz_k W_k) * [a_prev > 0], and the BN sums of dy_prev -----------------------------
// Same pipeline as sa_mid_fwd_kernel: the next tile's z_k (and dense dy) rows are prefetched with hand-issued loads
// while the matrix cores work, dz is formed while staging, and the accumulators go through LDS so that the epilogue
// (mask by relu'(bn(z_prev)), BN sums, store) reads z_prev and writes dy_prev as full rows.
// PREFETCH: register prefetch (off for CK = 256, where it would cost the second resident workgroup).
// ALIAS: the output tile reuses the staging buffer (one more barrier, 18 KB less LDS; CK = 256).
// L1 (first-layer fusion, SA1): z_prev is the first layer's pre-activation.  Its weight gradient
//   dW1[c, d] = sum_r dz1[r, c] in_d(r),  dz1 = g dy1 + k0 - k1 z1,  in = (rel x, rel y, rel z, inline feature)
// is linear in three sums that do not need the (not yet known) BN-backward constants g, k0, k1:
//   S1[c,d] = sum dy1 in_d,  S2[d] = sum in_d,  S3[c,d] = sum z1 in_d   =>   dW1 = g S1 + k0 S2 - k1 S3.
// The epilogue accumulates them from the tile it already holds, dy1 is never written and the separate first-layer
// backward pass (read dy1 + z1: 536 MB at SA1) disappears.
struct L1Args {
  const float *feat, *xyz, *new_xyz;
  const int32_t *idx;
  float rdiv;
  int Np, N, S;
  float *part;  // [NPART][COB*8 + 4]
  const float *rel4;   // optional: the rows' inputs as the statistics pass stored them; then z_prev is rebuilt, not read (zp unused)
  L1In li;
  float *partW;        // WG only: [gridDim.x][CK][COB] per-workgroup partial sums of dW_k = dz_k^T relu(bn(z_prev))
};

// WG (with L1 + rel4, CK = COB = 64): the layer's WEIGHT gradient from the same pass.  The tile's dz_k is in LDS for the data
// gradient and the epilogue rebuilds a_prev = relu(bn(z_prev)) for its mask anyway: it leaves a_prev in the output tile's place
// and one more product per tile, dW_k += dz_k^T a_prev (contraction over the tile's 64 rows), replaces the separate weight-gradient
// kernel and its second read of dy and z_k (536 MB at SA1).
template <int CK, int NT, bool POOLED, bool PREFETCH, bool ALIAS, bool L1 = false, bool WG = false>
// (first-layer instances: two waves per SIMD asked for explicitly.  Left to itself the compiler spreads their state over 262 - 307
// VGPRs + AGPRs, one more than half the register file: ONE 256-thread workgroup per CU, 32 KB of loads in flight per CU, 182 us
// at SA1; bounded to 256 registers the plain instance needs no scratch and runs 142 us, the fused one 256 instead of 311)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(L1 ? 2 : 1))) void sa_dgrad_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ arg, int S,
                                                       const float *__restrict__ zk, const float *__restrict__ coef,
                                                       const float *__restrict__ Wk, int CP, const float *__restrict__ zp,
                                                       const float *__restrict__ st_p, long R, float *__restrict__ dyp,
                                                       double *__restrict__ part, const L1Args L = L1Args{}) {
  constexpr int LD = CK + 4, KS = CK / 4, C4 = CK / 4, NV = TM * C4 / 256, RSTEP = 256 / C4;
  constexpr int COB = 64 * NT, LDO = COB + 4, O4 = COB / 4, NO = TM * O4 / 256, OSTEP = 256 / O4;
  constexpr bool DENSE_PF = PREFETCH && !POOLED;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_a = smem;                               // [TM][LD]  dz (MFMA B operand)
  float *s_o = ALIAS ? smem : smem + TM * LD;      // [TM][LDO] output tile, row-major
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int cbb = blockIdx.y * COB, wc = w * 16 * NT, cb = cbb + wc;
  float wf[NT][KS];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[j][ks] = Wk[(size_t)(ks * 4 + lg) * CP + cb + 16 * j + l15];
  const int c4 = tid % C4, r0 = tid / C4, o4 = tid % O4, or0 = tid / O4;
  f32x4 g, k0, k1, pm, pi, ps, pb;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float *s = coef + (size_t)(c4 * 4 + u) * 4;
    g[u] = s[0], k0[u] = s[1], k1[u] = s[2];
    const float *q = st_p + (size_t)(cbb + o4 * 4 + u) * 4;
    pm[u] = q[0], pi[u] = q[1], ps[u] = q[2], pb[u] = q[3];
  }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  const int lgS = (S > 0 && (S & (S - 1)) == 0) ? __builtin_ctz((unsigned)S) : -1;
  const long ntiles = (R + TM - 1) / TM, nfull = R / TM;
  float *s_rel = smem + (ALIAS ? TM * LD : TM * LD + TM * LDO);  // [TM][4] first-layer inputs of the tile's rows (L1)
  f32x4 q1[L1 ? 4 : 1], q3[L1 ? 4 : 1], q2 = {0.f, 0.f, 0.f, 0.f};
  f32x4 l1x = {0.f, 0.f, 0.f, 0.f}, l1y = l1x, l1z = l1x, l1f = l1x;
  if (L1 && L.rel4) l1_weights(L.li, cbb + o4 * 4, l1x, l1y, l1z, l1f);
  if (L1) {
#pragma unroll
    for (int u = 0; u < 4; ++u) q1[u] = q3[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  static_assert(!WG || (L1 && !ALIAS && CK == 64 && NT == 1), "fused weight gradient: 64 x 64 first-layer instance only");
  f32x4 accw[WG ? 4 : 1];
#pragma unroll
  for (int n = 0; n < (WG ? 4 : 1); ++n) accw[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 pz[PREFETCH ? NV : 1], pd[DENSE_PF ? NV : 1];
  auto fetch = [&](long t) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      long grow = t * TM + r0 + i * RSTEP;
      grow = grow < R ? grow : R - 1;
      const float *src = zk + (size_t)grow * CK + c4 * 4;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(pz[i]) : "v"(src) : "memory");
      if (DENSE_PF) {
        const float *sd = dy + (size_t)grow * CK + c4 * 4;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(pd[i]) : "v"(sd) : "memory");
      }
    }
  };
  auto tile = [&](long t, auto full, bool stores_pending) {
    constexpr bool FULL = decltype(full)::value;
    const long row0 = t * TM;
    if (PREFETCH) {
      // (L1: the epilogue stores nothing, every outstanding operation is one of the prefetched loads)
      if (stores_pending && !L1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NO) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        asm volatile("" : "+v"(pz[i]));
        if (DENSE_PF) asm volatile("" : "+v"(pd[i]));
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = r0 + i * RSTEP;
      const long grow = row0 + row;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      if (FULL || grow < R) {
        const f32x4 z = PREFETCH ? pz[i] : ld4(zk + (size_t)grow * CK + c4 * 4);
        f32x4 d;
        if (POOLED) {
          const long grp = lgS >= 0 ? (grow >> lgS) : grow / S;
          const int sidx = (int)(grow - grp * S);
          const f32x4 dm = ld4(dy + (size_t)grp * CK + c4 * 4);
          const uchar4 am = *reinterpret_cast<const uchar4 *>(arg + (size_t)grp * CK + c4 * 4);
          d[0] = am.x == sidx ? dm[0] : 0.f;
          d[1] = am.y == sidx ? dm[1] : 0.f;
          d[2] = am.z == sidx ? dm[2] : 0.f;
          d[3] = am.w == sidx ? dm[3] : 0.f;
        } else {
          d = DENSE_PF ? pd[i] : ld4(dy + (size_t)grow * CK + c4 * 4);
        }
        a = g * d + k0 - k1 * z;
      }
      st4(&s_a[row * LD + c4 * 4], a);
    }
    if (L1 && tid < TM) {
      f32x4 in = {0.f, 0.f, 0.f, 0.f};
      const long grow = row0 + tid;
      if ((FULL || grow < R) && L.rel4) {
        in = ld4(L.rel4 + (size_t)grow * 4);
      } else if (FULL || grow < R) {
        const long b = row_div(grow, (long)L.N * L.S), gi = row_div(grow, L.S);
        const int p = L.idx[grow];
        const float *q = L.xyz + ((size_t)b * L.Np + p) * 3, *c = L.new_xyz + (size_t)gi * 3;
        in[0] = (q[0] - c[0]) / L.rdiv, in[1] = (q[1] - c[1]) / L.rdiv, in[2] = (q[2] - c[2]) / L.rdiv;
        if (L.feat) in[3] = L.feat[(size_t)b * L.Np + p];
      }
      st4(&s_rel[tid * 4], in);
#ifndef SPACAP_HACK_NOQ3
      q2 += in;
#endif
    }
    __syncthreads();
    if (PREFETCH && FULL) fetch(t + gridDim.x);
    f32x4 acc[TM / 16][NT];
#pragma unroll
    for (int mt = 0; mt < TM / 16; ++mt)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int mt = 0; mt < TM / 16; ++mt) {
          const float b = s_a[(mt * 16 + l15) * LD + ks * 4 + lg];
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[mt][j] = MFMA16(wf[j][ks], b, acc[mt][j]);
        }
      }
    }
    if (ALIAS) __syncthreads();  // every wave is done reading dz before the output tile overwrites it
#pragma unroll
    for (int mt = 0; mt < TM / 16; ++mt)
#pragma unroll
      for (int j = 0; j < NT; ++j) st4(&s_o[(mt * 16 + l15) * LDO + wc + 16 * j + 4 * lg], acc[mt][j]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NO; ++i) {
      const int row = or0 + i * OSTEP;
      if (FULL || row0 + row < R) {
        const size_t o = (size_t)(row0 + row) * CP + cbb + o4 * 4;
        const f32x4 da = ld4(&s_o[row * LDO + o4 * 4]);
        f32x4 z;
        if (L1 && L.rel4) z = l1_row(l1x, l1y, l1z, l1f, ld4(&s_rel[row * 4]), L.li.has_feat != 0);
        else z = ld4(zp + o);
        f32x4 d;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float pre = (z[u] - pm[u]) * ps[u] + pb[u];
          d[u] = pre > 0.f ? da[u] : 0.f;
          s1[u] += d[u];
          s2[u] += d[u] * ((z[u] - pm[u]) * pi[u]);
        }
        if (L1) {
          const f32x4 in = ld4(&s_rel[row * 4]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            q1[u] += in * d[u];
#ifndef SPACAP_HACK_NOQ3
            q3[u] += in * z[u];
#endif
          }
        } else {
          st4(dyp + o, d);
        }
        if (WG) {   // a_prev over this thread's own element of the output tile (nobody else reads it in this loop)
          f32x4 ap;
#pragma unroll
          for (int u = 0; u < 4; ++u) ap[u] = fmaxf((z[u] - pm[u]) * ps[u] + pb[u], 0.f);
          st4(&s_o[row * LDO + o4 * 4], ap);
        }
      } else if (WG) {
        st4(&s_o[row * LDO + o4 * 4], f32x4{0.f, 0.f, 0.f, 0.f});   // (rows past the end: dz is zero there as well)
      }
    }
    if (WG) {
      __syncthreads();
      // dW[ck = 16 w + .][cp = 16 n + .] += sum over the tile's rows of dz[row][ck] a_prev[row][cp]
#pragma unroll
      for (int ks = 0; ks < TM / 4; ++ks) {
        const float af = s_a[(ks * 4 + lg) * LD + 16 * w + l15];
#pragma unroll
        for (int n = 0; n < 4; ++n) accw[n] = MFMA16(af, s_o[(ks * 4 + lg) * LDO + 16 * n + l15], accw[n]);
      }
      __syncthreads();   // before the next tile is staged over dz / the output tile
    }
    if (ALIAS) __syncthreads();  // the output tile is consumed before the next tile is staged over it
  };
  bool pending = false;
  if (PREFETCH && (long)blockIdx.x < nfull) fetch(blockIdx.x);
  for (long t = blockIdx.x; t < nfull; t += gridDim.x) {
    tile(t, std::true_type{}, pending);
    pending = true;
  }
  if (PREFETCH) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      asm volatile("" ::"v"(pz[i]));
      if (DENSE_PF) asm volatile("" ::"v"(pd[i]));
    }
  }
  if (nfull < ntiles && (long)blockIdx.x == nfull % gridDim.x) {  // ragged last tile
    if (PREFETCH) {
      fetch(nfull);
    }
    tile(nfull, std::false_type{}, false);
  }
  // BN sums: this thread owns columns cbb + 4*o4 .. +3 for the rows or0 + k*OSTEP; combine the OSTEP row groups
  __syncthreads();
  float *s_red = smem;  // [2][OSTEP][COB]
  st4(&s_red[(0 * OSTEP + or0) * COB + o4 * 4], s1);
  st4(&s_red[(1 * OSTEP + or0) * COB + o4 * 4], s2);
  __syncthreads();
  if (tid < 2 * COB) {
    const int k = tid / COB, c = tid % COB;
    float a = 0.f;
    for (int i = 0; i < OSTEP; ++i) a += s_red[(k * OSTEP + i) * COB + c];
    part[((size_t)blockIdx.x * 2 + k) * CP + cbb + c] = (double)a;
    for (int pr = blockIdx.x + gridDim.x; pr < NPART; pr += gridDim.x) part[((size_t)pr * 2 + k) * CP + cbb + c] = 0.0;
  }
  if (WG) {
    float *o = L.partW + (size_t)blockIdx.x * CK * COB;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int u = 0; u < 4; ++u) o[(size_t)(16 * w + 4 * lg + u) * COB + 16 * n + l15] = accw[n][u];
  }
  if (L1) {  // S1 / S3: combine the OSTEP row groups; S2: combine the TM row slots
    __syncthreads();
    float *s_q = smem;  // [OSTEP][COB][8]
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      st4(&s_q[((or0 * COB) + o4 * 4 + u) * 8], q1[u]);
      st4(&s_q[((or0 * COB) + o4 * 4 + u) * 8 + 4], q3[u]);
    }
    __syncthreads();
    float *out = L.part + (size_t)blockIdx.x * (COB * 8 + 4);
    for (int e = tid; e < COB * 8; e += 256) {
      float a = 0.f;
      for (int i = 0; i < OSTEP; ++i) a += s_q[i * COB * 8 + e];
      out[e] = a;
    }
    __syncthreads();
    if (tid < TM) st4(&s_q[tid * 4], q2);
    __syncthreads();
    if (tid < 4) {
      float a = 0.f;
      for (int i = 0; i < TM; ++i) a += s_q[i * 4 + tid];
      out[COB * 8 + tid] = a;
    }
    for (int pr = blockIdx.x + gridDim.x; pr < NPART; pr += gridDim.x)
      for (int e = tid; e < COB * 8 + 4; e += 256) L.part[(size_t)pr * (COB * 8 + 4) + e] = 0.f;
  }
}

// ---- weight gradient: dW_k[ck, cp] = sum_r dz_k[r, ck] a_prev[r, cp]; one partial per row slab -----------------
template <int CKB, int CP, bool POOLED>
__global__ __launch_bounds__(256) void sa_wgrad_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ arg, int S,
                                                       const float *__restrict__ zk, const float *__restrict__ coef, int CK,
                                                       const float *__restrict__ zp, const float *__restrict__ st_p, long R,
                                                       float *__restrict__ partW, L1In li = L1In{nullptr, 0, 0}) {
  constexpr int LDZ = CKB + 16, LDA = CP + 16;
  constexpr int MT = CKB / 64, NTT = CP / 16;           // m-tiles per wave, n-tiles
  constexpr int Z4 = CKB / 4, A4 = CP / 4;
  constexpr int NVZ = TW * Z4 / 256, NVA = TW * A4 / 256, RZ = 256 / Z4, RA = 256 / A4;
  __shared__ __attribute__((aligned(16))) float s_dz[TW * LDZ];
  __shared__ __attribute__((aligned(16))) float s_a[TW * LDA];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int ckb0 = blockIdx.y * CKB;
  const int z4 = tid % Z4, zr0 = tid / Z4, a4 = tid % A4, ar0 = tid / A4;
  f32x4 g, k0, k1, pm, ps, pb;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float *s = coef + (size_t)(ckb0 + z4 * 4 + u) * 4;
    g[u] = s[0], k0[u] = s[1], k1[u] = s[2];
    const float *q = st_p + (size_t)(a4 * 4 + u) * 4;
    pm[u] = q[0], ps[u] = q[2], pb[u] = q[3];
  }
  f32x4 acc[MT][NTT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long ntiles = (R + TW - 1) / TW;
  const int lgS = (S > 0 && (S & (S - 1)) == 0) ? __builtin_ctz((unsigned)S) : -1;
  // software pipeline: the raw operands of tile t + gridDim.x are loaded into registers while tile t is multiplied
  f32x4 rz[NVZ], rd[NVZ], ra[NVA];
  uchar4 rg[POOLED ? NVZ : 1];
  int rs[POOLED ? NVZ : 1];
  auto fetch = [&](long t) {
    const long row0 = t * TW;
#pragma unroll
    for (int i = 0; i < NVZ; ++i) {
      long grow = row0 + zr0 + i * RZ;
      grow = grow < R ? grow : R - 1;
      rz[i] = ld4(zk + (size_t)grow * CK + ckb0 + z4 * 4);
      if (POOLED) {
        const long grp = lgS >= 0 ? (grow >> lgS) : grow / S;
        rs[i] = (int)(grow - grp * S);
        rd[i] = ld4(dy + (size_t)grp * CK + ckb0 + z4 * 4);
        rg[i] = *reinterpret_cast<const uchar4 *>(arg + (size_t)grp * CK + ckb0 + z4 * 4);
      } else {
        rd[i] = ld4(dy + (size_t)grow * CK + ckb0 + z4 * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      long grow = row0 + ar0 + i * RA;
      grow = grow < R ? grow : R - 1;
      ra[i] = li.W1 ? ld4(zp + (size_t)grow * 4) : ld4(zp + (size_t)grow * CP + a4 * 4);
    }
  };
  f32x4 l1x = {0.f, 0.f, 0.f, 0.f}, l1y = l1x, l1z = l1x, l1f = l1x;
  if (li.W1) l1_weights(li, a4 * 4, l1x, l1y, l1z, l1f);
  if ((long)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long row0 = t * TW;
#pragma unroll
    for (int i = 0; i < NVZ; ++i) {
      const int row = zr0 + i * RZ;
      f32x4 d = rd[i];
      if (POOLED) {
        d[0] = rg[i].x == rs[i] ? d[0] : 0.f;
        d[1] = rg[i].y == rs[i] ? d[1] : 0.f;
        d[2] = rg[i].z == rs[i] ? d[2] : 0.f;
        d[3] = rg[i].w == rs[i] ? d[3] : 0.f;
      }
      f32x4 a = g * d + k0 - k1 * rz[i];
      if (row0 + row >= R) a = f32x4{0.f, 0.f, 0.f, 0.f};
      st4(&s_dz[row * LDZ + z4 * 4], a);
    }
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int row = ar0 + i * RA;
      f32x4 a;
      const f32x4 zv = li.W1 ? l1_row(l1x, l1y, l1z, l1f, ra[i], li.has_feat != 0) : ra[i];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = fmaxf((zv[u] - pm[u]) * ps[u] + pb[u], 0.f);
      if (row0 + row >= R) a = f32x4{0.f, 0.f, 0.f, 0.f};
      st4(&s_a[row * LDA + a4 * 4], a);
    }
    __syncthreads();
    if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
#pragma unroll
    for (int ks = 0; ks < TW / 4; ++ks) {
      float af[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = s_dz[(ks * 4 + lg) * LDZ + (w * MT + m) * 16 + l15];
#pragma unroll
      for (int n = 0; n < NTT; ++n) {
        const float b = s_a[(ks * 4 + lg) * LDA + n * 16 + l15];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][n] = MFMA16(af[m], b, acc[m][n]);
      }
    }
    __syncthreads();
  }
  float *o = partW + ((size_t)blockIdx.x * CK + ckb0) * CP;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTT; ++n)
#pragma unroll
      for (int u = 0; u < 4; ++u) o[(size_t)((w * MT + m) * 16 + 4 * lg + u) * CP + n * 16 + l15] = acc[m][n][u];
}

// ---- layer 1 backward: dz1 (in place over dy1), dW1 partials [NPART][C1][4], optional d rel [R][3] -------------
template <int C1>
__global__ __launch_bounds__(256) void sa_l1_bwd_kernel(float *__restrict__ dy1, const float *__restrict__ z1,
                                                        const float *__restrict__ coef, const float *__restrict__ feat,
                                                        const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                        const int32_t *__restrict__ idx, const float *__restrict__ W1,
                                                        int ldw, float rdiv, int Np, int N, int S, long R,
                                                        float *__restrict__ partW, float *__restrict__ drel, int write_dz) {
  constexpr int C4 = C1 / 4, RP = 256 / C4;
  __shared__ float s_red[4][RP][C1];
  const int tid = threadIdx.x, c4 = tid % C4, rs = tid / C4;
  f32x4 g, k0, k1, wx, wy, wz;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float *s = coef + (size_t)(c4 * 4 + u) * 4;
    g[u] = s[0], k0[u] = s[1], k1[u] = s[2];
    const float *w = W1 + (size_t)(c4 * 4 + u) * ldw;
    wx[u] = w[0], wy[u] = w[1], wz[u] = w[2];
  }
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 ax = zero, ay = zero, az = zero, af = zero;
  const long NS = (long)N * S;
  for (long r0 = (long)blockIdx.x * RP; r0 < R; r0 += (long)gridDim.x * RP) {
    const long r = r0 + rs;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (r < R) {
      const long b = row_div(r, NS), gi = row_div(r, S);
      const int p = idx[r];
      const float *q = xyz + ((size_t)b * Np + p) * 3, *c = new_xyz + (size_t)gi * 3;
      const float rx = (q[0] - c[0]) / rdiv, ry = (q[1] - c[1]) / rdiv, rz = (q[2] - c[2]) / rdiv;
      const size_t o = (size_t)r * C1 + c4 * 4;
      const f32x4 dz = g * ld4(dy1 + o) + k0 - k1 * ld4(z1 + o);
      if (write_dz) st4(dy1 + o, dz);
      ax += dz * rx, ay += dz * ry, az += dz * rz;
      if (feat) af += dz * feat[(size_t)b * Np + p];
      if (drel) {
        px = dz[0] * wx[0] + dz[1] * wx[1] + dz[2] * wx[2] + dz[3] * wx[3];
        py = dz[0] * wy[0] + dz[1] * wy[1] + dz[2] * wy[2] + dz[3] * wy[3];
        pz = dz[0] * wz[0] + dz[1] * wz[1] + dz[2] * wz[2] + dz[3] * wz[3];
      }
    }
    if (drel) {  // sum over the C4 lanes that share the row (C4 = 16 or 32, aligned inside the wave)
#pragma unroll
      for (int o = 1; o < C4; o <<= 1) px += __shfl_xor(px, o), py += __shfl_xor(py, o), pz += __shfl_xor(pz, o);
      if (c4 == 0 && r < R) drel[r * 3] = px / rdiv, drel[r * 3 + 1] = py / rdiv, drel[r * 3 + 2] = pz / rdiv;
    }
  }
  st4(&s_red[0][rs][c4 * 4], ax);
  st4(&s_red[1][rs][c4 * 4], ay);
  st4(&s_red[2][rs][c4 * 4], az);
  st4(&s_red[3][rs][c4 * 4], af);
  __syncthreads();
  for (int e = tid; e < 4 * C1; e += 256) {
    const int c = e / 4, k = e % 4;
    float a = 0.f;
    for (int i = 0; i < RP; ++i) a += s_red[k][i][c];
    partW[((size_t)blockIdx.x * C1 + c) * 4 + k] = a;
  }
}

// ---- rows -> source points: dY[b, p, :] = sum over the rows r with idx[r] = p of dz1[r, :] --------------------
__global__ __launch_bounds__(256) void rows_keys_kernel(const int32_t *__restrict__ idx, int Np, long E, long total,
                                                        unsigned *__restrict__ keys, int *__restrict__ vals) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  keys[i] = (unsigned)((i / E) * Np + idx[i]);
  vals[i] = (int)i;
}

__global__ __launch_bounds__(256) void rows_offsets_kernel(const unsigned *__restrict__ sorted, long total, long K,
                                                           int *__restrict__ off) {
  const long k = (long)blockIdx.x * 256 + threadIdx.x;
  if (k > K) return;
  long lo = 0, hi = total;
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if ((long)sorted[mid] < k) lo = mid + 1; else hi = mid;
  }
  off[k] = (int)lo;
}

// The same index without a sort, one launch: a workgroup of 16 waves per scene.  Wave w owns the w-th stretch of the scene's
// rows.  (1) per-wave histograms h[w][p] (16-bit counters, two per LDS word), (2) per point: counts -> exclusive prefix over the
// waves, totals -> exclusive scan over the points = off, (3) every wave walks its stretch in order, 64 rows at a time; a row's
// slot is off[p] + rows of p in earlier waves + in earlier iterations of this wave + in lower lanes of this iteration (one
// ballot per distinct point among the 64 rows).  Ascending row order within a point by construction: the result equals the
// stable sort's.  Np <= 4096, E <= 65535 (LDS, 16-bit counters); other shapes take the radix sort below.
constexpr int RI_WAVES = 16;
__global__ __launch_bounds__(1024) void rows_index_kernel(const int32_t *__restrict__ idx, int Np, int E, int B,
                                                          int *__restrict__ off, int *__restrict__ order) {
  extern __shared__ unsigned s_ri[];
  const int NpW = (Np + 1) / 2;
  unsigned *h = s_ri;                                        // [16][NpW]
  int *start = reinterpret_cast<int *>(h + RI_WAVES * NpW);  // [2 NpW]
  int *wsum = start + 2 * NpW;                               // [1024]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x;
  const int32_t *ib = idx + (size_t)b * E;
  const int seg = ((E + RI_WAVES * 64 - 1) / (RI_WAVES * 64)) * 64, e_beg = w * seg, e_end = min(E, e_beg + seg);
  for (int i = tid; i < RI_WAVES * NpW; i += 1024) h[i] = 0u;
  __syncthreads();
  for (int e = e_beg + lane; e < e_end; e += 64) {
    const int p = ib[e];
    atomicAdd(&h[w * NpW + (p >> 1)], 1u << (16 * (p & 1)));
  }
  __syncthreads();
  for (int q = tid; q < NpW; q += 1024) {   // one word = two points: counts -> prefix over the waves, totals -> start
    unsigned a0 = 0, a1 = 0;
    for (int v = 0; v < RI_WAVES; ++v) {
      const unsigned c = h[v * NpW + q];
      h[v * NpW + q] = a0 | (a1 << 16);
      a0 += c & 0xffffu, a1 += c >> 16;
    }
    start[2 * q] = (int)a0, start[2 * q + 1] = (int)a1;
  }
  __syncthreads();
  {   // exclusive scan of start[0 .. 2 NpW): CH consecutive entries per thread, then the 1024 partial sums
    const int CH = (2 * NpW + 1023) / 1024, lo = min(tid * CH, 2 * NpW), hi = min(lo + CH, 2 * NpW);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += start[i];
    wsum[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int v = tid >= d ? wsum[tid - d] : 0;
      __syncthreads();
      wsum[tid] += v;
      __syncthreads();
    }
    int run = wsum[tid] - sum;
    for (int i = lo; i < hi; ++i) {
      const int c = start[i];
      start[i] = run;
      if (i < Np) off[(size_t)b * Np + i] = b * E + run;
      run += c;
    }
    if (b == B - 1 && tid == 0) off[(size_t)B * Np] = B * E;
  }
  __syncthreads();
  for (int e0 = e_beg; e0 < e_end; e0 += 64) {
    const int e = e0 + lane;
    const bool active = e < e_end;
    const int p = active ? ib[e] : -1;
    unsigned long long todo = __ballot(active);
    int rank = 0, n = 0;
    bool lead = false;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int k = __builtin_amdgcn_readlane(p, leader);
      const unsigned long long m = __ballot(p == k);
      if (p == k) rank = __popcll(m & ((1ull << lane) - 1ull)), n = __popcll(m), lead = lane == leader;
      todo &= ~m;
    }
    if (active) {
      const unsigned word = h[w * NpW + (p >> 1)];
      const int pos = start[p] + (int)((word >> (16 * (p & 1))) & 0xffffu) + rank;
      order[(size_t)b * E + pos] = b * E + e;
    }
    if (lead) atomicAdd(&h[w * NpW + (p >> 1)], (unsigned)n << (16 * (p & 1)));
  }
}

__global__ __launch_bounds__(256) void rows_gather_sum_kernel(const float *__restrict__ dz, const int *__restrict__ off,
                                                              const int *__restrict__ order, long K, int C,
                                                              float *__restrict__ out) {
  const int C4 = C / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= K * C4) return;
  const long k = i / C4;
  const int c4 = (int)(i % C4);
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  const int beg = off[k], end = off[k + 1];
  // (order[p] -> row is a dependent chain: four of them in flight per iteration; the sum keeps its order)
  int p = beg;
  for (; p + 3 < end; p += 4) {
    const int o0 = order[p], o1 = order[p + 1], o2 = order[p + 2], o3 = order[p + 3];
    const f32x4 r0 = ld4(dz + (size_t)o0 * C + c4 * 4), r1 = ld4(dz + (size_t)o1 * C + c4 * 4),
                r2 = ld4(dz + (size_t)o2 * C + c4 * 4), r3 = ld4(dz + (size_t)o3 * C + c4 * 4);
    a += r0;
    a += r1;
    a += r2;
    a += r3;
  }
  for (; p < end; ++p) a += ld4(dz + (size_t)order[p] * C + c4 * 4);
  st4(out + (size_t)k * C + c4 * 4, a);
}

// Gradient of the relative coordinates (rel = (xyz[idx] - new_xyz) / r, drel [B * E][3], E = N S rows per scene) routed to
// both of its sources in one launch, fixed summation orders (the autograd composition is a zero-fill + an int64 copy of idx +
// an atomic scatter_add_ + a sum + a neg):
//   threads 0 .. K-1 (K = B Np):   dxyz[k][:] = sum of drel[r][:] over the rows r that reference source point k, ascending r
//   threads K .. K + B N - 1:      dnew[g][:] = - sum_s drel[g S + s][:]
__global__ __launch_bounds__(256) void sa_drel_sums_kernel(const float *__restrict__ drel, const int *__restrict__ off,
                                                           const int *__restrict__ order, long K, long G, int S,
                                                           float *__restrict__ dxyz, float *__restrict__ dnew) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < K) {
    if (!dxyz) return;
    float ax = 0.f, ay = 0.f, az = 0.f;
    const int beg = off[i], end = off[i + 1];
    int p = beg;
    for (; p + 3 < end; p += 4) {   // (order[p] -> row is a dependent chain: four in flight, the sum keeps its order)
      const float *r0 = drel + (size_t)order[p] * 3, *r1 = drel + (size_t)order[p + 1] * 3, *r2 = drel + (size_t)order[p + 2] * 3,
                  *r3 = drel + (size_t)order[p + 3] * 3;
      const float x0 = r0[0], y0 = r0[1], z0 = r0[2], x1 = r1[0], y1 = r1[1], z1 = r1[2];
      const float x2 = r2[0], y2 = r2[1], z2 = r2[2], x3 = r3[0], y3 = r3[1], z3 = r3[2];
      ax += x0, ay += y0, az += z0;
      ax += x1, ay += y1, az += z1;
      ax += x2, ay += y2, az += z2;
      ax += x3, ay += y3, az += z3;
    }
    for (; p < end; ++p) {
      const float *r = drel + (size_t)order[p] * 3;
      ax += r[0], ay += r[1], az += r[2];
    }
    dxyz[i * 3] = ax, dxyz[i * 3 + 1] = ay, dxyz[i * 3 + 2] = az;
  } else if (i < K + G && dnew) {
    const long g = i - K;
    const float *r = drel + (size_t)g * S * 3;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int sidx = 0; sidx < S; ++sidx) ax += r[sidx * 3], ay += r[sidx * 3 + 1], az += r[sidx * 3 + 2];
    dnew[g * 3] = -ax, dnew[g * 3 + 1] = -ay, dnew[g * 3 + 2] = -az;
  }
}

// First-layer weight gradient of a module with point features, assembled in one launch from the two sets of partial results:
//   dW1[c][0..2]    = sum over the n1 slabs of pw1 [n1][C1][4]   (relative coordinates; column 3 = the inline feature, unused here)
//   dW1[c][3 + j]   = sum over the nf slabs of pf  [nf][C1][Cf]  (the feature product over the source points)
// with sum_slabs_kernel's grouping (four runs of slabs, combined as (s0 + s1) + (s2 + s3)): the same values as the two slab sums
// and the concatenation it replaces.
__global__ __launch_bounds__(256) void sa_dw1_assemble_kernel(const float *__restrict__ pw1, int n1, const float *__restrict__ pf, int nf,
                                                              int C1, int Cf, float *__restrict__ out) {
  __shared__ float s_g[4][64];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long e = (long)blockIdx.x * 64 + c, ne = (long)C1 * (3 + Cf);
  float a = 0.f;
  if (e < ne) {
    const int ch = (int)(e / (3 + Cf)), j = (int)(e % (3 + Cf));
    const float *src = j < 3 ? pw1 + (size_t)ch * 4 + j : pf + (size_t)ch * Cf + (j - 3);
    const size_t stride = j < 3 ? (size_t)C1 * 4 : (size_t)C1 * Cf;
    const int ns = j < 3 ? n1 : nf, per = (ns + 3) / 4, s0 = grp * per, s1 = min(ns, s0 + per);
    // (a dependent chain of strided loads: 32 in flight per thread, added in slab order -- 2 round trips instead of 8 at 256 slabs)
    int k = s0;
    for (; k + 32 <= s1; k += 32) {
      float t[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) t[u] = src[(size_t)(k + u) * stride];
#pragma unroll
      for (int u = 0; u < 32; ++u) a += t[u];
    }
#pragma unroll 8
    for (; k < s1; ++k) a += src[(size_t)k * stride];
  }
  s_g[grp][c] = a;
  __syncthreads();
  if (grp == 0 && e < ne) out[e] = (s_g[0][c] + s_g[1][c]) + (s_g[2][c] + s_g[3][c]);
}

struct RowsLayout {
  size_t total, K, keys_in, keys_out, vals_in, vals_out, off, cub, cub_bytes, bytes;
  int bits;
};
bool rows_layout(int B, int Np, long E, RowsLayout &L) {
  if (B <= 0 || Np <= 0 || E <= 0) return false;
  L.total = (size_t)B * E;
  L.K = (size_t)B * Np;
  if (L.K >= (1ull << 31) || L.total >= (1ull << 31)) return false;
  L.bits = 1;
  while ((1ull << L.bits) < L.K) ++L.bits;
  size_t cub = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, cub, (const unsigned *)nullptr, (unsigned *)nullptr,
                                         (const int *)nullptr, (int *)nullptr, (int)L.total, 0, L.bits,
                                         (hipStream_t)0) != hipSuccess)
    cub = 0;
  (void)hipGetLastError();
  if (cub == 0) cub = 16 * L.total + (1 << 20);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = 0;
  L.keys_in = o; o += up(4 * L.total);
  L.keys_out = o; o += up(4 * L.total);
  L.vals_in = o; o += up(4 * L.total);
  L.vals_out = o; o += up(4 * L.total);
  L.off = o; o += up(4 * (L.K + 1));
  L.cub = o; L.cub_bytes = cub; o += up(cub);
  L.bytes = o;
  return true;
}

// Workgroups (256 threads) of `kernel` that are resident on the whole device at once.  The GEMM kernels are
// persistent: a grid larger than this would run its surplus as a second, nearly empty round.
// CUs left to the side stream by the FORWARD layer kernels (spacap_sa_reserve_cus).  While the next batch's sampling chain
// runs beside the step (one workgroup per scene: 8 CUs for the first ~4.4 of the step's ~9.8 ms, i.e. during the backbone's
// forward), a persistent grid sized to ALL CUs leaves its last workgroups waiting for a free CU: they run as a second round
// and the kernel takes 1.3 - 2.1x as long (tools/lab/fps_interference.py; the side-stream work cost the main stream 0.86 ms
// per step, tools/lab/step_without_side_stream.py).  A grid of (CUs - 8) workgroups costs one more tile round in nine
// when nothing runs beside it.
// The library's one environment switch: SPACAP_SA_F32MFMA=1 keeps every shared-MLP product (forward layers, data gradient,
// plain row products, pooling candidates from the epilogue) on the fp32-MFMA kernels instead of the split-bf16 streaming ones
// -- the reference implementation the split kernels are gated against (tests/test_sa_gemm_kernels_gpu.py).
inline bool f32_mfma_only() {
  static const bool on = getenv("SPACAP_SA_F32MFMA") != nullptr && atoi(getenv("SPACAP_SA_F32MFMA")) != 0;
  return on;
}
static std::atomic<int> g_reserved_cus{0};
inline int reserved_cus() { return g_reserved_cus.load(std::memory_order_relaxed); }
inline int device_cus() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    (void)hipGetLastError();
    // LAB: the CUs the step's stream may use when it was created with a CU mask (tools/lab/cumask_step.py)
    if (const char *e = getenv("SPACAP_LAB_CUS")) { const int v = atoi(e); if (v >= 8 && v <= cus) cus = v; }
    return cus;
  }();
  return n;
}
template <typename K>
int resident_blocks(K kernel, size_t lds) {
  int per = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, 256, lds) != hipSuccess || per < 1) per = 1;
  (void)hipGetLastError();
  return per * device_cus();
}
// resident workgroups of a FORWARD kernel with the reserved CUs left out (used by the relation tail; the HBM-bound 64-channel
// fp32 layer kernel runs 1.3x as long beside the sampling kernel with or without it and only loses from a smaller grid when
// alone, so it keeps the full one).  Kernels with several workgroups per CU need slack
// beyond the occupied CUs themselves (the dispatcher does not pack the rest perfectly: measured, the relation tail beside the
// sampling kernel runs 1.5x as long with 8 CUs left out and 1.04x with 24)
inline int fwd_resident(int resident) {
  const int per = resident / device_cus();
  return resident - per * reserved_cus() * (per > 1 ? 3 : 1);
}
// The BACKWARD kernels with persistent grids leave the reserved CUs out as well: the sampling chain of the next batch now runs
// beside the first ~5.8 ms of a ~7.4 ms step, i.e. beside the captioner's and most of the detector's backward, and a whole-CU
// workgroup that finds its CU taken runs as a second round (tools/lab/beside.py: 1.5 - 1.9x beside ANY 8 resident workgroups).
// (LAB knob SPACAP_LAB_BWD_RESERVE: 0 = full grids as before, 1 = per-CU count x reserved, 3 = the forward formula)
inline int bwd_resident(int resident) {
  static const int mode = [] { const char *e = getenv("SPACAP_LAB_BWD_RESERVE"); return e ? atoi(e) : 1; }();
  if (mode == 0) return resident;
  if (mode == 3) return fwd_resident(resident);
  const int per = resident / device_cus();
  return resident - per * reserved_cus();
}
inline int bwd_cus() { return bwd_resident(device_cus()); }
inline int grid_rows(int resident, int gy, long tiles) {
  long g = resident / gy;
  if (g > NPART) g = NPART;
  if (g > tiles) g = tiles;
  return (int)(g < 1 ? 1 : g);
}

inline unsigned nblocks(long work, int per) {
  long g = (work + per - 1) / per;
  return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

// ===========================================================================================================
extern "C" int spacap_sa_nparts(void) { return NPART; }
// row slabs (= partial results = workgroups per output tile) of the weight-gradient GEMM: what is resident at
// once, at most 64 MB of partials
namespace {
int wgrad_resident(int CK, int CP, bool pooled) {
#define WR(CKB, CPV, PV)                                                        \
  {                                                                             \
    static const int res = resident_blocks(sa_wgrad_kernel<CKB, CPV, PV>, 0);   \
    return res;                                                                 \
  }
  if (pooled && CK % 128 == 0 && CP == 64) WR(128, 64, true)
  if (pooled && CK % 128 == 0 && CP == 128) WR(128, 128, true)
  if (!pooled && CK == 64 && CP == 64) WR(64, 64, false)
  if (!pooled && CK == 128 && CP == 128) WR(128, 128, false)
#undef WR
  return 512;
}
}  // namespace
extern "C" int spacap_sa_wgrad_slabs(long R, int CK, int CP, int pooled) {
  const int gy = CK >= 128 ? CK / 128 : 1;
  long n = bwd_resident(wgrad_resident(CK, CP, pooled != 0)) / gy, cap = (16L << 20) / ((long)CK * CP), tiles = (R + TW - 1) / TW;
  if (n > cap) n = cap;
  if (n > tiles) n = tiles;
  return (int)(n < 1 ? 1 : n);
}

extern "C" int spacap_sa_mlp_supported(int C1, int C2, int C3) {
  const bool sa1 = (C1 == 64 && C2 == 64 && C3 == 128);
  const bool big = (C1 == 128 && C2 == 128 && (C3 == 128 || C3 == 256));
  return (sa1 || big) ? 1 : 0;
}

extern "C" int spacap_sa_l1_fwd_f32(const float *Y, const float *feat, const float *xyz, const float *new_xyz,
                                    const int32_t *idx, const float *W1, int ldw, float rdiv, int B, int Np, int N,
                                    int S, int C1, float *z1, double *part, spacap_stream_t stream) {
  const char *what = "spacap_sa_l1_fwd_f32";
  SPACAP_REQUIRE(B >= 1 && Np >= 1 && N >= 1 && S >= 1 && S <= 255, "%s: bad sizes", what);
  SPACAP_REQUIRE(C1 == 64 || C1 == 128, "%s: C1=%d unsupported", what, C1);
  SPACAP_REQUIRE(xyz && new_xyz && idx && W1 && z1 && part && ldw >= (feat ? 4 : 3) && rdiv > 0.f, "%s: bad arguments", what);
  const long R = (long)B * N * S;
  hipStream_t s = spacap::as_stream(stream);
  if (C1 == 64)
    hipLaunchKernelGGL((sa_l1_fwd_kernel<64>), dim3(NPART), dim3(256), 0, s, Y, feat, xyz, new_xyz, idx, W1, ldw, rdiv, Np, N, S, R, z1, part);
  else
    hipLaunchKernelGGL((sa_l1_fwd_kernel<128>), dim3(NPART), dim3(256), 0, s, Y, feat, xyz, new_xyz, idx, W1, ldw, rdiv, Np, N, S, R, z1, part);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// The same pass WITHOUT the z1 store: the BatchNorm sums of the first layer and rel4 [R][4] = each grouped row's four inputs
// (relative x, y, z / rdiv, inline feature or 0).  The later passes rebuild z1 from rel4 and W1 (spacap_sa_*_l1in_f32): 16 bytes
// per row instead of 4 C1 written once and read three times.  Modules whose first layer also has point features (Y) keep z1.
extern "C" int spacap_sa_l1_stats_f32(const float *feat, const float *xyz, const float *new_xyz, const int32_t *idx, const float *W1,
                                      int ldw, float rdiv, int B, int Np, int N, int S, int C1, float *rel4, double *part,
                                      spacap_stream_t stream) {
  const char *what = "spacap_sa_l1_stats_f32";
  SPACAP_REQUIRE(B >= 1 && Np >= 1 && N >= 1 && S >= 1 && S <= 255, "%s: bad sizes", what);
  SPACAP_REQUIRE(C1 == 64, "%s: C1=%d unsupported", what, C1);
  SPACAP_REQUIRE(xyz && new_xyz && idx && W1 && rel4 && part && ldw >= (feat ? 4 : 3) && rdiv > 0.f, "%s: bad arguments", what);
  const long R = (long)B * N * S;
  hipLaunchKernelGGL((sa_l1_fwd_kernel<64>), dim3(NPART), dim3(256), 0, spacap::as_stream(stream), (const float *)nullptr, feat, xyz,
                     new_xyz, idx, W1, ldw, rdiv, Np, N, S, R, (float *)nullptr, part, rel4);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// The same pass in closed form (sa_l1_moments_kernel): rel4 f32 [B*N*S, 4] and the moments of the rows' four inputs,
// mom f64 [spacap_sa_nparts()][16]; spacap_sa_l1_moments_finalize_f32 turns them into the first layer's statistics.
extern "C" int spacap_sa_l1_moments_f32(const float *feat, const float *xyz, const float *new_xyz, const int32_t *idx, float rdiv, int B,
                                        int Np, int N, int S, float *rel4, double *mom, spacap_stream_t stream) {
  const char *what = "spacap_sa_l1_moments_f32";
  SPACAP_REQUIRE(B >= 1 && Np >= 1 && N >= 1 && S >= 1 && S <= 255, "%s: bad sizes", what);
  SPACAP_REQUIRE(xyz && new_xyz && idx && rel4 && mom && rdiv > 0.f && (reinterpret_cast<uintptr_t>(rel4) & 15) == 0, "%s: bad arguments", what);
  const long R = (long)B * N * S;
  hipLaunchKernelGGL(sa_l1_moments_kernel, dim3(NPART), dim3(256), 0, spacap::as_stream(stream), feat, xyz, new_xyz, idx, rdiv, Np, N, S, R,
                     rel4, mom);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_sa_l1_moments_finalize_f32(const double *mom, const float *W1, int ldw, int has_feat, int C1, long count, float eps,
                                                 float momentum, const float *gamma, const float *beta, float *running_mean,
                                                 float *running_var, float *stats, spacap_stream_t stream) {
  const char *what = "spacap_sa_l1_moments_finalize_f32";
  SPACAP_REQUIRE(mom && W1 && gamma && beta && stats && C1 >= 1 && C1 <= 1024 && count >= 1 && ldw >= (has_feat ? 4 : 3), "%s: bad arguments", what);
  hipLaunchKernelGGL(sa_l1_moments_finalize_kernel, dim3(1), dim3(1024), 0, spacap::as_stream(stream), mom, NPART, W1, ldw, has_feat, C1,
                     (double)count, eps, momentum, gamma, beta, running_mean, running_var, stats);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// Second layer (64 -> 64) reading rel4 instead of z1: z2 = relu(bn1(W1 in)) W2^T, BatchNorm sums of z2 in `part`
extern "C" int spacap_sa_mid_fwd_l1in_f32(const float *rel4, const float *W1, int ldw, int has_feat, const float *st_in, const float *W,
                                          long R, float *zout, double *part, spacap_stream_t stream) {
  const char *what = "spacap_sa_mid_fwd_l1in_f32";
  SPACAP_REQUIRE(rel4 && W1 && st_in && W && zout && part && R >= 1 && ldw >= (has_feat ? 4 : 3), "%s: bad arguments", what);
  const size_t lds = (size_t)TM * ((64 + 8) + (64 + 4)) * sizeof(float);
  const long tiles = (R + TM - 1) / TM;
  static const int res = resident_blocks(sa_mid_fwd_kernel<64, 1, false>, lds);
  hipLaunchKernelGGL((sa_mid_fwd_kernel<64, 1, false>), dim3(grid_rows(res, 1, tiles), 1), dim3(256), lds, spacap::as_stream(stream),
                     rel4, st_in, W, 64, R, zout, part, TailArgs{}, L1In{W1, ldw, has_feat});
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// Weight gradient of the second layer (64 x 64, dense) with a_prev = relu(bn1(W1 in)) rebuilt from rel4
extern "C" int spacap_sa_wgrad_l1in_f32(const float *dy, const float *zk, const float *coef, const float *rel4, const float *W1, int ldw,
                                        int has_feat, const float *st_p, long R, float *partW, spacap_stream_t stream) {
  const char *what = "spacap_sa_wgrad_l1in_f32";
  SPACAP_REQUIRE(dy && zk && coef && rel4 && W1 && st_p && partW && R >= 1 && ldw >= (has_feat ? 4 : 3), "%s: bad arguments", what);
  const int nslab = spacap_sa_wgrad_slabs(R, 64, 64, 0);
  hipLaunchKernelGGL((sa_wgrad_kernel<64, 64, false>), dim3(nslab, 1), dim3(256), 0, spacap::as_stream(stream), dy, (const uint8_t *)nullptr,
                     0, zk, coef, 64, rel4, st_p, R, partW, L1In{W1, ldw, has_feat});
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_sa_bn_finalize_f32(const double *part, int C, long count, float eps, float momentum,
                                         const float *gamma, const float *beta, float *running_mean,
                                         float *running_var, float *stats, spacap_stream_t stream) {
  const char *what = "spacap_sa_bn_finalize_f32";
  SPACAP_REQUIRE(part && gamma && beta && stats && C >= 1 && count >= 1, "%s: bad arguments", what);
  hipLaunchKernelGGL(sa_bn_finalize_kernel, dim3((C + 7) / 8), dim3(1024), 0, spacap::as_stream(stream), part, NPART, C,
                     (double)count, eps, momentum, gamma, beta, running_mean, running_var, stats);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_sa_bwd_finalize_f32(const double *part, int C, long count, const float *stats, float *coef,
                                          float *dgamma, float *dbeta, spacap_stream_t stream) {
  const char *what = "spacap_sa_bwd_finalize_f32";
  SPACAP_REQUIRE(part && stats && coef && dgamma && dbeta && C >= 1 && count >= 1, "%s: bad arguments", what);
  hipLaunchKernelGGL(sa_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(1024), 0, spacap::as_stream(stream), part, NPART, C,
                     (double)count, stats, coef, dgamma, dbeta);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_sa_mid_fwd_f32(const float *zin, const float *st_in, const float *W, long R, int Cin, int Cout,
                                     float *zout, double *part, spacap_stream_t stream) {
  const char *what = "spacap_sa_mid_fwd_f32";
  SPACAP_REQUIRE(zin && st_in && W && zout && part && R >= 1, "%s: bad arguments", what);
  hipStream_t s = spacap::as_stream(stream);
  const int nt = (Cin == 64 && Cout == 64) ? 1 : 2;
  // default: the streaming split-bf16 kernel (fp32-equivalent products); SPACAP_SA_F32MFMA=1 (the
  // library's one switch) keeps every shared-MLP product on the fp32-MFMA kernels
  if (!f32_mfma_only() && (Cin == 64 || Cin == 128) && Cout % 128 == 0) {
    static const int cus = resident_blocks(sa_mid_fwd_bf3s_kernel<128, 1>, 100 * 1024);   // = CUs: one workgroup per CU
    const size_t ldss = bf3s_lds_bytes(Cin);
    const long wtiles = (R + 31) / 32;
    const int gy = Cout / 128;
    long gx = (cus - reserved_cus()) / gy;
    gx = gx > NPART ? NPART : gx;
    gx = gx > (wtiles + 7) / 8 ? (wtiles + 7) / 8 : gx;
    if (Cin == 64)
      hipLaunchKernelGGL((sa_mid_fwd_bf3s_kernel<64, 1>), dim3((unsigned)gx, gy), dim3(512), ldss, s, zin, st_in, W, Cout, R, zout,
                         part, PoolArgs{});
    else
      hipLaunchKernelGGL((sa_mid_fwd_bf3s_kernel<128, 1>), dim3((unsigned)gx, gy), dim3(512), ldss, s, zin, st_in, W, Cout, R, zout,
                         part, PoolArgs{});
    SPACAP_CHECK_LAUNCH(what);
    return SPACAP_OK;
  }
  const size_t lds = (size_t)TM * ((Cin + 8) + (64 * nt + 4)) * sizeof(float);
  const long tiles = (R + TM - 1) / TM;
#define MF(CI, NTV, GY)                                                                                              \
  {                                                                                                                  \
    static const int res = resident_blocks(sa_mid_fwd_kernel<CI, NTV, false>, lds);                                  \
    hipLaunchKernelGGL((sa_mid_fwd_kernel<CI, NTV, false>), dim3(grid_rows(res, GY, tiles), GY), dim3(256), lds, s,  \
                       zin, st_in, W, Cout, R, zout, part, TailArgs{});                                              \
  }
  if (Cin == 64 && Cout == 64) MF(64, 1, 1)
  else if (Cin == 64 && Cout % 128 == 0) MF(64, 2, Cout / 128)
  else if (Cin == 128 && Cout % 128 == 0) MF(128, 2, Cout / 128)
  else
    SPACAP_REQUIRE(false, "%s: (Cin=%d, Cout=%d) unsupported", what, Cin, Cout);
#undef MF
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ---- pooling from the candidates the streaming layer kernel left (PoolArgs in sa_bf3.inc) -----------------------------
// per (group, channel): merge the sub-groups' candidate lists (S = 64: two halves, the second with indices + 32), evaluate the
// activation of the best and of the runner-up, and keep the reference's first maximum:
//   equal activations -> smaller row index;  activation 0 (or a constant channel) -> row 0, as a strict ">" scan from -1 does.
__global__ __launch_bounds__(256) void sa_pool_finalize_kernel(const float *__restrict__ cand_v, const uint8_t *__restrict__ cand_i,
                                                               const float *__restrict__ st, const float *__restrict__ gamma,
                                                               long G, int S, int C, float *__restrict__ out,
                                                               uint8_t *__restrict__ arg, float *__restrict__ zmax) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= G * C) return;
  const long g = i / C;
  const int c = (int)(i % C);
  const int nsub = S > 32 ? S / 32 : 1;
  size_t o = ((size_t)(g * nsub) * C + c) * 2;
  float t1 = cand_v[o], t2 = cand_v[o + 1];
  int i1 = cand_i[o], i2 = cand_i[o + 1];
  for (int h = 1; h < nsub; ++h) {   // later sub-groups: all their indices are larger, ties go to what we have
    o = ((size_t)(g * nsub + h) * C + c) * 2;
    const float u1 = cand_v[o], u2 = cand_v[o + 1];
    const int j1 = cand_i[o] + 32 * h, j2 = cand_i[o + 1] + 32 * h;
    const bool mine = !(u1 > t1);
    const float w1 = mine ? t1 : u1, l1 = mine ? u1 : t1;
    const int wi = mine ? i1 : j1, li = mine ? j1 : i1;
    float c2 = l1 < w1 ? l1 : -INFINITY;
    int ci = li;
    if (t2 > c2 || (t2 == c2 && i2 < ci)) c2 = t2, ci = i2;
    if (u2 > c2 || (u2 == c2 && j2 < ci)) c2 = u2, ci = j2;
    t1 = w1, i1 = wi, t2 = c2, i2 = ci;
  }
  const float *s = st + (size_t)c * 4;
  const float mean = s[0], sc = s[2], be = s[3], sg = gamma[c] >= 0.f ? 1.f : -1.f;
  const float v1 = fmaxf((sg * t1 - mean) * sc + be, 0.f);
  int a = i1;
  float za = sg * t1;     // the pre-activation of the chosen row
  if (t2 > -INFINITY) {
    const float v2 = fmaxf((sg * t2 - mean) * sc + be, 0.f);
    if (v2 == v1 && i2 < i1) a = i2, za = sg * t2;
  }
  if (!(v1 > 0.f) || sc == 0.f) a = 0;
  out[i] = v1;
  arg[i] = (uint8_t)a;
  // what the pooled layer's BatchNorm backward needs of z (xhat at the arg-max row) when z itself is not stored.  Where the
  // choice fell back to row 0 -- the activation is 0: no gradient; or gamma == 0 exactly: a constant channel -- this is the
  // best row's value, not row 0's (only d gamma of an exactly-zero gamma could tell the difference).
  if (zmax) zmax[i] = za;
}

/* 1 when spacap_sa_mid_fwd_pool_f32 has a kernel for this layer (the streaming split-bf16 kernel is the active one). */
/* out[R][Cout] = x[R][Cin] W[Cout][Cin]^T, fp32 in / out / accumulate, on the streaming split-bf16 kernel (every fp32 product as
   six bf16 matrix products): the relation head's dhid1 = dz2 W2 (models/transformer_captioner.py:319-326 backward), 524 288
   rows.  _supported: 1 for Cin in {64, 128}, Cout a multiple of 128 (and the split kernels not switched off). */
extern "C" int spacap_gemm_rows_supported(int Cin, int Cout) {
  return !f32_mfma_only() && (Cin == 64 || Cin == 128) && Cout % 128 == 0;
}
extern "C" int spacap_gemm_rows_f32(const float *x, const float *W, long R, int Cin, int Cout, float *out,
                                    spacap_stream_t stream) {
  const char *what = "spacap_gemm_rows_f32";
  SPACAP_REQUIRE(x && W && out && R >= 1, "%s: bad arguments", what);
  SPACAP_REQUIRE(spacap_gemm_rows_supported(Cin, Cout), "%s: (Cin=%d, Cout=%d) unsupported", what, Cin, Cout);
  hipStream_t s = spacap::as_stream(stream);
  const size_t ldss = bf3s_lds_bytes(Cin);
  const long wtiles = (R + 31) / 32;
  const int gy = Cout / 128;
  long gx = bwd_cus() / gy;
  gx = gx > NPART ? NPART : gx;
  gx = gx > (wtiles + 7) / 8 ? (wtiles + 7) / 8 : gx;
  if (Cin == 64)
    hipLaunchKernelGGL((sa_mid_fwd_bf3s_kernel<64, 1, false, true>), dim3((unsigned)gx, gy), dim3(512), ldss, s, x,
                       (const float *)nullptr, W, Cout, R, out, (double *)nullptr, PoolArgs{});
  else
    hipLaunchKernelGGL((sa_mid_fwd_bf3s_kernel<128, 1, false, true>), dim3((unsigned)gx, gy), dim3(512), ldss, s, x,
                       (const float *)nullptr, W, Cout, R, out, (double *)nullptr, PoolArgs{});
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

namespace spacap {
int sa_reserved_cus() { return reserved_cus(); }
int device_cus() { return ::device_cus(); }
}  // namespace spacap

/* n CUs are left free by the forward layer kernels' persistent grids (0 <= n <= 64): set by a caller that runs other work
   (the next batch's sampling chain) beside the forward pass. */
extern "C" int spacap_sa_reserve_cus(int n) {
  if (n < 0 || n > 64) return SPACAP_E_INVALID;
  g_reserved_cus.store(n, std::memory_order_relaxed);
  return SPACAP_OK;
}

extern "C" int spacap_sa_mid_fwd_pool_supported(int Cin, int Cout, int S) {
  return !f32_mfma_only() && (Cin == 64 || Cin == 128) && Cout % 128 == 0 && (S == 16 || S == 32 || S == 64);
}

/* spacap_sa_mid_fwd_f32 for the LAST layer of a shared MLP whose output is max-pooled over groups of S consecutive rows:
   also leaves the pooling candidates (cand_v f32 / cand_i u8, [R / min(S,32)][Cout][2]) for spacap_sa_pool_finalize_f32.
   gamma_out: BatchNorm weight of this layer's output (its sign orders the activations). */
extern "C" int spacap_sa_mid_fwd_pool_f32(const float *zin, const float *st_in, const float *W, const float *gamma_out, long R,
                                          int Cin, int Cout, int S, float *zout, double *part, float *cand_v,
                                          uint8_t *cand_i, spacap_stream_t stream) {
  const char *what = "spacap_sa_mid_fwd_pool_f32";
  SPACAP_REQUIRE(zin && st_in && W && gamma_out && part && cand_v && cand_i && R >= 1, "%s: bad arguments", what);   // (zout may be NULL)
  SPACAP_REQUIRE(spacap_sa_mid_fwd_pool_supported(Cin, Cout, S) && R % S == 0, "%s: (Cin=%d, Cout=%d, S=%d) unsupported", what,
                 Cin, Cout, S);
  hipStream_t s = spacap::as_stream(stream);
  static const int cus = resident_blocks(sa_mid_fwd_bf3s_kernel<128, 1>, 100 * 1024);   // = CUs: one workgroup per CU
  const size_t ldss = bf3s_lds_bytes(Cin);
  const long wtiles = (R + 31) / 32;
  const int gy = Cout / 128;
  long gx = (cus - reserved_cus()) / gy;
  gx = gx > NPART ? NPART : gx;
  gx = gx > (wtiles + 7) / 8 ? (wtiles + 7) / 8 : gx;
  const PoolArgs pa{gamma_out, S, cand_v, cand_i};
  if (Cin == 64)
    hipLaunchKernelGGL((sa_mid_fwd_bf3s_kernel<64, 1, true>), dim3((unsigned)gx, gy), dim3(512), ldss, s, zin, st_in, W, Cout,
                       R, zout, part, pa);
  else
    hipLaunchKernelGGL((sa_mid_fwd_bf3s_kernel<128, 1, true>), dim3((unsigned)gx, gy), dim3(512), ldss, s, zin, st_in, W, Cout,
                       R, zout, part, pa);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

/* out[g,c] = max_s relu(bn(z[g*S+s,c])) (first maximum) and its arg, from the candidates of spacap_sa_mid_fwd_pool_f32. */
extern "C" int spacap_sa_pool_finalize_f32(const float *cand_v, const uint8_t *cand_i, const float *stats, const float *gamma,
                                           long G, int S, int C, float *out, uint8_t *arg, float *zmax, spacap_stream_t stream) {
  const char *what = "spacap_sa_pool_finalize_f32";
  SPACAP_REQUIRE(cand_v && cand_i && stats && gamma && out && arg && G >= 1 && C >= 1, "%s: bad arguments", what);
  SPACAP_REQUIRE(S == 16 || S == 32 || S == 64, "%s: S=%d unsupported", what, S);
  hipLaunchKernelGGL(sa_pool_finalize_kernel, dim3(nblocks(G * C, 256)), dim3(256), 0, spacap::as_stream(stream), cand_v, cand_i,
                     stats, gamma, G, S, C, out, arg, zmax);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_sa_pool_fwd_f32(const float *z, const float *stats, long G, int S, int C, float *out,
                                      uint8_t *arg, spacap_stream_t stream) {
  const char *what = "spacap_sa_pool_fwd_f32";
  SPACAP_REQUIRE(z && stats && out && arg && G >= 1 && S >= 1 && S <= 255 && C % 4 == 0, "%s: bad arguments", what);
  hipLaunchKernelGGL(sa_pool_fwd_kernel, dim3(nblocks(G * (C / 4), 256)), dim3(256), 0, spacap::as_stream(stream), z, stats,
                     G, S, C, out, arg);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_sa_pool_bwd_f32(const float *dout, const float *out, const uint8_t *arg, const float *z, const float *zmax,
                                      const float *stats, long G, int S, int C, float *dym, double *part,
                                      spacap_stream_t stream) {
  const char *what = "spacap_sa_pool_bwd_f32";
  SPACAP_REQUIRE(dout && out && arg && (z || zmax) && stats && dym && part && G >= 1, "%s: bad arguments", what);
  SPACAP_REQUIRE(C == 64 || C == 128 || C == 256, "%s: C=%d unsupported", what, C);
  hipLaunchKernelGGL(sa_pool_bwd_kernel, dim3(NPART), dim3(256), 0, spacap::as_stream(stream), dout, out, arg, z, zmax, stats,
                     G, S, C, dym, part);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ---- pooled layer's weight gradient from z2 alone: partial layout + reduction (sa_l3bwd.inc) ------------------------------------
/* floats per workgroup partial of spacap_sa_wgrad_pool_f32: [C3][C2] | [C2][C2] | [C2] */
extern "C" long spacap_sa_l3bwd_part_floats(int C2, int C3) { return (long)C3 * C2 + (long)C2 * C2 + C2; }

/* dW3 [C3][C2] from the workgroups' partials: sums (double scratch, spacap_sa_l3bwd_part_floats entries) then
   dW3 = S + k0 (x) colsum(a2) - diag(k1) W3 Gram. */
extern "C" int spacap_sa_l3bwd_dw_f32(const float *partW, int nparts, const float *coef3, const float *W3, int C3, int C2, double *sums,
                                      float *dW3, spacap_stream_t stream) {
  const char *what = "spacap_sa_l3bwd_dw_f32";
  SPACAP_REQUIRE(partW && coef3 && W3 && sums && dW3 && nparts >= 1 && C3 >= 1 && C2 >= 4 && C2 % 4 == 0, "%s: bad arguments", what);
  SPACAP_REQUIRE((reinterpret_cast<uintptr_t>(partW) & 15) == 0, "%s: partW must be 16-byte aligned", what);
  const long n = spacap_sa_l3bwd_part_floats(C2, C3);
  hipStream_t s = spacap::as_stream(stream);
  SPACAP_REQUIRE(C2 <= 1024 && 1024 % C2 == 0, "%s: C2=%d unsupported", what, C2);
  hipLaunchKernelGGL(sa_l3_sum_kernel, dim3(nblocks(n, 64)), dim3(256), 0, s, partW, nparts, n, sums);
  hipLaunchKernelGGL(sa_l3_dw_kernel, dim3(C3), dim3(1024), 0, s, sums, coef3, W3, C3, C2, dW3);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ---- pooled layer's weight gradient from z2 alone (sa_l3bwd.inc: sa_wgrad_pool_kernel) ------------------------------------
extern "C" int spacap_sa_wgrad_pool_supported(int C2, int C3, int S) { return wgrad_pool_shape(C2, C3, S) ? 1 : 0; }
namespace {
int wgrad_pool_grid(long R, int C2, int S) {
  // C2 = 64: two workgroups of four waves per CU; C2 = 128: one of eight (its partial is 197 KB: fewer, larger partials)
  long g = (long)bwd_cus() * (C2 == 64 ? 2 : 1), tiles = (R + S - 1) / S;
  if (g > NPART) g = NPART;
  if (g > tiles) g = tiles;
  return (int)(g < 1 ? 1 : g);
}
}  // namespace
/* workgroups (= partials, each spacap_sa_l3bwd_part_floats(C2, C3) floats) of spacap_sa_wgrad_pool_f32 */
extern "C" int spacap_sa_wgrad_pool_parts(long R, int C2, int C3, int S) {
  return wgrad_pool_shape(C2, C3, S) && R >= 1 ? wgrad_pool_grid(R, C2, S) : 0;
}
/* partial sums of dW3 = (g d)^T a2 + k0 (x) colsum(a2) - diag(k1) W3 (a2^T a2) of a pooled layer from (dym, arg), z2 and layer
   2's statistics: z3 is not read.  partW [spacap_sa_wgrad_pool_parts][spacap_sa_l3bwd_part_floats]; spacap_sa_l3bwd_dw_f32
   turns it into dW3. */
extern "C" int spacap_sa_wgrad_pool_f32(const float *dym, const uint8_t *arg, int S, const float *coef3, const float *z2,
                                        const float *st2, long R, int C3, int C2, float *partW, spacap_stream_t stream) {
  const char *what = "spacap_sa_wgrad_pool_f32";
  SPACAP_REQUIRE(dym && arg && coef3 && z2 && st2 && partW && R >= 1, "%s: bad arguments", what);
  SPACAP_REQUIRE(wgrad_pool_shape(C2, C3, S) && R % S == 0, "%s: (C2=%d, C3=%d, S=%d) unsupported", what, C2, C3, S);
  SPACAP_REQUIRE((reinterpret_cast<uintptr_t>(arg) & 3) == 0 && ((reinterpret_cast<uintptr_t>(z2) | reinterpret_cast<uintptr_t>(dym)) & 15) == 0,
                 "%s: unaligned pointer (arg: 4 bytes, z2 / dym: 16 bytes)", what);
  const WPArgs a{dym, arg, coef3, z2, st2, R, partW};
  const int grid = wgrad_pool_grid(R, C2, S);
  hipStream_t s = spacap::as_stream(stream);
#define WP(C2V, C3V, SV) \
  hipLaunchKernelGGL((sa_wgrad_pool_kernel<C2V, C3V, SV, SV>), dim3(grid), dim3(C2V * 4), wgrad_pool_lds_bytes(C2V, C3V, SV, SV), s, a)
  if (C2 == 64) WP(64, 128, 64);
  else if (C3 == 128) WP(128, 128, 32);
  else WP(128, 256, 32);
#undef WP
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// dy: dense [R, CK] when arg == NULL, else the masked pooled gradient [R / S, CK] with its arg-max map
extern "C" int spacap_sa_dgrad_f32(const float *dy, const uint8_t *arg, int S, const float *zk, const float *coef,
                                   const float *Wk, const float *zp, const float *st_p, long R, int CK, int CP,
                                   float *dyp, double *part, spacap_stream_t stream) {
  const char *what = "spacap_sa_dgrad_f32";
  SPACAP_REQUIRE(dy && zk && coef && Wk && zp && st_p && dyp && part && R >= 1, "%s: bad arguments", what);
  SPACAP_REQUIRE(!arg || (S >= 1 && R % S == 0), "%s: bad S", what);
  hipStream_t s = spacap::as_stream(stream);
  // default: the streaming split-bf16 kernel (sa_bf3_dgrad.inc); SPACAP_SA_F32MFMA=1: the fp32-MFMA kernels
  if (!f32_mfma_only() && (CK == 128 || CK == 256) && CP % 64 == 0 && R >= 49152) {   // (below: too few tiles per wave to pay for the weight staging)
    static const int cus = resident_blocks(sa_mid_fwd_bf3s_kernel<128, 1>, 100 * 1024);   // = CUs: one workgroup per CU
    const size_t ldsd = bf3s_dgrad_lds_bytes(CK);
    const long wtiles = (R + 31) / 32;
    const int gy = CP / 64;
    long gx = bwd_resident(cus) / gy;
    gx = gx > NPART ? NPART : gx;
    gx = gx > (wtiles + 7) / 8 ? (wtiles + 7) / 8 : gx;
#define DS(CKV, PV)                                                                                                   \
  hipLaunchKernelGGL((sa_dgrad_bf3s_kernel<CKV, PV>), dim3((unsigned)gx, gy), dim3(512), ldsd, s, dy, arg, S, zk, coef, Wk, CP, zp, \
                     st_p, R, dyp, part)
    if (CK == 128) { if (arg) DS(128, true); else DS(128, false); }
    else { if (arg) DS(256, true); else DS(256, false); }
#undef DS
    SPACAP_CHECK_LAUNCH(what);
    return SPACAP_OK;
  }
#define DG(CKV, NTV, PV, PF, AL, GY)                                                                                 \
  {                                                                                                                  \
    const size_t lds = (size_t)TM * ((CKV + 4) + ((AL) ? 0 : (64 * NTV + 4))) * sizeof(float);                       \
    if (lds > 65536)                                                                                                 \
      SPACAP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_dgrad_kernel<CKV, NTV, PV, PF, AL>),   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), what);            \
    static const int res = resident_blocks(sa_dgrad_kernel<CKV, NTV, PV, PF, AL>, lds);                              \
    hipLaunchKernelGGL((sa_dgrad_kernel<CKV, NTV, PV, PF, AL>), dim3(grid_rows(bwd_resident(res), GY, (R + TM - 1) / TM), GY), \
                       dim3(256), lds, s, dy, arg, S, zk, coef, Wk, CP, zp, st_p, R, dyp, part);                      \
  }
  if (arg && CK == 128 && CP == 64) DG(128, 1, true, true, false, 1)
  else if (arg && CK == 256 && CP == 128) DG(256, 1, true, false, true, 2)
  else if (arg && CK == 128 && CP == 128) DG(128, 2, true, true, false, 1)
  else if (!arg && CK == 64 && CP == 64) DG(64, 1, false, true, false, 1)
  else if (!arg && CK == 128 && CP == 128) DG(128, 2, false, true, false, 1)
  else SPACAP_REQUIRE(false, "%s: (CK=%d, CP=%d, pooled=%d) unsupported", what, CK, CP, arg ? 1 : 0);
#undef DG
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// Layer-2 data gradient of an SA module whose first layer reads (rel xyz, one inline feature) directly (SA1):
// as spacap_sa_dgrad_f32 with dense dy [R,64], but dy_prev is not written; instead part_l1 f32
// [spacap_sa_nparts()][64*8+4] receives per-workgroup sums (per channel c: S1[c,0:4], S3[c,0:4]; then S2[0:4]) from which
// the caller forms dW1 = g S1 + k0 S2 - k1 S3 once the layer-1 constants are known.
extern "C" int spacap_sa_dgrad_l1_f32(const float *dy, const float *zk, const float *coef, const float *Wk, const float *zp,
                                      const float *st_p, const float *feat, const float *xyz, const float *new_xyz,
                                      const int32_t *idx, float rdiv, int B, int Np, int N, int S, int CK, int CP,
                                      double *part, float *part_l1, spacap_stream_t stream) {
  const char *what = "spacap_sa_dgrad_l1_f32";
  SPACAP_REQUIRE(dy && zk && coef && Wk && zp && st_p && xyz && new_xyz && idx && part && part_l1 && rdiv > 0.f,
                 "%s: bad arguments", what);
  SPACAP_REQUIRE(CK == 64 && CP == 64, "%s: (CK=%d, CP=%d) unsupported", what, CK, CP);
  const long R = (long)B * N * S;
  hipStream_t s = spacap::as_stream(stream);
  const size_t lds = (size_t)TM * ((64 + 4) + (64 + 4) + 4) * sizeof(float);
  static const int res = resident_blocks(sa_dgrad_kernel<64, 1, false, true, false, true>, lds);
  L1Args L{feat, xyz, new_xyz, idx, rdiv, Np, N, S, part_l1, nullptr, L1In{nullptr, 0, 0}, nullptr};
  hipLaunchKernelGGL((sa_dgrad_kernel<64, 1, false, true, false, true>), dim3(grid_rows(bwd_resident(res), 1, (R + TM - 1) / TM), 1),
                     dim3(256), lds, s, dy, (const uint8_t *)nullptr, 0, zk, coef, Wk, CP, zp, st_p, R, (float *)nullptr, part, L);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// spacap_sa_dgrad_l1_f32 with the rows' inputs read from rel4 and z1 rebuilt from them (nothing of size R x 64 is read but dy, zk)
extern "C" int spacap_sa_dgrad_l1in_f32(const float *dy, const float *zk, const float *coef, const float *Wk, const float *rel4,
                                        const float *W1, int ldw, int has_feat, const float *st_p, int B, int N, int S, double *part,
                                        float *part_l1, spacap_stream_t stream) {
  const char *what = "spacap_sa_dgrad_l1in_f32";
  SPACAP_REQUIRE(dy && zk && coef && Wk && rel4 && W1 && st_p && part && part_l1 && ldw >= (has_feat ? 4 : 3), "%s: bad arguments", what);
  const long R = (long)B * N * S;
  const size_t lds = (size_t)TM * ((64 + 4) + (64 + 4) + 4) * sizeof(float);
  static const int res = resident_blocks(sa_dgrad_kernel<64, 1, false, true, false, true>, lds);
  L1Args L{nullptr, nullptr, nullptr, nullptr, 1.f, 1, N, S, part_l1, rel4, L1In{W1, ldw, has_feat}, nullptr};
  hipLaunchKernelGGL((sa_dgrad_kernel<64, 1, false, true, false, true>), dim3(grid_rows(bwd_resident(res), 1, (R + TM - 1) / TM), 1),
                     dim3(256), lds, spacap::as_stream(stream), dy, (const uint8_t *)nullptr, 0, zk, coef, Wk, 64, (const float *)nullptr,
                     st_p, R, (float *)nullptr, part, L);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// spacap_sa_dgrad_l1in_f32 that ALSO leaves the layer's weight-gradient partials: partW f32 [spacap_sa_dgrad_wgrad_l1in_slabs(R)][64][64],
// summed by the caller in slab order (what spacap_sa_wgrad_l1in_f32 computes from a second pass over dy and zk)
namespace {
int dgrad_wgrad_l1in_grid(long R) {
  const size_t lds = (size_t)TM * ((64 + 4) + (64 + 4) + 4) * sizeof(float);
  static const int res = resident_blocks(sa_dgrad_kernel<64, 1, false, true, false, true, true>, lds);
  return grid_rows(bwd_resident(res), 1, (R + TM - 1) / TM);
}
}  // namespace
extern "C" int spacap_sa_dgrad_wgrad_l1in_slabs(long R) { return R >= 1 ? dgrad_wgrad_l1in_grid(R) : 0; }
extern "C" int spacap_sa_dgrad_wgrad_l1in_f32(const float *dy, const float *zk, const float *coef, const float *Wk, const float *rel4,
                                              const float *W1, int ldw, int has_feat, const float *st_p, int B, int N, int S,
                                              double *part, float *part_l1, float *partW, spacap_stream_t stream) {
  const char *what = "spacap_sa_dgrad_wgrad_l1in_f32";
  SPACAP_REQUIRE(dy && zk && coef && Wk && rel4 && W1 && st_p && part && part_l1 && partW && ldw >= (has_feat ? 4 : 3), "%s: bad arguments", what);
  const long R = (long)B * N * S;
  const size_t lds = (size_t)TM * ((64 + 4) + (64 + 4) + 4) * sizeof(float);
  L1Args L{nullptr, nullptr, nullptr, nullptr, 1.f, 1, N, S, part_l1, rel4, L1In{W1, ldw, has_feat}, partW};
  hipLaunchKernelGGL((sa_dgrad_kernel<64, 1, false, true, false, true, true>), dim3(dgrad_wgrad_l1in_grid(R), 1), dim3(256), lds,
                     spacap::as_stream(stream), dy, (const uint8_t *)nullptr, 0, zk, coef, Wk, 64, (const float *)nullptr, st_p, R,
                     (float *)nullptr, part, L);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// dW1 of the fused first-layer backward from its three sums (L1Args): part_l1 [nparts][C1 * 8 + 4] = per workgroup
// (S1 [C1][4] | S3 [C1][4]) interleaved per channel as [c][2][4], then S2 [4]; coef [C1][4] = (g, k0, k1, .) of layer 1:
//   dW1[c][d] = g[c] S1[c][d] + k0[c] S2[d] - k1[c] S3[c][d],  sums over the partials in double, d < ldw columns written.
namespace {
// one workgroup per channel c: its 12 sums (S1[c][0..3], S3[c][0..3], S2[0..3]) over the partial rows, 16 lanes per sum with
// independent loads (the first version ran the nparts loads of every sum as one dependent chain on ONE workgroup: 262 us),
// lanes combined in a fixed order in double
__global__ __launch_bounds__(256) void sa_l1_dw_kernel(const float *__restrict__ part_l1, int nparts, const float *__restrict__ coef,
                                                       int C1, int ldw, float *__restrict__ dW1) {
  __shared__ double s_lane[12][17];
  __shared__ double s_sum[12];
  const int c = blockIdx.x, tid = threadIdx.x, sidx = tid >> 4, l = tid & 15;
  const int n = C1 * 8 + 4;
  if (sidx < 12) {
    const int e = sidx < 8 ? c * 8 + sidx : C1 * 8 + (sidx - 8);   // [c][2][4] then S2 [4]
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int p = l;
    for (; p + 48 < nparts; p += 64) {
      a0 += (double)part_l1[(size_t)p * n + e];
      a1 += (double)part_l1[(size_t)(p + 16) * n + e];
      a2 += (double)part_l1[(size_t)(p + 32) * n + e];
      a3 += (double)part_l1[(size_t)(p + 48) * n + e];
    }
    for (; p < nparts; p += 16) a0 += (double)part_l1[(size_t)p * n + e];
    s_lane[sidx][l] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  if (tid < 12) {
    double a = 0.0;
    for (int i = 0; i < 16; ++i) a += s_lane[tid][i];
    s_sum[tid] = a;
  }
  __syncthreads();
  if (tid < ldw) {
    const double g = coef[c * 4 + 0], k0 = coef[c * 4 + 1], k1 = coef[c * 4 + 2];
    dW1[c * ldw + tid] = (float)(g * s_sum[tid] + k0 * s_sum[8 + tid] - k1 * s_sum[4 + tid]);
  }
}
}  // namespace
extern "C" int spacap_sa_l1_dw_f32(const float *part_l1, int nparts, const float *coef, int C1, int ldw, float *dW1,
                                   spacap_stream_t stream) {
  const char *what = "spacap_sa_l1_dw_f32";
  SPACAP_REQUIRE(part_l1 && coef && dW1 && nparts >= 1 && C1 >= 1 && C1 <= 512 && ldw >= 1 && ldw <= 4, "%s: bad arguments", what);
  hipLaunchKernelGGL(sa_l1_dw_kernel, dim3(C1), dim3(256), 0, spacap::as_stream(stream), part_l1, nparts, coef, C1, ldw, dW1);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// partW: [spacap_sa_wgrad_slabs(R,CK,CP,pooled)][CK][CP] partial weight gradients, summed by the caller in slab order
extern "C" int spacap_sa_wgrad_f32(const float *dy, const uint8_t *arg, int S, const float *zk, const float *coef,
                                   const float *zp, const float *st_p, long R, int CK, int CP, float *partW,
                                   spacap_stream_t stream) {
  const char *what = "spacap_sa_wgrad_f32";
  SPACAP_REQUIRE(dy && zk && coef && zp && st_p && partW && R >= 1, "%s: bad arguments", what);
  SPACAP_REQUIRE(!arg || (S >= 1 && R % S == 0), "%s: bad S", what);
  hipStream_t s = spacap::as_stream(stream);
  const int nslab = spacap_sa_wgrad_slabs(R, CK, CP, arg ? 1 : 0);
#define WG(CKB, CPV, PV) \
  hipLaunchKernelGGL((sa_wgrad_kernel<CKB, CPV, PV>), dim3(nslab, CK / CKB), dim3(256), 0, s, dy, arg, S, zk, coef, CK, zp, st_p, R, partW)
  if (arg && CK % 128 == 0 && CP == 64) WG(128, 64, true);
  else if (arg && CK % 128 == 0 && CP == 128) WG(128, 128, true);
  else if (!arg && CK == 64 && CP == 64) WG(64, 64, false);
  else if (!arg && CK == 128 && CP == 128) WG(128, 128, false);
  else SPACAP_REQUIRE(false, "%s: (CK=%d, CP=%d, pooled=%d) unsupported", what, CK, CP, arg ? 1 : 0);
#undef WG
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// dy1 is overwritten with dz1 when write_dz != 0 (only the rows -> source points scatter of the feature gradient
// needs it); partW [spacap_sa_nparts()][C1][4] (columns: rel x, y, z, inline feature)
extern "C" int spacap_sa_l1_bwd_f32(float *dy1, const float *z1, const float *coef, const float *feat, const float *xyz,
                                    const float *new_xyz, const int32_t *idx, const float *W1, int ldw, float rdiv, int B,
                                    int Np, int N, int S, int C1, float *partW, float *drel, int write_dz,
                                    spacap_stream_t stream) {
  const char *what = "spacap_sa_l1_bwd_f32";
  SPACAP_REQUIRE(dy1 && z1 && coef && xyz && new_xyz && idx && W1 && partW && rdiv > 0.f, "%s: bad arguments", what);
  SPACAP_REQUIRE(C1 == 64 || C1 == 128, "%s: C1=%d unsupported", what, C1);
  const long R = (long)B * N * S;
  hipStream_t s = spacap::as_stream(stream);
  if (C1 == 64)
    hipLaunchKernelGGL((sa_l1_bwd_kernel<64>), dim3(NPART), dim3(256), 0, s, dy1, z1, coef, feat, xyz, new_xyz, idx, W1, ldw, rdiv, Np, N, S, R, partW, drel, write_dz);
  else
    hipLaunchKernelGGL((sa_l1_bwd_kernel<128>), dim3(NPART), dim3(256), 0, s, dy1, z1, coef, feat, xyz, new_xyz, idx, W1, ldw, rdiv, Np, N, S, R, partW, drel, write_dz);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" size_t spacap_sa_rows_scatter_workspace_bytes(int B, int Np, long E) {
  RowsLayout L;
  return rows_layout(B, Np, E, L) ? L.bytes : 0;
}

// The inverted index of a grouping (which rows reference each source point, ascending): a function of idx alone, i.e.
// of the input coordinates -- a trainer can build it ahead of the step (detector.geometry_pyramid) and the backward
// then only gathers.  `workspace` (spacap_sa_rows_scatter_workspace_bytes) holds the index afterwards.
extern "C" int spacap_sa_rows_index_f32(const int32_t *idx, int B, int Np, long E, void *workspace, spacap_stream_t stream) {
  const char *what = "spacap_sa_rows_index_f32";
  RowsLayout L;
  SPACAP_REQUIRE(idx && workspace && rows_layout(B, Np, E, L), "%s: bad arguments", what);
  hipStream_t s = spacap::as_stream(stream);
  char *ws = reinterpret_cast<char *>(workspace);
  unsigned *keys_in = reinterpret_cast<unsigned *>(ws + L.keys_in), *keys_out = reinterpret_cast<unsigned *>(ws + L.keys_out);
  int *vals_in = reinterpret_cast<int *>(ws + L.vals_in), *vals_out = reinterpret_cast<int *>(ws + L.vals_out);
  int *off = reinterpret_cast<int *>(ws + L.off);
  if (Np <= 4096 && E <= 65535 && B <= 65535) {   // one launch, no sort
    const size_t lds = (size_t)(RI_WAVES * ((Np + 1) / 2) + 2 * ((Np + 1) / 2) + 1024) * 4;
    static unsigned long long lds_ok = 0;
    SPACAP_CHECK_HIP(spacap::allow_dynamic_lds(reinterpret_cast<const void *>(&rows_index_kernel), 160 * 1024, lds_ok), what);
    hipLaunchKernelGGL(rows_index_kernel, dim3(B), dim3(1024), lds, s, idx, Np, (int)E, B, off, vals_out);
    SPACAP_CHECK_LAUNCH(what);
    return SPACAP_OK;
  }
  hipLaunchKernelGGL(rows_keys_kernel, dim3(nblocks((long)L.total, 256)), dim3(256), 0, s, idx, Np, E, (long)L.total, keys_in, vals_in);
  size_t cub_bytes = L.cub_bytes;
  SPACAP_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(ws + L.cub, cub_bytes, keys_in, keys_out, vals_in, vals_out,
                                                      (int)L.total, 0, L.bits, s), what);
  hipLaunchKernelGGL(rows_offsets_kernel, dim3(nblocks((long)L.K + 1, 256)), dim3(256), 0, s, keys_out, (long)L.total, (long)L.K, off);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// out[b, p, :] = sum of dz[r, :] over the rows r = (b, e) with idx[b, e] = p, ascending r, from the index
// spacap_sa_rows_index_f32 left in `workspace` (same B, Np, E)
extern "C" int spacap_sa_rows_gather_f32(const float *dz, int B, int Np, long E, int C, const void *workspace, float *out,
                                         spacap_stream_t stream) {
  const char *what = "spacap_sa_rows_gather_f32";
  RowsLayout L;
  SPACAP_REQUIRE(dz && out && workspace && C % 4 == 0 && rows_layout(B, Np, E, L), "%s: bad arguments", what);
  const char *ws = reinterpret_cast<const char *>(workspace);
  hipLaunchKernelGGL(rows_gather_sum_kernel, dim3(nblocks((long)L.K * (C / 4), 256)), dim3(256), 0, spacap::as_stream(stream), dz,
                     reinterpret_cast<const int *>(ws + L.off), reinterpret_cast<const int *>(ws + L.vals_out), (long)L.K, C, out);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// drel f32 [B, N S, 3] -> dxyz f32 [B, Np, 3] (sum over the rows that reference each source point, ascending, from the index
// spacap_sa_rows_index_f32 left in `workspace`; may be NULL) and dnew f32 [B, N, 3] = - sum over each group's S rows (may be NULL)
extern "C" int spacap_sa_drel_sums_f32(const float *drel, int B, int Np, int N, int S, const void *workspace, float *dxyz, float *dnew,
                                       spacap_stream_t stream) {
  const char *what = "spacap_sa_drel_sums_f32";
  RowsLayout L;
  SPACAP_REQUIRE(drel && (dxyz || dnew) && N >= 1 && S >= 1 && rows_layout(B, Np, (long)N * S, L) && (!dxyz || workspace),
                 "%s: bad arguments", what);
  const char *ws = reinterpret_cast<const char *>(workspace);
  const long K = (long)L.K, G = (long)B * N;
  hipLaunchKernelGGL(sa_drel_sums_kernel, dim3(nblocks(K + G, 256)), dim3(256), 0, spacap::as_stream(stream), drel,
                     ws ? reinterpret_cast<const int *>(ws + L.off) : nullptr, ws ? reinterpret_cast<const int *>(ws + L.vals_out) : nullptr,
                     K, G, S, dxyz, dnew);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// dW1 f32 [C1, 3 + Cf] from pw1 f32 [n1][C1][4] (spacap_sa_l1_bwd_f32) and pf f32 [nf][C1 * Cf] (spacap_linear_wgrad_f32 of the
// feature product): columns 0..2 and 3.. in one launch, the values of spacap_sum_slabs_f32 on each + a concatenation
extern "C" int spacap_sa_dw1_assemble_f32(const float *pw1, int n1, const float *pf, int nf, int C1, int Cf, float *dW1,
                                          spacap_stream_t stream) {
  const char *what = "spacap_sa_dw1_assemble_f32";
  SPACAP_REQUIRE(pw1 && pf && dW1 && n1 >= 1 && nf >= 1 && C1 >= 1 && Cf >= 1, "%s: bad arguments", what);
  hipLaunchKernelGGL(sa_dw1_assemble_kernel, dim3(nblocks((long)C1 * (3 + Cf), 64)), dim3(256), 0, spacap::as_stream(stream), pw1, n1, pf,
                     nf, C1, Cf, dW1);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// out[b, p, :] = sum of dz[r, :] over the rows r = (b, e) with idx[b, e] = p, ascending r (E = rows per scene)
extern "C" int spacap_sa_rows_scatter_f32(const float *dz, const int32_t *idx, int B, int Np, long E, int C, float *out,
                                          void *workspace, spacap_stream_t stream) {
  SPACAP_REQUIRE(dz && out && C % 4 == 0, "spacap_sa_rows_scatter_f32: bad arguments");
  const int rc = spacap_sa_rows_index_f32(idx, B, Np, E, workspace, stream);
  return rc ? rc : spacap_sa_rows_gather_f32(dz, B, Np, E, C, workspace, out, stream);
}

// ===========================================================================================================
// Weight + bias gradient of a Linear layer:  dW[ck, cp] = sum_r g[r, ck] x[r, cp],  db[ck] = sum_r g[r, ck]
// (torch.nn.Linear backward: models/transformer_captioner.py's projections and feed-forward layers).  The BLAS
// path runs these [<= 2048 rows] x [128..2048]^2 reductions as a memset + a split-K GEMM + a separate column-sum
// kernel (30 us of mostly latency for 67 MFLOP); here one launch produces per-slab partials of both (the bias
// gradient falls out of the same staged tile as one more MFMA column against a constant 1), summed in slab order
// by the caller.
namespace {
// (bx, gx): slab index / number of slabs; by, bz: 128-wide blocks of CK and CP
template <bool WITH_BIAS>
__device__ __forceinline__ void linear_wgrad_body(const float *__restrict__ g, const float *__restrict__ x, int CK, int CP,
                                                  long R, float *__restrict__ part, int bx, int by, int bz, int gx) {
  constexpr int CB = 128, LDG = CB + 16;
  __shared__ __attribute__((aligned(16))) float s_g[TW * LDG];
  __shared__ __attribute__((aligned(16))) float s_x[TW * LDG];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int ck0 = by * CB, cp0 = bz * CB;
  const int c4 = tid & 31, r0 = tid >> 5;  // 32 float4 per 128-wide row, 8 rows per pass
  f32x4 acc[2][8], accb[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const long ntiles = (R + TW - 1) / TW;
  // register pipeline: the next tile's rows are in flight while this tile's MFMAs run (a workgroup per CU has nobody
  // else to hide the load latency behind)
  f32x4 pa[TW / 8], pb[TW / 8];
  auto fetch = [&](long t) {
    const long row0 = t * TW;
#pragma unroll
    for (int i = 0; i < TW / 8; ++i) {
      const int row = r0 + 8 * i;
      pa[i] = pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < ntiles && row0 + row < R) {
        pa[i] = ld4(g + (size_t)(row0 + row) * CK + ck0 + c4 * 4);
        pb[i] = ld4(x + (size_t)(row0 + row) * CP + cp0 + c4 * 4);
      }
    }
  };
  fetch(bx);
  for (long t = bx; t < ntiles; t += gx) {
#pragma unroll
    for (int i = 0; i < TW / 8; ++i) {
      const int row = r0 + 8 * i;
      st4(&s_g[row * LDG + c4 * 4], pa[i]);
      st4(&s_x[row * LDG + c4 * 4], pb[i]);
    }
    __syncthreads();
    fetch(t + gx);
#pragma unroll
    for (int ks = 0; ks < TW / 4; ++ks) {
      float af[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) af[m] = s_g[(ks * 4 + lg) * LDG + (w * 2 + m) * 16 + l15];
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const float b = s_x[(ks * 4 + lg) * LDG + n * 16 + l15];
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][n] = MFMA16(af[m], b, acc[m][n]);
      }
      if (WITH_BIAS) {
#pragma unroll
        for (int m = 0; m < 2; ++m) accb[m] = MFMA16(af[m], 1.0f, accb[m]);
      }
    }
    __syncthreads();
  }
  // partial layout per slab: [CK][CP] weights, then [CK] bias
  float *o = part + (size_t)bx * ((size_t)CK * CP + (WITH_BIAS ? CK : 0));
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        o[(size_t)(ck0 + (w * 2 + m) * 16 + 4 * lg + u) * CP + cp0 + n * 16 + l15] = acc[m][n][u];
  if (WITH_BIAS && bz == 0 && l15 == 0) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int u = 0; u < 4; ++u) o[(size_t)CK * CP + ck0 + (w * 2 + m) * 16 + 4 * lg + u] = accb[m][u];
  }
}
template <bool WITH_BIAS>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float *__restrict__ g, const float *__restrict__ x, int CK,
                                                           int CP, long R, float *__restrict__ part) {
  linear_wgrad_body<WITH_BIAS>(g, x, CK, CP, R, part, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// Many weight gradients in ONE launch: the backward of a training step produces ~50 of them, most far too small to
// fill the chip (the decoder's: 8 slabs x 1 - 16 blocks) and each a launch of its own; nothing but the optimizer
// reads them, so they can all run together when the backward is over.  Job table by value in the kernel arguments
// (hipGraph-capturable as it is); a workgroup finds its job by binary search over the first-block prefix.
constexpr int WG_JOB_MAX = 72;
struct WgradJob {
  const float *g, *x;
  float *part;
  long R;
  int CK, CP, gx, gy, with_bias, block0;
};
struct WgradTable {
  int njobs, pad;
  WgradJob job[WG_JOB_MAX];
};
#include "wgrad_bf3.inc"
// (the split-bf16 body, wgrad_bf3.inc: the default; SPACAP_SA_F32MFMA=1 keeps the fp32-MFMA body)
__global__ __launch_bounds__(256) void linear_wgrad_bf3_batched_kernel(const WgradTable T) {
  int lo = 0, hi = T.njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (T.job[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WgradJob J = T.job[lo];
  const int local = (int)blockIdx.x - J.block0;
  const int bx = local % J.gx, by = (local / J.gx) % J.gy, bz = local / (J.gx * J.gy);
  linear_wgrad_bf3_body(J.with_bias != 0, J.g, J.x, J.CK, J.CP, J.R, J.part, bx, by, bz, J.gx);
}
__global__ __launch_bounds__(256) void linear_wgrad_batched_kernel(const WgradTable T) {
  int lo = 0, hi = T.njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (T.job[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WgradJob J = T.job[lo];
  const int local = (int)blockIdx.x - J.block0;
  const int bx = local % J.gx, by = (local / J.gx) % J.gy, bz = local / (J.gx * J.gy);
  if (J.with_bias) linear_wgrad_body<true>(J.g, J.x, J.CK, J.CP, J.R, J.part, bx, by, bz, J.gx);
  else linear_wgrad_body<false>(J.g, J.x, J.CK, J.CP, J.R, J.part, bx, by, bz, J.gx);
}
}  // namespace

// number of row slabs (= partial results) for a (rows, CK, CP) problem; 0 when the shape has no kernel
extern "C" int spacap_linear_wgrad_slabs(long R, int CK, int CP) {
  if (R < 1 || CK < 128 || CP < 128 || CK % 128 || CP % 128) return 0;
  const long tiles = (R + TW - 1) / TW, yz = (long)(CK / 128) * (CP / 128);
  long n = 1024 / yz, cap = (4L << 20) / ((long)CK * CP);
  if (n > cap) n = cap;
  if (n > tiles) n = tiles;
  return (int)(n < 1 ? 1 : n);
}

// part f32 [spacap_linear_wgrad_slabs(R,CK,CP)][CK*CP (+ CK when with_bias)]
extern "C" int spacap_linear_wgrad_f32(const float *g, const float *x, long R, int CK, int CP, int with_bias, float *part,
                                       spacap_stream_t stream) {
  const char *what = "spacap_linear_wgrad_f32";
  const int nslab = spacap_linear_wgrad_slabs(R, CK, CP);
  SPACAP_REQUIRE(nslab > 0, "%s: (R=%ld, CK=%d, CP=%d) unsupported", what, R, CK, CP);
  SPACAP_REQUIRE(g && x && part, "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
  const dim3 grid(nslab, CK / 128, CP / 128);
  if (!f32_mfma_only()) {
    if (with_bias) hipLaunchKernelGGL((linear_wgrad_bf3_kernel<true>), grid, dim3(256), 0, s, g, x, CK, CP, R, part);
    else hipLaunchKernelGGL((linear_wgrad_bf3_kernel<false>), grid, dim3(256), 0, s, g, x, CK, CP, R, part);
  } else if (with_bias) hipLaunchKernelGGL((linear_wgrad_kernel<true>), grid, dim3(256), 0, s, g, x, CK, CP, R, part);
  else hipLaunchKernelGGL((linear_wgrad_kernel<false>), grid, dim3(256), 0, s, g, x, CK, CP, R, part);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// The same with the number of row slabs chosen by the caller (1 <= nslab <= row tiles): part f32 [nslab][CK*CP (+ CK)].  For the
// wide relation head of the stress configuration (4.2 M pair rows x 512 x 512): 64 slabs fill the chip twice over, where
// spacap_linear_wgrad_slabs' 4 M-element cap on a partial set would leave it 16.
extern "C" int spacap_linear_wgrad_nslab_f32(const float *g, const float *x, long R, int CK, int CP, int with_bias, int nslab, float *part,
                                             spacap_stream_t stream) {
  const char *what = "spacap_linear_wgrad_nslab_f32";
  SPACAP_REQUIRE(spacap_linear_wgrad_slabs(R, CK, CP) > 0 && nslab >= 1 && nslab <= 65535 && nslab <= (R + TW - 1) / TW,
                 "%s: (R=%ld, CK=%d, CP=%d, nslab=%d) unsupported", what, R, CK, CP, nslab);
  SPACAP_REQUIRE(g && x && part, "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
  const dim3 grid(nslab, CK / 128, CP / 128);
  if (!f32_mfma_only()) {
    if (with_bias) hipLaunchKernelGGL((linear_wgrad_bf3_kernel<true>), grid, dim3(256), 0, s, g, x, CK, CP, R, part);
    else hipLaunchKernelGGL((linear_wgrad_bf3_kernel<false>), grid, dim3(256), 0, s, g, x, CK, CP, R, part);
  } else if (with_bias) hipLaunchKernelGGL((linear_wgrad_kernel<true>), grid, dim3(256), 0, s, g, x, CK, CP, R, part);
  else hipLaunchKernelGGL((linear_wgrad_kernel<false>), grid, dim3(256), 0, s, g, x, CK, CP, R, part);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// njobs independent weight gradients in one launch (same values as njobs calls of spacap_linear_wgrad_f32 with the
// same arguments when nslabs[i] = spacap_linear_wgrad_slabs(...); any other slab count only changes how the rows are
// grouped).  All arrays are HOST arrays, read before the call returns; part[i] holds nslabs[i] partial results.
// slabs per job when many jobs share one launch: the batch fills the chip, so a workgroup can take 8 row tiles
// (fewer partial results to write and to add up; one slab = the result itself for the decoder's 256 rows)
extern "C" int spacap_linear_wgrad_slabs_batched(long R, int CK, int CP) {
  const int single = spacap_linear_wgrad_slabs(R, CK, CP);
  if (single == 0) return 0;
  const long tiles = (R + TW - 1) / TW;
  long n = tiles / 8;
  if (n < 1) n = 1;
  return (int)(n < single ? n : single);
}

extern "C" int spacap_linear_wgrad_batched_f32(const float *const *g, const float *const *x, const long *R, const int *CK,
                                               const int *CP, const int *with_bias, const int *nslabs, float *const *part,
                                               int njobs, spacap_stream_t stream) {
  const char *what = "spacap_linear_wgrad_batched_f32";
  SPACAP_REQUIRE(njobs >= 0 && (njobs == 0 || (g && x && R && CK && CP && with_bias && nslabs && part)), "%s: bad arguments",
                 what);
  hipStream_t s = spacap::as_stream(stream);
  int i = 0;
  while (i < njobs) {
    WgradTable T;
    T.njobs = 0, T.pad = 0;
    long blocks = 0;
    for (; i < njobs && T.njobs < WG_JOB_MAX; ++i) {
      const int nslab = nslabs[i];
      SPACAP_REQUIRE(spacap_linear_wgrad_slabs(R[i], CK[i], CP[i]) > 0 && nslab >= 1 && g[i] && x[i] && part[i],
                     "%s: job %d: (R=%ld, CK=%d, CP=%d, slabs=%d) unsupported or null pointer", what, i, R[i], CK[i], CP[i], nslab);
      WgradJob &J = T.job[T.njobs++];
      J.g = g[i], J.x = x[i], J.part = part[i], J.R = R[i], J.CK = CK[i], J.CP = CP[i];
      J.gx = nslab, J.gy = CK[i] / 128, J.with_bias = with_bias[i], J.block0 = (int)blocks;
      blocks += (long)nslab * (CK[i] / 128) * (CP[i] / 128);
      SPACAP_REQUIRE(blocks < 2147483647L, "%s: too many blocks", what);
    }
    if (!f32_mfma_only()) hipLaunchKernelGGL(linear_wgrad_bf3_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, s, T);
    else hipLaunchKernelGGL(linear_wgrad_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, s, T);
  }
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ===========================================================================================================
// Weight gradient of a 1x1 convolution on CHANNEL-MAJOR tensors (the vote net and the feature-propagation MLPs:
// models/voting_module.py:33-60, lib/pointnet2/pointnet2_modules.py:376-421; Conv1d/Conv2d k = 1 on (B, C, N)):
//   dW[co, ci] = sum_b sum_n g[b, co, n] x[b, ci, n]
// The convolution library runs this as an implicit-GEMM weight-gradient kernel (46 - 60 us for 256 x 256 over
// 8 x 1 024 points) or as one small GEMM per scene; here both operands are read as [channel][32 points] panels
// (contiguous along n), each (scene, point range) slab accumulates a 128 x 128 block by MFMA and writes a partial
// result; the caller adds the slabs in order (spacap_sum_slabs_f32).
namespace {
// with_bias: the partial row is [CO * CI | CO rounded up to 4] and its tail receives db[co] = sum over the slab's points of g
// (a column of ones beside x; written by the workgroups of the first input-channel block)
__device__ __forceinline__ void conv1x1_wgrad_body(const float *__restrict__ g, const float *__restrict__ x, int CO, int CI,
                                                   int N, int nsplit, float *__restrict__ part, int bx, int by, int bz,
                                                   int with_bias = 0) {
  constexpr int CB = 128, KT = 32, LDK = KT + 4;
  __shared__ __attribute__((aligned(16))) float s_g[CB * LDK];
  __shared__ __attribute__((aligned(16))) float s_x[CB * LDK];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int b = bx / nsplit, sl = bx % nsplit;
  const int co0 = by * CB, ci0 = bz * CB;
  const int tiles = N / KT, t_begin = (int)((long)tiles * sl / nsplit), t_end = (int)((long)tiles * (sl + 1) / nsplit);
  const float *gb = g + ((size_t)b * CO + co0) * N, *xb = x + ((size_t)b * CI + ci0) * N;
  const int k4 = tid & 7, c0 = tid >> 3;   // 8 float4 per 32-point row, 32 channels per pass
  f32x4 acc[2][8];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const bool wb = with_bias && bz == 0;
  for (int t = t_begin; t < t_end; ++t) {
    const int n0 = t * KT;
#pragma unroll
    for (int i = 0; i < CB / 32; ++i) {
      const int c = c0 + 32 * i;   // (channels past the end re-read the last one: their products are never stored)
      st4(&s_g[c * LDK + k4 * 4], ld4(gb + (size_t)min(c, CO - 1 - co0) * N + n0 + k4 * 4));
      st4(&s_x[c * LDK + k4 * 4], ld4(xb + (size_t)min(c, CI - 1 - ci0) * N + n0 + k4 * 4));
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KT / 4; ++ks) {
      float af[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) af[m] = s_g[((w * 2 + m) * 16 + l15) * LDK + ks * 4 + lg];
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const float bb = s_x[(n * 16 + l15) * LDK + ks * 4 + lg];
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][n] = MFMA16(af[m], bb, acc[m][n]);
      }
      if (wb) {
#pragma unroll
        for (int m = 0; m < 2; ++m) accb[m] = MFMA16(af[m], 1.0f, accb[m]);
      }
    }
    __syncthreads();
  }
  float *o = part + (size_t)bx * ((size_t)CO * CI + (with_bias ? (size_t)((CO + 3) & ~3) : 0));
  if (wb && l15 == 0) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = co0 + (w * 2 + m) * 16 + 4 * lg + u;
        if (row < ((CO + 3) & ~3)) o[(size_t)CO * CI + row] = row < CO ? accb[m][u] : 0.f;
      }
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = co0 + (w * 2 + m) * 16 + 4 * lg + u, col = ci0 + n * 16 + l15;
        if (row < CO && col < CI) o[(size_t)row * CI + col] = acc[m][n][u];
      }
}

__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const float *__restrict__ g, const float *__restrict__ x, int CO,
                                                            int CI, int N, int nsplit, float *__restrict__ part) {
  conv1x1_wgrad_body(g, x, CO, CI, N, nsplit, part, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several 1x1-convolution weight gradients in one launch (end of a backward pass: see linear_wgrad_batched_kernel)
constexpr int CV_JOB_MAX = 64;
struct ConvJob {
  const float *g, *x;
  float *part;
  int CO, CI, N, nsplit, gx, gy, block0, with_bias;
};
struct ConvTable {
  int njobs, pad;
  ConvJob job[CV_JOB_MAX];
};
__global__ __launch_bounds__(256) void conv1x1_wgrad_batched_kernel(const ConvTable T) {
  int lo = 0, hi = T.njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (T.job[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ConvJob J = T.job[lo];
  const int local = (int)blockIdx.x - J.block0;
  conv1x1_wgrad_body(J.g, J.x, J.CO, J.CI, J.N, J.nsplit, J.part, local % J.gx, (local / J.gx) % J.gy, local / (J.gx * J.gy),
                     J.with_bias);
}

// (the split-bf16 bodies of wgrad_bf3.inc: the default; SPACAP_SA_F32MFMA=1 keeps the fp32-MFMA ones)
__global__ __launch_bounds__(256) void conv1x1_wgrad_bf3_kernel(const float *__restrict__ g, const float *__restrict__ x, int CO,
                                                                int CI, int N, int nsplit, float *__restrict__ part) {
  conv1x1_wgrad_bf3_body(g, x, CO, CI, N, nsplit, part, blockIdx.x, blockIdx.y, blockIdx.z, 0);
}
__global__ __launch_bounds__(256) void conv1x1_wgrad_bf3_batched_kernel(const ConvTable T) {
  int lo = 0, hi = T.njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (T.job[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ConvJob J = T.job[lo];
  const int local = (int)blockIdx.x - J.block0;
  conv1x1_wgrad_bf3_body(J.g, J.x, J.CO, J.CI, J.N, J.nsplit, J.part, local % J.gx, (local / J.gx) % J.gy, local / (J.gx * J.gy),
                         J.with_bias);
}

inline int conv1x1_nsplit(int B, int CO, int CI, int N) {
  const long yz = (long)((CO + 127) / 128) * ((CI + 127) / 128), tiles = N / 32;
  long n = 512 / (yz * B), cap = (4L << 20) / ((long)CO * CI * B);
  if (n > cap) n = cap;
  if (n > tiles) n = tiles;
  return (int)(n < 1 ? 1 : n);
}
}  // namespace

// number of partial results (= B x point ranges) for a (B, CO, CI, N) problem; 0 when the shape has no kernel
extern "C" int spacap_conv1x1_wgrad_slabs(int B, int CO, int CI, int N) {
  if (B < 1 || N < 32 || N % 32 || CO < 1 || CI < 1) return 0;   // (any widths: 128 x 128 tiles with clamped tails)
  return B * conv1x1_nsplit(B, CO, CI, N);
}

// slabs per job inside a batch (the batch fills the chip: ~16 point tiles per workgroup)
extern "C" int spacap_conv1x1_wgrad_slabs_batched(int B, int CO, int CI, int N) {
  if (spacap_conv1x1_wgrad_slabs(B, CO, CI, N) == 0) return 0;
  // (512 points per workgroup: half the partial-sum traffic of 256 -- 65 instead of 130 MB per step at cfg2 -- and still ~1 000
  // workgroups in the step's batch; 6.60 -> 6.57 ms same box, 1 024 points gives it back)
  int nsplit = N / 512;
  if (nsplit < 1) nsplit = 1;
  const int single = conv1x1_nsplit(B, CO, CI, N);
  return B * (nsplit < single ? nsplit : single);
}

// njobs independent 1x1-convolution weight gradients in one launch; all arrays are HOST arrays (read before the call
// returns); part[i] receives nslabs[i] = B[i] x (point ranges) partial results (add in order).
extern "C" int spacap_conv1x1_wgrad_batched_f32(const float *const *g, const float *const *x, const int *B, const int *CO,
                                                const int *CI, const int *N, const int *nslabs, const int *with_bias,
                                                float *const *part, int njobs, spacap_stream_t stream) {
  const char *what = "spacap_conv1x1_wgrad_batched_f32";
  SPACAP_REQUIRE(njobs >= 0 && (njobs == 0 || (g && x && B && CO && CI && N && nslabs && part)), "%s: bad arguments", what);
  hipStream_t s = spacap::as_stream(stream);
  int i = 0;
  while (i < njobs) {
    ConvTable T;
    T.njobs = 0, T.pad = 0;
    long blocks = 0;
    for (; i < njobs && T.njobs < CV_JOB_MAX; ++i) {
      SPACAP_REQUIRE(spacap_conv1x1_wgrad_slabs(B[i], CO[i], CI[i], N[i]) > 0 && nslabs[i] >= B[i] && nslabs[i] % B[i] == 0 &&
                         g[i] && x[i] && part[i],
                     "%s: job %d: (B=%d, CO=%d, CI=%d, N=%d, slabs=%d) unsupported or null pointer", what, i, B[i], CO[i], CI[i],
                     N[i], nslabs[i]);
      ConvJob &J = T.job[T.njobs++];
      J.g = g[i], J.x = x[i], J.part = part[i], J.CO = CO[i], J.CI = CI[i], J.N = N[i];
      J.nsplit = nslabs[i] / B[i], J.gx = nslabs[i], J.gy = (CO[i] + 127) / 128, J.block0 = (int)blocks, J.with_bias = with_bias ? with_bias[i] : 0;
      blocks += (long)nslabs[i] * ((CO[i] + 127) / 128) * ((CI[i] + 127) / 128);
      SPACAP_REQUIRE(blocks < 2147483647L, "%s: too many blocks", what);
    }
    if (!f32_mfma_only()) hipLaunchKernelGGL(conv1x1_wgrad_bf3_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, s, T);
    else hipLaunchKernelGGL(conv1x1_wgrad_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, s, T);
  }
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// g f32 [B,CO,N], x f32 [B,CI,N] dense; part f32 [spacap_conv1x1_wgrad_slabs(B,CO,CI,N)][CO*CI]
extern "C" int spacap_conv1x1_wgrad_f32(const float *g, const float *x, int B, int CO, int CI, int N, float *part,
                                        spacap_stream_t stream) {
  const char *what = "spacap_conv1x1_wgrad_f32";
  const int nslab = spacap_conv1x1_wgrad_slabs(B, CO, CI, N);
  SPACAP_REQUIRE(nslab > 0, "%s: (B=%d, CO=%d, CI=%d, N=%d) unsupported", what, B, CO, CI, N);
  SPACAP_REQUIRE(g && x && part, "%s: null pointer", what);
  if (!f32_mfma_only())
    hipLaunchKernelGGL(conv1x1_wgrad_bf3_kernel, dim3(nslab, (CO + 127) / 128, (CI + 127) / 128), dim3(256), 0, spacap::as_stream(stream), g, x,
                       CO, CI, N, nslab / B, part);
  else
    hipLaunchKernelGGL(conv1x1_wgrad_kernel, dim3(nslab, (CO + 127) / 128, (CI + 127) / 128), dim3(256), 0, spacap::as_stream(stream), g, x, CO, CI,
                       N, nslab / B, part);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ===========================================================================================================
// Data gradient of the feed-forward block's second Linear fused with the backward of relu + dropout:
//   dx[r, n] = (y[r, n] > 0) ? scale * sum_k g[r, k] W[k, n] : 0        g [R, 128], W [128, CP] (= w_2.weight), y [R, CP]
// (models/transformer_captioner.py:117-126: w_2(dropout(relu(w_1 x))); y is the saved dropout(relu(.)) output, which is
// positive exactly where the unit was active and kept).  The BLAS library runs this row-major x row-major product at
// 28 TFLOP/s (37 us for 2048 x 2048 x 128) and the mask is one more pass over the 16 MB result.  Here: weights
// stationary in registers (K = 128), one 64-row tile per workgroup and column block, accumulators transposed through
// LDS so that y is read and dx written as full rows.
namespace {
__global__ __launch_bounds__(256) void linear_dgrad_mask_kernel(const float *__restrict__ g, const float *__restrict__ W,
                                                                const float *__restrict__ y, float scale, long R, int CP,
                                                                float *__restrict__ dx) {
  constexpr int CK = 128, NT = 2, LD = CK + 4, KS = CK / 4, C4 = CK / 4, NV = TM * C4 / 256, RSTEP = 256 / C4;
  constexpr int COB = 64 * NT, LDO = COB + 4, O4 = COB / 4, NO = TM * O4 / 256, OSTEP = 256 / O4;
  __shared__ __attribute__((aligned(16))) float s_a[TM * LD];
  __shared__ __attribute__((aligned(16))) float s_o[TM * LDO];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int cbb = blockIdx.y * COB, wc = w * 16 * NT, cb = cbb + wc;
  const long row0 = (long)blockIdx.x * TM;
  const int c4 = tid % C4, r0 = tid / C4, o4 = tid % O4, or0 = tid / O4;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int row = r0 + i * RSTEP;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (row0 + row < R) a = ld4(g + (size_t)(row0 + row) * CK + c4 * 4);
    st4(&s_a[row * LD + c4 * 4], a);
  }
  float wf[NT][KS];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[j][ks] = W[(size_t)(ks * 4 + lg) * CP + cb + 16 * j + l15];
  __syncthreads();
  f32x4 acc[TM / 16][NT];
#pragma unroll
  for (int mt = 0; mt < TM / 16; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int mt = 0; mt < TM / 16; ++mt) {
      const float b = s_a[(mt * 16 + l15) * LD + ks * 4 + lg];
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[mt][j] = MFMA16(wf[j][ks], b, acc[mt][j]);
    }
  }
#pragma unroll
  for (int mt = 0; mt < TM / 16; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j) st4(&s_o[(mt * 16 + l15) * LDO + wc + 16 * j + 4 * lg], acc[mt][j]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int row = or0 + i * OSTEP;
    if (row0 + row < R) {
      const size_t o = (size_t)(row0 + row) * CP + cbb + o4 * 4;
      const f32x4 d = ld4(&s_o[row * LDO + o4 * 4]), yv = ld4(y + o);
      f32x4 r;
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = yv[u] > 0.f ? d[u] * scale : 0.f;
      st4(dx + o, r);
    }
  }
}
}  // namespace

// g f32 [R,128], W f32 [128,CP] (CP a multiple of 128), y f32 [R,CP], dx f32 [R,CP]; all dense
extern "C" int spacap_linear_dgrad_mask_f32(const float *g, const float *W, const float *y, float scale, long R, int CK,
                                            int CP, float *dx, spacap_stream_t stream) {
  const char *what = "spacap_linear_dgrad_mask_f32";
  SPACAP_REQUIRE(R >= 0 && CK == 128 && CP >= 128 && CP % 128 == 0, "%s: (R=%ld, CK=%d, CP=%d) unsupported", what, R, CK, CP);
  if (R == 0) return SPACAP_OK;
  SPACAP_REQUIRE(g && W && y && dx, "%s: null pointer", what);
  const long tiles = (R + TM - 1) / TM;
  SPACAP_REQUIRE(tiles <= 2147483647L, "%s: too many rows", what);
  hipLaunchKernelGGL(linear_dgrad_mask_kernel, dim3((unsigned)tiles, CP / 128), dim3(256), 0, spacap::as_stream(stream), g, W, y,
                     scale, R, CP, dx);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ===========================================================================================================
// Relation head, layers 2 and 3 (models/transformer_captioner.py:319-326, 392-397) on R = B*K*K pair rows:
//   hid2 = relu(hid1 W2^T + b2) [R,128],  pred = hid2 W3^T + b3 [R,NO3 = 9]
// Forward: sa_mid_fwd_kernel<128, 2, TAIL> (one pass: read hid1, write hid2 and pred; the composition of a BLAS GEMM,
// a ReLU pass and a second GEMM moves 5x the bytes).  Backward, first stage (this kernel): one streaming pass over
// hid2 that produces dz2 = (dpred W3) * (hid2 > 0) and per-workgroup partial sums of dW3 = dpred^T hid2,
// db2 = sum dz2 and db3 = sum dpred -- replacing a GEMM, a transposed GEMM, a masking pass and two column sums, each
// a full pass over a 268 MB tensor.  dhid1 = dz2 W2 and dW2 = dz2^T hid1 stay BLAS GEMMs (MFMA-bound).
namespace {
constexpr int RT_NO = 9, RT_TM = 64;
// part f32 [gridDim.x][RT_NO*128 + 128 + 16]: dW3 (row-major [9][128]), db2 [128], db3 [9 (+7 pad)]
__global__ __launch_bounds__(256) void rel_tail_bwd_kernel(const float *__restrict__ dpred, const float *__restrict__ W3,
                                                           const float *__restrict__ hid2, long R, float *__restrict__ dz2,
                                                           float *__restrict__ part) {
  constexpr int C = 128, PW = RT_NO * C + C + 16;
  __shared__ __attribute__((aligned(16))) float s_dp[RT_TM * RT_NO];
  __shared__ float s_red[8 * 32 * 41];
  const int tid = threadIdx.x, c4 = tid & 31, r0 = tid >> 5;
  float w3[RT_NO][4], aw[RT_NO][4], ab2[4] = {0.f, 0.f, 0.f, 0.f}, ab3[RT_NO];
#pragma unroll
  for (int o = 0; o < RT_NO; ++o) {
    ab3[o] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) w3[o][u] = W3[o * C + c4 * 4 + u], aw[o][u] = 0.f;
  }
  const long ntiles = (R + RT_TM - 1) / RT_TM;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long row0 = t * RT_TM;
    __syncthreads();
    for (int i = tid; i < RT_TM * RT_NO; i += 256) s_dp[i] = (row0 * RT_NO + i < R * RT_NO) ? dpred[row0 * RT_NO + i] : 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RT_TM / 8; ++i) {
      const int row = r0 + 8 * i;
      if (row0 + row >= R) continue;
      const f32x4 h = ld4(hid2 + (size_t)(row0 + row) * C + c4 * 4);
      float d[RT_NO];
#pragma unroll
      for (int o = 0; o < RT_NO; ++o) d[o] = s_dp[row * RT_NO + o];
      f32x4 dz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < RT_NO; ++o)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          dz[u] = fmaf(d[o], w3[o][u], dz[u]);
          aw[o][u] = fmaf(d[o], h[u], aw[o][u]);
        }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        dz[u] = h[u] > 0.f ? dz[u] : 0.f;
        ab2[u] += dz[u];
      }
      st4(dz2 + (size_t)(row0 + row) * C + c4 * 4, dz);
      if (c4 == 0) {
#pragma unroll
        for (int o = 0; o < RT_NO; ++o) ab3[o] += d[o];
      }
    }
  }
  // the 8 row groups of a column quad, added in a fixed order
  __syncthreads();
  float *mine = &s_red[(r0 * 32 + c4) * 41];
#pragma unroll
  for (int o = 0; o < RT_NO; ++o)
#pragma unroll
    for (int u = 0; u < 4; ++u) mine[o * 4 + u] = aw[o][u];
#pragma unroll
  for (int u = 0; u < 4; ++u) mine[36 + u] = ab2[u];
  __syncthreads();
  float *o_part = part + (size_t)blockIdx.x * PW;
  for (int i = tid; i < 32 * 40; i += 256) {
    const int q = i / 40, e = i % 40;
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) a += s_red[(g * 32 + q) * 41 + e];
    if (e < 36) o_part[(e >> 2) * C + q * 4 + (e & 3)] = a;
    else o_part[RT_NO * C + q * 4 + (e - 36)] = a;
  }
  __syncthreads();
  if (c4 == 0) {
#pragma unroll
    for (int o = 0; o < RT_NO; ++o) s_red[r0 * 16 + o] = ab3[o];
  }
  __syncthreads();
  if (tid < 16) {
    float a = 0.f;
    if (tid < RT_NO)
#pragma unroll
      for (int g = 0; g < 8; ++g) a += s_red[g * 16 + tid];
    o_part[RT_NO * C + C + tid] = a;
  }
}
}  // namespace

// hid1 f32 [R,128], W2 f32 [128,128], b2 f32 [128], W3 f32 [9,128], b3 f32 [9] -> hid2 f32 [R,128], pred f32 [R,9]
extern "C" int spacap_rel_tail_fwd_f32(const float *hid1, const float *W2, const float *b2, const float *W3, const float *b3,
                                       long R, float *hid2, float *pred, spacap_stream_t stream) {
  const char *what = "spacap_rel_tail_fwd_f32";
  SPACAP_REQUIRE(hid1 && W2 && b2 && W3 && b3 && hid2 && pred && R >= 1, "%s: bad arguments", what);
  const size_t lds = (size_t)TM * ((128 + 8) + (128 + 4)) * sizeof(float);
  const long tiles = (R + TM - 1) / TM;
  static const int res = resident_blocks(sa_mid_fwd_kernel<128, 2, true>, lds);
  hipLaunchKernelGGL((sa_mid_fwd_kernel<128, 2, true>), dim3(grid_rows(fwd_resident(res), 1, tiles), 1), dim3(256), lds,
                     spacap::as_stream(stream), hid1, (const float *)nullptr, W2, 128, R, hid2, (double *)nullptr,
                     TailArgs{b2, W3, b3, pred, RT_NO});
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// number of partial rows the backward writes (each 9*128 + 128 + 16 floats)
extern "C" int spacap_rel_tail_bwd_nparts(long R) {
  const long tiles = (R + RT_TM - 1) / RT_TM;
  return (int)(tiles < 1024 ? (tiles < 1 ? 1 : tiles) : 1024);
}

// dpred f32 [R,9], W3 f32 [9,128], hid2 f32 [R,128] -> dz2 f32 [R,128], part f32 [nparts][9*128 + 128 + 16]
extern "C" int spacap_rel_tail_bwd_f32(const float *dpred, const float *W3, const float *hid2, long R, float *dz2, float *part,
                                       spacap_stream_t stream) {
  const char *what = "spacap_rel_tail_bwd_f32";
  SPACAP_REQUIRE(dpred && W3 && hid2 && dz2 && part && R >= 1, "%s: bad arguments", what);
  hipLaunchKernelGGL(rel_tail_bwd_kernel, dim3(spacap_rel_tail_bwd_nparts(R)), dim3(256), 0, spacap::as_stream(stream), dpred, W3,
                     hid2, R, dz2, part);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ===========================================================================================================
// Row-panel products of the Transformer's d_model = 128 projections (models/transformer_captioner.py:63-99):
//   out[r, n] = sum_k a[r, k] Wop[k, n] (+ bias[n])
//   Wop[k, n] = TRANS_W ? W[n, k]  (forward  y = x W^T,  W [CO, K])
//                       : W[k, n]  (data gradient dx = g W,  W [K, CO])
// The BLAS library's heuristics pick one or two 128 x 256 macro tiles for these shapes when R is a few hundred rows
// (the caption decoder: 8 x 32 tokens; 20 - 60 us per product on 1 - 3 workgroups).  Here: one (16 MT rows) x 64 column tile per workgroup, each wave 16 columns, the
// activations of a 128-wide K chunk in LDS and that chunk's weights in registers, so a few-hundred-row product is tens
// of workgroups of ~100 MFMA each (3 - 7 us).  Same summation order for every row: results do not depend on R.
// a dense [R, K], K a multiple of 128.  (The vocabulary projection's gradients, 248 x 3 001, were tried in this
// form too -- K-split partial sums for dx, a transposed loader for dW -- and lost to the BLAS kernels, 40 vs 35 us.)
namespace {
template <bool TRANS_W, int MT>
__global__ __launch_bounds__(256) void linear_rows_kernel(const float *__restrict__ a, const float *__restrict__ W,
                                                          const float *__restrict__ bias, long R, int K, int CO,
                                                          float *__restrict__ out) {
  constexpr int KC = 128, LD = KC + 4, KS = KC / 4, TMR = 16 * MT, C4 = KC / 4, RSTEP = 256 / C4, NV = TMR * C4 / 256;
  __shared__ __attribute__((aligned(16))) float s_a[TMR * LD];
  __shared__ __attribute__((aligned(16))) float s_w[TRANS_W ? 64 * LD : 4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int cbb = blockIdx.y * 64, cb = cbb + w * 16;
  const long row0 = (long)blockIdx.x * TMR;
  const int c4 = tid % C4, r0 = tid / C4;
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < K; kc += KC) {
    if (kc) __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = r0 + i * RSTEP;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row0 + row < R) v = ld4(a + (size_t)(row0 + row) * K + kc + c4 * 4);
      st4(&s_a[row * LD + c4 * 4], v);
    }
    float wf[KS];
    if (TRANS_W) {
      // the 64 x 128 weight panel through LDS: full-row loads instead of 16-byte runs per lane
#pragma unroll
      for (int i = 0; i < 64 * C4 / 256; ++i) {
        const int n = r0 + i * RSTEP;
        st4(&s_w[n * LD + c4 * 4], ld4(W + (size_t)(cbb + n) * K + kc + c4 * 4));
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[ks] = W[(size_t)(kc + ks * 4 + lg) * CO + cb + l15];
    }
    __syncthreads();
    if (TRANS_W) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[ks] = s_w[(w * 16 + l15) * LD + ks * 4 + lg];
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = MFMA16(wf[ks], s_a[(mt * 16 + l15) * LD + ks * 4 + lg], acc[mt]);
  }
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = ld4(bias + cb + 4 * lg);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long row = row0 + mt * 16 + l15;
    if (row < R) st4(out + (size_t)row * CO + cb + 4 * lg, acc[mt] + bv);
  }
}
}  // namespace

// 1 when (R, K, CO) has a dense row-panel kernel (spacap_linear_rows_f32)
extern "C" int spacap_linear_rows_supported(long R, int K, int CO) {
  return R >= 1 && K >= 128 && K <= 512 && K % 128 == 0 && CO >= 64 && CO % 64 == 0;
}

// a f32 [R,K], W f32 [CO,K] (trans_w) or [K,CO], bias f32 [CO] or null, out f32 [R,CO]; all dense, 16-byte aligned
extern "C" int spacap_linear_rows_f32(const float *a, const float *W, const float *bias, long R, int K, int CO, int trans_w,
                                      float *out, spacap_stream_t stream) {
  const char *what = "spacap_linear_rows_f32";
  SPACAP_REQUIRE(R >= 0 && K >= 128 && K <= 512 && K % 128 == 0 && CO >= 64 && CO % 64 == 0,
                 "%s: (R=%ld, K=%d, CO=%d) unsupported", what, R, K, CO);
  if (R == 0) return SPACAP_OK;
  SPACAP_REQUIRE(a && W && out, "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
  const bool small = R <= 1024;
  const long tiles = small ? (R + 31) / 32 : (R + 63) / 64;
  SPACAP_REQUIRE(tiles <= 2147483647L, "%s: too many rows", what);
  const dim3 grid((unsigned)tiles, CO / 64);
  if (trans_w) {
    if (small) hipLaunchKernelGGL((linear_rows_kernel<true, 2>), grid, dim3(256), 0, s, a, W, bias, R, K, CO, out);
    else hipLaunchKernelGGL((linear_rows_kernel<true, 4>), grid, dim3(256), 0, s, a, W, bias, R, K, CO, out);
  } else {
    if (small) hipLaunchKernelGGL((linear_rows_kernel<false, 2>), grid, dim3(256), 0, s, a, W, bias, R, K, CO, out);
    else hipLaunchKernelGGL((linear_rows_kernel<false, 4>), grid, dim3(256), 0, s, a, W, bias, R, K, CO, out);
  }
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
