// Fused sub-layer kernels of the Transformer captioner (d_model = 128) for gfx950 (MI355X).
//
// Reference: models/transformer_captioner.py -- SublayerConnection :115-127 (x + dropout(sublayer(norm(x)))),
// LayerNorm :102-113 (unbiased std, eps on std), PositionwiseFeedForward :72-81, MultiHeadedAttention's four
// Linear(128,128) :39-70, EncoderLayer :180-191, DecoderLayer :209-225 (early guide: no cross-attention).
//
// One encoder / decoder layer is ~10 launches forward and ~14 backward as separate operators (LayerNorm, packed q|k|v
// GEMM, attention, output projection, dropout-add, LayerNorm, w_1, relu-dropout, w_2, dropout-add and their
// gradients), each a few microseconds on 1 MB tensors.  Rows (tokens) are independent everywhere except inside the
// attention kernel, so everything between two attention calls is ONE row-tile kernel here:
//
//   tf_rows_kernel<FWD>:  acc = A1 W1^T + b1            (optional GEMM, K1 = 128 .. 2048: output projection or w_2)
//                         x' = res + dropout(acc)       (stored: the residual stream)
//                         n  = LayerNorm(x')            (stored for the weight gradients; statistics for the backward)
//                         out2 = n W2^T + b2            (optional GEMM, N2 = 384: the NEXT attention's packed q|k|v)
//   tf_rows_kernel<BWD>:  dn = A1 W1                    (optional GEMM: dqkv Wqkv, or dhid W_1 with K1 = 2048)
//                         dx = LayerNorm'(dn) + addend  (stored; per-workgroup partials of the LayerNorm parameter sums)
//                         dy = dropout'(dx)             (stored: gradient of the sub-layer output in front of the residual)
//                         out2 = dy W2                  (optional GEMM, N2 = 128: through the output projection)
//   tf_ffn1_kernel:       h = dropout(relu(n W_1^T + b_1))   [R, d_ff]   (weights stationary, 64 x 128 tiles)
//
// The feed-forward hidden layer's backward mask / data gradient is spacap_linear_dgrad_mask_f32 (sa_mlp.hip), the
// weight gradients are the batched kernel of sa_mlp.hip (deferred to the end of the backward pass).
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: bit-for-bit an fp32 fma chain), computed transposed
// (weights as the A operand) so that a lane ends up with consecutive columns of one row.  Operands reach the
// registers as 16-byte loads through a permutation of k inside every group of 16: lane (l15, lg) loads
// X[l15][16 q + 4 lg .. + 3] and the i-th MFMA of the group uses element i of both operands.
#include <math.h>

#include "common.hpp"

namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x2 = float __attribute__((ext_vector_type(2)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int D = 128;        // d_model
constexpr int BM = 16;        // rows per workgroup of the row kernel
constexpr int LDT = D + 8;    // LDS row stride of a [rows][128] fp32 tile read with ds_read_b128 (conflict-free)

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x2 ld2(const float *p) { return *reinterpret_cast<const f32x2 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

// ---- dropout keep mask: the counter hash of elementwise.hip (seed word per call + device-resident step counter) ----
struct DropSeed {
  unsigned lo, hi;
};
__device__ __forceinline__ DropSeed make_seed(unsigned long long seed, const unsigned long long *seed_dev) {
  const unsigned long long s = seed + (seed_dev ? *seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
  return DropSeed{(unsigned)s, (unsigned)(s >> 32)};
}
__device__ __forceinline__ unsigned hash32(unsigned long long idx, DropSeed s) {
  unsigned h = (unsigned)idx ^ s.lo;
  h += ((unsigned)(idx >> 32) ^ s.hi) * 0x9E3779B1u;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

struct TfRowsArgs {
  long R;
  const float *A1;       // [R][K1], or [nparts][R][128] partial sums of the first product when nparts > 0, or null
  const float *W1;       // FWD: [128][K1];  BWD: [K1][128]
  const float *bias1;    // FWD: [128] or null
  int K1;
  unsigned drop_thresh;  // 0: no dropout
  float drop_scale;
  float eps;
  unsigned long long seed;
  const unsigned long long *seed_dev;
  const float *res;      // FWD: residual rows [R][128];  BWD: addend [R][128] or null
  float *x_out;          // FWD: res + dropout(gemm1) or null;  BWD: dx
  const float *ln_a, *ln_b;
  float *n_out;          // FWD: LayerNorm output rows or null;  BWD: dy = dropout'(dx) or null
  float *stats;          // FWD: out [R][2] (mean, 1 / (std + eps)) or null;  BWD: in
  const float *x_ln;     // BWD: the rows the LayerNorm normalised
  const float *G;        // BWD with A1 == null: gradient w.r.t. the LayerNorm output [R][128]
  float *part;           // BWD: [gridDim.x][256] partial sums of (d ln_a, d ln_b)
  const float *W2;       // FWD: [N2][128];  BWD: [128][N2 = 128]
  const float *bias2;    // FWD: [N2] or null
  float *out2;           // [R][N2]
  int N2;                // 0: no second product
  int nparts;            // > 0: A1 holds that many partial results of the first product (added in order), no GEMM here
  const float *attn_out; // BWD with N2 == 128 (out2 = gradient of the attention output a): a itself [R][128], and
  float *delta_out;      //   delta_out [R / Lq][8][Lq] receives sum_d out2[r, 16 head + d] a[r, 16 head + d] per (row, head):
  int Lq;                //   the softmax-backward row term of the attention kernel (h = 8 heads of 16), rows r = b Lq + q
};

// Row-tile kernel: see the file header.  256 threads = 4 waves; wave w owns columns [32 w, 32 w + 32) of the
// 128-wide row through GEMM1 and the row epilogue, and N2 / 4 columns of the second product.
template <bool BWD>
__global__ __launch_bounds__(256) void tf_rows_kernel(const TfRowsArgs P) {
  __shared__ __attribute__((aligned(16))) float s_t[BM * LDT];
  __shared__ float s_red[2][4][BM];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const long row0 = (long)blockIdx.x * BM, row = row0 + l15;
  const bool valid = row < P.R;
  const long rowc = valid ? row : P.R - 1;

  // ---- GEMM1 -> v[g][t]: the lane's 8 values of row l15, two groups of 4 consecutive columns cb[g] .. cb[g] + 3
  float v[2][4];
  int cb[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) cb[g] = BWD ? 32 * w + 8 * lg + 4 * g : 32 * w + 16 * g + 4 * lg;
  if (P.nparts > 0) {
    // the first product was formed by tf_gemm_kernel as partial sums over slices of k: add them in slice order
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float *pp = P.A1 + (size_t)rowc * D + cb[g];
      const size_t ps = (size_t)P.R * D;
      f32x4 a = ld4(pp);
      int sidx = 1;
      for (; sidx + 7 < P.nparts; sidx += 8) {   // eight loads in flight (one round trip for d_ff = 2048's 16 slices, both column
        f32x4 t[8];                              // groups interleaved), added in slice order
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ld4(pp + (sidx + u) * ps);
#pragma unroll
        for (int u = 0; u < 8; ++u) a += t[u];
      }
      for (; sidx + 3 < P.nparts; sidx += 4) {   // four loads in flight
        const f32x4 t0 = ld4(pp + sidx * ps), t1 = ld4(pp + (sidx + 1) * ps), t2 = ld4(pp + (sidx + 2) * ps),
                    t3 = ld4(pp + (sidx + 3) * ps);
        a += t0, a += t1, a += t2, a += t3;
      }
      for (; sidx < P.nparts; ++sidx) a += ld4(pp + sidx * ps);
#pragma unroll
      for (int t = 0; t < 4; ++t) v[g][t] = a[t];
    }
  } else if (P.A1 != nullptr) {
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int K1 = P.K1;
    const float *ap = P.A1 + (size_t)rowc * K1 + 4 * lg;
    if (!BWD) {
      // weights [128][K1]: tile j of the wave = output columns 32 w + 16 j + (0..15)
      const float *wp0 = P.W1 + (size_t)(32 * w + l15) * K1 + 4 * lg, *wp1 = wp0 + (size_t)16 * K1;
      // two register sets, loaded one chunk (128 of k) ahead of their use: no copies, so the loads of chunk c + 1 stay in
      // flight under the MFMAs of chunk c
      f32x4 a0[8], p0[8], q0[8], a1[8], p1[8], q1[8];
      auto load = [&](f32x4 *a, f32x4 *x, f32x4 *y, int kc) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = ld4(ap + kc + 16 * q), x[q] = ld4(wp0 + kc + 16 * q), y[q] = ld4(wp1 + kc + 16 * q);
      };
      auto mma = [&](const f32x4 *a, const f32x4 *x, const f32x4 *y) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[0] = MFMA16(x[q][i], a[q][i], acc[0]);
            acc[1] = MFMA16(y[q][i], a[q][i], acc[1]);
          }
      };
      load(a0, p0, q0, 0);
      for (int kc = 0; kc < K1; kc += 256) {
        if (kc + 128 < K1) load(a1, p1, q1, kc + 128);
        mma(a0, p0, q0);
        if (kc + 128 < K1) {
          if (kc + 256 < K1) load(a0, p0, q0, kc + 256);
          mma(a1, p1, q1);
        }
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) v[g][u] = acc[g][u];
    } else {
      // weights [K1][128]: the wave's 32 columns as two interleaved tiles, tile i = columns 32 w + 2 n + i
      const float *wp = P.W1 + (size_t)(4 * lg) * D + 32 * w + 2 * l15;
      f32x4 a0[8], a1[8];
      f32x2 x0[32], x1[32];
      auto load = [&](f32x4 *a, f32x2 *x, int kc) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          a[q] = ld4(ap + kc + 16 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) x[q * 4 + i] = ld2(wp + (size_t)(kc + 16 * q + i) * D);
        }
      };
      auto mma = [&](const f32x4 *a, const f32x2 *x) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[0] = MFMA16(x[q * 4 + i][0], a[q][i], acc[0]);
            acc[1] = MFMA16(x[q * 4 + i][1], a[q][i], acc[1]);
          }
      };
      load(a0, x0, 0);
      for (int kc = 0; kc < K1; kc += 256) {
        if (kc + 128 < K1) load(a1, x1, kc + 128);
        mma(a0, x0);
        if (kc + 128 < K1) {
          if (kc + 256 < K1) load(a0, x0, kc + 256);
          mma(a1, x1);
        }
      }
      // acc[i][u] is column 32 w + 2 (4 lg + u) + i = cb[u >> 1] + 2 (u & 1) + i
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u >> 1][2 * (u & 1) + i] = acc[i][u];
    }
  }

  // The second product's weights depend on nothing computed here: its first registers are requested NOW, so that their
  // latency runs under the row epilogue (three barriers and a round of global loads) instead of after it.
  f32x4 x0[8], y0[8];   // FWD: first pair of 16-column tiles of W2
  f32x2 wv2[32];        // BWD: W2 [128][128], the wave's two interleaved tiles
  // forward with N2 > 128: blockIdx.y owns N2 / gridDim.y of the second product's columns (the rows' first part is then formed
  // by every one of those workgroups, stored by the first)
  const int n2y = P.N2 / (int)gridDim.y, nw2 = n2y >> 2, n00 = (int)blockIdx.y * n2y + w * nw2;
  const bool first_y = blockIdx.y == 0;
  const float *wp2 = BWD ? P.W2 + (size_t)(4 * lg) * D + 32 * w + 2 * l15 : P.W2 + (size_t)(n00 + l15) * D + 4 * lg;
  if (P.N2 != 0) {
    if (!BWD) {
#pragma unroll
      for (int q = 0; q < 8; ++q) x0[q] = ld4(wp2 + 16 * q), y0[q] = ld4(wp2 + 16 * D + 16 * q);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) wv2[q * 4 + i] = ld2(wp2 + (size_t)(16 * q + i) * D);
    }
  }

  const DropSeed sd = make_seed(P.seed, P.seed_dev);
  float t_out[2][4];   // what the second product consumes: FWD LayerNorm output, BWD dy

  if (!BWD) {
    // ---- x' = res + dropout(acc + b1);  n = LayerNorm(x') ----------------------------------------------------------
    float xo[2][4], s = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 r4 = ld4(P.res + (size_t)rowc * D + cb[g]);
      if (P.A1 != nullptr) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (P.bias1) b4 = ld4(P.bias1 + cb[g]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float a = v[g][t] + b4[t];
          if (P.drop_thresh != 0u) {
            const bool keep = hash32((unsigned long long)row * D + cb[g] + t, sd) >= P.drop_thresh;
            a = keep ? a * P.drop_scale : 0.f;
          }
          xo[g][t] = r4[t] + a;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) xo[g][t] = r4[t];
      }
      if (P.x_out && valid && first_y) st4(P.x_out + (size_t)row * D + cb[g], f32x4{xo[g][0], xo[g][1], xo[g][2], xo[g][3]});
#pragma unroll
      for (int t = 0; t < 4; ++t) s += xo[g][t];
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (lg == 0) s_red[0][w][l15] = s;
    __syncthreads();
    const float mu = ((s_red[0][0][l15] + s_red[0][1][l15]) + (s_red[0][2][l15] + s_red[0][3][l15])) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float c = xo[g][t] - mu;
        q += c * c;
      }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (lg == 0) s_red[1][w][l15] = q;
    __syncthreads();
    const float var = ((s_red[1][0][l15] + s_red[1][1][l15]) + (s_red[1][2][l15] + s_red[1][3][l15])) / (float)(D - 1);
    const float r = 1.0f / (sqrtf(var) + P.eps);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 a4 = ld4(P.ln_a + cb[g]), b4 = ld4(P.ln_b + cb[g]);
#pragma unroll
      for (int t = 0; t < 4; ++t) t_out[g][t] = a4[t] * ((xo[g][t] - mu) * r) + b4[t];
      if (P.n_out && valid && first_y)
        st4(P.n_out + (size_t)row * D + cb[g], f32x4{t_out[g][0], t_out[g][1], t_out[g][2], t_out[g][3]});
    }
    if (P.stats && valid && w == 0 && lg == 0 && first_y) {
      P.stats[row * 2] = mu;
      P.stats[row * 2 + 1] = r;
    }
  } else {
    // ---- dx = LayerNorm'(dn) + addend;  partial sums of the LayerNorm parameter gradients;  dy = dropout'(dx) -----
    const float mu = P.stats[rowc * 2], r = P.stats[rowc * 2 + 1];
    float dn[2][4], xc[2][4], dxh[2][4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (P.A1 == nullptr) {
        const f32x4 g4 = ld4(P.G + (size_t)rowc * D + cb[g]);
#pragma unroll
        for (int t = 0; t < 4; ++t) v[g][t] = g4[t];
      }
      const f32x4 x4 = ld4(P.x_ln + (size_t)rowc * D + cb[g]), a4 = ld4(P.ln_a + cb[g]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        dn[g][t] = valid ? v[g][t] : 0.f;
        xc[g][t] = x4[t] - mu;
        dxh[g][t] = dn[g][t] * a4[t];
        s1 += dxh[g][t];
        s2 += dxh[g][t] * xc[g][t];
      }
    }
    s1 += __shfl_xor(s1, 16);
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 16);
    s2 += __shfl_xor(s2, 32);
    if (lg == 0) s_red[0][w][l15] = s1, s_red[1][w][l15] = s2;
    __syncthreads();
    s1 = (s_red[0][0][l15] + s_red[0][1][l15]) + (s_red[0][2][l15] + s_red[0][3][l15]);
    s2 = (s_red[1][0][l15] + s_red[1][1][l15]) + (s_red[1][2][l15] + s_red[1][3][l15]);
    const float sdv = 1.0f / r - P.eps;                            // the unbiased std
    const float c2 = -(s2 * r * r) / (sdv * (float)(D - 1));      // 0/0 = NaN on a constant row, as autograd gives
    const float c1 = r * s1 / (float)D;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 ad = {0.f, 0.f, 0.f, 0.f};
      if (P.res) ad = ld4(P.res + (size_t)rowc * D + cb[g]);
      float dx[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        dx[t] = r * dxh[g][t] + c2 * xc[g][t] - c1 + ad[t];
        float y = dx[t];
        if (P.drop_thresh != 0u) {
          const bool keep = hash32((unsigned long long)row * D + cb[g] + t, sd) >= P.drop_thresh;
          y = keep ? y * P.drop_scale : 0.f;
        }
        t_out[g][t] = valid ? y : 0.f;
      }
      if (valid) {
        st4(P.x_out + (size_t)row * D + cb[g], f32x4{dx[0], dx[1], dx[2], dx[3]});
        if (P.n_out) st4(P.n_out + (size_t)row * D + cb[g], f32x4{t_out[g][0], t_out[g][1], t_out[g][2], t_out[g][3]});
      }
      // column sums over the 16 rows of the tile (lanes l15 = 0..15 of one lg), fixed tree order
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float pa = dn[g][t] * (xc[g][t] * r), pb = dn[g][t];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          pa += __shfl_xor(pa, o);
          pb += __shfl_xor(pb, o);
        }
        if (l15 == 0) {
          P.part[(size_t)blockIdx.x * 2 * D + cb[g] + t] = pa;
          P.part[(size_t)blockIdx.x * 2 * D + D + cb[g] + t] = pb;
        }
      }
    }
  }

  if (P.N2 == 0) return;   // (uniform: kernel argument)

  // ---- second product on the rows just formed -------------------------------------------------------------------------
#pragma unroll
  for (int g = 0; g < 2; ++g) st4(&s_t[l15 * LDT + cb[g]], f32x4{t_out[g][0], t_out[g][1], t_out[g][2], t_out[g][3]});
  __syncthreads();
  f32x4 at[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) at[q] = ld4(&s_t[l15 * LDT + 16 * q + 4 * lg]);
  if (!BWD) {
    // W2 [N2][128], N2 a multiple of 128: the wave's N2 / 4 columns in pairs of 16-column tiles
    const int N2 = P.N2, nw = nw2;
    const float *wp = wp2;
    // pairs of tiles, two register sets loaded one pair ahead (no copies: the loads stay in flight under the MFMAs)
    f32x4 x1[8], y1[8];
    auto load = [&](f32x4 *x, f32x4 *y, int jt) {
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = ld4(wp + (size_t)jt * D + 16 * q), y[q] = ld4(wp + (size_t)(jt + 16) * D + 16 * q);
    };
    auto pair = [&](const f32x4 *x, const f32x4 *y, int jt) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc0 = MFMA16(x[q][i], at[q][i], acc0);
          acc1 = MFMA16(y[q][i], at[q][i], acc1);
        }
      const int c0 = n00 + jt + 4 * lg;
      if (P.bias2) acc0 += ld4(P.bias2 + c0), acc1 += ld4(P.bias2 + c0 + 16);
      if (valid) {
        st4(P.out2 + (size_t)row * N2 + c0, acc0);
        st4(P.out2 + (size_t)row * N2 + c0 + 16, acc1);
      }
    };
    for (int jt = 0; jt < nw; jt += 64) {   // (pair 0 was requested before the row epilogue)
      if (jt + 32 < nw) load(x1, y1, jt + 32);
      pair(x0, y0, jt);
      if (jt + 32 < nw) {
        if (jt + 64 < nw) load(x0, y0, jt + 64);
        pair(x1, y1, jt + 32);
      }
    }
  } else {
    // W2 [128][128] (N2 == 128): out2 = dy W2, the wave's 32 columns as two interleaved tiles
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[0] = MFMA16(wv2[q * 4 + i][0], at[q][i], acc[0]);
        acc[1] = MFMA16(wv2[q * 4 + i][1], at[q][i], acc[1]);
      }
    float dsum = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // columns cb[g] + e: e = 2 (u & 1) + i with u >> 1 == g
      const f32x4 o = {acc[0][2 * g], acc[1][2 * g], acc[0][2 * g + 1], acc[1][2 * g + 1]};
      if (valid) st4(P.out2 + (size_t)row * D + cb[g], o);
      if (P.delta_out) {
        const f32x4 a4 = ld4(P.attn_out + (size_t)rowc * D + cb[g]);
        dsum += (o[0] * a4[0] + o[1] * a4[1]) + (o[2] * a4[2] + o[3] * a4[3]);
      }
    }
    if (P.delta_out) {
      // the lane's 8 columns 32 w + 8 lg .. + 7 are half of head 2 w + (lg >> 1); the other half sits in lane ^ 16
      dsum += __shfl_xor(dsum, 16);
      if (valid && (lg & 1) == 0) {
        const long bi = row / P.Lq, qi = row - bi * P.Lq;
        P.delta_out[((size_t)bi * 8 + 2 * w + (lg >> 1)) * P.Lq + qi] = dsum;
      }
    }
  }
}

// Tiled product with the weights stationary in registers and a choice of epilogue (fp32 MFMA, 64-row x 128-column tile
// per workgroup, K walked in chunks of 128 through LDS; blockIdx.z = slice of K for the split products):
//   EPI 0  out[z][r][n] = sum_{k in slice z} x[r][k] Wop[k][n]            partial sums, added in order by tf_rows_kernel
//          (w_2 / the data gradient through w_1: K = d_ff = 2 048 and only 128 output columns -- a row-tile kernel alone would
//          occupy R / 16 workgroups for 2 048 dependent k-steps each; split over K the product fills the chip)
//   EPI 1  out[r][n] = dropout(relu(. + bias[n]))                          models/transformer_captioner.py:80 (w_1)
//   EPI 2  out[r][n] = y[r][n] > 0 ? scale * . : 0                         backward of EPI 1 through w_2 (y = the saved EPI-1 output)
// NN: Wop[k][n] = W[k][n] (W [K][ldw >= N]: data gradients), else Wop[k][n] = W[n][k] (W [N][ldw >= K]: forward products).
constexpr int FM = 64;
struct TfGemmArgs {
  const float *x, *W, *bias, *y;
  float *out;
  long R;
  int ldx, ldw, N, KS;
  unsigned thresh;
  float scale;
  unsigned long long seed;
  const unsigned long long *seed_dev;
};
template <int EPI, bool NN>
__global__ __launch_bounds__(256) void tf_gemm_kernel(const TfGemmArgs P) {
  __shared__ __attribute__((aligned(16))) float s_a[FM * LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const long row0 = (long)blockIdx.x * FM;
  const int n0 = blockIdx.y * 128 + 32 * w;
  const int k0 = blockIdx.z * P.KS;
  const int c4 = tid & 31, r0 = tid >> 5;
  f32x4 acc[FM / 16][2];
#pragma unroll
  for (int mt = 0; mt < FM / 16; ++mt) acc[mt][0] = acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kc = k0; kc < k0 + P.KS; kc += 128) {
    if (kc != k0) __syncthreads();
    {
      // (rows past the end read the last row: branch-free, so the eight loads are in flight together; those rows of the
      // tile are computed and never stored)
      f32x4 stg[FM / 8];
#pragma unroll
      for (int i = 0; i < FM / 8; ++i) {
        const long rr = min(row0 + r0 + 8 * i, P.R - 1);
        stg[i] = ld4(P.x + (size_t)rr * P.ldx + kc + c4 * 4);
      }
#pragma unroll
      for (int i = 0; i < FM / 8; ++i) st4(&s_a[(r0 + 8 * i) * LDT + c4 * 4], stg[i]);
    }
    f32x4 w0[8], w1[8];   // NN: w0[q][i] / w1[q][i] = the two interleaved tiles' weights for k = kc + 16 q + 4 lg + i
    if (!NN) {
      const float *wp = P.W + (size_t)(n0 + l15) * P.ldw + kc + 4 * lg;
#pragma unroll
      for (int q = 0; q < 8; ++q) w0[q] = ld4(wp + 16 * q), w1[q] = ld4(wp + (size_t)16 * P.ldw + 16 * q);
    } else {
      const float *wp = P.W + (size_t)(kc + 4 * lg) * P.ldw + n0 + 2 * l15;
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2 t = ld2(wp + (size_t)(16 * q + i) * P.ldw);
          w0[q][i] = t[0], w1[q][i] = t[1];
        }
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < FM / 16; ++mt)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 a = ld4(&s_a[(mt * 16 + l15) * LDT + 16 * q + 4 * lg]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[mt][0] = MFMA16(w0[q][i], a[i], acc[mt][0]);
          acc[mt][1] = MFMA16(w1[q][i], a[i], acc[mt][1]);
        }
      }
  }
  // the lane's columns of row (mt, l15): two runs of four, starting at c0 and c1
  const int c0 = NN ? n0 + 8 * lg : n0 + 4 * lg, c1 = NN ? c0 + 4 : c0 + 16;
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
  if (EPI == 1 && P.bias) b0 = ld4(P.bias + c0), b1 = ld4(P.bias + c1);
  const DropSeed sd = make_seed(P.seed, P.seed_dev);
  float *out = P.out + (EPI == 0 ? (size_t)blockIdx.z * P.R * P.N : (size_t)0);
#pragma unroll
  for (int mt = 0; mt < FM / 16; ++mt) {
    const long row = row0 + mt * 16 + l15;
    if (row >= P.R) continue;
    f32x4 o0, o1;
    if (NN) {
      o0 = f32x4{acc[mt][0][0], acc[mt][1][0], acc[mt][0][1], acc[mt][1][1]};
      o1 = f32x4{acc[mt][0][2], acc[mt][1][2], acc[mt][0][3], acc[mt][1][3]};
    } else {
      o0 = acc[mt][0], o1 = acc[mt][1];
    }
    if (EPI == 1) {
      o0 += b0, o1 += b1;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float a0 = fmaxf(o0[u], 0.f), a1 = fmaxf(o1[u], 0.f);
        if (P.thresh != 0u) {
          const unsigned long long e = (unsigned long long)row * P.N;
          a0 = hash32(e + c0 + u, sd) >= P.thresh ? a0 * P.scale : 0.f;
          a1 = hash32(e + c1 + u, sd) >= P.thresh ? a1 * P.scale : 0.f;
        }
        o0[u] = a0, o1[u] = a1;
      }
    } else if (EPI == 2) {
      const f32x4 y0 = ld4(P.y + (size_t)row * P.N + c0), y1 = ld4(P.y + (size_t)row * P.N + c1);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        o0[u] = y0[u] > 0.f ? o0[u] * P.scale : 0.f;
        o1[u] = y1[u] > 0.f ? o1[u] * P.scale : 0.f;
      }
    }
    st4(out + (size_t)row * P.N + c0, o0);
    st4(out + (size_t)row * P.N + c1, o1);
  }
}

// The feed-forward block as ONE launch per direction, chained through LDS (models/transformer_captioner.py:72-81):
//   forward   h_c = dropout(relu(n W1_c^T + b1_c))   [64 rows x 128 hidden units c]   stored (the backward needs it)
//             part[c] = h_c W2[:, c]^T                [64 x 128]   partial sum of w_2's output over this slice of d_ff
//   backward  dhid_c = (dy W2[:, c]) * [h_c > 0] * scale            stored (the weight gradient of w_1 needs it)
//             part[c] = dhid_c W1_c                   [64 x 128]   partial sum of the gradient w.r.t. the LayerNorm output
// One workgroup per (64-row tile, 128-wide slice of d_ff): the hidden tile goes from the first product's accumulators
// through the epilogue into LDS and is the second product's activation tile; both weight blocks are in registers
// before the first MFMA (the second block's load latency hides under the first product).  tf_rows_kernel adds the
// d_ff / 128 partial sums in slice order.
struct TfFfnArgs {
  const float *x, *Wa, *Wb, *bias, *y;
  float *hid, *part;
  long R;
  int dff;
  unsigned thresh;
  float scale;
  unsigned long long seed;
  const unsigned long long *seed_dev;
};
template <bool BWD, int MT>   // MT row tiles of 16 per workgroup: 4, or 1 for the few hundred rows of the caption decoder
__global__ __launch_bounds__(256) void tf_ffn_kernel(const TfFfnArgs P) {
  constexpr int FM = 16 * MT, NST = FM >= 8 ? FM / 8 : 1;
  __shared__ __attribute__((aligned(16))) float s_a[FM * LDT];
  __shared__ __attribute__((aligned(16))) float s_h[FM * LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const long row0 = (long)blockIdx.x * FM;
  const int c0 = blockIdx.y * 128, dff = P.dff;
  const int c4 = tid & 31, r0 = tid >> 5;
  f32x4 stg[NST];   // (rows past the end read the last row: branch-free, all loads in flight together; never stored)
#pragma unroll
  for (int i = 0; i < NST; ++i) stg[i] = ld4(P.x + (size_t)min(row0 + r0 + 8 * i, P.R - 1) * D + c4 * 4);
  // element [q][i] of a weight register = Wop[k = 16 q + 4 lg + i][the wave's column l15 of tile 0 / 1]
  f32x4 a0[8], a1[8], b0[8], b1[8];
  if (!BWD) {
    const float *wa = P.Wa + (size_t)(c0 + 32 * w + l15) * D + 4 * lg;          // W1 [dff][128]: rows = hidden units
    const float *wb = P.Wb + (size_t)(32 * w + l15) * dff + c0 + 4 * lg;        // W2 [128][dff]: k = hidden units
#pragma unroll
    for (int q = 0; q < 8; ++q) a0[q] = ld4(wa + 16 * q), a1[q] = ld4(wa + 16 * D + 16 * q);
#pragma unroll
    for (int q = 0; q < 8; ++q) b0[q] = ld4(wb + 16 * q), b1[q] = ld4(wb + (size_t)16 * dff + 16 * q);
  } else {
    const float *wa = P.Wa + (size_t)(4 * lg) * dff + c0 + 32 * w + 2 * l15;    // W2 [128][dff]: k = model channels
    const float *wb = P.Wb + (size_t)(c0 + 4 * lg) * D + 32 * w + 2 * l15;      // W1 [dff][128]: k = hidden units
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 t = ld2(wa + (size_t)(16 * q + i) * dff);
        a0[q][i] = t[0], a1[q][i] = t[1];
      }
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 t = ld2(wb + (size_t)(16 * q + i) * D);
        b0[q][i] = t[0], b1[q][i] = t[1];
      }
  }
#pragma unroll
  for (int i = 0; i < NST; ++i) st4(&s_a[(r0 + 8 * i) * LDT + c4 * 4], stg[i]);
  // the lane's columns inside the 128-wide block: two runs of four starting at e0 and e1
  const int e0 = BWD ? 32 * w + 8 * lg : 32 * w + 4 * lg, e1 = BWD ? e0 + 4 : e0 + 16;
  f32x4 bb0 = {0.f, 0.f, 0.f, 0.f}, bb1 = {0.f, 0.f, 0.f, 0.f};
  if (!BWD && P.bias) bb0 = ld4(P.bias + c0 + e0), bb1 = ld4(P.bias + c0 + e1);
  const DropSeed sd = make_seed(P.seed, P.seed_dev);
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < FM / 16; ++mt) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 a = ld4(&s_a[(mt * 16 + l15) * LDT + 16 * q + 4 * lg]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0 = MFMA16(a0[q][i], a[i], acc0);
        acc1 = MFMA16(a1[q][i], a[i], acc1);
      }
    }
    const long row = row0 + mt * 16 + l15;
    const bool valid = row < P.R;
    f32x4 o0, o1;
    if (BWD) {
      o0 = f32x4{acc0[0], acc1[0], acc0[1], acc1[1]};
      o1 = f32x4{acc0[2], acc1[2], acc0[3], acc1[3]};
      f32x4 y0 = {0.f, 0.f, 0.f, 0.f}, y1 = {0.f, 0.f, 0.f, 0.f};
      if (valid) y0 = ld4(P.y + (size_t)row * dff + c0 + e0), y1 = ld4(P.y + (size_t)row * dff + c0 + e1);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        o0[u] = y0[u] > 0.f ? o0[u] * P.scale : 0.f;
        o1[u] = y1[u] > 0.f ? o1[u] * P.scale : 0.f;
      }
    } else {
      o0 = acc0 + bb0, o1 = acc1 + bb1;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v0 = fmaxf(o0[u], 0.f), v1 = fmaxf(o1[u], 0.f);
        if (P.thresh != 0u) {
          const unsigned long long e = (unsigned long long)row * dff + c0;
          v0 = hash32(e + e0 + u, sd) >= P.thresh ? v0 * P.scale : 0.f;
          v1 = hash32(e + e1 + u, sd) >= P.thresh ? v1 * P.scale : 0.f;
        }
        o0[u] = valid ? v0 : 0.f, o1[u] = valid ? v1 : 0.f;
      }
    }
    if (valid && P.hid) {   // (inference passes no buffer: nothing is kept for a backward)
      st4(P.hid + (size_t)row * dff + c0 + e0, o0);
      st4(P.hid + (size_t)row * dff + c0 + e1, o1);
    }
    st4(&s_h[(mt * 16 + l15) * LDT + e0], o0);
    st4(&s_h[(mt * 16 + l15) * LDT + e1], o1);
  }
  __syncthreads();
  float *out = P.part + (size_t)blockIdx.y * P.R * D;
#pragma unroll
  for (int mt = 0; mt < FM / 16; ++mt) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 a = ld4(&s_h[(mt * 16 + l15) * LDT + 16 * q + 4 * lg]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0 = MFMA16(b0[q][i], a[i], acc0);
        acc1 = MFMA16(b1[q][i], a[i], acc1);
      }
    }
    const long row = row0 + mt * 16 + l15;
    if (row < P.R) {
      if (BWD) {
        st4(out + (size_t)row * D + e0, f32x4{acc0[0], acc1[0], acc0[1], acc1[1]});
        st4(out + (size_t)row * D + e1, f32x4{acc0[2], acc1[2], acc0[3], acc1[3]});
      } else {
        st4(out + (size_t)row * D + e0, acc0);
        st4(out + (size_t)row * D + e1, acc1);
      }
    }
  }
}

// ---- the chained feed-forward kernel on split-bf16 products (tall inputs: the encoder's 2 048 rows, greedy decoding) ---------
// Same contract as tf_ffn_kernel; both products as bf16 x 3 (three bf16 pieces per operand, the six piece products above 2^-24
// on v_mfma_f32_16x16x32_bf16: fp32-equivalent at 6/16 of the fp32-MFMA time).  The WEIGHTS arrive pre-split (four piece
// images per layer, made by one batched launch per stack and forward: spacap_tf_ffn_split_f32) in the two layouts the products
// need -- rows = the product's output unit, 8 consecutive contraction indices per 16-byte operand load -- and go from L2 straight
// into matrix-core operand registers, one 32-deep contraction step ahead; the 64-row activation tile is split once into three
// LDS images [64][128 + 8] which the hidden tile overwrites between the two products (everything of product 1 is in
// accumulators by then).  52 KB of LDS, ~130 registers: three workgroups per CU.
typedef __bf16 ff_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ff_bf16x4 __attribute__((ext_vector_type(4)));
#define FF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
constexpr int FF_LD = D + 8, FF_IMG = 64 * FF_LD;
struct TfFfnBf3Args {
  const float *x, *bias, *y;
  const __bf16 *Wa, *Wb;   // the two products' weight piece images [3][dff 128], in operand order (tf_ffn_split_kernel)
  float *hid, *part;
  long R;
  int dff;
  unsigned thresh;
  float scale;
  unsigned long long seed;
  const unsigned long long *seed_dev;
};
__device__ __forceinline__ void ff_split4(f32x4 v, ff_bf16x4 &p0, ff_bf16x4 &p1, ff_bf16x4 &p2) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const __bf16 h = (__bf16)v[u];
    const float r = v[u] - (float)h;
    const __bf16 m = (__bf16)r;
    p0[u] = h, p1[u] = m, p2[u] = (__bf16)(r - (float)m);
  }
}
template <bool BWD>
__global__ __launch_bounds__(256, 3) void tf_ffn_bf3_kernel(const TfFfnBf3Args P) {
  __shared__ __attribute__((aligned(16))) __bf16 s_t[3 * FF_IMG];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  // workgroup -> (row tile, slice of d_ff): workgroups are dealt round robin to the 8 XCDs, each with its own L2.  An XCD takes
  // nslice / 8 slices for ALL row tiles, so it fetches 2 x 2 x 96 KB of weight images and the 1 MB of rows instead of every
  // slice's images (3 MB per XCD: 24 MB of the launch's fetches, rocprofv3 FETCH_SIZE) for a few row tiles.
  const int dff = P.dff, nslice = dff / 128;
  int slice;
  long tile;
  if (nslice % 8 == 0) {
    const int spx = nslice / 8;
    const long j = blockIdx.x / 8;
    slice = (int)(blockIdx.x % 8) * spx + (int)(j % spx);
    tile = j / spx;
  } else {
    const long ntiles = (P.R + 63) / 64;
    tile = blockIdx.x % ntiles, slice = (int)(blockIdx.x / ntiles);
  }
  const long row0 = tile * 64;
  const int c0 = slice * 128;
  const int c4 = tid & 31, r0 = tid >> 5;
  // ---- the rows' tile: fp32 -> three bf16 images
  {
    f32x4 stg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) stg[i] = ld4(P.x + (size_t)min(row0 + r0 + 8 * i, P.R - 1) * D + c4 * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ff_bf16x4 p0, p1, p2;
      ff_split4(stg[i], p0, p1, p2);
      __bf16 *d = s_t + (r0 + 8 * i) * FF_LD + 4 * c4;
      *reinterpret_cast<ff_bf16x4 *>(d) = p0;
      *reinterpret_cast<ff_bf16x4 *>(d + FF_IMG) = p1;
      *reinterpret_cast<ff_bf16x4 *>(d + 2 * FF_IMG) = p2;
    }
  }
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};   // smallest piece products first
  // weight operand of a product, step kc: wq[t][piece] = 8 consecutive contraction indices of output unit 32 w + 16 t + l15.  The
  // piece images are stored in OPERAND ORDER (tf_ffn_split_kernel): the 64 lanes of one load read 1 KB of consecutive bytes
  // (per-lane 16-byte reads at a row pitch of 256 B / 4 KB ran the kernel at the L1's line rate, not the matrix rate)
  auto wload = [&](const __bf16 *base, size_t img, int kc, ff_bf16x8 (*wq)[3]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) wq[t][p] = *reinterpret_cast<const ff_bf16x8 *>(base + p * img + (size_t)(((t * 4 + kc) * 64 + lane) * 8));
  };
  // one product: acc[t][mt][u] = out[row 16 mt + l15][unit 32 w + 16 t + 4 lg + u] (weights as the A operand: a lane ends up with
  // four consecutive units of one row)
  auto product = [&](const __bf16 *wbase, size_t img, f32x4 (*acc)[4]) {
    ff_bf16x8 wq[2][2][3];
    wload(wbase, img, 0, wq[0]);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      if (kc + 1 < 4) wload(wbase, img, kc + 1, wq[(kc + 1) & 1]);
      ff_bf16x8 a[4][3];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[mt][p] = *reinterpret_cast<const ff_bf16x8 *>(s_t + p * FF_IMG + (16 * mt + l15) * FF_LD + 32 * kc + 8 * lg);
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[t][mt] = FF_MFMA(wq[kc & 1][t][PA[q]], a[mt][PB[q]], acc[t][mt]);
    }
  };
  f32x4 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const size_t wimg = (size_t)dff * D, wofs = (size_t)(slice * 4 + w) * 4096;   // (slice, wave) block of 2 x 4 x 64 x 8 elements
  product(P.Wa + wofs, wimg, acc);
  f32x4 bb[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (!BWD && P.bias) bb[0] = ld4(P.bias + c0 + 32 * w + 4 * lg), bb[1] = ld4(P.bias + c0 + 32 * w + 16 + 4 * lg);
  const DropSeed sd = make_seed(P.seed, P.seed_dev);
  __syncthreads();   // every wave has read the rows' images for the last time: the hidden tile takes their place
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const long row = row0 + 16 * mt + l15;
    const bool valid = row < P.R;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int e = 32 * w + 16 * t + 4 * lg;   // the lane's four hidden units inside the slice
      f32x4 o = acc[t][mt];
      if (BWD) {
        f32x4 yv = {0.f, 0.f, 0.f, 0.f};
        if (valid) yv = ld4(P.y + (size_t)row * dff + c0 + e);
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = yv[u] > 0.f ? o[u] * P.scale : 0.f;
      } else {
        o += bb[t];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float v = fmaxf(o[u], 0.f);
          if (P.thresh != 0u) v = hash32((unsigned long long)row * dff + c0 + e + u, sd) >= P.thresh ? v * P.scale : 0.f;
          o[u] = valid ? v : 0.f;
        }
      }
      if (valid && P.hid) st4(P.hid + (size_t)row * dff + c0 + e, o);
      ff_bf16x4 p0, p1, p2;
      ff_split4(o, p0, p1, p2);
      __bf16 *d = s_t + (16 * mt + l15) * FF_LD + e;
      *reinterpret_cast<ff_bf16x4 *>(d) = p0;
      *reinterpret_cast<ff_bf16x4 *>(d + FF_IMG) = p1;
      *reinterpret_cast<ff_bf16x4 *>(d + 2 * FF_IMG) = p2;
      acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  product(P.Wb + wofs, wimg, acc);
  float *out = P.part + (size_t)slice * P.R * D;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const long row = row0 + 16 * mt + l15;
    if (row < P.R) {
#pragma unroll
      for (int t = 0; t < 2; ++t) st4(out + (size_t)row * D + 32 * w + 16 * t + 4 * lg, acc[t][mt]);
    }
  }
}

// the four piece images of up to 16 feed-forward layers in one launch: per layer W1 f32 [dff][128], W2 f32 [128][dff] ->
//   out + 0 img: first-product operands of the forward  (unit = hidden h, contraction k = model channel:  W1[h][k])
//   out + 1 img: second-product operands of the forward (unit = model channel m, contraction k = hidden:  W2[m][k])
//   out + 2 img: first-product operands of the backward (unit = hidden h, contraction k = model channel:  W2[k][h])
//   out + 3 img: second-product operands of the backward (unit = model channel m, contraction k = hidden: W1[k][m])
// img = 3 dff 128 elements (three pieces of dff 128).  Inside a piece the elements are in OPERAND ORDER of tf_ffn_bf3_kernel:
//   index = ((((slice c, wave w), t, kc), lane), e)  <->  unit 32 w + 16 t + (lane % 16) [+ 128 c for hidden units],
//   contraction index 32 kc + 8 (lane / 16) + e [+ 128 c when it runs over hidden units]
constexpr int FF_SPLIT_MAX = 16;
struct FfSplitTable {
  int nlayers, dff;
  const float *w1[FF_SPLIT_MAX], *w2[FF_SPLIT_MAX];
  __bf16 *out[FF_SPLIT_MAX];
};
__global__ __launch_bounds__(256) void tf_ffn_split_kernel(const FfSplitTable T) {
  const int layer = blockIdx.y, dff = T.dff;
  const size_t n = (size_t)dff * D, img = 3 * n;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63), kc = (int)((idx >> 9) & 3), t = (int)((idx >> 11) & 1),
            w = (int)((idx >> 12) & 3), c = (int)(idx >> 14);
  const int unit = 32 * w + 16 * t + (lane & 15), kk = 32 * kc + 8 * (lane >> 4) + e;   // inside the 128 x 128 block of slice c
  const float *W1 = T.w1[layer], *W2 = T.w2[layer];
  const int h_u = 128 * c + unit, h_k = 128 * c + kk;
  const float v[4] = {W1[(size_t)h_u * D + kk], W2[(size_t)unit * dff + h_k], W2[(size_t)kk * dff + h_u], W1[(size_t)h_k * D + unit]};
  __bf16 *o = T.out[layer];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 hh = (__bf16)v[i];
    const float r = v[i] - (float)hh;
    const __bf16 m = (__bf16)r;
    __bf16 *d = o + i * img + idx;
    d[0] = hh, d[n] = m, d[2 * n] = (__bf16)(r - (float)m);
  }
}

// One greedy-decoding step of self-attention over a key / value cache (models/transformer_captioner.py:402-453: the
// reference re-runs the whole decoder prefix for every new word; with pre-norm layers and a causal mask the newest row of
// that recomputation equals this incremental step).  One workgroup per sequence: the new token's k, v (from its packed
// q|k|v row) are appended at position t of the caches [R][T][h*16], thread (head, key) forms one logit, a 32-lane softmax
// per head, then thread (head, d) accumulates sum_key p[key] v[key][d] over coalesced 64-byte reads.  h = 8, d_k = 16, T <= 32.
__global__ __launch_bounds__(256) void decode_attn_kernel(const float *__restrict__ qkv, float *__restrict__ kc,
                                                          float *__restrict__ vc, int T, int t, float scale,
                                                          float *__restrict__ out) {
  constexpr int HD = 128;
  __shared__ float s_p[8][32];
  const int tid = threadIdx.x, hh = tid >> 5, tk = tid & 31;
  const size_t row = blockIdx.x;
  const float *me = qkv + row * 3 * HD;
  float *kr = kc + row * (size_t)T * HD, *vr = vc + row * (size_t)T * HD;
  if (tid < 32) st4(kr + (size_t)t * HD + tid * 4, ld4(me + HD + tid * 4));
  else if (tid < 64) st4(vr + (size_t)t * HD + (tid - 32) * 4, ld4(me + 2 * HD + (tid - 32) * 4));
  float logit = -INFINITY;
  if (tk <= t) {
    const float *kp = tk == t ? me + HD + hh * 16 : kr + (size_t)tk * HD + hh * 16;   // (position t: straight from the row)
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 kv = ld4(kp + 4 * q), qv = ld4(me + hh * 16 + 4 * q);
      a += qv[0] * kv[0] + qv[1] * kv[1] + qv[2] * kv[2] + qv[3] * kv[3];
    }
    logit = a * scale;
  }
  float mx = logit;
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  const float e = tk <= t ? __expf(logit - mx) : 0.f;
  float sum = e;
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
  s_p[hh][tk] = e / sum;
  __syncthreads();
  // thread (head, half, d): keys of its parity
  const int d = tk & 15, par = tk >> 4;
  float acc = 0.f;
  for (int k2 = par; k2 <= t; k2 += 2) {
    const float v = k2 == t ? me[2 * HD + hh * 16 + d] : vr[(size_t)k2 * HD + hh * 16 + d];
    acc += s_p[hh][k2] * v;
  }
  acc += __shfl_xor(acc, 16);
  if (par == 0) out[row * HD + hh * 16 + d] = acc;
}

inline bool drop_params(float p, unsigned &thresh, float &scale) {
  if (!(p >= 0.f && p < 1.f)) return false;
  thresh = p > 0.f ? (unsigned)((double)p * 4294967296.0) : 0u;
  scale = 1.0f / (1.0f - p);
  return true;
}
inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int spacap_tf_rows_f32(const spacap_tf_rows_args *a, spacap_stream_t stream) {
  const char *what = "spacap_tf_rows_f32";
  SPACAP_REQUIRE(a != nullptr, "%s: null argument block", what);
  SPACAP_REQUIRE(a->R >= 0 && (a->mode == 0 || a->mode == 1), "%s: bad R / mode", what);
  if (a->R == 0) return SPACAP_OK;
  TfRowsArgs P;
  P.R = a->R;
  P.A1 = a->a1, P.W1 = a->w1, P.bias1 = a->bias1, P.K1 = a->k1;
  SPACAP_REQUIRE(drop_params(a->drop_p, P.drop_thresh, P.drop_scale), "%s: bad dropout probability %f", what, (double)a->drop_p);
  P.eps = a->eps, P.seed = a->seed, P.seed_dev = reinterpret_cast<const unsigned long long *>(a->seed_dev);
  P.res = a->res, P.x_out = a->x_out, P.ln_a = a->ln_a, P.ln_b = a->ln_b, P.n_out = a->n_out, P.stats = a->stats;
  P.x_ln = a->x_ln, P.G = a->g, P.part = a->part, P.W2 = a->w2, P.bias2 = a->bias2, P.out2 = a->out2, P.N2 = a->n2;
  P.nparts = a->nparts;
  P.attn_out = a->attn_out, P.delta_out = a->delta_out, P.Lq = a->lq;
  SPACAP_REQUIRE(!P.delta_out || (a->mode == 1 && P.N2 == 128 && P.attn_out && P.Lq >= 1 && a->R % P.Lq == 0 && al16(P.attn_out)),
                 "%s: delta_out needs mode 1, n2 = 128, attn_out and lq dividing R", what);
  SPACAP_REQUIRE(P.nparts >= 0 && (P.nparts == 0 || P.A1), "%s: nparts = %d needs the partial sums in a1", what, P.nparts);
  const bool bwd = a->mode == 1;
  if (P.A1 && P.nparts == 0) SPACAP_REQUIRE(P.W1 && P.K1 >= 128 && P.K1 % 128 == 0, "%s: first product needs weights and K1 a multiple of 128 (K1=%d)", what, P.K1);
  SPACAP_REQUIRE(P.ln_a && (bwd || P.ln_b), "%s: LayerNorm parameters missing", what);
  if (!bwd) {
    SPACAP_REQUIRE(P.res, "%s: forward needs the residual rows", what);
    SPACAP_REQUIRE(P.N2 == 0 || (P.N2 % 128 == 0 && P.W2 && P.out2), "%s: N2=%d unsupported (multiple of 128) or null pointer", what, P.N2);
  } else {
    SPACAP_REQUIRE(P.A1 || P.G, "%s: backward needs a1 or g", what);
    SPACAP_REQUIRE(P.x_ln && P.stats && P.x_out && P.part, "%s: backward needs x_ln, stats, x_out (dx) and part", what);
    SPACAP_REQUIRE(P.N2 == 0 || (P.N2 == 128 && P.W2 && P.out2), "%s: backward N2 must be 0 or 128", what);
  }
  SPACAP_REQUIRE(al16(P.A1) && al16(P.W1) && al16(P.bias1) && al16(P.res) && al16(P.x_out) && al16(P.ln_a) && al16(P.ln_b) &&
                     al16(P.n_out) && al16(P.x_ln) && al16(P.G) && al16(P.W2) && al16(P.bias2) && al16(P.out2),
                 "%s: pointers must be 16-byte aligned", what);
  const long tiles = (P.R + BM - 1) / BM;
  SPACAP_REQUIRE(tiles <= 2147483647L, "%s: too many rows", what);
  hipStream_t s = spacap::as_stream(stream);
  // forward, N2 = 384 (the next attention's packed q|k|v) on FEW row tiles (the decoder's 8 x ~30 tokens = 16 tiles): its three
  // 128-column blocks on three workgroups per row tile -- 17 -> 11 us per launch there; with the encoder's 128 row tiles the
  // repeated first part costs what the shorter second part saves (measured 17.5 us both ways), so those stay one workgroup
  const unsigned ny = (!bwd && P.N2 > 128 && tiles <= 64) ? (unsigned)(P.N2 / 128) : 1u;
  if (bwd) hipLaunchKernelGGL((tf_rows_kernel<true>), dim3((unsigned)tiles), dim3(256), 0, s, P);
  else hipLaunchKernelGGL((tf_rows_kernel<false>), dim3((unsigned)tiles, ny), dim3(256), 0, s, P);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_tf_rows_parts(long R) { return (int)((R + BM - 1) / BM); }

namespace {
template <int EPI>
int launch_gemm(const char *what, const TfGemmArgs &P, bool nn, long R, int N, int nsplit, hipStream_t s) {
  const long tiles = (R + FM - 1) / FM;
  SPACAP_REQUIRE(tiles <= 2147483647L, "%s: too many rows", what);
  const dim3 grid((unsigned)tiles, N / 128, nsplit);
  if (nn) hipLaunchKernelGGL((tf_gemm_kernel<EPI, true>), grid, dim3(256), 0, s, P);
  else hipLaunchKernelGGL((tf_gemm_kernel<EPI, false>), grid, dim3(256), 0, s, P);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
}  // namespace

extern "C" int spacap_tf_ffn1_f32(const float *x, const float *W, const float *bias, long R, int N, float drop_p, uint64_t seed,
                                  const uint64_t *seed_dev, float *h, spacap_stream_t stream) {
  const char *what = "spacap_tf_ffn1_f32";
  TfGemmArgs P = {};
  SPACAP_REQUIRE(R >= 0 && N >= 128 && N % 128 == 0 && drop_params(drop_p, P.thresh, P.scale), "%s: (R=%ld, N=%d, p=%f) unsupported",
                 what, R, N, (double)drop_p);
  if (R == 0) return SPACAP_OK;
  SPACAP_REQUIRE(x && W && h && al16(x) && al16(W) && al16(bias) && al16(h), "%s: null or unaligned pointer", what);
  P.x = x, P.W = W, P.bias = bias, P.out = h, P.R = R, P.ldx = D, P.ldw = D, P.N = N, P.KS = D;
  P.seed = seed, P.seed_dev = reinterpret_cast<const unsigned long long *>(seed_dev);
  return launch_gemm<1>(what, P, false, R, N, 1, spacap::as_stream(stream));
}

// mode 0: hid = dropout(relu(x Wa^T + bias)) [R,dff] (Wa = w_1 [dff,128]), part[c] = hid[:, c] Wb[:, c]^T (Wb = w_2 [128,dff]);
// mode 1: hid = (x Wa[:, c]) * [y > 0] / (1 - p) (Wa = w_2, y = the saved forward hid), part[c] = hid[:, c] Wb[c] (Wb = w_1).
extern "C" int spacap_tf_ffn_f32(int mode, const float *x, const float *Wa, const float *Wb, const float *bias, const float *y, long R,
                                 int dff, float drop_p, uint64_t seed, const uint64_t *seed_dev, float *hid, float *part,
                                 spacap_stream_t stream) {
  const char *what = "spacap_tf_ffn_f32";
  TfFfnArgs P = {};
  SPACAP_REQUIRE((mode == 0 || mode == 1) && R >= 0 && dff >= 128 && dff % 128 == 0 && drop_params(drop_p, P.thresh, P.scale),
                 "%s: (mode=%d, R=%ld, dff=%d, p=%f) unsupported", what, mode, R, dff, (double)drop_p);
  if (R == 0) return SPACAP_OK;
  SPACAP_REQUIRE(x && Wa && Wb && (hid || mode == 0) && part && (mode == 0 || y) && al16(x) && al16(Wa) && al16(Wb) && al16(bias) && al16(y) &&
                     al16(hid) && al16(part), "%s: null or unaligned pointer", what);
  P.x = x, P.Wa = Wa, P.Wb = Wb, P.bias = bias, P.y = y, P.hid = hid, P.part = part, P.R = R, P.dff = dff;
  P.seed = seed, P.seed_dev = reinterpret_cast<const unsigned long long *>(seed_dev);
  const bool small = R <= 512;   // few rows: 16-row tiles, so that the launch still has a few hundred workgroups
  const long tiles = small ? (R + 15) / 16 : (R + 63) / 64;
  SPACAP_REQUIRE(tiles <= 2147483647L, "%s: too many rows", what);
  const dim3 grid((unsigned)tiles, dff / 128);
  hipStream_t s = spacap::as_stream(stream);
  if (mode == 0) {
    if (small) hipLaunchKernelGGL((tf_ffn_kernel<false, 1>), grid, dim3(256), 0, s, P);
    else hipLaunchKernelGGL((tf_ffn_kernel<false, 4>), grid, dim3(256), 0, s, P);
  } else {
    if (small) hipLaunchKernelGGL((tf_ffn_kernel<true, 1>), grid, dim3(256), 0, s, P);
    else hipLaunchKernelGGL((tf_ffn_kernel<true, 4>), grid, dim3(256), 0, s, P);
  }
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

/* bf16 elements of the piece images of ONE feed-forward layer (four images of 3 dff 128 elements, see tf_ffn_split_kernel) */
extern "C" long spacap_tf_ffn_pieces_elems(int dff) { return dff >= 128 && dff % 128 == 0 ? 4L * 3 * dff * D : 0; }
/* pieces[l] (device, spacap_tf_ffn_pieces_elems(dff) bf16 elements each) <- the split images of w1[l] f32 [dff,128], w2[l] f32 [128,dff];
   the pointer arrays are HOST arrays, read before the call returns; one launch per 16 layers. */
extern "C" int spacap_tf_ffn_split_f32(const float *const *w1, const float *const *w2, void *const *pieces, int nlayers, int dff,
                                       spacap_stream_t stream) {
  const char *what = "spacap_tf_ffn_split_f32";
  SPACAP_REQUIRE(nlayers >= 0 && dff >= 128 && dff % 128 == 0 && (nlayers == 0 || (w1 && w2 && pieces)), "%s: bad arguments", what);
  hipStream_t s = spacap::as_stream(stream);
  for (int l0 = 0; l0 < nlayers; l0 += FF_SPLIT_MAX) {
    FfSplitTable T;
    T.nlayers = std::min(FF_SPLIT_MAX, nlayers - l0), T.dff = dff;
    for (int l = 0; l < T.nlayers; ++l) {
      SPACAP_REQUIRE(w1[l0 + l] && w2[l0 + l] && pieces[l0 + l], "%s: null pointer (layer %d)", what, l0 + l);
      T.w1[l] = w1[l0 + l], T.w2[l] = w2[l0 + l], T.out[l] = static_cast<__bf16 *>(pieces[l0 + l]);
    }
    hipLaunchKernelGGL(tf_ffn_split_kernel, dim3((unsigned)((size_t)dff * D / 256), T.nlayers), dim3(256), 0, s, T);
  }
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
/* spacap_tf_ffn_f32 on split-bf16 products (fp32-equivalent), for R > 512 rows: `pieces` = the layer's images from
   spacap_tf_ffn_split_f32 (mode 0 reads the W1 / W2 images, mode 1 the transposed ones); everything else as spacap_tf_ffn_f32. */
extern "C" int spacap_tf_ffn_bf3_f32(int mode, const float *x, const void *pieces, const float *bias, const float *y, long R, int dff,
                                     float drop_p, uint64_t seed, const uint64_t *seed_dev, float *hid, float *part, spacap_stream_t stream) {
  const char *what = "spacap_tf_ffn_bf3_f32";
  TfFfnBf3Args P = {};
  SPACAP_REQUIRE((mode == 0 || mode == 1) && R >= 0 && dff >= 128 && dff % 128 == 0 && drop_params(drop_p, P.thresh, P.scale),
                 "%s: (mode=%d, R=%ld, dff=%d, p=%f) unsupported", what, mode, R, dff, (double)drop_p);
  if (R == 0) return SPACAP_OK;
  SPACAP_REQUIRE(x && pieces && (hid || mode == 0) && part && (mode == 0 || y) && al16(x) && al16(pieces) && al16(bias) && al16(y) && al16(hid) &&
                     al16(part), "%s: null or unaligned pointer", what);
  const size_t img = (size_t)3 * dff * D;
  const __bf16 *pc = static_cast<const __bf16 *>(pieces);
  P.x = x, P.bias = bias, P.y = y, P.hid = hid, P.part = part, P.R = R, P.dff = dff;
  P.Wa = pc + (mode == 0 ? 0 : 2) * img, P.Wb = pc + (mode == 0 ? 1 : 3) * img;
  P.seed = seed, P.seed_dev = reinterpret_cast<const unsigned long long *>(seed_dev);
  const long tiles = (R + 63) / 64;
  SPACAP_REQUIRE(tiles * (dff / 128) <= 2147483647L, "%s: too many rows", what);
  const dim3 grid((unsigned)(tiles * (dff / 128)));
  hipStream_t s = spacap::as_stream(stream);
  if (mode == 0) hipLaunchKernelGGL((tf_ffn_bf3_kernel<false>), grid, dim3(256), 0, s, P);
  else hipLaunchKernelGGL((tf_ffn_bf3_kernel<true>), grid, dim3(256), 0, s, P);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_decode_attn_f32(const float *qkv, float *kcache, float *vcache, long R, int h, int d_k, int T, int t, float scale,
                                      float *out, spacap_stream_t stream) {
  const char *what = "spacap_decode_attn_f32";
  SPACAP_REQUIRE(R >= 0 && h == 8 && d_k == 16 && T >= 1 && T <= 32 && t >= 0 && t < T, "%s: (R=%ld, h=%d, d_k=%d, T=%d, t=%d) unsupported",
                 what, R, h, d_k, T, t);
  if (R == 0) return SPACAP_OK;
  SPACAP_REQUIRE(qkv && kcache && vcache && out && al16(qkv) && al16(kcache) && al16(vcache) && al16(out), "%s: null or unaligned pointer", what);
  SPACAP_REQUIRE(R <= 2147483647L, "%s: too many sequences", what);
  hipLaunchKernelGGL(decode_attn_kernel, dim3((unsigned)R), dim3(256), 0, spacap::as_stream(stream), qkv, kcache, vcache, T, t, scale, out);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ---- greedy decoding: vocabulary projection + arg-max without the logits, and the next token's embedding ---------------------
// (models/transformer_captioner.py:93-100 Generator: log_softmax(proj(x)) and :441-447: `_, next_word = torch.max(prob, dim=1)`;
// the arg-max of the log-softmax is the arg-max of the logits.)  Round 4 ran F.linear on 2 048 x 3 001 logits + torch.argmax for
// each of the 31 words.  Here a workgroup owns 16 sequences and one slice of the vocabulary: the slice's weight rows go through
// LDS 64 at a time (next chunk's loads in flight), logits come out of v_mfma_f32_16x16x4_f32 (exact fp32 products) 16 words per
// wave, and every lane keeps the running (best logit, first index) of its rows; the slices' winners [R][NS] are merged by
// decode_next_kernel, which also writes the word into the caption and forms the next input row lut[word] sqrt(d) + pe[t].
// Arithmetic: split-bf16 (three bf16 pieces per operand, the six piece products above 2^-24 on v_mfma_f32_16x16x32_bf16:
// fp32-equivalent logits at 6/16 of the fp32-MFMA time -- 1.6 GFLOP per word on the fp32 pipe alone is 10 us).  The weight's
// pieces Wp bf16 [3][V][128] are made once per decoding call (spacap_gemm_bf3_split_w_f32).
typedef __bf16 va_bf16x8 __attribute__((ext_vector_type(8)));
#define VA_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
constexpr int VA_CHUNK = 64, VA_LDB = D + 8, VA_ROWS = 16;
__global__ __launch_bounds__(256) void vocab_argmax_kernel(const float *__restrict__ x, const __bf16 *__restrict__ Wp, const float *__restrict__ bias,
                                                           long R, int V, int per_slice, float *__restrict__ best_v, int *__restrict__ best_i) {
  __shared__ __attribute__((aligned(16))) __bf16 s_w[3 * VA_CHUNK * VA_LDB];
  __shared__ float s_bv[4][VA_ROWS][17];
  __shared__ int s_bi[4][VA_ROWS][17];
  constexpr int IMGW = VA_CHUNK * VA_LDB;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const long row0 = (long)blockIdx.x * VA_ROWS;
  const int ns = gridDim.y, sl = blockIdx.y;
  const int v_beg = sl * per_slice, v_end = min(V, v_beg + per_slice);
  // the sequences' rows as the A operand, split once: a[kc][piece] = pieces of x[row0 + l15][32 kc + 8 lg .. + 7]
  va_bf16x8 a[D / 32][3];
  {
    const float *xr = x + (size_t)min(row0 + l15, R - 1) * D + 8 * lg;
#pragma unroll
    for (int kc = 0; kc < D / 32; ++kc) {
      const f32x4 lo = ld4(xr + 32 * kc), hi = ld4(xr + 32 * kc + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? lo[e] : hi[e - 4];
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        a[kc][0][e] = h, a[kc][1][e] = m, a[kc][2][e] = (__bf16)(r1 - (float)m);
      }
    }
  }
  float bv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {v_beg, v_beg, v_beg, v_beg};
  // staging: per piece 64 rows x 16 sixteen-byte pieces = 1 024 loads: 4 per thread and piece
  const int c8 = tid & 15, r0 = tid >> 4;
  const size_t wimg = (size_t)V * D;
  va_bf16x8 stg[3][4];
  auto fetch = [&](int v0) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int v = v0 + r0 + 16 * i;
        stg[p][i] = *reinterpret_cast<const va_bf16x8 *>(Wp + p * wimg + (size_t)min(v, V - 1) * D + 8 * c8);
      }
  };
  fetch(v_beg < V ? v_beg : 0);
  for (int v0 = v_beg; v0 < v_end; v0 += VA_CHUNK) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<va_bf16x8 *>(s_w + p * IMGW + (r0 + 16 * i) * VA_LDB + 8 * c8) = stg[p][i];
    __syncthreads();
    if (v0 + VA_CHUNK < v_end) fetch(v0 + VA_CHUNK);
    const int v = v0 + 16 * w + l15;                    // this lane's word of the chunk
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int kc = 0; kc < D / 32; ++kc) {
      va_bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const va_bf16x8 *>(s_w + p * IMGW + (16 * w + l15) * VA_LDB + 32 * kc + 8 * lg);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        if (kc & 1) acc2 = VA_MFMA(a[kc][PA[q]], b[PB[q]], acc2);
        else acc = VA_MFMA(a[kc][PA[q]], b[PB[q]], acc);
      }
    }
    if (v < v_end) {
      const float bsv = bias[v];
#pragma unroll
      for (int u = 0; u < 4; ++u) {      // logit of sequence row0 + 4 lg + u, word v; words arrive in increasing order
        const float val = (acc[u] + acc2[u]) + bsv;
        if (val > bv[u]) bv[u] = val, bi[u] = v;
      }
    }
  }
  // merge across the 16 lanes and the 4 waves; ties go to the smaller word index (torch.max: first maximum)
#pragma unroll
  for (int u = 0; u < 4; ++u) s_bv[w][4 * lg + u][l15] = bv[u], s_bi[w][4 * lg + u][l15] = bi[u];
  __syncthreads();
  if (tid < 64) {
    const int row = tid >> 2, ww = tid & 3;          // four threads per sequence, one wave's 16 candidates each
    float m = -INFINITY;
    int mi = 0x7fffffff;
#pragma unroll
    for (int l = 0; l < 16; ++l) {
      const float val = s_bv[ww][row][l];
      const int idx = s_bi[ww][row][l];
      if (val > m || (val == m && idx < mi)) m = val, mi = idx;
    }
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
      const float om = __shfl_xor(m, o);
      const int oi = __shfl_xor(mi, o);
      if (om > m || (om == m && oi < mi)) m = om, mi = oi;
    }
    if (ww == 0 && row0 + row < R) {
      best_v[(size_t)(row0 + row) * ns + sl] = m;
      best_i[(size_t)(row0 + row) * ns + sl] = mi;
    }
  }
}

// word[r] = the best of the NS slice winners (first maximum); ys[r][t_out] = word; x[r, :] = lut[word] * scale + pe_row
__global__ __launch_bounds__(256) void decode_next_kernel(const float *__restrict__ best_v, const int *__restrict__ best_i, int ns, long R,
                                                          const float *__restrict__ lut, float scale, const float *__restrict__ pe_row,
                                                          long long *__restrict__ ys, int ys_ld, int t_out, float *__restrict__ x) {
  const long r = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int c4 = threadIdx.x & 31;
  if (r >= R) return;
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int s = 0; s < ns; ++s) {
    const float val = best_v[(size_t)r * ns + s];
    const int idx = best_i[(size_t)r * ns + s];
    if (val > m || (val == m && idx < mi)) m = val, mi = idx;
  }
  if (c4 == 0) ys[(size_t)r * ys_ld + t_out] = mi;
  const f32x4 e = ld4(lut + (size_t)mi * D + 4 * c4), p = ld4(pe_row + 4 * c4);
  st4(x + (size_t)r * D + 4 * c4, f32x4{e[0] * scale + p[0], e[1] * scale + p[1], e[2] * scale + p[2], e[3] * scale + p[3]});
}

/* One greedy-decoding step's word choice (models/transformer_captioner.py:441-447 with the Generator of :93-100): x f32 [R,128] the
   decoder's output rows, Wp bf16 [3][V][128] = the pieces of the projection weight (spacap_gemm_bf3_split_w_f32), bias f32 [V]
   -> ys i64 [R][ys_ld] column t_out = arg-max word (first maximum), and the next
   step's input rows x_next f32 [R,128] = lut[word] * scale + pe_row (lut f32 [V,128], pe_row f32 [128]).
   workspace: spacap_decode_word_workspace_bytes(R, V) bytes (the vocabulary slices' winners). */
namespace {
inline int va_slices(long R, int V) {
  const long tiles = (R + VA_ROWS - 1) / VA_ROWS;
  long ns = (4L * spacap::device_cus() + tiles - 1) / tiles;   // ~4 workgroups per CU
  const long most = (V + VA_CHUNK - 1) / VA_CHUNK;
  if (ns > most) ns = most;
  if (ns > 64) ns = 64;
  return (int)(ns < 1 ? 1 : ns);
}
}  // namespace
extern "C" size_t spacap_decode_word_workspace_bytes(long R, int V) { return R > 0 && V > 0 ? (size_t)R * va_slices(R, V) * 8 : 0; }
extern "C" int spacap_decode_word_f32(const float *x, const void *W, const float *bias, long R, int V, const float *lut, float scale,
                                      const float *pe_row, int64_t *ys, int ys_ld, int t_out, float *x_next, void *workspace,
                                      spacap_stream_t stream) {
  const char *what = "spacap_decode_word_f32";
  SPACAP_REQUIRE(R >= 0 && V >= 1 && ys_ld >= 1 && t_out >= 0 && t_out < ys_ld, "%s: bad sizes", what);
  if (R == 0) return SPACAP_OK;
  SPACAP_REQUIRE(x && W && bias && lut && pe_row && ys && x_next && workspace && al16(x) && al16(W) && al16(lut) && al16(pe_row) && al16(x_next),
                 "%s: null or unaligned pointer", what);
  SPACAP_REQUIRE(R <= 16L * 2147483647L, "%s: too many sequences", what);
  const int ns = va_slices(R, V);
  const int per = ((V + ns - 1) / ns + VA_CHUNK - 1) / VA_CHUNK * VA_CHUNK;   // whole chunks per slice
  float *bv = static_cast<float *>(workspace);
  int *bi = reinterpret_cast<int *>(bv + (size_t)R * ns);
  hipStream_t s = spacap::as_stream(stream);
  hipLaunchKernelGGL(vocab_argmax_kernel, dim3((unsigned)((R + VA_ROWS - 1) / VA_ROWS), ns), dim3(256), 0, s, x, static_cast<const __bf16 *>(W),
                     bias, R, V, per, bv, bi);
  hipLaunchKernelGGL(decode_next_kernel, dim3((unsigned)((R + 7) / 8)), dim3(256), 0, s, bv, bi, ns, R, lut, scale, pe_row,
                     reinterpret_cast<long long *>(ys), ys_ld, t_out, x_next);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// number of K slices that fills the chip with 64 x 128 tiles (a divisor of K / 128)
extern "C" int spacap_tf_gemm_splits(long R, int K, int N) {
  if (R < 1 || K < 128 || K % 128 || N < 128 || N % 128) return 0;
  const long tiles = ((R + FM - 1) / FM) * (N / 128);
  const int chunks = K / 128;
  long want = 512 / tiles;
  if (want < 1) want = 1;
  int best = 1;
  for (int sidx = 1; sidx <= chunks; ++sidx)
    if (chunks % sidx == 0 && sidx <= want) best = sidx;
  return best;
}

extern "C" int spacap_tf_gemm_f32(const float *a, const float *W, long R, int K, int N, int trans_w, int nsplit, float *out,
                                  spacap_stream_t stream) {
  const char *what = "spacap_tf_gemm_f32";
  SPACAP_REQUIRE(R >= 0 && K >= 128 && K % 128 == 0 && N >= 128 && N % 128 == 0 && nsplit >= 1 && (K / 128) % nsplit == 0,
                 "%s: (R=%ld, K=%d, N=%d, nsplit=%d) unsupported", what, R, K, N, nsplit);
  if (R == 0) return SPACAP_OK;
  SPACAP_REQUIRE(a && W && out && al16(a) && al16(W) && al16(out), "%s: null or unaligned pointer", what);
  TfGemmArgs P = {};
  P.x = a, P.W = W, P.out = out, P.R = R, P.ldx = K, P.ldw = trans_w ? K : N, P.N = N, P.KS = K / nsplit;
  return launch_gemm<0>(what, P, !trans_w, R, N, nsplit, spacap::as_stream(stream));
}

extern "C" int spacap_tf_dgrad_mask_f32(const float *g, const float *W, const float *y, float scale, long R, int K, int N,
                                        float *out, spacap_stream_t stream) {
  const char *what = "spacap_tf_dgrad_mask_f32";
  SPACAP_REQUIRE(R >= 0 && K == 128 && N >= 128 && N % 128 == 0, "%s: (R=%ld, K=%d, N=%d) unsupported", what, R, K, N);
  if (R == 0) return SPACAP_OK;
  SPACAP_REQUIRE(g && W && y && out && al16(g) && al16(W) && al16(y) && al16(out), "%s: null or unaligned pointer", what);
  TfGemmArgs P = {};
  P.x = g, P.W = W, P.y = y, P.out = out, P.R = R, P.ldx = K, P.ldw = N, P.N = N, P.KS = K, P.scale = scale;
  return launch_gemm<2>(what, P, true, R, N, 1, spacap::as_stream(stream));
}
