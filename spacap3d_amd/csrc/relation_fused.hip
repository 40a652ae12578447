// The relation head of the spatiality-guided encoder end to end on chip, forward and backward, for gfx950 (MI355X).
//
// Reference: models/transformer_captioner.py:319-326 (relation_proposal = Linear(128,128)-ReLU-Linear(128,128)-ReLU-Linear(128,9))
// and :392-397 (relation feature R[b,i,j,h*16+d] = P[b,h,i,j] V[b,h,j,d] of the last encoder layer, then the MLP on all B*K*K
// proposal pairs).  With U[b,j,h,:] = V[b,h,j,:] W1[:, 16h:16h+16]^T (a tiny product done by the caller, csrc/relation.hip):
//     hid1[b,i,j,:] = relu(b1 + sum_h P[b,h,i,j] U[b,j,h,:])      hid2 = relu(hid1 W2^T + b2)      pred = hid2 W3^T + b3
// As separate operators the head moves hid1 (268 MB at B = 8, K = 256) five times and the equally large dz2 / dhid1 twice
// each: 2.9 GB of HBM traffic around 51 GFLOP.  Here a workgroup owns 8 key columns j of one scene (U[b, j-block] stays in
// registers for the whole launch) and walks the queries in blocks of 8: one tile = 8 x 8 = 64 pair rows.
//   forward   hid1 tile -> LDS -> layer 2 (weights in registers) -> + b2, ReLU -> hid2 (stored: the backward's only large
//             input) -> layer 3 (9 outputs) from the same registers -> pred.
//   backward  hid1 tile recomputed from P, U; dz2 = (dpred W3) * [hid2 > 0] -> LDS; dW3 += dpred^T hid2; dW2 += dz2^T hid1;
//             dhid1 = dz2 W2; dz1 = dhid1 * [hid1 > 0]; then layer 1's gradients from the tile: dP[b,h,i,j] =
//             dz1 . U[b,j,h,:] (stored), dU[b,j,h,:] += P dz1 (registers, written once at the end); the bias sums ride the
//             same products as a column of ones; all parameter sums leave as per-workgroup partials (fixed order, no atomics).
// Every product, the small per-key ones included (block-diagonal operands, two key columns per 16-row tile), runs on the
// matrix cores: the backward has one wave per SIMD, so VALU phases would leave the matrix pipe idle.
// Arithmetic: the three 128 x 128 products (layer 2, dhid1, dW2: 94 % of the flops) as split-bf16, three bf16 pieces per
// operand and the six piece products above 2^-24 on v_mfma_f32_16x16x32_bf16 (fp32-equivalent, 6/16 of the fp32-MFMA time);
// the operands are split once by their producer and live in LDS as bf16 images, row-major where the contraction runs over
// channels and channel-major (the producing MFMA issued with its operands swapped) where it runs over the rows.  The small
// products use v_mfma_f32_16x16x4_f32 (exact fp32).  H = 8 heads, 128 channels, 9 outputs, K a multiple of 8.
#include <stdlib.h>
#include <atomic>

#include "common.hpp"

namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA_B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

constexpr int H = 8, C = 128, NO = 9, LDT = C + 8, TI = 8, TJ = 8, TR = TI * TJ;   // 64 pair rows per tile
constexpr int DPL = 12;                      // row stride of 9-wide tiles in LDS (9 values + 3 zeros: three k steps of 4)
constexpr int LDB = C + 8, IMG = TR * LDB;   // row-major bf16 piece [64 rows][128 + 8]: 272-byte rows, 16-byte reads conflict free
constexpr int LDR = TR + 8, IMGT = C * LDR;  // channel-major bf16 piece [128 channels][64 + 8]: 144-byte rows
constexpr int PART = C * C + NO * C + C + C + 16;   // per-workgroup partial: dW2 | dW3 | db1 | db2 | db3 (padded)

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }

// Tile rows are ordered key-major: row rr = jj * 8 + ii is the pair (query i0 + ii, key j0 + jj), so a 16-row matrix-core tile
// holds the 8 queries of two key columns and wave w owns the key columns 2 w, 2 w + 1 in every per-key product.
__device__ __forceinline__ size_t pair_row(int b, int K, int i0, int j0, int rr) { return ((size_t)b * K + i0 + (rr & 7)) * K + j0 + (rr >> 3); }

// ---- split-bf16 operands: x = x0 + x1 + x2 (three bf16 pieces, 24 mantissa bits), a product = the six piece products whose
// weight is above 2^-24, each exact in the fp32 accumulator of v_mfma_f32_16x16x32_bf16: 6/16 of the fp32-MFMA time ----------
__device__ __forceinline__ void split4(f32x4 v, bf16x4 &p0, bf16x4 &p1, bf16x4 &p2) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const __bf16 h = (__bf16)v[u];
    const float r = v[u] - (float)h;
    const __bf16 m = (__bf16)r;
    p0[u] = h, p1[u] = m, p2[u] = (__bf16)(r - (float)m);
  }
}
__device__ __forceinline__ void split8(f32x4 lo, f32x4 hi, bf16x8 *p) {
  bf16x4 a[3], c[3];
  split4(lo, a[0], a[1], a[2]);
  split4(hi, c[0], c[1], c[2]);
#pragma unroll
  for (int q = 0; q < 3; ++q) p[q] = bf16x8{a[q][0], a[q][1], a[q][2], a[q][3], c[q][0], c[q][1], c[q][2], c[q][3]};
}
__device__ __forceinline__ f32x4 mfma6(const bf16x8 *a, const bf16x8 *b, f32x4 acc) {   // smallest terms first
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
  for (int q = 0; q < 6; ++q) acc = MFMA_B(a[PA[q]], b[PB[q]], acc);
  return acc;
}
__device__ __forceinline__ void st_pieces(__bf16 *dst, int piece_stride, f32x4 v) {   // 3 x 8 bytes
  bf16x4 p0, p1, p2;
  split4(v, p0, p1, p2);
  *reinterpret_cast<bf16x4 *>(dst) = p0;
  *reinterpret_cast<bf16x4 *>(dst + piece_stride) = p1;
  *reinterpret_cast<bf16x4 *>(dst + 2 * piece_stride) = p2;
}
__device__ __forceinline__ void ld_pieces(const __bf16 *src, int piece_stride, bf16x8 *p) {   // 3 x 16 bytes
#pragma unroll
  for (int q = 0; q < 3; ++q) p[q] = *reinterpret_cast<const bf16x8 *>(src + q * piece_stride);
}

// the attention values of the next tile, two per thread (key index fastest in memory), stored as s_p[h][jj * 8 + ii]
__device__ __forceinline__ void p_tile_request(const float *__restrict__ P, int b, int K, int i0, int j0, float *pp) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = threadIdx.x + 256 * e, h = idx >> 6, ii = (idx >> 3) & 7, jj = idx & 7;
    pp[e] = P[(((size_t)b * H + h) * K + i0 + ii) * K + j0 + jj];
  }
}
__device__ __forceinline__ void p_tile_store(const float *pp, float *s_p) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = threadIdx.x + 256 * e, h = idx >> 6, ii = (idx >> 3) & 7, jj = idx & 7;
    s_p[h * TR + jj * TI + ii] = pp[e];
  }
}

// hid1 rows 16 w .. 16 w + 15 = relu(b1 + sum_h P U) on the matrix cores (exact fp32 products): contraction index k = (key
// column of the pair, head) with the attention operand block diagonal (zero where the row's key column is not k's) and the
// bias as a fifth k step.  uA[n][ks] = U[b, j0 + 2 w + (ks >> 1), 4 (ks & 1) + lg, 16 n + l15], bA[n] = lg == 0 ? b1[16 n + l15] : 0.
// TRANSPOSED = false: the lane gets 4 channels of row l15 -> row-major bf16 pieces [64][LDB]; true: 4 rows of channel l15 ->
// channel-major pieces [128][LDR].
template <bool TRANSPOSED>
__device__ __forceinline__ void hid1_rows(const float (*uA)[4], const float *bA, const float *s_p, int w, int l15, int lg, __bf16 *img) {
  const float p0 = s_p[lg * TR + 16 * w + l15], p1 = s_p[(4 + lg) * TR + 16 * w + l15];
  const bool hi = l15 >> 3;
  const float pB[4] = {hi ? 0.f : p0, hi ? 0.f : p1, hi ? p0 : 0.f, hi ? p1 : 0.f};
  const float one = lg == 0 ? 1.f : 0.f;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (TRANSPOSED) {
      acc = MFMA16(one, bA[n], acc);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = MFMA16(pB[ks], uA[n][ks], acc);
      st_pieces(img + (16 * n + l15) * LDR + 16 * w + 4 * lg, IMGT, relu4(acc));
    } else {
      acc = MFMA16(bA[n], one, acc);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = MFMA16(uA[n][ks], pB[ks], acc);
      st_pieces(img + (16 * w + l15) * LDB + 16 * n + 4 * lg, IMG, relu4(acc));
    }
  }
}

__global__ __launch_bounds__(256) void rel_fused_fwd_kernel(const float *__restrict__ P, const float *__restrict__ Um,
                                                            const float *__restrict__ b1, const float *__restrict__ W2,
                                                            const float *__restrict__ b2, const float *__restrict__ W3,
                                                            const float *__restrict__ b3, int B, int K, float *__restrict__ hid2,
                                                            float *__restrict__ pred) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16 *s_a = reinterpret_cast<__bf16 *>(smem_raw);                       // hid1 pieces [3][64][LDB]
  float *s_pr = reinterpret_cast<float *>(smem_raw + 3 * IMG * 2);          // layer 3 partial sums [4 waves][64][DPL]
  float *s_p = s_pr + 4 * TR * DPL;                                         // [8][64]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  // this workgroup's contiguous range of tile units (unit = (scene, key block, query block), query block fastest)
  const int NI = K / TI, NJ = K / TJ;
  const long U = (long)B * NJ * NI, u_beg = (long)blockIdx.x * U / gridDim.x, u_end = ((long)blockIdx.x + 1) * U / gridDim.x;
  // layer 2 weights of the wave's 32 output channels as pieces: wA[t][kc] = W2[32 w + 16 t + l15][32 kc + 8 lg .. + 7]
  bf16x8 wA[2][4][3];
  float w3r[2][4];
  f32x4 bb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float *wp = W2 + (size_t)(32 * w + 16 * t + l15) * C + 8 * lg;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) split8(ld4(wp + 32 * kc), ld4(wp + 32 * kc + 4), wA[t][kc]);
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) w3r[t][uu] = l15 < NO ? W3[l15 * C + 32 * w + 16 * t + 4 * lg + uu] : 0.f;
    bb[t] = ld4(b2 + 32 * w + 16 * t + 4 * lg);
  }
  float pp[2];
  for (long u = u_beg; u < u_end;) {
    const int grp = (int)(u / NI), ib0 = (int)(u - (long)grp * NI), ibe = (int)min((long)NI, ib0 + (u_end - u));
    const int b = grp / NJ, j0 = (grp - b * NJ) * TJ, ibeg = ib0 * TI, iend = ibe * TI;
    u += ibe - ib0;
    float uA[8][4], bA[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) uA[n][ks] = Um[(((size_t)b * K + j0 + 2 * w + (ks >> 1)) * H + 4 * (ks & 1) + lg) * C + 16 * n + l15];
      bA[n] = lg == 0 ? b1[16 * n + l15] : 0.f;
    }
    p_tile_request(P, b, K, ibeg, j0, pp);
    auto pred_out = [&](int i0) {   // the four waves' layer 3 partial sums + b3 -> pred, 8 runs (one per query) of 72 floats
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int idx = tid + 256 * e, ii = idx / (TJ * NO), off = idx - ii * (TJ * NO), j = off / NO, o = off - j * NO;
        if (idx < TR * NO) {
          const float *q = s_pr + (j * TI + ii) * DPL + o;
          pred[(((size_t)b * K + i0 + ii) * K + j0) * NO + off] = ((q[0] + q[TR * DPL]) + (q[2 * TR * DPL] + q[3 * TR * DPL])) + b3[o];
        }
      }
    };

    for (int i0 = ibeg; i0 < iend; i0 += TI) {
      __syncthreads();   // the previous tile's readers of s_p / s_a and writers of s_pr are done
      p_tile_store(pp, s_p);
      if (i0 > ibeg) pred_out(i0 - TI);
      if (i0 + TI < iend) p_tile_request(P, b, K, i0 + TI, j0, pp);
      __syncthreads();
      hid1_rows<false>(uA, bA, s_p, w, l15, lg, s_a);
      __syncthreads();
#pragma unroll 2
      for (int mt = 0; mt < TR / 16; ++mt) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          bf16x8 a[3];
          ld_pieces(s_a + (mt * 16 + l15) * LDB + 32 * kc + 8 * lg, IMG, a);
          acc[0] = mfma6(wA[0][kc], a, acc[0]);
          acc[1] = mfma6(wA[1][kc], a, acc[1]);
        }
        const int rr = mt * 16 + l15;
        float *hrow = hid2 + pair_row(b, K, i0, j0, rr) * C + 32 * w + 4 * lg;
        f32x4 a3 = {0.f, 0.f, 0.f, 0.f}, a3b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 h = relu4(acc[t] + bb[t]);
          st4(hrow + 16 * t, h);
          // layer 3 straight from the registers: k = the wave's channel 32 w + 16 t + 4 lg + uu
          a3 = MFMA16(w3r[t][0], h[0], a3);
          a3b = MFMA16(w3r[t][1], h[1], a3b);
          a3 = MFMA16(w3r[t][2], h[2], a3);
          a3b = MFMA16(w3r[t][3], h[3], a3b);
        }
        if (lg < 3) st4(&s_pr[(w * TR + rr) * DPL + 4 * lg], a3 + a3b);
      }
    }
    __syncthreads();
    pred_out(iend - TI);
  }
}

__global__ __launch_bounds__(256) void rel_fused_bwd_kernel(const float *__restrict__ dpred, const float *__restrict__ hid2,
                                                            const float *__restrict__ P, const float *__restrict__ U,
                                                            const float *__restrict__ b1, const float *__restrict__ W2,
                                                            const float *__restrict__ W3, int B, int K, int Z,
                                                            float *__restrict__ dP, float *__restrict__ dU, float *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16 *s_h1T = reinterpret_cast<__bf16 *>(smem_raw);                  // hid1 pieces, channel-major [3][128][LDR]
  __bf16 *s_zT = s_h1T + 3 * IMGT;                                       // dz2 pieces, channel-major; the same bytes later
  __bf16 *s_z = s_zT;                                                    //   hold them row-major [3][64][LDB]
  float *s_x = reinterpret_cast<float *>(smem_raw + 2 * 3 * IMGT * 2);   // [64][LDT]: the hid2 tile, then dz1
  float *s_p = s_x + TR * LDT, *s_dp = s_p + H * TR;                     // [8][64], [64][DPL]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int jj = tid >> 5, c4 = tid & 31;
  // this workgroup's contiguous range of tile units (unit = (scene, key block, query block), query block fastest)
  const int NI = K / TI, NJ = K / TJ;
  const long NU = (long)B * NJ * NI, u_beg = (long)blockIdx.x * NU / gridDim.x, u_end = ((long)blockIdx.x + 1) * NU / gridDim.x;
  // ---- operands that stay in registers for the whole launch ------------------------------------------------------------
  float bA[8], w3A[8][3];
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    bA[n] = lg == 0 ? b1[16 * n + l15] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) w3A[n][ks] = 4 * ks + lg < NO ? W3[(4 * ks + lg) * C + 16 * n + l15] : 0.f;
  }
  // dhid1 = dz2 W2 for the wave's 32 INPUT channels: wA[t][kc] = W2[32 kc + 8 lg .. + 7][32 w + 16 t + l15] as pieces
  bf16x8 wA[2][4][3];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const float *wp = W2 + (size_t)(32 * kc + 8 * lg) * C + 32 * w + 16 * t + l15;
      split8(f32x4{wp[0], wp[C], wp[2 * C], wp[3 * C]}, f32x4{wp[4 * C], wp[5 * C], wp[6 * C], wp[7 * C]}, wA[t][kc]);
    }
  const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
  // ---- running sums ----------------------------------------------------------------------------------------------------
  f32x4 gw[2][8], gb2[2], g3[2], d1[2], gb3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    gb2[t] = g3[t] = d1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 8; ++n) gw[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int idx = tid; idx < TR * DPL; idx += 256) s_dp[idx] = 0.f;   // (the three pad columns stay zero)

  // ---- the next tile's inputs travel while the current one is worked on -------------------------------------------------
  f32x4 hp[TI];
  float pp[2], dq[3];
  for (long u = u_beg; u < u_end;) {
  // ---- one key block (8 key columns of one scene): the stretch of its query blocks that falls into this workgroup's range --
  const int grp = (int)(u / NI), ib0 = (int)(u - (long)grp * NI), ibe = (int)min((long)NI, ib0 + (u_end - u));
  const int b = grp / NJ, j0 = (grp - b * NJ) * TJ, ibeg = ib0 * TI, iend = ibe * TI;
  u += ibe - ib0;
  float uA[8][4];
  f32x4 uP[8], du[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) uA[n][ks] = U[(((size_t)b * K + j0 + 2 * w + (ks >> 1)) * H + 4 * (ks & 1) + lg) * C + 16 * n + l15];
    // dP: row (key column l15 >> 3, head l15 & 7) of U, contraction over the channels
    uP[n] = ld4(U + (((size_t)b * K + j0 + 2 * w + (l15 >> 3)) * H + (l15 & 7)) * C + 16 * n + 4 * lg);
    du[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto request = [&](int i0) {
    p_tile_request(P, b, K, i0, j0, pp);
#pragma unroll
    for (int e = 0; e < 3; ++e) {   // dpred rows of the tile: 8 runs (one per query) of 72 contiguous floats
      const int idx = tid + 256 * e, ii = idx / (TJ * NO), off = idx - ii * (TJ * NO);
      dq[e] = idx < TR * NO ? dpred[(((size_t)b * K + i0 + ii) * K + j0) * NO + off] : 0.f;
    }
#pragma unroll
    for (int ii = 0; ii < TI; ++ii) hp[ii] = ld4(hid2 + pair_row(b, K, i0, j0, jj * TI + ii) * C + 4 * c4);
  };
  request(ibeg);

  for (int i0 = ibeg; i0 < iend; i0 += TI) {
    __syncthreads();   // the previous tile's readers of every LDS buffer are done
    p_tile_store(pp, s_p);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int idx = tid + 256 * e, ii = idx / (TJ * NO), off = idx - ii * (TJ * NO), j = off / NO, o = off - j * NO;
      if (idx < TR * NO) s_dp[(j * TI + ii) * DPL + o] = dq[e];
    }
#pragma unroll
    for (int ii = 0; ii < TI; ++ii) st4(&s_x[(jj * TI + ii) * LDT + 4 * c4], hp[ii]);
    if (i0 + TI < iend) request(i0 + TI);
    __syncthreads();
    float dB[3];   // dpred of row 16 w + l15, three k steps of 4 outputs (either operand side: same lane structure)
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) dB[ks] = s_dp[(16 * w + l15) * DPL + 4 * ks + lg];
    hid1_rows<true>(uA, bA, s_p, w, l15, lg, s_h1T);
    {
      // dz2 = (dpred W3) * [hid2 > 0] of the wave's 16 rows, the lane holding 4 ROWS of channel 16 n + l15 -> channel-major pieces
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) acc = MFMA16(dB[ks], w3A[n][ks], acc);
        const float *hx = s_x + (16 * w + 4 * lg) * LDT + 16 * n + l15;
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) acc[uu] = hx[uu * LDT] > 0.f ? acc[uu] : 0.f;
        st_pieces(s_zT + (16 * n + l15) * LDR + 16 * w + 4 * lg, IMGT, acc);
      }
      // dW3[o][c] += dpred^T hid2 for the channels 32 w .. 32 w + 31, db3 (wave 0) with a column of ones
#pragma unroll 4
      for (int ks = 0; ks < TR / 4; ++ks) {
        const float a = l15 < NO ? s_dp[(4 * ks + lg) * DPL + l15] : 0.f;
        const float *hr = s_x + (4 * ks + lg) * LDT + 32 * w + l15;
        g3[0] = MFMA16(a, hr[0], g3[0]);
        g3[1] = MFMA16(a, hr[16], g3[1]);
        if (w == 0) gb3 = MFMA16(a, 1.f, gb3);
      }
    }
    __syncthreads();   // both channel-major images complete
#pragma unroll 1
    for (int kc = 0; kc < TR / 32; ++kc) {   // dW2 += dz2^T hid1 (contraction over the 64 rows), db2 with a column of ones
      bf16x8 za[2][3];
      ld_pieces(s_zT + (32 * w + l15) * LDR + 32 * kc + 8 * lg, IMGT, za[0]);
      ld_pieces(s_zT + (32 * w + 16 + l15) * LDR + 32 * kc + 8 * lg, IMGT, za[1]);
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        bf16x8 hb[3];
        ld_pieces(s_h1T + (16 * n + l15) * LDR + 32 * kc + 8 * lg, IMGT, hb);
        gw[0][n] = mfma6(za[0], hb, gw[0][n]);
        gw[1][n] = mfma6(za[1], hb, gw[1][n]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 2; q >= 0; --q) gb2[t] = MFMA_B(za[t][q], ones, gb2[t]);
    }
    __syncthreads();   // the channel-major dz2 image is consumed: its bytes take the row-major one
    {
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) acc = MFMA16(w3A[n][ks], dB[ks], acc);
        const f32x4 h2 = ld4(&s_x[(16 * w + l15) * LDT + 16 * n + 4 * lg]);
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) acc[uu] = h2[uu] > 0.f ? acc[uu] : 0.f;
        st_pieces(s_z + (16 * w + l15) * LDB + 16 * n + 4 * lg, IMG, acc);
      }
    }
    __syncthreads();   // row-major dz2 complete; hid2 (s_x) consumed
#pragma unroll 1
    for (int mt = 0; mt < TR / 16; ++mt) {   // dhid1 = dz2 W2, masked by hid1 > 0 -> dz1 (s_x), db1
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        bf16x8 z[3];
        ld_pieces(s_z + (mt * 16 + l15) * LDB + 32 * kc + 8 * lg, IMG, z);
        acc[0] = mfma6(wA[0][kc], z, acc[0]);
        acc[1] = mfma6(wA[1][kc], z, acc[1]);
      }
      const int rr = mt * 16 + l15;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // hid1 > 0 <=> its leading piece is not zero (ReLU output; bf16 keeps the fp32 exponent range)
        const unsigned short *hb = reinterpret_cast<const unsigned short *>(s_h1T) + (32 * w + 16 * t + 4 * lg) * LDR + rr;
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) acc[t][uu] = (hb[uu * LDR] & 0x7fff) ? acc[t][uu] : 0.f;
        d1[t] += acc[t];
        st4(&s_x[rr * LDT + 32 * w + 16 * t + 4 * lg], acc[t]);
      }
    }
    __syncthreads();   // dz1 complete
    {
      // dP[h, i, j] = dz1[(i, j), :] . U[j, h, :] for the wave's two key columns: result row (key column, head), valid where
      // the row's key column is the pair's
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 a = ld4(&s_x[(16 * w + l15) * LDT + 16 * q + 4 * lg]);
        acc = MFMA16(uP[q][0], a[0], acc);
        accb = MFMA16(uP[q][1], a[1], accb);
        acc = MFMA16(uP[q][2], a[2], acc);
        accb = MFMA16(uP[q][3], a[3], accb);
      }
      acc += accb;
      if ((lg >> 1) == (l15 >> 3)) {
        float *o = dP + (((size_t)b * H + 4 * (lg & 1)) * K + i0 + (l15 & 7)) * K + j0 + 2 * w + (l15 >> 3);
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) o[(size_t)uu * K * K] = acc[uu];
      }
      // dU[j, h, c] += sum_i P[h, i, j] dz1[(i, j), c]: block-diagonal attention operand again, now on the row side
      const float q0 = s_p[(l15 & 7) * TR + 16 * w + (l15 >> 3) * TI + lg], q1 = s_p[(l15 & 7) * TR + 16 * w + (l15 >> 3) * TI + 4 + lg];
      const bool hi = l15 >> 3;
      const float pA[4] = {hi ? 0.f : q0, hi ? 0.f : q1, hi ? q0 : 0.f, hi ? q1 : 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float *zr = s_x + (16 * w + 4 * ks + lg) * LDT + l15;
#pragma unroll
        for (int n = 0; n < 8; ++n) du[n] = MFMA16(pA[ks], zr[16 * n], du[n]);
      }
    }
  }
  {   // this stretch's share of dU: du[n][uu] = dU[key column 2 w + (lg >> 1), head 4 (lg & 1) + uu, channel 16 n + l15].
      // Slot = position of this workgroup among those that share the key block; the one that ends it clears the unused slots.
    const long first = ((long)grp * NI + 1) * gridDim.x;
    const int slot = (int)(blockIdx.x - ((first + NU - 1) / NU - 1));
    const size_t slab = (size_t)B * K * H * C;
    float *duo = dU + (size_t)slot * slab + (((size_t)b * K + j0 + 2 * w + (lg >> 1)) * H + 4 * (lg & 1)) * C + l15;
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) duo[uu * C + 16 * n] = du[n][uu];
    if (ibe == NI)
      for (int z = slot + 1; z < Z; ++z)
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
          for (int uu = 0; uu < 4; ++uu) duo[(size_t)(z - slot) * slab + uu * C + 16 * n] = 0.f;
  }
  }
  // ---- results of the whole launch --------------------------------------------------------------------------------------
  float *po = part + (size_t)blockIdx.x * PART;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) po[(size_t)(32 * w + 16 * t + 4 * lg + uu) * C + 16 * n + l15] = gw[t][n][uu];
  float *p3 = po + C * C, *pb1 = p3 + NO * C, *pb2 = pb1 + C, *pb3 = pb2 + C;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) {
      if (4 * lg + uu < NO) p3[(4 * lg + uu) * C + 32 * w + 16 * t + l15] = g3[t][uu];
      if (l15 == 0) pb2[32 * w + 16 * t + 4 * lg + uu] = gb2[t][uu];
    }
  if (w == 0 && l15 == 0) st4(pb3 + 4 * lg, gb3);
  __syncthreads();
  float *s_sum = s_x;   // db1: the 16 row lanes' sums of channels 32 w + 16 t + 4 lg .. + 3, added in order
  st4(&s_sum[l15 * C + 32 * w + 4 * lg], d1[0]);
  st4(&s_sum[l15 * C + 32 * w + 16 + 4 * lg], d1[1]);
  __syncthreads();
  if (tid < C) {
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) a += s_sum[g * C + tid];
    pb1[tid] = a;
  }
}

constexpr size_t FWD_LDS = (size_t)3 * IMG * 2 + (size_t)(4 * TR * DPL + H * TR) * sizeof(float);
constexpr size_t BWD_LDS = (size_t)2 * 3 * IMGT * 2 + (size_t)(TR * LDT + H * TR + TR * DPL) * sizeof(float);

// Grid: one workgroup per CU the caller has not reserved for side-stream work (spacap_sa_reserve_cus), each with a contiguous
// range of tile units.  A grid sized to ALL CUs would leave its last workgroups waiting behind the sampling chain's (one
// scene per CU for the first half of the step) and run them as a second round: twice the time.
// CUs the head's persistent grids leave free IN ADDITION to spacap_sa_reserve_cus: a caller that runs the caption decoder on
// another stream beside the head (engine.Trainer: the decoder only needs the encoder's output) sets this to what the decoder's
// launches need to be resident at once (spacap_relation_fused_leave_cus).  Below ~56 the decoder's 64-workgroup launches queue
// behind the head's persistent workgroups and the two chains serialise again (measured: 7.25 against 6.97 ms per step).
static std::atomic<int> g_rel_leave{0};
inline int grid_size(int B, int K, int per_cu) {
  const long units = (long)B * (K / TJ) * (K / TI);
  // (with two workgroups per CU the dispatcher needs slack beyond the occupied CUs themselves: sa_mlp.hip, fwd_resident)
  const int extra = g_rel_leave.load(std::memory_order_relaxed);   // CUs left to a stream that runs beside the head (below)
  const long g = (long)per_cu * std::max(1, spacap::device_cus() - extra - spacap::sa_reserved_cus() * (per_cu > 1 ? 3 : 1));
  return (int)std::min(units, g);
}
// dU slots: the largest number of workgroups whose ranges meet one key block
inline int du_slots(int B, int K, int G) {
  const long NI = K / TI, NU = (long)B * (K / TJ) * NI;
  int z = 1;
  for (long g = 0; g < (long)B * (K / TJ); ++g) {
    const long wf = ((g * NI + 1) * G + NU - 1) / NU - 1, wl = (((g + 1) * NI) * G + NU - 1) / NU - 1;
    z = std::max(z, (int)(wl - wf + 1));
  }
  return z;
}

}  // namespace

extern "C" int spacap_relation_fused_leave_cus(int n) {
  if (n < 0 || n > spacap::device_cus() / 2) return SPACAP_E_INVALID;
  g_rel_leave.store(n, std::memory_order_relaxed);
  return SPACAP_OK;
}
extern "C" int spacap_relation_fused_supported(int H_, int K, int C_, int NO_) {
  return H_ == H && C_ == C && NO_ == NO && K >= 8 && K % 8 == 0;
}
extern "C" int spacap_relation_fused_nparts(int B, int K) { return B > 0 && K >= 8 ? grid_size(B, K, 1) : 0; }
extern "C" int spacap_relation_fused_zsplit(int B, int K, int nparts) { return B > 0 && K >= 8 && nparts > 0 ? du_slots(B, K, nparts) : 0; }
extern "C" int spacap_relation_fused_part_floats(void) { return PART; }

extern "C" int spacap_relation_fused_fwd_f32(const float *P, const float *U, const float *b1, const float *W2, const float *b2,
                                             const float *W3, const float *b3, int B, int K, float *hid2, float *pred,
                                             spacap_stream_t stream) {
  const char *what = "spacap_relation_fused_fwd_f32";
  SPACAP_REQUIRE(B >= 0 && K >= 8 && K % 8 == 0 && B <= 65535, "%s: (B=%d, K=%d) unsupported", what, B, K);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(P && U && b1 && W2 && b2 && W3 && b3 && hid2 && pred, "%s: null pointer", what);
  static unsigned long long lds_ok = 0;
  SPACAP_CHECK_HIP(spacap::allow_dynamic_lds(reinterpret_cast<const void *>(&rel_fused_fwd_kernel), (int)FWD_LDS, lds_ok), what);
  // (two workgroups per CU overlap each other's phases)
  hipLaunchKernelGGL(rel_fused_fwd_kernel, dim3(grid_size(B, K, 2)), dim3(256), FWD_LDS, spacap::as_stream(stream), P, U, b1, W2, b2, W3,
                     b3, B, K, hid2, pred);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_relation_fused_bwd_f32(const float *dpred, const float *hid2, const float *P, const float *U, const float *b1,
                                             const float *W2, const float *W3, int B, int K, int nparts, int zslots, float *dP,
                                             float *dU, float *part, spacap_stream_t stream) {
  const char *what = "spacap_relation_fused_bwd_f32";
  SPACAP_REQUIRE(B >= 0 && K >= 8 && K % 8 == 0 && B <= 65535, "%s: (B=%d, K=%d) unsupported", what, B, K);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(dpred && hid2 && P && U && b1 && W2 && W3 && dP && dU && part, "%s: null pointer", what);
  static unsigned long long lds_ok = 0;
  SPACAP_CHECK_HIP(spacap::allow_dynamic_lds(reinterpret_cast<const void *>(&rel_fused_bwd_kernel), (int)BWD_LDS, lds_ok), what);
  SPACAP_REQUIRE(nparts >= 1 && nparts <= B * (K / TJ) * (K / TI) && zslots >= du_slots(B, K, nparts),
                 "%s: (nparts=%d, zslots=%d) do not fit (B=%d, K=%d)", what, nparts, zslots, B, K);
  hipLaunchKernelGGL(rel_fused_bwd_kernel, dim3(nparts), dim3(256), BWD_LDS, spacap::as_stream(stream), dpred, hid2, P, U, b1, W2, W3, B, K,
                     zslots, dP, dU, part);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
