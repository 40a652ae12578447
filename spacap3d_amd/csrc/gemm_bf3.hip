// Tiled split-bf16 ("bf16 x 3") row products for WIDE layers, for gfx950 (MI355X):
//
//     out[r, n] = epi( sum_k A[r, k] W[n, k] )          A f32 [R, K], W f32 [N, K] (y = x W^T), out f32 [R, N]
//
// Callers: the relation head of the 512-wide / 32-head stress configuration (BASELINE.json config 5; reference
// models/transformer_captioner.py:319-326, 392-398: Linear(d_model, d_model) - ReLU - Linear(d_model, d_model) - ReLU -
// Linear(d_model, 9) on all B K K proposal pairs) -- 16 x 512 x 512 = 4.19 M pair rows x 512 x 512 = 2.2 TFLOP per product,
// which round 4 sent to rocBLAS as fp32 GEMMs (0.85 - 0.98 of the fp32-MFMA peak: matrix bound).  The fused relation kernels
// (csrc/relation_fused.hip) keep a 128 x 128 weight matrix in registers; a 512 x 512 one (1.5 MB as three bf16 images) has to
// stream, so this is a classic tiled GEMM instead.
//
// Arithmetic: every fp32 operand is x0 + x1 + x2 with three bf16 pieces (24 mantissa bits); a product is the six piece products
// whose weight is above 2^-24, each exact in the fp32 accumulator of v_mfma_f32_16x16x32_bf16: fp32-equivalent results at 6/16
// of the fp32-MFMA time (the arithmetic of csrc/relation_fused.hip and csrc/sa_bf3.inc).
// Organisation: 128 x 128 output tile per workgroup of 4 waves (2 x 2, 64 x 64 each), K in steps of 32.  The WEIGHTS are split
// once per call into three bf16 images in HBM ([3][N][K], 1.5 MB: L2 resident) and go to LDS as they are; the ACTIVATIONS are
// read as fp32, split in registers and written to LDS as three images [128][32 + 8] (80-byte rows: conflict-free 16-byte
// operand reads).  The next step's global loads are in flight while the current step is multiplied (register staging, one LDS
// buffer, two barriers per step); two workgroups per CU (61 KB of LDS, <= 256 registers) overlap each other's staging and matrix
// phases.  Workgroup -> tile mapping: the column tiles of one row tile run back to back ON THE SAME XCD (workgroups are dealt
// round robin to the 8 XCDs), so an activation tile is fetched from HBM once and re-read from that XCD's L2.
// Epilogues: bias + ReLU (forward); none (data gradient: the consumer masks).  Split over rows for the weight gradient
//     dW[n, k] = sum_r G[r, n] X[r, k]
// (contraction over the rows: both operands are staged TRANSPOSED, channel-major bf16 images, so that the matrix cores read 16
// bytes of consecutive rows per lane); per-slab partial results, added in slab order by the caller.
#include "common.hpp"

namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define MFMA_B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

constexpr int BM = 128, BN = 128, BK = 32, LDS_ROW = BK + 8;   // halfs per LDS row (80 bytes)
constexpr int IMG = BM * LDS_ROW;                               // one bf16 image of a 128 x 32 tile

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

__device__ __forceinline__ void split4(f32x4 v, bf16x4 &p0, bf16x4 &p1, bf16x4 &p2) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const __bf16 h = (__bf16)v[u];
    const float r = v[u] - (float)h;
    const __bf16 m = (__bf16)r;
    p0[u] = h, p1[u] = m, p2[u] = (__bf16)(r - (float)m);
  }
}

// W f32 [N][K] (trans == 0) or [K][N] (trans != 0: the image is of W^T) at row stride ldw -> Wp bf16 [3][N][K]
__global__ __launch_bounds__(256) void gemm_bf3_split_w_kernel(const float *__restrict__ W, long ldw, int N, int K, int trans,
                                                               __bf16 *__restrict__ Wp) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * K) return;
  const int n = (int)(e / K), k = (int)(e % K);
  const float v = trans ? W[(size_t)k * ldw + n] : W[(size_t)n * ldw + k];
  const __bf16 h = (__bf16)v;
  const float r = v - (float)h;
  const __bf16 m = (__bf16)r;
  Wp[e] = h, Wp[(size_t)N * K + e] = m, Wp[(size_t)2 * N * K + e] = (__bf16)(r - (float)m);
}

// logical tile id of a workgroup: consecutive ids on the same XCD (see the header comment)
__device__ __forceinline__ long xcd_local_id(long bid, long total) {
  const long per = total / 8;
  return bid < per * 8 ? (bid % 8) * per + bid / 8 : bid;
}

template <bool RELU>
__global__ __launch_bounds__(256, 2) void gemm_bf3_kernel(const float *__restrict__ A, long lda, const __bf16 *__restrict__ Wp,
                                                          const float *__restrict__ bias, long R, int K, int N,
                                                          float *__restrict__ out, long ldo) {
  __shared__ __attribute__((aligned(16))) __bf16 sA[3 * IMG];
  __shared__ __attribute__((aligned(16))) __bf16 sB[3 * IMG];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int wm = w >> 1, wn = w & 1;
  const int nbn = N / BN;
  const long L = xcd_local_id(blockIdx.x, gridDim.x);
  const long row0 = (L / nbn) * BM;
  const int col0 = (int)(L % nbn) * BN;
  // staging: two threads per tile row (A) / tile column (B), 16 k each
  const int srow = tid >> 1, sk = 16 * (tid & 1);
  const long garow = row0 + srow < R ? row0 + srow : R - 1;
  const float *ag = A + (size_t)garow * lda + sk;
  const __bf16 *bg = Wp + (size_t)(col0 + srow) * K + sk;
  const size_t wimg = (size_t)N * K;
  f32x4 ra[4];
  bf16x8 rb[3][2];
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = ld4(ag + kt * BK + 4 * i);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) rb[p][i] = *reinterpret_cast<const bf16x8 *>(bg + p * wimg + kt * BK + 8 * i);
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = K / BK;
  fetch(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();   // the previous step's operand reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x4 p0, p1, p2;
      split4(ra[i], p0, p1, p2);
      __bf16 *d = sA + srow * LDS_ROW + sk + 4 * i;
      *reinterpret_cast<bf16x4 *>(d) = p0;
      *reinterpret_cast<bf16x4 *>(d + IMG) = p1;
      *reinterpret_cast<bf16x4 *>(d + 2 * IMG) = p2;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<bf16x8 *>(sB + p * IMG + srow * LDS_ROW + sk + 8 * i) = rb[p][i];
    __syncthreads();
    if (kt + 1 < nk) fetch(kt + 1);
    bf16x8 a[4][3];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[mt][p] = *reinterpret_cast<const bf16x8 *>(sA + p * IMG + (64 * wm + 16 * mt + l15) * LDS_ROW + 8 * lg);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8 *>(sB + p * IMG + (64 * wn + 16 * nt + l15) * LDS_ROW + 8 * lg);
      // the six piece products, smallest first; four independent accumulators between two dependent instructions
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = MFMA_B(a[mt][PA[q]], b[PB[q]], acc[mt][nt]);
    }
  }
  // acc[mt][nt][u] = out[row0 + 64 wm + 16 mt + 4 lg + u][col0 + 64 wn + 16 nt + l15]: 64-byte runs per 16 lanes
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int c = col0 + 64 * wn + 16 * nt + l15;
    const float bv = bias ? bias[c] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long r = row0 + 64 * wm + 16 * mt + 4 * lg + u;
        float v = acc[mt][nt][u] + bv;
        if (RELU) v = fmaxf(v, 0.f);
        if (r < R) out[(size_t)r * ldo + c] = v;
      }
  }
}

// ---- weight gradient: part[slab][n][k] = sum over the slab's rows of G[r][n] X[r][k] ------------------------------------------
// 128 (n) x 128 (k) tile of dW per workgroup, rows in steps of 32 (= the contraction depth of one MFMA).  Both operands arrive
// row-major (a thread: 16 consecutive channels of one row) and are needed channel-major (a lane: 8 consecutive rows of one
// channel): the split pieces are written to LDS transposed, two bytes at a time, into [128 channels][32 + 8 rows] images.
constexpr int LDT_ROW = 32 + 8;   // halfs per channel row of a transposed image (80 bytes: conflict-free 16-byte reads)
__global__ __launch_bounds__(256, 2) void gemm_bf3_wgrad_kernel(const float *__restrict__ G, long ldg, const float *__restrict__ X, long ldx,
                                                                long R, int N, int K, float *__restrict__ part) {
  __shared__ __attribute__((aligned(16))) __bf16 sG[3 * 128 * LDT_ROW];
  __shared__ __attribute__((aligned(16))) __bf16 sX[3 * 128 * LDT_ROW];
  constexpr int TIMG = 128 * LDT_ROW;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int wm = w >> 1, wn = w & 1;
  const int n0 = blockIdx.y * 128, k0 = blockIdx.z * 128;
  const int nslab = gridDim.x;
  const long nsteps = (R + 31) / 32, per = (nsteps + nslab - 1) / nslab;
  const long sbeg = (long)blockIdx.x * per, send = sbeg + per < nsteps ? sbeg + per : nsteps;
  // staging: thread = (row tid / 8 of the 32-row step, channel pairs 2 (tid % 8) + 16 i, i < 8): 8-byte loads, 64 contiguous bytes
  // per 8 lanes; the transposed two-byte LDS writes of a wave then fall into 32 distinct words (channel pitch 20 words: the 8
  // lanes of a row are 40 words apart, the 8 rows of a wave share 4 words)
  const int srow = tid >> 3, sq = 2 * (tid & 7);
  float2 rg[8], rx[8];
  auto fetch = [&](long s) {
    const long r = s * 32 + srow;
    const bool ok = s < send && r < R;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      rg[i] = ok ? *reinterpret_cast<const float2 *>(G + (size_t)r * ldg + n0 + sq + 16 * i) : make_float2(0.f, 0.f);
      rx[i] = ok ? *reinterpret_cast<const float2 *>(X + (size_t)r * ldx + k0 + sq + 16 * i) : make_float2(0.f, 0.f);
    }
  };
  auto put = [&](__bf16 *img, const float2 *v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float x = u ? v[i].y : v[i].x;
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        __bf16 *d = img + (sq + 16 * i + u) * LDT_ROW + srow;
        d[0] = h, d[TIMG] = m, d[2 * TIMG] = (__bf16)(r1 - (float)m);
      }
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  fetch(sbeg);
  for (long s = sbeg; s < send; ++s) {
    __syncthreads();
    put(sG, rg);
    put(sX, rx);
    __syncthreads();
    fetch(s + 1);
    bf16x8 a[4][3];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[mt][p] = *reinterpret_cast<const bf16x8 *>(sG + p * TIMG + (64 * wm + 16 * mt + l15) * LDT_ROW + 8 * lg);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      bf16x8 b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8 *>(sX + p * TIMG + (64 * wn + 16 * nt + l15) * LDT_ROW + 8 * lg);
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = MFMA_B(a[mt][PA[q]], b[PB[q]], acc[mt][nt]);
    }
  }
  float *o = part + (size_t)blockIdx.x * N * K;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        o[(size_t)(n0 + 64 * wm + 16 * mt + 4 * lg + u) * K + k0 + 64 * wn + 16 * nt + l15] = acc[mt][nt][u];
}

// ---- tail of the relation head's backward at any width C (multiple of 4, C / 4 dividing 256):
//   dz2 = (dpred W3) * [hid2 > 0],  dW3 = dpred^T hid2,  db2 = sum dz2,  db3 = sum dpred
// (models/transformer_captioner.py:319-326: the last Linear(C, 9) and the ReLU in front of it).  One pass over hid2 -> dz2,
// 9 multiply-adds per element each way, HBM bound; per-workgroup partial sums [9 C | C | 16], added in order by the caller.
constexpr int WT_NO = 9, WT_TM = 64;
__global__ __launch_bounds__(256) void rel_wide_tail_bwd_kernel(const float *__restrict__ dpred, const float *__restrict__ W3,
                                                                const float *__restrict__ hid2, long R, int C, float *__restrict__ dz2,
                                                                float *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C4 = C / 4, RG = 256 / C4;           // row groups of the workgroup
  float *s_dp = smem;                            // [WT_TM][9]
  float *s_red = smem + WT_TM * WT_NO;           // [RG][C4][41]
  const int tid = threadIdx.x, c4 = tid % C4, r0 = tid / C4;
  float w3[WT_NO][4], aw[WT_NO][4], ab2[4] = {0.f, 0.f, 0.f, 0.f}, ab3[WT_NO];
#pragma unroll
  for (int o = 0; o < WT_NO; ++o) {
    ab3[o] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) w3[o][u] = W3[(size_t)o * C + c4 * 4 + u], aw[o][u] = 0.f;
  }
  const long ntiles = (R + WT_TM - 1) / WT_TM;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long row0 = t * WT_TM;
    __syncthreads();
    for (int i = tid; i < WT_TM * WT_NO; i += 256) s_dp[i] = (row0 * WT_NO + i < R * WT_NO) ? dpred[row0 * WT_NO + i] : 0.f;
    __syncthreads();
    for (int row = r0; row < WT_TM; row += RG) {
      if (row0 + row >= R) break;
      const f32x4 h = ld4(hid2 + (size_t)(row0 + row) * C + c4 * 4);
      float d[WT_NO];
#pragma unroll
      for (int o = 0; o < WT_NO; ++o) d[o] = s_dp[row * WT_NO + o];
      f32x4 dz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < WT_NO; ++o)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          dz[u] = __builtin_fmaf(d[o], w3[o][u], dz[u]);
          aw[o][u] = __builtin_fmaf(d[o], h[u], aw[o][u]);
        }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        dz[u] = h[u] > 0.f ? dz[u] : 0.f;
        ab2[u] += dz[u];
      }
      *reinterpret_cast<f32x4 *>(dz2 + (size_t)(row0 + row) * C + c4 * 4) = dz;
      if (c4 == 0) {
#pragma unroll
        for (int o = 0; o < WT_NO; ++o) ab3[o] += d[o];
      }
    }
  }
  __syncthreads();
  float *mine = &s_red[(size_t)(r0 * C4 + c4) * 41];
#pragma unroll
  for (int o = 0; o < WT_NO; ++o)
#pragma unroll
    for (int u = 0; u < 4; ++u) mine[o * 4 + u] = aw[o][u];
#pragma unroll
  for (int u = 0; u < 4; ++u) mine[36 + u] = ab2[u];
  if (c4 == 0) {
#pragma unroll
    for (int o = 0; o < WT_NO; ++o) s_dp[r0 * 16 + o] = ab3[o];   // (s_dp is free now)
  }
  __syncthreads();
  const size_t PW = (size_t)WT_NO * C + C + 16;
  float *o_part = part + (size_t)blockIdx.x * PW;
  for (int i = tid; i < C4 * 40; i += 256) {
    const int q = i / 40, e = i % 40;
    float a = 0.f;
    for (int g = 0; g < RG; ++g) a += s_red[(size_t)(g * C4 + q) * 41 + e];
    if (e < 36) o_part[(size_t)(e >> 2) * C + q * 4 + (e & 3)] = a;
    else o_part[(size_t)WT_NO * C + q * 4 + (e - 36)] = a;
  }
  if (tid < 16) {
    float a = 0.f;
    if (tid < WT_NO)
      for (int g = 0; g < RG; ++g) a += s_dp[g * 16 + tid];
    o_part[(size_t)WT_NO * C + C + tid] = a;
  }
}

bool wide_tail_shape(int C) { return C >= 16 && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0; }

// ---- first layer of the relation head at wide C, forward and backward, on the matrix cores -----------------------------------
//   hid1[b,i,j,:] = relu(b1 + sum_h P[b,h,i,j] U[b,j,h,:])        (models/transformer_captioner.py:392-397 + the first Linear of
//   :319-326, factored through U[b,j,h,:] = W1[:, 16 h : 16 h + 16] V[b,h,j,:])
// csrc/relation.hip does this on the vector pipe with U[b,j] in registers (4 channels per thread): at C = 512 / H = 32 that is
// 256 registers per thread, one wave per SIMD, and its backward ran 24 ms of the stress configuration's step.  Here one
// workgroup owns ONE key column (b, j) and walks the queries in tiles of 16, every product on v_mfma_f32_16x16x4_f32 (exact fp32
// products):      forward   hid1 tile [16 x C]   = Pt_j^T [16 x H] U_j [H x C]
//                 backward  dP tile [16 x H]     = dz1 [16 x C] U_j^T [C x H]      (each wave a quarter of C, summed through LDS)
//                           dU_j [H x C]        += Pt_j [H x 16] dz1 [16 x C]      (registers, written once: no partial slabs)
//                           db1                 += column sums of dz1              (per workgroup partial)
// with dz1 = dhid1 * [hid1 > 0].  The attention map is read TRANSPOSED, Pt[b,j,h,i] = P[b,h,i,j] (one tiled transposition
// per direction, 0.5 GB): a key column's [H x K] block is then contiguous instead of 16 384 four-byte reads at stride K.
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// out[((b K + j) H + h) K + i] = in[((b H + h) K + i) K + j]  (to_t != 0)   or the inverse (to_t == 0); 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void rel_wide_transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int K, int to_t) {
  __shared__ float tile[32][33];
  const int bh = blockIdx.z, b = bh / H, h = bh % H;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;   // x: the input's fastest index
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  // to_t: input (b,h,i,j): y = i, x = j; output (b,j,h,i).   else: input (b,j,h,i): blockIdx.z = b * H + h, y = j, x = i
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int y = y0 + r, x = x0 + tx;
    if (y < K && x < K)
      tile[r][tx] = to_t ? in[(((size_t)b * H + h) * K + y) * K + x] : in[(((size_t)b * K + y) * H + h) * K + x];
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int x = x0 + r, y = y0 + tx;   // output row index x (the input's fast index), fast index y
    if (y < K && x < K) {
      if (to_t) out[(((size_t)b * K + x) * H + h) * K + y] = tile[tx][r];
      else out[(((size_t)b * H + h) * K + x) * K + y] = tile[tx][r];
    }
  }
}

template <int HS, int NT>   // H = 4 HS heads, C = 64 NT channels (each of the 4 waves owns 16 NT of them)
__global__ __launch_bounds__(256) void rel_wide_l1_fwd_kernel(const float *__restrict__ Pt, const float *__restrict__ U,
                                                              const float *__restrict__ b1, int K, float *__restrict__ hid1) {
  constexpr int H = 4 * HS, C = 64 * NT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const long bj = blockIdx.x;                 // b * K + j
  const long b = bj / K;
  const int j = (int)(bj - b * K);
  const int cw = w * 16 * NT;
  const float *uj = U + (size_t)bj * H * C;
  float uB[NT][HS];
  f32x4 bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int ks = 0; ks < HS; ++ks) uB[nt][ks] = uj[(size_t)(4 * ks + lg) * C + cw + 16 * nt + l15];
    const float bv = b1[cw + 16 * nt + l15];
    bias[nt] = f32x4{bv, bv, bv, bv};
  }
  const float *pj = Pt + (size_t)bj * H * K;
  float a[HS], an[HS];
#pragma unroll
  for (int ks = 0; ks < HS; ++ks) an[ks] = l15 < K ? pj[(size_t)(4 * ks + lg) * K + l15] : 0.f;
  for (int i0 = 0; i0 < K; i0 += 16) {
#pragma unroll
    for (int ks = 0; ks < HS; ++ks) a[ks] = an[ks];
    if (i0 + 16 < K) {
#pragma unroll
      for (int ks = 0; ks < HS; ++ks) an[ks] = i0 + 16 + l15 < K ? pj[(size_t)(4 * ks + lg) * K + i0 + 16 + l15] : 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 acc = bias[nt];
#pragma unroll
      for (int ks = 0; ks < HS; ++ks) acc = MFMA16(a[ks], uB[nt][ks], acc);
      // acc[u] = hid1[query i0 + 4 lg + u][channel cw + 16 nt + l15]
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 4 * lg + u;
        if (i < K) hid1[(((size_t)b * K + i) * K + j) * C + cw + 16 * nt + l15] = fmaxf(acc[u], 0.f);
      }
    }
  }
}

template <int HS, int NT>
__global__ __launch_bounds__(256, 2) void rel_wide_l1_bwd_kernel(const float *__restrict__ dh1, const float *__restrict__ hid1,
                                                                 const float *__restrict__ Pt, const float *__restrict__ U, int K,
                                                                 float *__restrict__ dPt, float *__restrict__ dU, float *__restrict__ db_part) {
  constexpr int H = 4 * HS, C = 64 * NT, LDZ = C + 4, NH = (H + 15) / 16, KSW = C / 16;   // KSW: k steps of a wave's channel quarter
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_z = smem;                    // [16][LDZ]: dz1 of the tile
  float *s_dp = smem + 16 * LDZ;        // [4 waves][16 queries][16 NH + 1]
  constexpr int LDP = 16 * NH + 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const long bj = blockIdx.x;
  const long b = bj / K;
  const int j = (int)(bj - b * K);
  const int cw = w * 16 * NT;
  const float *uj = U + (size_t)bj * H * C;
  // dP: B[k = channel][n = head]: lane (n = l15, k = 4 ks + lg) of the wave's quarter
  float uP[KSW][NH];
#pragma unroll
  for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
    for (int nh = 0; nh < NH; ++nh) uP[ks][nh] = 16 * nh + l15 < H ? uj[(size_t)(16 * nh + l15) * C + cw + 4 * ks + lg] : 0.f;
  f32x4 du[NH][NT];
#pragma unroll
  for (int nh = 0; nh < NH; ++nh)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) du[nh][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging: thread = (row tid / (C / 4) + RG * pass, float4 column tid % (C / 4))
  constexpr int C4 = C / 4, RG = 256 / C4, NP = 16 / RG;
  const int c4 = tid % C4, r0 = tid / C4;
  f32x4 db = {0.f, 0.f, 0.f, 0.f};
  const float *pj = Pt + (size_t)bj * H * K;
  float *dpj = dPt + (size_t)bj * H * K;
  for (int i0 = 0; i0 < K; i0 += 16) {
    __syncthreads();   // the previous tile's readers of s_z / s_dp are done
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int row = r0 + RG * p, i = i0 + row;
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (i < K) {
        const size_t off = (((size_t)b * K + i) * K + j) * C + 4 * c4;
        g = ld4(dh1 + off);
        const f32x4 hv = ld4(hid1 + off);
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = hv[u] > 0.f ? g[u] : 0.f;
      }
      db += g;
      *reinterpret_cast<f32x4 *>(s_z + row * LDZ + 4 * c4) = g;
    }
    // the attention operand of dU: A[m = head][k = query]: lane (m = l15, k = 4 ks + lg)
    float pA[NH][4];
#pragma unroll
    for (int nh = 0; nh < NH; ++nh)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        pA[nh][ks] = (16 * nh + l15 < H && i0 + 4 * ks + lg < K) ? pj[(size_t)(16 * nh + l15) * K + i0 + 4 * ks + lg] : 0.f;
    __syncthreads();
    // dP partial of this wave's channel quarter: [16 queries x H]
    {
      f32x4 acc[NH];
#pragma unroll
      for (int nh = 0; nh < NH; ++nh) acc[nh] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) {
        const float a = s_z[l15 * LDZ + cw + 4 * ks + lg];
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) acc[nh] = MFMA16(a, uP[ks][nh], acc[nh]);
      }
      // acc[nh][u] = dP[query 4 lg + u][head 16 nh + l15]
#pragma unroll
      for (int nh = 0; nh < NH; ++nh)
#pragma unroll
        for (int u = 0; u < 4; ++u) s_dp[(w * 16 + 4 * lg + u) * LDP + 16 * nh + l15] = acc[nh][u];
    }
    // dU += Pt_tile dz1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bz = s_z[(4 * ks + lg) * LDZ + cw + 16 * nt + l15];
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) du[nh][nt] = MFMA16(pA[nh][ks], bz, du[nh][nt]);
      }
    }
    __syncthreads();   // the four partial dP tiles are complete
    for (int e = tid; e < 16 * H; e += 256) {
      const int h = e / 16, q = e % 16;      // consecutive threads: consecutive queries of one head (contiguous in dPt)
      const float v = (s_dp[(0 * 16 + q) * LDP + h] + s_dp[(1 * 16 + q) * LDP + h]) + (s_dp[(2 * 16 + q) * LDP + h] + s_dp[(3 * 16 + q) * LDP + h]);
      if (i0 + q < K) dpj[(size_t)h * K + i0 + q] = v;
    }
  }
  // du[nh][nt][u] = dU[head 16 nh + 4 lg + u][channel cw + 16 nt + l15]
  float *duj = dU + (size_t)bj * H * C;
#pragma unroll
  for (int nh = 0; nh < NH; ++nh)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (16 * nh + 4 * lg + u < H) duj[(size_t)(16 * nh + 4 * lg + u) * C + cw + 16 * nt + l15] = du[nh][nt][u];
  __syncthreads();
  *reinterpret_cast<f32x4 *>(s_z + r0 * LDZ + 4 * c4) = db;
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float sacc = 0.f;
#pragma unroll
    for (int r = 0; r < RG; ++r) sacc += s_z[r * LDZ + c];
    db_part[(size_t)bj * C + c] = sacc;
  }
}

bool wide_l1_shape(int H, int K, int C) { return (H == 8 || H == 16 || H == 32) && (C == 128 || C == 256 || C == 512) && K >= 1; }

bool bf3_shape(int K, int N) { return K >= 128 && N >= 128 && K % 128 == 0 && N % 128 == 0 && K <= 4096 && N <= 4096; }

}  // namespace

extern "C" int spacap_gemm_bf3_supported(int K, int N) { return bf3_shape(K, N) ? 1 : 0; }

/* Wp bf16 [3][N][K] (3 N K two-byte elements) = the three split pieces of W [N][K] (trans == 0; row stride ldw) or of the
   transpose of W [K][N] (trans != 0). */
extern "C" int spacap_gemm_bf3_split_w_f32(const float *W, long ldw, int N, int K, int trans, void *Wp, spacap_stream_t stream) {
  const char *what = "spacap_gemm_bf3_split_w_f32";
  SPACAP_REQUIRE(W && Wp && N >= 1 && K >= 1 && ldw >= (trans ? N : K), "%s: bad arguments", what);
  hipLaunchKernelGGL(gemm_bf3_split_w_kernel, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, spacap::as_stream(stream), W, ldw,
                     N, K, trans, static_cast<__bf16 *>(Wp));
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

/* out f32 [R][N] (row stride ldo) = A f32 [R][K] (row stride lda, 16-byte aligned rows) W^T (+ bias f32 [N], nullable), then ReLU
   when relu != 0; Wp from spacap_gemm_bf3_split_w_f32.  K, N multiples of 128; any R >= 0. */
extern "C" int spacap_gemm_bf3_f32(const float *A, long lda, const void *Wp, const float *bias, long R, int K, int N, int relu, float *out,
                                   long ldo, spacap_stream_t stream) {
  const char *what = "spacap_gemm_bf3_f32";
  SPACAP_REQUIRE(R >= 0 && bf3_shape(K, N) && lda >= K && ldo >= N && lda % 4 == 0, "%s: (R=%ld, K=%d, N=%d) unsupported", what, R, K, N);
  if (R == 0) return SPACAP_OK;
  SPACAP_REQUIRE(A && Wp && out && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(Wp) & 15) == 0,
                 "%s: null / unaligned pointer", what);
  const long tiles = ((R + BM - 1) / BM) * (N / BN);
  SPACAP_REQUIRE(tiles < 2147483647L, "%s: too many tiles", what);
  hipStream_t s = spacap::as_stream(stream);
  if (relu)
    hipLaunchKernelGGL(gemm_bf3_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, s, A, lda, static_cast<const __bf16 *>(Wp), bias, R, K, N, out, ldo);
  else
    hipLaunchKernelGGL(gemm_bf3_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, s, A, lda, static_cast<const __bf16 *>(Wp), bias, R, K, N, out, ldo);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

/* row slabs (= partial results) of spacap_gemm_bf3_wgrad_f32 for (R, N, K) */
extern "C" int spacap_gemm_bf3_wgrad_slabs(long R, int N, int K) {
  if (R < 1 || !bf3_shape(K, N)) return 0;
  const long steps = (R + 31) / 32, yz = (long)(N / 128) * (K / 128);
  long n = 1024 / yz, cap = (16L << 20) / ((long)N * K);   // <= 64 MB of partial results
  if (n > cap) n = cap;
  if (n > steps / 8) n = steps / 8;
  return (int)(n < 1 ? 1 : n);
}

/* part f32 [nslab][N][K]: per row slab, dW[n][k] = sum_r G[r][n] X[r][k] (G f32 [R][N] at stride ldg, X f32 [R][K] at stride
   ldx, 16-byte aligned rows); the caller adds the slabs in order.  N, K multiples of 128. */
extern "C" int spacap_gemm_bf3_wgrad_f32(const float *G, long ldg, const float *X, long ldx, long R, int N, int K, int nslab, float *part,
                                         spacap_stream_t stream) {
  const char *what = "spacap_gemm_bf3_wgrad_f32";
  SPACAP_REQUIRE(R >= 1 && bf3_shape(K, N) && nslab >= 1 && nslab <= 65535 && ldg >= N && ldx >= K && ldg % 4 == 0 && ldx % 4 == 0,
                 "%s: (R=%ld, N=%d, K=%d, nslab=%d) unsupported", what, R, N, K, nslab);
  SPACAP_REQUIRE(G && X && part && ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(X)) & 15) == 0, "%s: null / unaligned pointer", what);
  hipLaunchKernelGGL(gemm_bf3_wgrad_kernel, dim3(nslab, N / 128, K / 128), dim3(256), 0, spacap::as_stream(stream), G, ldg, X, ldx, R, N, K, part);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

/* ---- tail of the relation head's backward at any width (see rel_wide_tail_bwd_kernel) ---------------------------------------
   dpred f32 [R,9], W3 f32 [9,C], hid2 f32 [R,C] -> dz2 f32 [R,C], part f32 [nparts][9 C + C + 16] (dW3 | db2 | db3). */
extern "C" int spacap_rel_wide_tail_supported(int C) { return wide_tail_shape(C) ? 1 : 0; }
extern "C" int spacap_rel_wide_tail_nparts(long R) {
  const long tiles = (R + WT_TM - 1) / WT_TM;
  const long g = 4L * spacap::device_cus();
  return (int)(tiles < g ? (tiles < 1 ? 1 : tiles) : g);
}
extern "C" int spacap_rel_wide_tail_bwd_f32(const float *dpred, const float *W3, const float *hid2, long R, int C, int nparts, float *dz2,
                                            float *part, spacap_stream_t stream) {
  const char *what = "spacap_rel_wide_tail_bwd_f32";
  SPACAP_REQUIRE(dpred && W3 && hid2 && dz2 && part && R >= 1 && nparts >= 1 && wide_tail_shape(C), "%s: bad arguments (C=%d)", what, C);
  const size_t lds = sizeof(float) * ((size_t)WT_TM * WT_NO + (size_t)256 * 41);
  hipLaunchKernelGGL(rel_wide_tail_bwd_kernel, dim3(nparts), dim3(256), lds, spacap::as_stream(stream), dpred, W3, hid2, R, C, dz2, part);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

/* ---- first layer of the relation head at wide C on the matrix cores (see rel_wide_l1_*_kernel) ---------------------------------
   Pt f32 [B,K,H,K] = the attention map transposed (spacap_rel_wide_transpose_f32(P, ..., to_t = 1)), U f32 [B,K,H,C], b1 f32 [C]
   -> hid1 f32 [B,K,K,C].  Backward: dh1, hid1 f32 [B,K,K,C] -> dPt f32 [B,K,H,K] (transpose back with to_t = 0), dU f32 [B,K,H,C],
   db_part f32 [B K][C] (per key column; the caller adds the rows in order).  H in {8,16,32}, C in {128,256,512}. */
extern "C" int spacap_rel_wide_l1_supported(int H, int K, int C) { return wide_l1_shape(H, K, C) ? 1 : 0; }
extern "C" int spacap_rel_wide_transpose_f32(const float *in, float *out, int B, int H, int K, int to_t, spacap_stream_t stream) {
  const char *what = "spacap_rel_wide_transpose_f32";
  SPACAP_REQUIRE(in && out && B >= 1 && H >= 1 && K >= 1 && (long)B * H <= 65535, "%s: bad arguments", what);
  const unsigned t = (unsigned)((K + 31) / 32);
  hipLaunchKernelGGL(rel_wide_transpose_kernel, dim3(t, t, (unsigned)(B * H)), dim3(256), 0, spacap::as_stream(stream), in, out, H, K, to_t);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
#define WIDE_L1_DISPATCH(CALL)                                                          \
  if (H == 8 && C == 128) { CALL(2, 2) } else if (H == 8 && C == 256) { CALL(2, 4) } else if (H == 8 && C == 512) { CALL(2, 8) }   \
  else if (H == 16 && C == 128) { CALL(4, 2) } else if (H == 16 && C == 256) { CALL(4, 4) } else if (H == 16 && C == 512) { CALL(4, 8) } \
  else if (H == 32 && C == 128) { CALL(8, 2) } else if (H == 32 && C == 256) { CALL(8, 4) } else { CALL(8, 8) }
extern "C" int spacap_rel_wide_l1_fwd_f32(const float *Pt, const float *U, const float *b1, int B, int H, int K, int C, float *hid1,
                                          spacap_stream_t stream) {
  const char *what = "spacap_rel_wide_l1_fwd_f32";
  SPACAP_REQUIRE(B >= 0 && wide_l1_shape(H, K, C), "%s: unsupported shape H=%d K=%d C=%d", what, H, K, C);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(Pt && U && b1 && hid1 && (long)B * K < 2147483647L, "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
#define CALL(HS, NT) hipLaunchKernelGGL((rel_wide_l1_fwd_kernel<HS, NT>), dim3((unsigned)((long)B * K)), dim3(256), 0, s, Pt, U, b1, K, hid1);
  WIDE_L1_DISPATCH(CALL)
#undef CALL
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
extern "C" int spacap_rel_wide_l1_bwd_f32(const float *dh1, const float *hid1, const float *Pt, const float *U, int B, int H, int K, int C,
                                          float *dPt, float *dU, float *db_part, spacap_stream_t stream) {
  const char *what = "spacap_rel_wide_l1_bwd_f32";
  SPACAP_REQUIRE(B >= 0 && wide_l1_shape(H, K, C), "%s: unsupported shape H=%d K=%d C=%d", what, H, K, C);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(dh1 && hid1 && Pt && U && dPt && dU && db_part && (long)B * K < 2147483647L, "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
  const int NH = (H + 15) / 16;
  const size_t lds = sizeof(float) * ((size_t)16 * (C + 4) + (size_t)4 * 16 * (16 * NH + 1));
#define CALL(HS, NT) hipLaunchKernelGGL((rel_wide_l1_bwd_kernel<HS, NT>), dim3((unsigned)((long)B * K)), dim3(256), lds, s, dh1, hid1, Pt, U, K, dPt, dU, db_part);
  WIDE_L1_DISPATCH(CALL)
#undef CALL
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
