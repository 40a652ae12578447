// Dense row products of any shape on point-major (row-major) operands, for gfx950 (MI355X).
//
//     out[r, n] = sum_k A[r, k] Wop[k, n] (+ bias[n]),     Wop[k, n] = trans_w ? W[n, k] : W[k, n]
//
// Callers (all were rocBLAS GEMMs inside the training step until round 4):
//   * the first layer of a set-abstraction module's shared MLP commuted with the grouping gather: Y = F W1[:, 3:]^T over the
//     SOURCE points and its data gradient dF = dY W1[:, 3:] (lib/pointnet2/pointnet2_modules.py:241-259 -> QueryAndGroup +
//     the first Conv2d of SharedMLP; W1[:, 3:] is a column slice: row stride 3 + Cf, rows not 16-byte aligned);
//   * the vocabulary projection of the caption head, forward / data gradient (models/transformer_captioner.py:93-100;
//     3 001 words: neither K nor N a multiple of anything);
//   * the relation head's per-head value projection U (models/transformer_captioner.py:319-326, first Linear) and the token
//     projection of the d_model = 512 stress configuration.
// Organisation (the row-panel kernel of the Transformer projections, csrc/sa_mlp.hip: linear_rows_kernel, made shape-agnostic):
// one (16 MT rows) x 64 column tile per workgroup, each wave 16 columns; K in chunks of 128: the activations of a chunk in LDS,
// that chunk's weights in registers (through LDS for the transposed case); v_mfma_f32_16x16x4_f32 (exact fp32 products).  Rows,
// columns and the K tail are guarded (zero fill), every operand has its own row stride, vector loads are used where the
// addresses allow and scalar ones elsewhere.  Two-level rows: row r = (r / rows_per_group, r % rows_per_group) lives at
// group * group_stride + (within + skip) * ld -- the caption head reads positions 1.. of every sequence without a slice copy.
// Optional split over K (gridDim.z slices, partial results [z][R][ldo] summed by the caller in order).
#include "common.hpp"

namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

struct RowsArgs {
  const float *a;
  const float *W;
  const float *bias;
  float *out;
  long R, lda, ldw, ldo;
  int K, CO;
  long a_grp, a_gstride, a_skip;   // rows per group of A (0: plain rows), elements between groups, leading rows skipped per group
  long o_grp, o_gstride, o_skip;   // the same for out; with o_zero the skipped leading rows of every group are written as zeros
  int o_zero;
  int kchunks_per_slice;           // split over K: 128-wide chunks per gridDim.z slice (all of them when z is a batch index)
  long slice_stride;               // elements between the slices' results
  long a_zstride, w_zstride;       // batch mode (split_k == 0): elements between the z-th operands
  int split_k;
  int vec_a, vec_w, vec_o;         // 16-byte accesses allowed for A / W / out (alignment of base, stride and offsets)
};

__device__ __forceinline__ size_t row_off(long r, long grp, long gstride, long skip, long ld) {
  if (grp <= 0) return (size_t)r * ld;
  long g, t;
  if (((unsigned long long)r | (unsigned long long)grp) >> 32) {
    g = r / grp, t = r - g * grp;
  } else {   // (a 64-bit division is ~200 instructions on this chip and sat on every element load of the small weight gradient: 35 of its 45 us)
    const unsigned rr = (unsigned)r, gg = (unsigned)grp, q = rr / gg;
    g = q, t = rr - q * gg;
  }
  return (size_t)g * gstride + (size_t)(t + skip) * ld;
}

template <bool TRANS_W, int MT>
__global__ __launch_bounds__(256) void dense_rows_kernel(const RowsArgs a) {
  constexpr int KC = 128, LD = KC + 4, KS = KC / 4, TMR = 16 * MT, C4 = KC / 4, RSTEP = 256 / C4, NV = TMR * C4 / 256;
  __shared__ __attribute__((aligned(16))) float s_a[TMR * LD];
  __shared__ __attribute__((aligned(16))) float s_w[TRANS_W ? 64 * LD : 4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int cbb = blockIdx.y * 64, cb = cbb + w * 16;
  const long row0 = (long)blockIdx.x * TMR;
  const int c4 = tid % C4, r0 = tid / C4;
  const int K = a.K, CO = a.CO;
  const int nchunks = (K + KC - 1) / KC;
  const int cbeg = a.split_k ? blockIdx.z * a.kchunks_per_slice : 0, cend = a.split_k ? min(nchunks, cbeg + a.kchunks_per_slice) : nchunks;
  const float *Ab = a.a + (size_t)blockIdx.z * a.a_zstride, *Wb = a.W + (size_t)blockIdx.z * a.w_zstride;
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = cbeg; c < cend; ++c) {
    const int kc = c * KC;
    if (c > cbeg) __syncthreads();
    const int k0 = kc + c4 * 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = r0 + i * RSTEP;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row0 + row < a.R && k0 < K) {
        const float *p = Ab + row_off(row0 + row, a.a_grp, a.a_gstride, a.a_skip, a.lda) + k0;
        if (a.vec_a && k0 + 3 < K) {
          v = ld4(p);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = k0 + u < K ? p[u] : 0.f;
        }
      }
      st4(&s_a[row * LD + c4 * 4], v);
    }
    float wf[KS];
    if (TRANS_W) {   // the 64 x 128 weight panel W[n][k] through LDS
      if (a.vec_w) {
#pragma unroll
        for (int i = 0; i < 64 * C4 / 256; ++i) {
          const int n = r0 + i * RSTEP;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (cbb + n < CO && k0 < K) {
            const float *p = Wb + (size_t)(cbb + n) * a.ldw + k0;
            if (k0 + 3 < K) {
              v = ld4(p);
            } else {
#pragma unroll
              for (int u = 0; u < 4; ++u) v[u] = k0 + u < K ? p[u] : 0.f;
            }
          }
          st4(&s_w[n * LD + c4 * 4], v);
        }
      } else {
        // rows that are not 16-byte aligned (a column slice of a wider matrix): 4-byte loads with consecutive lanes on
        // consecutive k -- two rows of 128 per wave instruction, every line fetched once
        float t[64 * KC / 256];
#pragma unroll
        for (int i = 0; i < 64 * KC / 256; ++i) {
          const int n = (tid >> 7) + 2 * i, k = kc + (tid & 127);
          t[i] = (cbb + n < CO && k < K) ? Wb[(size_t)(cbb + n) * a.ldw + k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 64 * KC / 256; ++i) s_w[((tid >> 7) + 2 * i) * LD + (tid & 127)] = t[i];
      }
    } else {         // W[k][n]: lane (l15, lg) takes column cb + l15 of the rows kc + 4 ks + lg
      const bool cok = cb + l15 < CO;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = kc + ks * 4 + lg;
        wf[ks] = (cok && k < K) ? Wb[(size_t)k * a.ldw + cb + l15] : 0.f;
      }
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
  // acc[mt][u] = out[row0 + 16 mt + l15][cb + 4 lg + u]
  const int n0 = cb + 4 * lg;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (a.bias && (blockIdx.z == 0 || !a.split_k)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) bv[u] = n0 + u < CO ? a.bias[n0 + u] : 0.f;
  }
  float *outz = a.out + (size_t)blockIdx.z * a.slice_stride;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long row = row0 + mt * 16 + l15;
    if (row < a.R && n0 < CO) {
      float *o = outz + row_off(row, a.o_grp, a.o_gstride, a.o_skip, a.ldo) + n0;
      const f32x4 v = acc[mt] + bv;
      if (a.vec_o && n0 + 3 < CO) {
        st4(o, v);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (n0 + u < CO) o[u] = v[u];
      }
      if (a.o_zero && a.o_grp > 0 && row % a.o_grp == 0) {   // the skipped leading rows of this row's group
        for (long s = 0; s < a.o_skip; ++s) {
          float *z = outz + (size_t)(row / a.o_grp) * a.o_gstride + (size_t)s * a.ldo + n0;
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (n0 + u < CO) z[u] = 0.f;
        }
      }
    }
  }
}

// out[e] = sum_z parts[z][e], e < n (the K slices in order)
__global__ __launch_bounds__(256) void dense_sum_slices_kernel(const float *__restrict__ parts, int S, long n, long stride,
                                                               float *__restrict__ out) {
  const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= n) return;
  if (e + 3 < n) {
    f32x4 v = ld4(parts + e);
    for (int z = 1; z < S; ++z) v += ld4(parts + (size_t)z * stride + e);
    st4(out + e, v);
  } else {
    for (long i = e; i < n; ++i) {
      float v = parts[i];
      for (int z = 1; z < S; ++z) v += parts[(size_t)z * stride + i];
      out[i] = v;
    }
  }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

/* K slices spacap_dense_rows_f32 should be called with for (R, K, CO): 1 unless the tile grid alone leaves most of the chip
   idle AND the reduction is long (the vocabulary projection's data gradient: 248 rows, K = 3 001). */
extern "C" int spacap_dense_rows_slices(long R, int K, int CO) {
  const long tiles = ((R + 31) / 32) * ((CO + 63) / 64);
  const int chunks = (K + 127) / 128;
  if (tiles >= 128 || chunks < 4) return 1;
  long s = 256 / (tiles < 1 ? 1 : tiles);
  if (s > chunks) s = chunks;
  if (s > 16) s = 16;
  return (int)(s < 1 ? 1 : s);
}

/* out[r, n] = sum_k A[r, k] Wop[k, n] (+ bias[n]); A f32 rows of K floats at stride lda, W f32 [CO, K] rows at stride ldw
   (trans_w) or [K, CO] rows at stride ldw, out rows of CO floats at stride ldo; any R >= 0, K >= 1, CO >= 1.
   Two-level rows (a_grp / o_grp > 0): row r of A lives at (r / a_grp) * a_gstride + (r % a_grp + a_skip) * lda, likewise for out;
   o_zero: the o_skip leading rows of every output group are written as zeros.  a_grp = o_grp = 0: plain rows.
   slices > 1, batch == 0: out receives `slices` partial results over K, `slice_stride` floats apart (bias in slice 0); sum them
   with spacap_dense_sum_slices_f32.  batch != 0: `slices` independent products, operand z at a + z a_zstride, W + z w_zstride,
   out + z slice_stride (the relation head's per-head value projection: 8 heads, 16-wide windows of one weight matrix). */
extern "C" int spacap_dense_rows_f32(const float *a, long lda, long a_grp, long a_gstride, long a_skip, const float *W, long ldw,
                                     int trans_w, const float *bias, long R, int K, int CO, float *out, long ldo, long o_grp,
                                     long o_gstride, long o_skip, int o_zero, int slices, long slice_stride, int batch,
                                     long a_zstride, long w_zstride, spacap_stream_t stream) {
  const char *what = "spacap_dense_rows_f32";
  SPACAP_REQUIRE(R >= 0 && K >= 1 && CO >= 1 && lda >= K && ldo >= CO && ldw >= (trans_w ? K : CO) && slices >= 1 && slices <= 64,
                 "%s: (R=%ld, K=%d, CO=%d, lda=%ld, ldw=%ld, ldo=%ld, slices=%d) unsupported", what, R, K, CO, lda, ldw, ldo, slices);
  if (R == 0) return SPACAP_OK;
  SPACAP_REQUIRE(a && W && out, "%s: null pointer", what);
  SPACAP_REQUIRE(a_grp >= 0 && o_grp >= 0 && a_skip >= 0 && o_skip >= 0 && (slices == 1 || slice_stride > 0), "%s: bad row groups", what);
  RowsArgs g;
  g.a = a, g.W = W, g.bias = bias, g.out = out, g.R = R, g.lda = lda, g.ldw = ldw, g.ldo = ldo, g.K = K, g.CO = CO;
  g.a_grp = a_grp, g.a_gstride = a_gstride, g.a_skip = a_skip, g.o_grp = o_grp, g.o_gstride = o_gstride, g.o_skip = o_skip;
  g.o_zero = o_zero;
  const int chunks = (K + 127) / 128;
  g.kchunks_per_slice = (chunks + slices - 1) / slices;
  g.slice_stride = slice_stride;
  g.split_k = batch ? 0 : 1;
  g.a_zstride = batch ? a_zstride : 0, g.w_zstride = batch ? w_zstride : 0;
  g.vec_a = aligned16(a) && lda % 4 == 0 && (a_grp == 0 || a_gstride % 4 == 0) && g.a_zstride % 4 == 0;
  g.vec_w = aligned16(W) && ldw % 4 == 0 && g.w_zstride % 4 == 0;
  g.vec_o = aligned16(out) && ldo % 4 == 0 && (o_grp == 0 || o_gstride % 4 == 0) && (slices == 1 || slice_stride % 4 == 0);
  hipStream_t s = spacap::as_stream(stream);
  const bool small = R <= 1024;
  const long tiles = small ? (R + 31) / 32 : (R + 63) / 64;
  SPACAP_REQUIRE(tiles <= 2147483647L, "%s: too many rows", what);
  const dim3 grid((unsigned)tiles, (unsigned)((CO + 63) / 64), (unsigned)slices);
  if (trans_w) {
    if (small) hipLaunchKernelGGL((dense_rows_kernel<true, 2>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((dense_rows_kernel<true, 4>), grid, dim3(256), 0, s, g);
  } else {
    if (small) hipLaunchKernelGGL((dense_rows_kernel<false, 2>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((dense_rows_kernel<false, 4>), grid, dim3(256), 0, s, g);
  }
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

/* out[e] = sum over the S slices of parts[z * stride + e], e < n, slices added in order. */
extern "C" int spacap_dense_sum_slices_f32(const float *parts, int S, long n, long stride, float *out, spacap_stream_t stream) {
  const char *what = "spacap_dense_sum_slices_f32";
  SPACAP_REQUIRE(parts && out && S >= 1 && n >= 0 && stride >= n && stride % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(parts) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                 "%s: bad arguments", what);
  if (n == 0) return SPACAP_OK;
  hipLaunchKernelGGL(dense_sum_slices_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, spacap::as_stream(stream), parts, S, n,
                     stride, out);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ---- weight + bias gradient of a row product with FEW rows and a wide output (the vocabulary projection: 248 rows, 3 001 x 128):
//   dW[m][n] = sum_r G[r][m] X[r][n],   db[m] = sum_r G[r][m]
// One 64 (m) x 128 (n) tile per workgroup, thread = 8 m x 4 n, the rows added in ascending order (no partial results).  X rows may
// be two-level (positions 1.. of every sequence).  Plain fp32 FMAs: 0.2 GFLOP, a latency-bound launch either way.
namespace {
template <int MT>   // m per thread: a workgroup covers 8 MT rows of dW x 128 columns
__global__ __launch_bounds__(256) void dense_wgrad_small_kernel(const float *__restrict__ G, long ldg, const float *__restrict__ X,
                                                                long ldx, long x_grp, long x_gstride, long x_skip, long R, int M, int N,
                                                                float *__restrict__ dW, float *__restrict__ db) {
  // 64 rows of both operands at a time through LDS (coalesced 4-byte loads: the rows of G are not 16-byte aligned when M is odd),
  // the next chunk's loads in flight while this one is multiplied (a chunk costs one round trip whatever its size: a few hundred
  // rows are four of them).  The launch is VALU work on few workgroups: with 8 m per
  // thread the vocabulary projection (M = 3 001) ran on 47 CUs for 54 us; 2 m per thread puts it on 188.
  constexpr int RB = 64, MB = 8 * MT, NG = (RB * MB + 255) / 256, NX = RB * 128 / 256;
  __shared__ __attribute__((aligned(16))) float s_g[RB][MB + 4];
  __shared__ __attribute__((aligned(16))) float s_x[RB][128 + 4];
  const int tid = threadIdx.x, n4 = tid & 31, mq = tid >> 5;
  const int mb = blockIdx.x * MB, nb = blockIdx.y * 128;
  const int m0 = mb + mq * MT, n0 = nb + n4 * 4;
  float acc[MT][4], sb[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    sb[i] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[i][u] = 0.f;
  }
  float tg[NG], tx[NX];
  __shared__ size_t s_off[2][RB];   // element offset of the chunk's rows of X: one division per ROW (two-level rows), not per element
  auto offsets = [&](long r0, int buf) {
    if (tid < RB) s_off[buf][tid] = row_off(r0 + tid < R ? r0 + tid : R - 1, x_grp, x_gstride, x_skip, ldx);
  };
  auto request = [&](long r0, int buf) {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int e = tid + 256 * i, j = e / MB, m = e % MB;
      tg[i] = (e < RB * MB && r0 + j < R && mb + m < M) ? G[(size_t)(r0 + j) * ldg + mb + m] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i, j = e >> 7, n = e & 127;
      tx[i] = (r0 + j < R && nb + n < N) ? X[s_off[buf][j] + nb + n] : 0.f;
    }
  };
  int buf = 0;
  if (R > 0) {
    offsets(0, 0);
    __syncthreads();
    request(0, 0);
  }
  for (long r0 = 0; r0 < R; r0 += RB, buf ^= 1) {
    if (r0 + RB < R) offsets(r0 + RB, buf ^ 1);
    __syncthreads();   // (the previous chunk's readers are done; the next chunk's row offsets are written)
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int e = tid + 256 * i;
      if (e < RB * MB) s_g[e / MB][e % MB] = tg[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) s_x[(tid + 256 * i) >> 7][(tid + 256 * i) & 127] = tx[i];
    __syncthreads();
    if (r0 + RB < R) request(r0 + RB, buf ^ 1);
    // eight rows' operands out of LDS first, then their products (issued one row at a time the compiler waits out an LDS
    // round trip per row: 11 us of this kernel's 25 at 248 rows); same order of additions
    for (int j0 = 0; j0 < RB; j0 += 8) {
      f32x4 x[8];
      float gv[8][MT];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x[j] = ld4(&s_x[j0 + j][n4 * 4]);
#pragma unroll
        for (int i = 0; i < MT; ++i) gv[j][i] = s_g[j0 + j][mq * MT + i];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          sb[i] += gv[j][i];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[i][u] = __builtin_fmaf(gv[j][i], x[j][u], acc[i][u]);
        }
    }
  }
  const bool nok = n0 < N;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (m0 + i < M) {
      if (nok) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (n0 + u < N) dW[(size_t)(m0 + i) * N + n0 + u] = acc[i][u];
      }
      if (db && blockIdx.y == 0 && n4 == 0) db[m0 + i] = sb[i];
    }
  }
}
}  // namespace

/* dW f32 [M,N] = G^T X, db f32 [M] (nullable) = column sums of G; G f32 rows of M floats at stride ldg, X rows of N floats at
   stride ldx (two-level rows as in spacap_dense_rows_f32), R rows added in ascending order.  For R up to a few thousand. */
extern "C" int spacap_dense_wgrad_small_f32(const float *G, long ldg, const float *X, long ldx, long x_grp, long x_gstride, long x_skip,
                                            long R, int M, int N, float *dW, float *db, spacap_stream_t stream) {
  const char *what = "spacap_dense_wgrad_small_f32";
  SPACAP_REQUIRE(R >= 0 && M >= 1 && N >= 1 && ldg >= M && ldx >= N && x_grp >= 0 && x_skip >= 0, "%s: bad sizes", what);
  SPACAP_REQUIRE(G && X && dW, "%s: null pointer", what);
  const unsigned gy = (unsigned)((N + 127) / 128);
  if ((long)((M + 63) / 64) * gy >= 200)   // enough workgroups already: 8 m per thread (fewer LDS reads per product)
    hipLaunchKernelGGL(dense_wgrad_small_kernel<8>, dim3((unsigned)((M + 63) / 64), gy), dim3(256), 0, spacap::as_stream(stream), G, ldg, X,
                       ldx, x_grp, x_gstride, x_skip, R, M, N, dW, db);
  else
    hipLaunchKernelGGL(dense_wgrad_small_kernel<2>, dim3((unsigned)((M + 15) / 16), gy), dim3(256), 0, spacap::as_stream(stream), G, ldg, X,
                       ldx, x_grp, x_gstride, x_skip, R, M, N, dW, db);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ---- block-diagonal weight gradient: dW[o][z N + n] = sum_r G[r][z M + o] X[r][z N + n], z < Z -------------------------------
// (the relation head's first Linear applied per head: G = dU [R][Z M], X = the value vectors [R][Z N], models/
// transformer_captioner.py:319-326.)  One (row slab, z) per workgroup, thread = one o and 8 columns n; per-slab partial results
// in dW's own layout [slab][M][Z N], to be added in slab order.
namespace {
__global__ __launch_bounds__(256) void dense_wgrad_blocks_kernel(const float *__restrict__ G, long ldg, const float *__restrict__ X,
                                                                 long ldx, long R, int Z, int M, int N, float *__restrict__ part) {
  const int s = blockIdx.x, z = blockIdx.y, nslab = gridDim.x;
  const long per = (R + nslab - 1) / nslab, rbeg = (long)s * per, rend = rbeg + per < R ? rbeg + per : R;
  float *po = part + (size_t)s * M * Z * N;
  for (int o = threadIdx.x % 128; o < M; o += 128) {
    for (int n0 = 8 * (threadIdx.x / 128); n0 < N; n0 += 16) {
      float acc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = 0.f;
      for (long rr = rbeg; rr < rend; rr += 8) {   // eight rows' operands in flight together
        float g[8], xv[8][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long r = rr + j < rend ? rr + j : rend - 1;
          g[j] = rr + j < rend ? G[(size_t)r * ldg + (size_t)z * M + o] : 0.f;
          const float *x = X + (size_t)r * ldx + (size_t)z * N + n0;
#pragma unroll
          for (int u = 0; u < 8; ++u) xv[j][u] = n0 + u < N ? x[u] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int u = 0; u < 8; ++u) acc[u] = __builtin_fmaf(g[j], xv[j][u], acc[u]);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (n0 + u < N) po[(size_t)o * Z * N + (size_t)z * N + n0 + u] = acc[u];
    }
  }
}
}  // namespace

/* row slabs (= partial results) spacap_dense_wgrad_blocks_f32 should be called with for R rows */
extern "C" int spacap_dense_wgrad_blocks_slabs(long R) {
  long n = (R + 127) / 128;
  if (n > 64) n = 64;
  return (int)(n < 1 ? 1 : n);
}

/* part f32 [nslab][M][Z N]: per row slab, dW[o][z N + n] = sum_r G[r][z M + o] X[r][z N + n] (block-diagonal product of G^T X);
   G rows of Z M floats at stride ldg, X rows of Z N floats at stride ldx.  The caller adds the slabs in order. */
extern "C" int spacap_dense_wgrad_blocks_f32(const float *G, long ldg, const float *X, long ldx, long R, int Z, int M, int N, int nslab,
                                             float *part, spacap_stream_t stream) {
  const char *what = "spacap_dense_wgrad_blocks_f32";
  SPACAP_REQUIRE(R >= 1 && Z >= 1 && Z <= 65535 && M >= 1 && N >= 1 && nslab >= 1 && ldg >= (long)Z * M && ldx >= (long)Z * N,
                 "%s: bad sizes", what);
  SPACAP_REQUIRE(G && X && part, "%s: null pointer", what);
  hipLaunchKernelGGL(dense_wgrad_blocks_kernel, dim3(nslab, Z), dim3(256), 0, spacap::as_stream(stream), G, ldg, X, ldx, R, Z, M, N, part);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ---- weight gradient of a row product with MANY rows and a narrow / odd-width result:  dW[m][n] = sum_r G[r][m] X[r][n] -------
// (the feature columns of a set-abstraction module's first layer when the source points carry 7 or 132 feature channels,
// BASELINE configs 3 and 4: G = dY [B Np][64], X = the point-major features [B Np][Cf], 320 000 rows into a 64 x Cf matrix;
// lib/pointnet2/pytorch_utils.py:11-36 -- the first Conv2d of SharedMLP.  Round 4 sent this shape to torch.bmm.)
// One row slab per workgroup (grid.x), 64 rows of dW per grid.y block, 16 NT columns per grid.z block: 32-row tiles of both
// operands through LDS with the next tile's loads in flight (coalesced 4-byte loads: rows of 7 or 132 floats are not 16-byte
// aligned), v_mfma_f32_16x16x4_f32 (exact fp32 products), wave w owns rows 16 w .. 16 w + 15 of the block.  Per-slab partial
// results [slab][M][N], added in slab order by the caller (spacap_sum_slabs_f32 / spacap_sa_dw1_assemble_f32).
namespace {
template <int NT>
__global__ __launch_bounds__(256) void dense_wgrad_tall_kernel(const float *__restrict__ G, long ldg, const float *__restrict__ X, long ldx,
                                                               long R, int M, int N, float *__restrict__ part) {
  constexpr int TR = 32, MB = 64, NB = 16 * NT, LG = MB + 4, LX = NB + 4;
  constexpr int NG = TR * MB / 256, NX = (TR * NB + 255) / 256;
  __shared__ __attribute__((aligned(16))) float s_g[TR * LG];
  __shared__ __attribute__((aligned(16))) float s_x[TR * LX];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int nslab = gridDim.x, m0 = blockIdx.y * MB, n0 = blockIdx.z * NB;
  const long ntiles = (R + TR - 1) / TR;
  const long per = (ntiles + nslab - 1) / nslab, tbeg = (long)blockIdx.x * per, tend = tbeg + per < ntiles ? tbeg + per : ntiles;
  f32x4 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float tg[NG], tx[NX];
  auto fetch = [&](long t) {
    const long r0 = t * TR;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int e = tid + 256 * i, j = e / MB, m = e % MB;
      tg[i] = (t < tend && r0 + j < R && m0 + m < M) ? G[(size_t)(r0 + j) * ldg + m0 + m] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i, j = e / NB, n = e % NB;
      tx[i] = (e < TR * NB && t < tend && r0 + j < R && n0 + n < N) ? X[(size_t)(r0 + j) * ldx + n0 + n] : 0.f;
    }
  };
  fetch(tbeg);
  for (long t = tbeg; t < tend; ++t) {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int e = tid + 256 * i;
      s_g[(e / MB) * LG + e % MB] = tg[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i;
      if (e < TR * NB) s_x[(e / NB) * LX + e % NB] = tx[i];
    }
    __syncthreads();
    fetch(t + 1);
#pragma unroll
    for (int ks = 0; ks < TR / 4; ++ks) {
      const float a = s_g[(ks * 4 + lg) * LG + w * 16 + l15];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = MFMA16(a, s_x[(ks * 4 + lg) * LX + n * 16 + l15], acc[n]);
    }
    __syncthreads();
  }
  float *o = part + (size_t)blockIdx.x * M * N;
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = m0 + w * 16 + 4 * lg + u, c = n0 + n * 16 + l15;
      if (m < M && c < N) o[(size_t)m * N + c] = acc[n][u];
    }
}
}  // namespace

/* row slabs (= partial results) spacap_dense_wgrad_tall_f32 should be called with for (R, M, N) */
extern "C" int spacap_dense_wgrad_tall_slabs(long R, int M, int N) {
  if (R < 1 || M < 1 || N < 1) return 0;
  const long tiles = (R + 31) / 32, yz = (long)((M + 63) / 64) * ((N + (N <= 16 ? 15 : 143)) / (N <= 16 ? 16 : 144));
  long n = 512 / yz, cap = (8L << 20) / ((long)M * N);   // <= 32 MB of partial results
  if (n > cap) n = cap;
  if (n > tiles / 4) n = tiles / 4;                      // at least four tiles per slab
  return (int)(n < 1 ? 1 : n);
}

/* part f32 [nslab][M][N]: per row slab, dW[m][n] = sum_r G[r][m] X[r][n] over the slab's rows (ascending); G rows of M floats at
   stride ldg, X rows of N floats at stride ldx; any R >= 1, M, N >= 1, nslab >= 1.  The caller adds the slabs in order. */
extern "C" int spacap_dense_wgrad_tall_f32(const float *G, long ldg, const float *X, long ldx, long R, int M, int N, int nslab, float *part,
                                           spacap_stream_t stream) {
  const char *what = "spacap_dense_wgrad_tall_f32";
  SPACAP_REQUIRE(R >= 1 && M >= 1 && N >= 1 && nslab >= 1 && nslab <= 65535 && ldg >= M && ldx >= N, "%s: bad sizes", what);
  SPACAP_REQUIRE(G && X && part, "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
  const unsigned gy = (unsigned)((M + 63) / 64);
  if (N <= 16)
    hipLaunchKernelGGL(dense_wgrad_tall_kernel<1>, dim3(nslab, gy, 1), dim3(256), 0, s, G, ldg, X, ldx, R, M, N, part);
  else
    hipLaunchKernelGGL(dense_wgrad_tall_kernel<9>, dim3(nslab, gy, (unsigned)((N + 143) / 144)), dim3(256), 0, s, G, ldg, X, ldx, R, M, N, part);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
