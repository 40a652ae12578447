// 1x1 convolutions on channel-major tensors -- forward and input gradient -- for gfx950 (MI355X).
//
// Reference: nn.Conv1d / nn.Conv2d with kernel size 1 in models/voting_module.py:33-60, models/proposal_module.py:41-55,
// lib/pointnet2/pointnet2_modules.py:376-421 (the feature-propagation SharedMLPs, lib/pointnet2/pytorch_utils.py:11-36) and
// the learned position embedding of models/transformer_captioner.py:251-258.  A few hundred to two thousand points per scene,
// 128 - 768 channels: 0.1 - 1 GFLOP each, launched ~30 times per training step.  As library calls they are 6 - 25 us each
// (some with transposed copies around them); here one kernel computes
//     C[b, m, n] = sum_k A[m, k] In[b, k, n]  (+ bias[m])
// on the layout the tensors already have (n contiguous): forward A = W [CO][CI] (m = output channel), input gradient
// A = W^T (m = input channel, k = output channel, read through LDS).  Workgroup tile 64 (m) x 64 (n), k in chunks of 32:
// the activation chunk goes through LDS ([32][64 + 4], 16-byte global loads, one chunk ahead in registers), each wave owns 16
// rows of m and four 16-column tiles; v_mfma_f32_16x16x4_f32 (exact fp32 products).  The k index of MFMA step s in lane group
// lg is 8 lg + s, so that a lane's weights for a chunk are two 16-byte loads.  Any M and K (tails clamped / zeroed), N a
// multiple of 64.
#include <stdlib.h>

#include "common.hpp"

namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

constexpr int TM = 64, TN = 64, KC = 32, LDN = TN + 4, LDW = TM + 4;

// TRANS_A: A[m][k] = W[k * lda + m] (input gradient), else A[m][k] = W[m * lda + k]
template <bool TRANS_A>
__global__ __launch_bounds__(256) void conv1x1_cm_kernel(const float *__restrict__ W, int lda, const float *__restrict__ in,
                                                         const float *__restrict__ bias, int M, int K, int N,
                                                         float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float s_in[2][KC * LDN];
  __shared__ __attribute__((aligned(16))) float s_w[TRANS_A ? 2 * KC * LDW : 4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * TN, m0 = blockIdx.y * TM, b = blockIdx.z;
  const float *inb = in + (size_t)b * K * N;
  // staging maps: the activation chunk as 512 float4 (two per thread), the transposed weight chunk likewise
  const int sr = tid >> 4, sc = (tid & 15) * 4;            // rows sr, sr + 16 of the chunk, columns sc .. sc + 3
  f32x4 pin[2], pw[2];
  auto request = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = k0 + sr + 16 * e;
      const f32x4 v = ld4(inb + (size_t)min(k, K - 1) * N + n0 + sc);
      pin[e] = k < K ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      if (TRANS_A) {   // W[k][m0 + sc ..]: 64 consecutive m per k row
        const float *p = W + (size_t)min(k, K - 1) * lda;
        f32x4 t;
        if (m0 + sc + 3 < M && (lda & 3) == 0) {
          t = ld4(p + m0 + sc);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) t[u] = p[min(m0 + sc + u, M - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = (k < K && m0 + sc + u < M) ? t[u] : 0.f;
        pw[e] = t;
      }
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      st4(&s_in[buf][(sr + 16 * e) * LDN + sc], pin[e]);
      if (TRANS_A) st4(&s_w[buf * KC * LDW + (sr + 16 * e) * LDW + sc], pw[e]);
    }
  };
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int mrow = min(m0 + 16 * w + l15, M - 1);
  const int nchunks = (K + KC - 1) / KC;
  request(0);
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1, k0 = c * KC;
    store(buf);
    f32x4 a0, a1;   // the lane's weights of this chunk: k = k0 + 8 lg + 0 .. 7
    if (!TRANS_A) {
      const float *p = W + (size_t)mrow * lda + k0 + 8 * lg;
      if (k0 + KC <= K && (lda & 3) == 0) {
        a0 = ld4(p), a1 = ld4(p + 4);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a0[u] = k0 + 8 * lg + u < K ? p[u] : 0.f;
          a1[u] = k0 + 8 * lg + 4 + u < K ? p[4 + u] : 0.f;
        }
      }
    }
    if (c + 1 < nchunks) request(k0 + KC);
    __syncthreads();   // chunk c is in LDS (and chunk c - 1's readers were done before its buffer was written again)
    const float *si = &s_in[buf][(8 * lg) * LDN + l15];
    if (TRANS_A) {
      const float *sw = &s_w[buf * KC * LDW + (8 * lg) * LDW + 16 * w + l15];
#pragma unroll
      for (int u = 0; u < 4; ++u) a0[u] = sw[u * LDW], a1[u] = sw[(4 + u) * LDW];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float av = s < 4 ? a0[s] : a1[s - 4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = MFMA16(av, si[s * LDN + 16 * t], acc[t]);
    }
  }
  // acc[t][u] = C[m0 + 16 w + 4 lg + u][n0 + 16 t + l15]
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int m = m0 + 16 * w + 4 * lg + u;
    if (m < M) {
      const float bv = bias ? bias[m] : 0.f;
      float *o = out + ((size_t)b * M + m) * N + n0 + l15;
#pragma unroll
      for (int t = 0; t < 4; ++t) o[16 * t] = acc[t][u] + bv;
    }
  }
}


// ---- the same product on the bf16 matrix cores with fp32-equivalent accuracy (the default) ---------------------------------
// The fp32-MFMA kernel above is matrix-pipe bound (1 GFLOP per layer at 1/16 of the bf16 rate, 20 us for 7.8 us of matrix
// time).  Here both operands are split into three bf16 pieces (x = x1 + x2 + x3, 24 significant bits) on their way in and a
// product is the six piece products above 2^-24 on v_mfma_f32_16x16x32_bf16 (6/16 of the fp32-MFMA time; the arithmetic of
// sa_bf3.inc / relation_fused.hip / wgrad_bf3.inc).  The activation chunk [32 k][64 n] stays row-major in LDS (n contiguous, as
// in memory) and the B fragments -- 8 consecutive k of one column n -- come out of it through ds_read_b64_tr_b16, the
// transposing LDS read of gfx950; the forward's weights are split in registers (a lane's 8 consecutive k of its row), the
// input gradient's W^T chunk goes through a second image that is read the same way.  Same tiling, grid and C layout as above.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BLD = TN + 8;                // bf16 elements per image row (144 bytes)
constexpr int BIMG = KC * BLD;             // one piece of one chunk

__device__ __forceinline__ void split4(f32x4 v, bf16x4 &p0, bf16x4 &p1, bf16x4 &p2) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const __bf16 h = (__bf16)v[u];
    const float r = v[u] - (float)h;
    const __bf16 m = (__bf16)r;
    p0[u] = h, p1[u] = m, p2[u] = (__bf16)(r - (float)m);
  }
}
// 16x16x32 fragment whose 16 outer indices are the image columns c0 .. c0 + 15 and whose contraction index is the chunk's 32
// rows: lane (g = lane >> 4, i = lane & 15) gets rows 8 g .. 8 g + 7 of column c0 + i (two transposing reads of 4 rows each;
// lane 4 q + p of a group supplies the address of row q, columns 4 p .. 4 p + 3 of the block)
__device__ __forceinline__ bf16x8 tr_frag(const __bf16 *img, int c0, int lane) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const __bf16 *a = img + (8 * g + q) * BLD + c0 + 4 * p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(a + 4 * BLD));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <bool TRANS_A>
__global__ __launch_bounds__(256) void conv1x1_cm_bf3_kernel(const float *__restrict__ W, int lda, const float *__restrict__ in,
                                                             const float *__restrict__ bias, int M, int K, int N,
                                                             float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) __bf16 s_b[2][3 * BIMG];
  __shared__ __attribute__((aligned(16))) __bf16 s_a[TRANS_A ? 2 * 3 * BIMG : 8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * TN, m0 = blockIdx.y * TM, b = blockIdx.z;
  const float *inb = in + (size_t)b * K * N;
  const int sr = tid >> 4, sc = (tid & 15) * 4;            // rows sr, sr + 16 of the chunk, columns sc .. sc + 3
  f32x4 pin[2], pw[2];
  auto request = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = k0 + sr + 16 * e;
      const f32x4 v = ld4(inb + (size_t)min(k, K - 1) * N + n0 + sc);
      pin[e] = k < K ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      if (TRANS_A) {   // W[k][m0 + sc ..]: 64 consecutive m per k row
        const float *p = W + (size_t)min(k, K - 1) * lda;
        f32x4 t;
        if (m0 + sc + 3 < M && (lda & 3) == 0) {
          t = ld4(p + m0 + sc);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) t[u] = p[min(m0 + sc + u, M - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = (k < K && m0 + sc + u < M) ? t[u] : 0.f;
        pw[e] = t;
      }
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int o = (sr + 16 * e) * BLD + sc;
      bf16x4 p0, p1, p2;
      split4(pin[e], p0, p1, p2);
      *reinterpret_cast<bf16x4 *>(&s_b[buf][o]) = p0;
      *reinterpret_cast<bf16x4 *>(&s_b[buf][BIMG + o]) = p1;
      *reinterpret_cast<bf16x4 *>(&s_b[buf][2 * BIMG + o]) = p2;
      if (TRANS_A) {
        split4(pw[e], p0, p1, p2);
        *reinterpret_cast<bf16x4 *>(&s_a[buf * 3 * BIMG + o]) = p0;
        *reinterpret_cast<bf16x4 *>(&s_a[buf * 3 * BIMG + BIMG + o]) = p1;
        *reinterpret_cast<bf16x4 *>(&s_a[buf * 3 * BIMG + 2 * BIMG + o]) = p2;
      }
    }
  };
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int mrow = min(m0 + 16 * w + l15, M - 1);
  const int nchunks = (K + KC - 1) / KC;
  f32x4 na0 = {0.f, 0.f, 0.f, 0.f}, na1 = na0;   // forward: the lane's weights of the NEXT chunk, k = k0 + 8 lg + 0 .. 7
  auto wrequest = [&](int k0) {
    if (TRANS_A) return;
    const float *p = W + (size_t)mrow * lda + k0 + 8 * lg;
    if (k0 + KC <= K && (lda & 3) == 0) {
      na0 = ld4(p), na1 = ld4(p + 4);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        na0[u] = k0 + 8 * lg + u < K ? p[u] : 0.f;
        na1[u] = k0 + 8 * lg + 4 + u < K ? p[4 + u] : 0.f;
      }
    }
  };
  request(0);
  wrequest(0);
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1, k0 = c * KC;
    store(buf);
    bf16x8 a[3];
    if (!TRANS_A) {
      bf16x4 l0, l1, l2, h0, h1, h2;
      split4(na0, l0, l1, l2);
      split4(na1, h0, h1, h2);
      a[0] = bf16x8{l0[0], l0[1], l0[2], l0[3], h0[0], h0[1], h0[2], h0[3]};
      a[1] = bf16x8{l1[0], l1[1], l1[2], l1[3], h1[0], h1[1], h1[2], h1[3]};
      a[2] = bf16x8{l2[0], l2[1], l2[2], l2[3], h2[0], h2[1], h2[2], h2[3]};
    }
    if (c + 1 < nchunks) {
      request(k0 + KC);
      wrequest(k0 + KC);
    }
    __syncthreads();   // chunk c is in LDS (and chunk c - 1's readers were done before its buffer was written again)
    if (TRANS_A) {
#pragma unroll
      for (int q = 0; q < 3; ++q) a[q] = tr_frag(&s_a[buf * 3 * BIMG + q * BIMG], 16 * w, lane);
    }
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};   // the six products above 2^-24, smallest first
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      bf16x8 bq[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) bq[q] = tr_frag(&s_b[buf][q * BIMG], 16 * t, lane);
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]], bq[PB[q]], acc[t], 0, 0, 0);
    }
  }
  // acc[t][u] = C[m0 + 16 w + 4 lg + u][n0 + 16 t + l15]
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int m = m0 + 16 * w + 4 * lg + u;
    if (m < M) {
      const float bv = bias ? bias[m] : 0.f;
      float *o = out + ((size_t)b * M + m) * N + n0 + l15;
#pragma unroll
      for (int t = 0; t < 4; ++t) o[16 * t] = acc[t][u] + bv;
    }
  }
}

// SPACAP_SA_F32MFMA=1 (the library's one switch, sa_mlp.hip) keeps the fp32-MFMA kernel
inline bool conv_f32_mfma_only() {
  static const bool on = getenv("SPACAP_SA_F32MFMA") != nullptr && atoi(getenv("SPACAP_SA_F32MFMA")) != 0;
  return on;
}

}  // namespace

extern "C" int spacap_conv1x1_cm_supported(int CI, int CO, long N) { return CI >= 1 && CO >= 1 && N >= 64 && N % 64 == 0; }

// mode 0: out[b, co, n] = sum_ci W[co, ci] in[b, ci, n] + bias[co]   (in [B,CI,N], out [B,CO,N], bias may be NULL)
// mode 1: out[b, ci, n] = sum_co W[co, ci] in[b, co, n]              (in [B,CO,N], out [B,CI,N]): the input gradient
// mode 2: mode 0 on the fp32-MFMA kernel (exact fp32 products; mode 0 is split-bf16 unless SPACAP_SA_F32MFMA=1)
extern "C" int spacap_conv1x1_cm_f32(int mode, const float *W, const float *in, const float *bias, int B, int CI, int CO, long N,
                                     float *out, spacap_stream_t stream) {
  const char *what = "spacap_conv1x1_cm_f32";
  SPACAP_REQUIRE((mode == 0 || mode == 1 || mode == 2) && B >= 0 && spacap_conv1x1_cm_supported(CI, CO, N) && N <= 2147483647L && B <= 65535,
                 "%s: (mode=%d, B=%d, CI=%d, CO=%d, N=%ld) unsupported", what, mode, B, CI, CO, N);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(W && in && out && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                 "%s: null or unaligned pointer", what);
  const bool f32fwd = mode == 2;   // mode 2: the forward of mode 0 on the fp32-MFMA kernel whatever the environment says
  if (f32fwd) mode = 0;
  const int M = mode == 0 ? CO : CI, K = mode == 0 ? CI : CO;
  const dim3 grid((unsigned)(N / TN), (unsigned)((M + TM - 1) / TM), (unsigned)B);
  hipStream_t s = spacap::as_stream(stream);
  // measured in the step (cfg2, 12 + 12 launches): forward 219 -> 192 us on the split-bf16 kernel; the input gradient, whose W^T
  // chunk must be split and staged as a second image, 232 -> 246 us: it stays on the fp32-MFMA kernel
  if (mode == 0 && !f32fwd && !conv_f32_mfma_only())
    hipLaunchKernelGGL(conv1x1_cm_bf3_kernel<false>, grid, dim3(256), 0, s, W, CI, in, bias, M, K, (int)N, out);
  else if (mode == 0) hipLaunchKernelGGL(conv1x1_cm_kernel<false>, grid, dim3(256), 0, s, W, CI, in, bias, M, K, (int)N, out);
  else hipLaunchKernelGGL(conv1x1_cm_kernel<true>, grid, dim3(256), 0, s, W, CI, in, (const float *)nullptr, M, K, (int)N, out);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
