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

}  // namespace

extern "C" int spacap_conv1x1_cm_supported(int CI, int CO, long N) { return CI >= 1 && CO >= 1 && N >= 64 && N % 64 == 0; }

// mode 0: out[b, co, n] = sum_ci W[co, ci] in[b, ci, n] + bias[co]   (in [B,CI,N], out [B,CO,N], bias may be NULL)
// mode 1: out[b, ci, n] = sum_co W[co, ci] in[b, co, n]              (in [B,CO,N], out [B,CI,N]): the input gradient
extern "C" int spacap_conv1x1_cm_f32(int mode, const float *W, const float *in, const float *bias, int B, int CI, int CO, long N,
                                     float *out, spacap_stream_t stream) {
  const char *what = "spacap_conv1x1_cm_f32";
  SPACAP_REQUIRE((mode == 0 || mode == 1) && B >= 0 && spacap_conv1x1_cm_supported(CI, CO, N) && N <= 2147483647L && B <= 65535,
                 "%s: (mode=%d, B=%d, CI=%d, CO=%d, N=%ld) unsupported", what, mode, B, CI, CO, N);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(W && in && out && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                 "%s: null or unaligned pointer", what);
  const int M = mode == 0 ? CO : CI, K = mode == 0 ? CI : CO;
  const dim3 grid((unsigned)(N / TN), (unsigned)((M + TM - 1) / TM), (unsigned)B);
  hipStream_t s = spacap::as_stream(stream);
  if (mode == 0) hipLaunchKernelGGL(conv1x1_cm_kernel<false>, grid, dim3(256), 0, s, W, CI, in, bias, M, K, (int)N, out);
  else hipLaunchKernelGGL(conv1x1_cm_kernel<true>, grid, dim3(256), 0, s, W, CI, in, (const float *)nullptr, M, K, (int)N, out);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
