// Pairwise relation feature of the spatiality-guided encoder, forward and backward, for gfx950 (MI355X).
//
// Replaces models/transformer_captioner.py:393-396
//     attn  = P.unsqueeze(-1).repeat(1,1,1,1,16)            # (B,h,K,K,16)  copy
//     value = V.unsqueeze(-3)                               # (B,h,1,K,16)
//     R     = (attn * value).transpose(1,2).transpose(2,3).contiguous().view(B,K,K,h*16)
// i.e.  R[b,i,j,h*D+d] = P[b,h,i,j] * V[b,h,j,d]  -- the largest tensor of the model (268 MB at B=8, K=256).
// The reference materialises it three times (repeat, product, contiguous); autograd then walks the same chain
// backwards (a copy, a product with two reductions).  Here one kernel writes R once in its final layout, and one
// kernel reads dR once and produces both dP (reduction over d) and dV (reduction over the query index i) with
// fixed summation order -- no atomics.
#include "common.hpp"

namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));

// grid (K, B): one workgroup writes the K x C slab of query i
__global__ __launch_bounds__(256) void relation_fwd_kernel(const float *__restrict__ P, const float *__restrict__ V,
                                                           long v_sb, long v_sh, long v_sl, int H, int K, int D,
                                                           float *__restrict__ R) {
  const int C = H * D, C4 = C / 4, RPI = 256 / C4;
  const int i = blockIdx.x, b = blockIdx.y;
  const int c4 = threadIdx.x % C4, jj = threadIdx.x / C4;
  const int h = (c4 * 4) / D, d = (c4 * 4) % D;
  const float *prow = P + (((size_t)b * H + h) * K + i) * K;
  float *out = R + (((size_t)b * K + i) * K) * C + c4 * 4;
  for (int j = jj; j < K; j += RPI) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(V + b * v_sb + h * v_sh + j * v_sl + d);
    const float p = prow[j];
    *reinterpret_cast<f32x4 *>(out + (size_t)j * C) = v * p;
  }
}

// grid (K / RPI, B): one workgroup owns RPI key columns j, walks all queries i; dV accumulates in registers
__global__ __launch_bounds__(256) void relation_bwd_kernel(const float *__restrict__ dR, const float *__restrict__ P,
                                                           const float *__restrict__ V, long v_sb, long v_sh,
                                                           long v_sl, int H, int K, int D, float *__restrict__ dP,
                                                           float *__restrict__ dV) {
  const int C = H * D, C4 = C / 4, RPI = 256 / C4, LPH = D / 4;  // LPH lanes share one head
  const int b = blockIdx.y;
  const int c4 = threadIdx.x % C4, jj = threadIdx.x / C4;
  const int j = blockIdx.x * RPI + jj;
  const int h = (c4 * 4) / D, d = (c4 * 4) % D;
  const bool ok = j < K;
  const int jc = ok ? j : K - 1;
  const f32x4 v = *reinterpret_cast<const f32x4 *>(V + b * v_sb + h * v_sh + jc * v_sl + d);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float *pcol = P + (((size_t)b * H + h) * K) * K + jc;
  float *dpcol = dP + (((size_t)b * H + h) * K) * K + jc;
  const float *g = dR + ((size_t)b * K * K + jc) * C + c4 * 4;
  for (int i = 0; i < K; ++i) {
    const f32x4 x = *reinterpret_cast<const f32x4 *>(g + (size_t)i * K * C);
    acc += x * pcol[(size_t)i * K];
    float s = x.x * v.x + x.y * v.y + x.z * v.z + x.w * v.w;
    for (int o = 1; o < LPH; o <<= 1) s += __shfl_xor(s, o);  // lanes of one head are adjacent
    if (ok && (c4 % LPH) == 0) dpcol[(size_t)i * K] = s;
  }
  if (ok) *reinterpret_cast<f32x4 *>(dV + (((size_t)b * K + j) * H + h) * D + d) = acc;
}

bool shape_ok(int H, int K, int D) {
  const int C = H * D;
  return H >= 1 && K >= 1 && D >= 4 && (D & (D - 1)) == 0 && D <= 64 && C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0;
}

}  // namespace

extern "C" int spacap_relation_feature_fwd_f32(const float *P, const float *V, long v_sb, long v_sh, long v_sl,
                                               int B, int H, int K, int D, float *R, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 0 && shape_ok(H, K, D), "spacap_relation_feature_fwd_f32: unsupported shape H=%d K=%d D=%d", H, K, D);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(P && V && R, "spacap_relation_feature_fwd_f32: null pointer");
  SPACAP_REQUIRE(B <= 65535 && (v_sb % 4) == 0 && (v_sh % 4) == 0 && (v_sl % 4) == 0 &&
                     (reinterpret_cast<uintptr_t>(V) & 15) == 0,
                 "spacap_relation_feature_fwd_f32: V must be 16-byte aligned with strides multiple of 4");
  hipLaunchKernelGGL(relation_fwd_kernel, dim3(K, B), dim3(256), 0, spacap::as_stream(stream), P, V, v_sb, v_sh, v_sl, H,
                     K, D, R);
  SPACAP_CHECK_LAUNCH("spacap_relation_feature_fwd_f32");
  return SPACAP_OK;
}

extern "C" int spacap_relation_feature_bwd_f32(const float *dR, const float *P, const float *V, long v_sb, long v_sh,
                                               long v_sl, int B, int H, int K, int D, float *dP, float *dV,
                                               spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 0 && shape_ok(H, K, D), "spacap_relation_feature_bwd_f32: unsupported shape H=%d K=%d D=%d", H, K, D);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(dR && P && V && dP && dV, "spacap_relation_feature_bwd_f32: null pointer");
  SPACAP_REQUIRE(B <= 65535 && (v_sb % 4) == 0 && (v_sh % 4) == 0 && (v_sl % 4) == 0 &&
                     (reinterpret_cast<uintptr_t>(V) & 15) == 0,
                 "spacap_relation_feature_bwd_f32: V must be 16-byte aligned with strides multiple of 4");
  const int RPI = 256 / (H * D / 4);
  hipLaunchKernelGGL(relation_bwd_kernel, dim3((K + RPI - 1) / RPI, B), dim3(256), 0, spacap::as_stream(stream), dR, P,
                     V, v_sb, v_sh, v_sl, H, K, D, dP, dV);
  SPACAP_CHECK_LAUNCH("spacap_relation_feature_bwd_f32");
  return SPACAP_OK;
}
