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

// ------------------------------------------------------------------------------------------------------------
// First layer of the relation MLP, fused with the feature:   H1 = relu(R W1^T + b1),  R = P (x) V
// (models/transformer_captioner.py:319-326,393-397).  Because R[b,i,j,h*D+d] = P[b,h,i,j] * V[b,h,j,d],
//     H1[b,i,j,o] = relu(b1[o] + sum_h P[b,h,i,j] * U[b,j,h,o]),     U[b,j,h,o] = sum_d V[b,h,j,d] * W1[o,h*D+d]
// so the 268 MB feature tensor never exists and the layer needs H (= 8) multiply-adds per output instead of H*D
// (= 128).  U (B,K,H,C: 8 MB) is a tiny batched GEMM done by the caller (whose autograd also turns dU into dV, dW1).
// One workgroup owns 256 / (C/4) key columns j and walks all queries i: U stays in registers, P is the only
// per-pair input (8 floats), H1 is written once; the backward reads dH1 and H1 once and produces dP, dU (register
// accumulation over i) and per-workgroup partial sums of db1 -- fixed summation order, no atomics.
namespace {

template <int H>
__global__ __launch_bounds__(256) void relation_l1_fwd_kernel(const float *__restrict__ P, const float *__restrict__ U,
                                                              const float *__restrict__ b1, int K, int C,
                                                              float *__restrict__ H1) {
  const int C4 = C / 4, RPI = 256 / C4;
  const int b = blockIdx.y;
  const int c4 = threadIdx.x % C4, jj = threadIdx.x / C4;
  const int j = blockIdx.x * RPI + jj;
  const bool ok = j < K;
  const int jc = ok ? j : K - 1;
  f32x4 u[H];
#pragma unroll
  for (int h = 0; h < H; ++h)
    u[h] = *reinterpret_cast<const f32x4 *>(U + (((size_t)b * K + jc) * H + h) * C + c4 * 4);
  const f32x4 bias = *reinterpret_cast<const f32x4 *>(b1 + c4 * 4);
  const float *pcol = P + ((size_t)b * H * K) * K + jc;
  float *out = H1 + ((size_t)b * K * K + jc) * C + c4 * 4;
  const int ichunk = (K + gridDim.z - 1) / gridDim.z, ibeg = blockIdx.z * ichunk, iend = min(K, ibeg + ichunk);
  for (int i = ibeg; i < iend; ++i) {
    f32x4 acc = bias;
#pragma unroll
    for (int h = 0; h < H; ++h) acc += u[h] * pcol[((size_t)h * K + i) * K];
    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    if (ok) *reinterpret_cast<f32x4 *>(out + (size_t)i * K * C) = acc;
  }
}

// Backward.  Thread (jj, c4) keeps U[b, j, :, 4 c4 .. 4 c4 + 3] in registers and walks the queries i in batches of
// NBI = 256 / (RPI * H): per (i, h) it forms its 4-term piece of dP[b,h,i,j] = sum_o g[o] * U[j,h,o]; the C/4 pieces of
// one output are added through LDS (one row per output, padded against bank conflicts) by the thread that owns that
// output -- 256 outputs per batch, one per thread -- instead of 5 shuffle steps per head and query.
template <int H>
__global__ __launch_bounds__(256) void relation_l1_bwd_kernel(const float *__restrict__ dH1, const float *__restrict__ H1,
                                                              const float *__restrict__ P, const float *__restrict__ U,
                                                              int K, int C, float *__restrict__ dP,
                                                              float *__restrict__ dU, float *__restrict__ db_part) {
  extern __shared__ float s_mem[];  // [256][C4 + 1] partial dots, then [RPI][C] for db
  const int C4 = C / 4, RPI = 256 / C4, NBI = 256 / (RPI * H), LD = C4 + 1;
  const int b = blockIdx.y;
  const int c4 = threadIdx.x % C4, jj = threadIdx.x / C4;
  const int j = blockIdx.x * RPI + jj;
  const bool ok = j < K;
  const int jc = ok ? j : K - 1;
  f32x4 u[H], du[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    u[h] = *reinterpret_cast<const f32x4 *>(U + (((size_t)b * K + jc) * H + h) * C + c4 * 4);
    du[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  f32x4 db = {0.f, 0.f, 0.f, 0.f};
  const float *pcol = P + ((size_t)b * H * K) * K + jc;
  const size_t base = ((size_t)b * K * K + jc) * C + c4 * 4;
  // the output this thread finishes in the reduction phase: (ii, jj2, h2)
  const int o_h = threadIdx.x % H, o_jj = (threadIdx.x / H) % RPI, o_ii = threadIdx.x / (H * RPI);
  const int o_j = blockIdx.x * RPI + o_jj;
  // the queries are cut into gridDim.z chunks (more workgroups in flight to hide the strided reads); every chunk
  // writes its own partial dU / db, summed by the caller in chunk order
  const int ichunk = (K + gridDim.z - 1) / gridDim.z, ibeg = blockIdx.z * ichunk, iend = min(K, ibeg + ichunk);
  for (int i0 = ibeg; i0 < iend; i0 += NBI) {
#pragma unroll 4
    for (int ii = 0; ii < NBI; ++ii) {
      const int i = i0 + ii;
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (ok && i < iend) {
        const size_t off = base + (size_t)i * K * C;
        g = *reinterpret_cast<const f32x4 *>(dH1 + off);
        const f32x4 a = *reinterpret_cast<const f32x4 *>(H1 + off);
        g.x = a.x > 0.f ? g.x : 0.f; g.y = a.y > 0.f ? g.y : 0.f; g.z = a.z > 0.f ? g.z : 0.f; g.w = a.w > 0.f ? g.w : 0.f;
      }
      db += g;
      const int ic = i < iend ? i : iend - 1;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        du[h] += g * pcol[((size_t)h * K + ic) * K];
        s_mem[((ii * RPI + jj) * H + h) * LD + c4] = g.x * u[h].x + g.y * u[h].y + g.z * u[h].z + g.w * u[h].w;
      }
    }
    __syncthreads();
    {
      const float *row = s_mem + threadIdx.x * LD;
      float s = 0.f;
      for (int q = 0; q < C4; ++q) s += row[q];
      const int i = i0 + o_ii;
      if (o_ii < NBI && i < iend && o_j < K) dP[(((size_t)b * H + o_h) * K + i) * K + o_j] = s;
    }
    __syncthreads();
  }
  if (ok) {
    float *duo = dU + (size_t)blockIdx.z * gridDim.y * K * H * C;  // partial of this i-chunk: [z][B][K][H][C]
#pragma unroll
    for (int h = 0; h < H; ++h) *reinterpret_cast<f32x4 *>(duo + (((size_t)b * K + j) * H + h) * C + c4 * 4) = du[h];
  }
  *reinterpret_cast<f32x4 *>(&s_mem[jj * C + c4 * 4]) = db;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {   // (C = 512: two channels per thread)
    float s = 0.f;
    for (int r = 0; r < RPI; ++r) s += s_mem[r * C + c];
    db_part[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * C + c] = s;
  }
}

bool l1_shape_ok(int H, int K, int C) {
  // (C = 512 / H = 32, the stress configuration: 128 threads per key column, two key columns per workgroup)
  return (H == 4 || H == 8 || H == 16 || H == 32) && K >= 1 && C % 4 == 0 && C >= 16 && C <= 512 && 256 % (C / 4) == 0 &&
         (C / 4) <= 128 && (256 / (C / 4)) * H <= 256;
}

}  // namespace

constexpr int L1_ISPLIT = 4;  // query chunks per key tile in the backward (partial dU / db per chunk)
extern "C" int spacap_relation_l1_isplit(void) { return L1_ISPLIT; }
extern "C" int spacap_relation_l1_supported(int H, int K, int C) { return l1_shape_ok(H, K, C) ? 1 : 0; }
extern "C" int spacap_relation_l1_blocks(int B, int K, int C) {
  return L1_ISPLIT * B * ((K + 256 / (C / 4) - 1) / (256 / (C / 4)));
}

extern "C" int spacap_relation_l1_fwd_f32(const float *P, const float *U, const float *b1, int B, int H, int K, int C,
                                          float *H1, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 0 && l1_shape_ok(H, K, C), "spacap_relation_l1_fwd_f32: unsupported shape H=%d K=%d C=%d", H, K, C);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(P && U && b1 && H1 && B <= 65535, "spacap_relation_l1_fwd_f32: null pointer");
  const int RPI = 256 / (C / 4);
  dim3 grid((K + RPI - 1) / RPI, B, 4);
  hipStream_t s = spacap::as_stream(stream);
#define L1F(HV) if (H == HV) hipLaunchKernelGGL((relation_l1_fwd_kernel<HV>), grid, dim3(256), 0, s, P, U, b1, K, C, H1);
  L1F(4) L1F(8) L1F(16) L1F(32)
#undef L1F
  SPACAP_CHECK_LAUNCH("spacap_relation_l1_fwd_f32");
  return SPACAP_OK;
}

extern "C" int spacap_relation_l1_bwd_f32(const float *dH1, const float *H1, const float *P, const float *U, int B, int H,
                                          int K, int C, float *dP, float *dU, float *db_part, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 0 && l1_shape_ok(H, K, C), "spacap_relation_l1_bwd_f32: unsupported shape H=%d K=%d C=%d", H, K, C);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(dH1 && H1 && P && U && dP && dU && db_part && B <= 65535, "spacap_relation_l1_bwd_f32: null pointer");
  const int RPI = 256 / (C / 4);
  dim3 grid((K + RPI - 1) / RPI, B, L1_ISPLIT);
  hipStream_t s = spacap::as_stream(stream);
  const size_t lds = sizeof(float) * (256 * (C / 4 + 1) > RPI * C ? 256 * (C / 4 + 1) : RPI * C);
  static unsigned long long lds_ok[4] = {0, 0, 0, 0};
#define L1B(HV, SLOT)                                                                                                         \
  if (H == HV) {                                                                                                              \
    if (lds > 65536)                                                                                                          \
      SPACAP_CHECK_HIP(spacap::allow_dynamic_lds(reinterpret_cast<const void *>(&relation_l1_bwd_kernel<HV>), 160 * 1024 - 512, \
                                                 lds_ok[SLOT]), "spacap_relation_l1_bwd_f32");                                \
    hipLaunchKernelGGL((relation_l1_bwd_kernel<HV>), grid, dim3(256), lds, s, dH1, H1, P, U, K, C, dP, dU, db_part);          \
  }
  L1B(4, 0) L1B(8, 1) L1B(16, 2) L1B(32, 3)
#undef L1B
  SPACAP_CHECK_LAUNCH("spacap_relation_l1_bwd_f32");
  return SPACAP_OK;
}
