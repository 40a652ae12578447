// gather_points / group_points and their gradients for gfx950 (MI355X).
//
// Replaces lib/pointnet2/_ext_src/src/sampling_gpu.cu:8-57 and src/group_points_gpu.cu:8-75.
// Forward ops are exact copies of selected elements (bit-exact).  The gradients scatter-add with
// float atomics exactly as the reference does (sampling_gpu.cu:42, group_points_gpu.cu:60), so their
// sums are order-dependent in the last bits, as the reference's are.
//
// Design: the reference launches one block per scene (group) and walks (channel, centre) pairs with
// the sample index innermost per thread, i.e. stride-S writes.  Here the flattened (centre, sample)
// index is the lane index -- index reads and output writes are fully coalesced -- each thread keeps
// its point index in a register and walks a slab of CHUNK channels, and the grid is
// (P*S / 256, C / CHUNK, B) so every CU has work.
#include "common.hpp"

namespace {

constexpr int CHUNK = 8;

// out[b,c,e] = points[b,c,idx[b,e]],  e in [0, E)   (E = m for gather, P*S for group)
__global__ __launch_bounds__(256) void index_select_kernel(const float *__restrict__ points,
                                                           const int32_t *__restrict__ idx, int C, int N,
                                                           int E, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int a = idx[(size_t)b * E + e];
  const int c0 = blockIdx.y * CHUNK;
  const int c1 = min(c0 + CHUNK, C);
  const float *__restrict__ p = points + ((size_t)b * C + c0) * N + a;
  float *__restrict__ o = out + ((size_t)b * C + c0) * E + e;
#pragma unroll 4
  for (int c = c0; c < c1; ++c, p += N, o += E) *o = *p;
}

// grad_points[b,c,idx[b,e]] += grad_out[b,c,e]
__global__ __launch_bounds__(256) void index_scatter_add_kernel(const float *__restrict__ grad_out,
                                                                const int32_t *__restrict__ idx, int C,
                                                                int N, int E,
                                                                float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int a = idx[(size_t)b * E + e];
  const int c0 = blockIdx.y * CHUNK;
  const int c1 = min(c0 + CHUNK, C);
  float *__restrict__ g = grad_points + ((size_t)b * C + c0) * N + a;
  const float *__restrict__ go = grad_out + ((size_t)b * C + c0) * E + e;
#pragma unroll 4
  for (int c = c0; c < c1; ++c, g += N, go += E) atomicAdd(g, *go);
}

int select(const char *what, const float *points, const int32_t *idx, int B, int C, int N, long E,
           float *out, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 0 && C >= 0 && N >= 0 && E >= 0, "%s: bad sizes", what);
  if (B == 0 || C == 0 || E == 0) return SPACAP_OK;
  SPACAP_REQUIRE(points && idx && out, "%s: null pointer", what);
  SPACAP_REQUIRE(E < (1L << 31) && B <= 65535, "%s: size out of range", what);
  dim3 grid((unsigned)((E + 255) / 256), (C + CHUNK - 1) / CHUNK, B);
  hipLaunchKernelGGL(index_select_kernel, grid, dim3(256), 0, spacap::as_stream(stream), points, idx, C, N,
                     (int)E, out);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

int scatter(const char *what, const float *grad_out, const int32_t *idx, int B, int C, int N, long E,
            float *grad_points, spacap_stream_t stream) {
  SPACAP_REQUIRE(B >= 0 && C >= 0 && N >= 0 && E >= 0, "%s: bad sizes", what);
  if (B == 0 || C == 0 || N == 0) return SPACAP_OK;
  SPACAP_REQUIRE(grad_points, "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
  SPACAP_CHECK_HIP(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)B * C * N, s), what);
  if (E == 0) return SPACAP_OK;
  SPACAP_REQUIRE(grad_out && idx, "%s: null pointer", what);
  SPACAP_REQUIRE(E < (1L << 31) && B <= 65535, "%s: size out of range", what);
  dim3 grid((unsigned)((E + 255) / 256), (C + CHUNK - 1) / CHUNK, B);
  hipLaunchKernelGGL(index_scatter_add_kernel, grid, dim3(256), 0, s, grad_out, idx, C, N, (int)E, grad_points);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

}  // namespace

extern "C" int spacap_gather_points_f32(const float *points, const int32_t *idx, int B, int C, int N, int m,
                                        float *out, spacap_stream_t stream) {
  return select("spacap_gather_points_f32", points, idx, B, C, N, m, out, stream);
}

extern "C" int spacap_gather_points_grad_f32(const float *grad_out, const int32_t *idx, int B, int C, int N,
                                             int m, float *grad_points, spacap_stream_t stream) {
  return scatter("spacap_gather_points_grad_f32", grad_out, idx, B, C, N, m, grad_points, stream);
}

extern "C" int spacap_group_points_f32(const float *points, const int32_t *idx, int B, int C, int N, int P,
                                       int S, float *out, spacap_stream_t stream) {
  return select("spacap_group_points_f32", points, idx, B, C, N, (long)P * S, out, stream);
}

extern "C" int spacap_group_points_grad_f32(const float *grad_out, const int32_t *idx, int B, int C, int N,
                                            int P, int S, float *grad_points, spacap_stream_t stream) {
  return scatter("spacap_group_points_grad_f32", grad_out, idx, B, C, N, (long)P * S, grad_points, stream);
}
