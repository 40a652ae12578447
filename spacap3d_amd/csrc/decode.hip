// Score decoding of the proposal head for gfx950 (MI355X): models/proposal_module.py:106-158 (decode_scores) and :81-104
// (decode_pred_box; the reference takes a GPU -> CPU -> numpy -> GPU round trip there).
//
// The head's output (B, CH, K) -- CH = 2 + 3 + 2 NH + 4 NS + NC channels per proposal -- is sliced into objectness scores,
// centre offset, heading scores / residuals, size scores / residuals and class scores; PyTorch runs the transpose, the centre
// add, two residual scalings, four arg-maxes, a gather and the float64 box corners as ~20 launches of a few microseconds on a
// 200 K-element tensor (and as many again for the slice gradients).  Here: one launch forward (one thread per proposal,
// channel-major reads coalesced over the proposals), one launch for the gradient.  Arithmetic as the composition: fp32 centre
// add and residual products (one rounding each), first-maximum arg-max, corners in float64.
#include <math.h>

#include "common.hpp"

namespace {

__global__ __launch_bounds__(64) void proposal_decode_fwd_kernel(const float *__restrict__ net, const float *__restrict__ agg_xyz,
                                                                 const float *__restrict__ msa, const double *__restrict__ msa64,
                                                                 int K, int NH, int NS, int NC,
                                                                 float *__restrict__ nt, float *__restrict__ center,
                                                                 float *__restrict__ hres, float *__restrict__ sres,
                                                                 double *__restrict__ corners, int64_t *__restrict__ bbox_mask,
                                                                 int64_t *__restrict__ sem_cls, int64_t *__restrict__ size_cls) {
  // the 64 proposals' channels through LDS: channel-major reads and proposal-major writes are both coalesced
  extern __shared__ float s_x[];                         // [64][CH + 1]
  const int b = blockIdx.y, k0 = blockIdx.x * 64, k = k0 + threadIdx.x;
  const int CH = 5 + 2 * NH + 4 * NS + NC, LD = CH + 1, nk = min(64, K - k0);
  for (int c = 0; c < CH; ++c)
    if (k < K) s_x[threadIdx.x * LD + c] = net[((size_t)b * CH + c) * K + k];
  __syncthreads();
  for (int i = threadIdx.x; i < nk * CH; i += 64) nt[((size_t)b * K + k0) * CH + i] = s_x[(i / CH) * LD + i % CH];
  if (k >= K) return;
  const float *row = s_x + threadIdx.x * LD;
  const size_t p = (size_t)b * K + k;
  bbox_mask[p] = row[1] > row[0] ? 1 : 0;                 // argmax over (no object, object): first maximum
  float cen[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) cen[d] = agg_xyz[p * 3 + d] + row[2 + d], center[p * 3 + d] = cen[d];
  const float hs = (float)(M_PI / (double)NH);
  for (int i = 0; i < NH; ++i) hres[p * NH + i] = row[5 + NH + i] * hs;
  const float *ss = row + 5 + 2 * NH, *sr = ss + NS;
  int sc = 0;
  float best = ss[0];
  for (int j = 1; j < NS; ++j)
    if (ss[j] > best) best = ss[j], sc = j;
  size_cls[p] = sc;
  float mine[3] = {0.f, 0.f, 0.f};
  for (int j = 0; j < NS; ++j)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float v = sr[3 * j + d] * msa[3 * j + d];
      sres[(p * NS + j) * 3 + d] = v;
      if (j == sc) mine[d] = v;
    }
  const float *cl = sr + 3 * NS;
  int am = 0;
  best = cl[0];
  for (int j = 1; j < NC; ++j)
    if (cl[j] > best) best = cl[j], am = j;
  sem_cls[p] = am;
  // utils/box_util.py:377-379 corner order (l on x, w on y, h on z); heading is 0 for this dataset configuration
  const double sx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sy[8] = {1, -1, -1, 1, 1, -1, -1, 1}, sz[8] = {1, 1, 1, 1, -1, -1, -1, -1};
  double half[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) half[d] = ((msa64 ? msa64[3 * sc + d] : (double)msa[3 * sc + d]) + (double)mine[d]) / 2.0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    corners[(p * 8 + c) * 3 + 0] = (double)cen[0] + sx[c] * half[0];
    corners[(p * 8 + c) * 3 + 1] = (double)cen[1] + sy[c] * half[1];
    corners[(p * 8 + c) * 3 + 2] = (double)cen[2] + sz[c] * half[2];
  }
}

// d net[b, c, k] = g_nt[b, k, c] (+ g_center on channels 2..4, + g_hres * pi / NH on the heading residuals, + g_sres * mean
// size on the size residuals); every gradient pointer may be null
__global__ __launch_bounds__(64) void proposal_decode_bwd_kernel(const float *__restrict__ g_nt, const float *__restrict__ g_center,
                                                                 const float *__restrict__ g_hres, const float *__restrict__ g_sres,
                                                                 const float *__restrict__ msa, int K, int NH, int NS, int NC,
                                                                 float *__restrict__ d_net) {
  extern __shared__ float s_x[];                         // [64][CH + 1]
  const int b = blockIdx.y, k0 = blockIdx.x * 64, k = k0 + threadIdx.x;
  const int CH = 5 + 2 * NH + 4 * NS + NC, LD = CH + 1, nk = min(64, K - k0);
  const float hs = (float)(M_PI / (double)NH);
  const int s0 = 5 + 2 * NH + NS;
  for (int i = threadIdx.x; i < nk * CH; i += 64) {      // proposal-major reads, coalesced
    const int kk = i / CH, c = i % CH;
    const size_t p = (size_t)b * K + k0 + kk;
    float v = g_nt ? g_nt[p * CH + c] : 0.f;
    if (g_center && c >= 2 && c < 5) v += g_center[p * 3 + (c - 2)];
    if (g_hres && c >= 5 + NH && c < 5 + 2 * NH) v += g_hres[p * NH + (c - 5 - NH)] * hs;
    if (g_sres && c >= s0 && c < s0 + 3 * NS) v += g_sres[p * NS * 3 + (c - s0)] * msa[c - s0];
    s_x[kk * LD + c] = v;
  }
  __syncthreads();
  if (k >= K) return;
  for (int c = 0; c < CH; ++c) d_net[((size_t)b * CH + c) * K + k] = s_x[threadIdx.x * LD + c];
}

}  // namespace

extern "C" int spacap_proposal_decode_fwd_f32(const float *net, const float *agg_xyz, const float *mean_size,
                                              const double *mean_size_f64, int B, int K, int NH, int NS, int NC, float *nt, float *center, float *heading_res, float *size_res,
                                              double *corners, int64_t *bbox_mask, int64_t *sem_cls, int64_t *size_cls,
                                              spacap_stream_t stream) {
  const char *what = "spacap_proposal_decode_fwd_f32";
  SPACAP_REQUIRE(B >= 0 && K >= 1 && NH >= 1 && NS >= 1 && NC >= 1, "%s: bad sizes", what);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(net && agg_xyz && mean_size && nt && center && heading_res && size_res && corners && bbox_mask && sem_cls && size_cls,
                 "%s: null pointer", what);
  const int CH = 5 + 2 * NH + 4 * NS + NC;
  SPACAP_REQUIRE((size_t)64 * (CH + 1) * sizeof(float) <= 64 * 1024, "%s: too many channels (%d)", what, CH);
  hipLaunchKernelGGL(proposal_decode_fwd_kernel, dim3((K + 63) / 64, B), dim3(64), (size_t)64 * (CH + 1) * sizeof(float),
                     spacap::as_stream(stream), net, agg_xyz, mean_size,
                     mean_size_f64, K, NH, NS, NC, nt, center, heading_res, size_res, corners, bbox_mask, sem_cls, size_cls);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_proposal_decode_bwd_f32(const float *g_nt, const float *g_center, const float *g_heading_res,
                                              const float *g_size_res, const float *mean_size, int B, int K, int NH, int NS, int NC,
                                              float *d_net, spacap_stream_t stream) {
  const char *what = "spacap_proposal_decode_bwd_f32";
  SPACAP_REQUIRE(B >= 0 && K >= 1 && NH >= 1 && NS >= 1 && NC >= 1, "%s: bad sizes", what);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(mean_size && d_net, "%s: null pointer", what);
  const int CH = 5 + 2 * NH + 4 * NS + NC;
  SPACAP_REQUIRE((size_t)64 * (CH + 1) * sizeof(float) <= 64 * 1024, "%s: too many channels (%d)", what, CH);
  hipLaunchKernelGGL(proposal_decode_bwd_kernel, dim3((K + 63) / 64, B), dim3(64), (size_t)64 * (CH + 1) * sizeof(float),
                     spacap::as_stream(stream), g_nt, g_center,
                     g_heading_res, g_size_res, mean_size, K, NH, NS, NC, d_net);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
