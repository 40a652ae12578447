// Detection losses of the training step as three launches forward and one backward, for gfx950 (MI355X).
//
// Replaces lib/loss_helper.py:35-197 of the reference (compute_vote_loss, compute_objectness_loss,
// compute_box_and_sem_cls_loss with utils/nn_distance.py:32-62): ~150 PyTorch launches forward and as many backward,
// every one a few microseconds of latency on (B, 256) tensors.  Same arithmetic in float32 without FMA contraction
// (this file is built with -ffp-contract=off): squared distances as ((dx^2 + dy^2) + dz^2), first minimum wins, the
// objectness thresholds are applied to sqrt(d + 1e-6) -- the integer outputs (objectness label, assignment) are
// therefore the ones the PyTorch composition produces.
//
// Forward: det_proposal_kernel (one workgroup of 1 024 threads per scene, four lanes per proposal) assigns proposals to ground-truth boxes and evaluates every
// per-proposal loss term together with the UNNORMALISED gradient of that term; det_vote_kernel does the same for the
// vote loss; det_finalize_kernel adds the per-scene partial sums (fixed order) into the eight losses and the inverse
// denominators.  Backward: det_scale_kernel multiplies the stored gradient numerators by (upstream gradient x inverse
// denominator) and lays them out as d(net) (B, K, CH), d(center), d(vote_xyz).
//
// loss index: 0 vote, 1 objectness, 2 center, 3 heading_cls, 4 heading_reg, 5 size_cls, 6 size_reg, 7 sem_cls
// denominators: 0: sum vote mask + 1e-6, 1: sum objectness mask + 1e-6, 2: n_obj + 1e-6, 3: sum box mask + 1e-6
#include <math.h>

#include "common.hpp"

namespace {

constexpr int MAXM = 256;   // ground-truth boxes per scene (the reference: 128)
constexpr int NPARTL = 12;  // partial sums per scene: 8 loss numerators (center split in two) + 3 counts

struct DetArgs {
  // proposal head output nt (B, K, CH) = [objectness 2 | centre offset 3 | heading scores NH | heading residuals NH |
  // size scores NS | size residuals NS*3 | class scores NC]
  const float *net;
  const float *center;       // (B,K,3) decoded centres
  const float *agg_xyz;      // (B,K,3) aggregated vote positions
  const float *gt_center;    // (B,M,3)
  const float *box_mask;     // (B,M)
  const int64_t *heading_cls_label, *size_cls_label, *sem_cls_label;  // (B,M)
  const float *heading_res_label;                                     // (B,M)
  const float *size_res_label;                                        // (B,M,3)
  const float *mean_size;                                             // (NS,3)
  int B, K, M, CH, NH, NS, NC;
  float near_thr, far_thr, w0, w1, heading_scale;  // heading_scale = pi / NH
  // outputs
  int64_t *obj_label, *assignment;  // (B,K)
  float *obj_mask;                  // (B,K)
  float *dnet;                      // (B,K,CH) gradient numerators (every element written)
  float *dcenter;                   // (B,K,6): [0:3] obj-weighted pred->gt term, [3:6] gt->pred term
  float *part;                      // (B, NPARTL)
};

__device__ __forceinline__ float sqdist(float ax, float ay, float az, const float *b) {
  const float dx = ax - b[0], dy = ay - b[1], dz = az - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ float huber1(float err, float &grad) {  // delta = 1 (lib/loss_helper.py:20-33)
  const float a = fabsf(err), q = fminf(a, 1.0f), lin = a - q;
  grad = fminf(fmaxf(err, -1.0f), 1.0f);
  return 0.5f * q * q + lin;
}

// cross-entropy over C logits with stride 1; writes (softmax - onehot) * scale into g
__device__ __forceinline__ float ce_grad(const float *s, int C, int label, float scale, float *g) {
  const float s_label = s[label];   // read first: g may alias s (rows staged in LDS are overwritten in place)
  float m = s[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, s[c]);
  float l = 0.f;
  for (int c = 0; c < C; ++c) l += expf(s[c] - m);
  const float lse = m + logf(l);
  for (int c = 0; c < C; ++c) g[c] = (expf(s[c] - lse) - (c == label ? 1.f : 0.f)) * scale;
  return lse - s_label;
}

__device__ float block_sum(float v, float *s_red) {  // 256 threads, fixed order
  const int tid = threadIdx.x;
  v = spacap::wave_sum_f32(v);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// 1 024 threads per scene, FOUR lanes per proposal (round 2 - 5: 256 threads, one per proposal, one wave per SIMD with nothing to
// hide an LDS round trip behind: 49 us on 8 CUs in the middle of the step's critical path).  The three scans over the scene's
// ground-truth boxes are split over the lanes (j = lane, lane + 4, ...) and joined by a lexicographic (distance, index) minimum --
// the first minimum of the serial scan; the tail is split by loss term (lane 0 objectness + centre, 1 heading + size residuals,
// 2 size class, 3 semantic class); the ground truth's nearest proposal (the other direction) takes eight lanes per box; per-box
// labels, mean sizes and vote positions are staged in LDS (no dependent global load in the tail); the twelve partial sums meet at
// ONE barrier.  38 us: the kernel is now bound by VALU issue on its one CU per scene (rocprofv3: 3 740 VALU instructions per wave,
// 16 waves -- the four tail branches diverge inside a wave and the ~98 000 distance evaluations per scene are 11 us of issue by
// themselves); more would take several workgroups per scene.
// STAGED: the scene's K x CH head outputs are copied into LDS with coalesced loads, every lane group works on its own row there
// (rows are CH floats apart: a walk over a row in global memory touches 64 cache lines per load instruction), the gradient
// numerators overwrite the row in place and leave with coalesced stores.
constexpr int DET_THREADS = 1024;

// (d, i) <- the lexicographically smaller of (d, i) and the partner lane's pair: joined over a lane group this is the FIRST
// minimum of the serial scan (NaN distances never win, as with `d < best`)
__device__ __forceinline__ void lexmin_xor(float &d, int &i, int o) {
  const float od = __shfl_xor(d, o);
  const int oi = __shfl_xor(i, o);
  if (od < d || (od == d && oi < i)) d = od, i = oi;
}

template <bool STAGED>
__global__ __launch_bounds__(DET_THREADS) void det_proposal_kernel(const DetArgs A) {
  extern __shared__ float smem[];
  float *s_gt = smem;                 // [M][3]
  float *s_c = s_gt + 3 * A.M;        // [K][3] predicted centres
  int *s_i2 = reinterpret_cast<int *>(s_c + 3 * A.K);  // [M] nearest proposal of every gt box
  float *s_w2 = reinterpret_cast<float *>(s_i2 + A.M); // [M] box mask
  // the per-box labels the tail reads through the assignment (dependent global loads otherwise: two L2 round trips per proposal)
  int *s_hl = reinterpret_cast<int *>(s_w2 + A.M);      // [M] heading class
  int *s_sl = s_hl + A.M;                               // [M] size class
  int *s_se = s_sl + A.M;                               // [M] semantic class
  float *s_hr = reinterpret_cast<float *>(s_se + A.M);  // [M] heading residual
  float *s_sr = s_hr + A.M;                             // [M][3] size residual
  float *s_ms = s_sr + 3 * A.M;                         // [NS][3] mean sizes
  float *s_ag = s_ms + 3 * A.NS;                        // [K][3] aggregated vote positions
  float *s_net = s_ag + 3 * A.K + ((4 - ((12 * A.M + 6 * A.K + 3 * A.NS) & 3)) & 3);   // [K][CH] (STAGED only), 16-byte aligned
  __shared__ float s_red[16][NPARTL];
  const int b = blockIdx.x, tid = threadIdx.x, K = A.K, M = A.M, CH = A.CH;
  if (STAGED) {
    const float *src = A.net + (size_t)b * K * CH;
    if ((((size_t)K * CH) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(s_net) & 15) == 0) {
      for (int i = tid; i < K * CH / 4; i += DET_THREADS)
        reinterpret_cast<float4 *>(s_net)[i] = reinterpret_cast<const float4 *>(src)[i];
    } else {
      for (int i = tid; i < K * CH; i += DET_THREADS) s_net[i] = src[i];
    }
  }
  for (int i = tid; i < 3 * M; i += DET_THREADS) s_gt[i] = A.gt_center[(size_t)b * M * 3 + i];
  for (int i = tid; i < 3 * K; i += DET_THREADS) s_c[i] = A.center[(size_t)b * K * 3 + i];
  for (int i = tid; i < 3 * K; i += DET_THREADS) s_ag[i] = A.agg_xyz[(size_t)b * K * 3 + i];
  for (int i = tid; i < 3 * M; i += DET_THREADS) s_sr[i] = A.size_res_label[(size_t)b * M * 3 + i];
  for (int i = tid; i < 3 * A.NS; i += DET_THREADS) s_ms[i] = A.mean_size[i];
  for (int j = tid; j < M; j += DET_THREADS) {
    const size_t bj = (size_t)b * M + j;
    s_w2[j] = A.box_mask[bj];
    s_hl[j] = (int)A.heading_cls_label[bj], s_sl[j] = (int)A.size_cls_label[bj], s_se[j] = (int)A.sem_cls_label[bj];
    s_hr[j] = A.heading_res_label[bj];
  }
  __syncthreads();
  // gt -> nearest predicted centre (dist2 / idx2 of nn_distance(center, gt_center)): 8 lanes per box
  float num_c2 = 0.f, n_box = 0.f;
  for (int j0 = 0; j0 < M; j0 += DET_THREADS / 8) {
    const int j = j0 + (tid >> 3), l8 = tid & 7;
    const bool vj = j < M;
    const float *gj = s_gt + 3 * (vj ? j : M - 1);
    float best = INFINITY;
    int bi = 0;
    for (int k = l8; k < K; k += 8) {
      const float d = sqdist(s_c[3 * k], s_c[3 * k + 1], s_c[3 * k + 2], gj);
      if (d < best) best = d, bi = k;
    }
    lexmin_xor(best, bi, 1); lexmin_xor(best, bi, 2); lexmin_xor(best, bi, 4);
    if (vj && l8 == 0) {
      s_i2[j] = bi;
      num_c2 += best * s_w2[j];
      n_box += s_w2[j];
    }
  }
  __syncthreads();
  float n_obj = 0.f, n_mask = 0.f, num_obj = 0.f, num_c1 = 0.f, num_hc = 0.f, num_hr = 0.f, num_sc = 0.f, num_sr = 0.f,
        num_sem = 0.f;
  const int o_hs = 5, o_hr = 5 + A.NH, o_ss = 5 + 2 * A.NH, o_sr = o_ss + A.NS, o_sem = o_sr + 3 * A.NS;
  for (int k0 = 0; k0 < K; k0 += DET_THREADS / 4) {
    const int k = min(k0 + (tid >> 2), K - 1), l4 = tid & 3;
    const bool vk = k0 + (tid >> 2) < K;
    const size_t bk = (size_t)b * K + k;
    // objectness label from the aggregated vote position (lib/loss_helper.py:117-133)
    const float ax = s_ag[3 * k], ay = s_ag[3 * k + 1], az = s_ag[3 * k + 2];
    float d1 = INFINITY;
    int i1 = 0;
    for (int j = l4; j < M; j += 4) {
      const float d = sqdist(ax, ay, az, s_gt + 3 * j);
      if (d < d1) d1 = d, i1 = j;
    }
    lexmin_xor(d1, i1, 1); lexmin_xor(d1, i1, 2);
    const float eu = sqrtf(d1 + 1e-6f);
    const int label = eu < A.near_thr ? 1 : 0;
    const float mask = (label || eu > A.far_thr) ? 1.f : 0.f, obj = (float)label;
    // centre: pred -> nearest gt, weighted by objectness
    const float cx = s_c[3 * k], cy = s_c[3 * k + 1], cz = s_c[3 * k + 2];
    float dc = INFINITY;
    int ic = 0;
    for (int j = l4; j < M; j += 4) {
      const float d = sqdist(cx, cy, cz, s_gt + 3 * j);
      if (d < dc) dc = d, ic = j;
    }
    lexmin_xor(dc, ic, 1); lexmin_xor(dc, ic, 2);
    float gx = 0.f, gy = 0.f, gz = 0.f;  // gt -> pred term: every gt box whose nearest proposal is k
    for (int j = l4; j < M; j += 4)
      if (s_i2[j] == k) {
        gx += 2.f * (cx - s_gt[3 * j]) * s_w2[j];
        gy += 2.f * (cy - s_gt[3 * j + 1]) * s_w2[j];
        gz += 2.f * (cz - s_gt[3 * j + 2]) * s_w2[j];
      }
    gx += __shfl_xor(gx, 1); gy += __shfl_xor(gy, 1); gz += __shfl_xor(gz, 1);
    gx += __shfl_xor(gx, 2); gy += __shfl_xor(gy, 2); gz += __shfl_xor(gz, 2);
    if (!vk) continue;   // (after the lane-group exchanges)
    const float *row = STAGED ? s_net + (size_t)k * CH : A.net + bk * CH;
    float *g = STAGED ? s_net + (size_t)k * CH : A.dnet + bk * CH;
    if (l4 == 0) {
      A.obj_label[bk] = label;
      A.obj_mask[bk] = mask;
      A.assignment[bk] = i1;
      n_obj += obj;
      n_mask += mask;
      // objectness: weighted 2-class cross-entropy
      num_obj += ce_grad(row, 2, label, (label ? A.w1 : A.w0) * mask, g) * (label ? A.w1 : A.w0) * mask;
      g[2] = g[3] = g[4] = 0.f;  // centre offsets: their gradient arrives through `center`
      num_c1 += dc * obj;
      float *dcn = A.dcenter + bk * 6;
      dcn[0] = 2.f * (cx - s_gt[3 * ic]) * obj;
      dcn[1] = 2.f * (cy - s_gt[3 * ic + 1]) * obj;
      dcn[2] = 2.f * (cz - s_gt[3 * ic + 2]) * obj;
      dcn[3] = gx, dcn[4] = gy, dcn[5] = gz;
    } else if (l4 == 1) {
      // heading
      const int hl = s_hl[i1];
      float hg;
      const float hv = huber1(row[o_hr + hl] - s_hr[i1] / A.heading_scale, hg);   // (g may alias row: read first)
      num_hc += ce_grad(row + o_hs, A.NH, hl, obj, g + o_hs) * obj;
      num_hr += hv * obj;
      for (int c = 0; c < A.NH; ++c) g[o_hr + c] = c == hl ? hg * obj : 0.f;
      // size residuals
      const int sl = s_sl[i1];
      float sr = 0.f;
      const float res[3] = {row[o_sr + 3 * sl], row[o_sr + 3 * sl + 1], row[o_sr + 3 * sl + 2]};   // (g may alias row)
      for (int c = 0; c < 3 * A.NS; ++c) g[o_sr + c] = 0.f;
      for (int d = 0; d < 3; ++d) {
        float sg;
        sr += huber1(res[d] - s_sr[3 * i1 + d] / s_ms[3 * sl + d], sg);
        g[o_sr + 3 * sl + d] = sg * obj / 3.0f;
      }
      num_sr += sr / 3.0f * obj;
    } else if (l4 == 2) {
      // size class
      num_sc += ce_grad(row + o_ss, A.NS, s_sl[i1], obj, g + o_ss) * obj;
    } else {
      // semantic class
      num_sem += ce_grad(row + o_sem, A.NC, s_se[i1], obj, g + o_sem) * obj;
    }
  }
  // the twelve partial sums of the scene: per wave by shuffles, the 16 waves' rows through LDS, one barrier, added in wave order
  const float v[NPARTL] = {0.f, num_obj, num_c1, num_c2, num_hc, num_hr, num_sc, num_sr, num_sem, n_obj, n_mask, n_box};
#pragma unroll
  for (int i = 1; i < NPARTL; ++i) {
    const float w = spacap::wave_sum_f32(v[i]);
    if ((tid & 63) == 0) s_red[tid >> 6][i] = w;
  }
  __syncthreads();   // (also: every row of s_net holds its gradient numerators)
  if (STAGED) {
    float *dst = A.dnet + (size_t)b * K * CH;
    if ((((size_t)K * CH) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(s_net) & 15) == 0) {
      for (int i = tid; i < K * CH / 4; i += DET_THREADS)
        reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(s_net)[i];
    } else {
      for (int i = tid; i < K * CH; i += DET_THREADS) dst[i] = s_net[i];
    }
  }
  if (tid >= 1 && tid < NPARTL) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) a += s_red[w][tid];
    A.part[(size_t)b * NPARTL + tid] = a;
  }
}

// vote loss (lib/loss_helper.py:35-114): per seed the smallest L1 distance between its vote and its 3 ground-truth votes
__global__ __launch_bounds__(256) void det_vote_kernel(const float *__restrict__ seed_xyz, const float *__restrict__ vote_xyz,
                                                       const int32_t *__restrict__ seed_inds,
                                                       const float *__restrict__ vote_label,
                                                       const int64_t *__restrict__ vote_mask, int NS, int N,
                                                       float *__restrict__ dvote, float *__restrict__ part) {
  __shared__ float s_red[4];
  const int b = blockIdx.x;
  float num = 0.f, cnt = 0.f;
  for (int s = threadIdx.x; s < NS; s += 256) {
    const size_t bs = (size_t)b * NS + s;
    const int p = seed_inds[bs];
    const float m = (float)vote_mask[(size_t)b * N + p];
    const float *vl = vote_label + ((size_t)b * N + p) * 9, *sx = seed_xyz + bs * 3, *vx = vote_xyz + bs * 3;
    float best = INFINITY;
    int bi = 0;
    for (int v = 0; v < 3; ++v) {
      const float d = (fabsf(vx[0] - (vl[3 * v] + sx[0])) + fabsf(vx[1] - (vl[3 * v + 1] + sx[1]))) +
                      fabsf(vx[2] - (vl[3 * v + 2] + sx[2]));
      if (d < best) best = d, bi = v;
    }
    num += best * m;
    cnt += m;
    for (int d = 0; d < 3; ++d) {
      const float e = vx[d] - (vl[3 * bi + d] + sx[d]);
      dvote[bs * 3 + d] = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * m;
    }
  }
  const float a = block_sum(num, s_red), c = block_sum(cnt, s_red);
  if (threadIdx.x == 0) part[(size_t)b * 2] = a, part[(size_t)b * 2 + 1] = c;
}

// losses[8], inv_den[4] from the per-scene partial sums (ascending scene order)
__global__ void det_finalize_kernel(const float *__restrict__ part, const float *__restrict__ vpart, int B,
                                    float *__restrict__ losses, float *__restrict__ inv_den) {
  if (threadIdx.x != 0) return;
  float s[NPARTL] = {0.f}, vn = 0.f, vc = 0.f;
  for (int b = 0; b < B; ++b) {
    for (int i = 1; i < NPARTL; ++i) s[i] += part[(size_t)b * NPARTL + i];
    vn += vpart[2 * b], vc += vpart[2 * b + 1];
  }
  const float d_vote = vc + 1e-6f, d_mask = s[10] + 1e-6f, d_obj = s[9] + 1e-6f, d_box = s[11] + 1e-6f;
  losses[0] = vn / d_vote;
  losses[1] = s[1] / d_mask;
  losses[2] = s[2] / d_obj + s[3] / d_box;
  losses[3] = s[4] / d_obj;
  losses[4] = s[5] / d_obj;
  losses[5] = s[6] / d_obj;
  losses[6] = s[7] / d_obj;
  losses[7] = s[8] / d_obj;
  inv_den[0] = 1.f / d_vote, inv_den[1] = 1.f / d_mask, inv_den[2] = 1.f / d_obj, inv_den[3] = 1.f / d_box;
}

// gradients = numerators x upstream gradient of the owning loss x inverse denominator
__global__ __launch_bounds__(256) void det_scale_kernel(const float *__restrict__ dnet_n, const float *__restrict__ dcen_n,
                                                        const float *__restrict__ dvote_n, const float *__restrict__ gout,
                                                        const float *__restrict__ inv_den, long rows, int CH, int NH, int NS,
                                                        long nvote, float *__restrict__ dnet, float *__restrict__ dcenter,
                                                        float *__restrict__ dvote) {
  const float g_obj = gout[1] * inv_den[1], g_hc = gout[3] * inv_den[2], g_hr = gout[4] * inv_den[2],
              g_sc = gout[5] * inv_den[2], g_sr = gout[6] * inv_den[2], g_sem = gout[7] * inv_den[2];
  const float g_c1 = gout[2] * inv_den[2], g_c2 = gout[2] * inv_den[3], g_v = gout[0] * inv_den[0];
  const int o_hr = 5 + NH, o_ss = 5 + 2 * NH, o_sr = o_ss + NS, o_sem = o_sr + 3 * NS;
  const long n_net = rows * CH, n_cen = rows * 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_net + n_cen + nvote; i += (long)gridDim.x * 256) {
    if (i < n_net) {
      const int c = (int)(i % CH);
      const float s = c < 2 ? g_obj : c < 5 ? 0.f : c < o_hr ? g_hc : c < o_ss ? g_hr : c < o_sr ? g_sc : c < o_sem ? g_sr : g_sem;
      dnet[i] = dnet_n[i] * s;
    } else if (i < n_net + n_cen) {
      const long j = i - n_net, r = j / 3;
      const int d = (int)(j % 3);
      dcenter[j] = dcen_n[r * 6 + d] * g_c1 + dcen_n[r * 6 + 3 + d] * g_c2;
    } else {
      const long j = i - n_net - n_cen;
      dvote[j] = dvote_n[j] * g_v;
    }
  }
}


// ---- relation loss (lib/loss_helper.py:240-289): 3-way cross-entropy per axis on the selected proposal pairs -----
// pred (B,K,K,9) = [x 3 | y 3 | z 3]; pair (i, j) counts iff both proposals are positive and assigned to valid boxes;
// label_a = rel_a[b, oa_i, oa_j].  One launch: losses / accuracies (per-workgroup partial sums) and the unnormalised
// gradient W * (softmax - onehot); rel_finalize_kernel adds the partials in order.
constexpr int REL_THREADS = 256;

__global__ __launch_bounds__(REL_THREADS) void rel_loss_kernel(const float *__restrict__ pred, const int64_t *__restrict__ oa,
                                                              const int64_t *__restrict__ box_mask_int,
                                                              const int64_t *__restrict__ obj_label,
                                                              const int64_t *__restrict__ xl, const int64_t *__restrict__ yl,
                                                              const int64_t *__restrict__ zl, int K, int M,
                                                              float *__restrict__ dnum, float *__restrict__ part) {
  __shared__ float s_red[4];
  const int b = blockIdx.y;
  const long pair = (long)blockIdx.x * REL_THREADS + threadIdx.x;
  float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // loss x,y,z, acc x,y,z, sum W
  if (pair < (long)K * K) {
    const int i = (int)(pair / K), j = (int)(pair - (long)i * K);
    const int oi = (int)oa[(size_t)b * K + i], oj = (int)oa[(size_t)b * K + j];
    const bool si = (box_mask_int[(size_t)b * M + oi] & obj_label[(size_t)b * K + i]) != 0;
    const bool sj = (box_mask_int[(size_t)b * M + oj] & obj_label[(size_t)b * K + j]) != 0;
    const float W = (si && sj) ? 1.f : 0.f;
    const size_t o = ((size_t)b * K * K + pair) * 9;
    const size_t lo = ((size_t)b * M + oi) * M + oj;
    const int lab[3] = {(int)xl[lo], (int)yl[lo], (int)zl[lo]};
    v[6] = W;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float s0 = pred[o + 3 * a], s1 = pred[o + 3 * a + 1], s2 = pred[o + 3 * a + 2];
      const float m = fmaxf(s0, fmaxf(s1, s2));
      const float e0 = expf(s0 - m), e1 = expf(s1 - m), e2 = expf(s2 - m);
      const float lse = m + logf((e0 + e1) + e2);
      const float sl = lab[a] == 0 ? s0 : (lab[a] == 1 ? s1 : s2);
      const int am = (s0 >= s1 && s0 >= s2) ? 0 : (s1 >= s2 ? 1 : 2);  // first maximum
      v[a] = (lse - sl) * W;
      v[3 + a] = (am == lab[a] ? 1.f : 0.f) * W;
      dnum[o + 3 * a] = (expf(s0 - lse) - (lab[a] == 0 ? 1.f : 0.f)) * W;
      dnum[o + 3 * a + 1] = (expf(s1 - lse) - (lab[a] == 1 ? 1.f : 0.f)) * W;
      dnum[o + 3 * a + 2] = (expf(s2 - lse) - (lab[a] == 2 ? 1.f : 0.f)) * W;
    }
  }
  float *p = part + ((size_t)b * gridDim.x + blockIdx.x) * 7;
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const float t = block_sum(v[q], s_red);
    if (threadIdx.x == 0) p[q] = t;
  }
}

// out[0..2] losses x,y,z; out[3..5] accuracies; out[6] = 1 / n
__global__ __launch_bounds__(256) void rel_finalize_kernel(const float *__restrict__ part, int nparts, float *__restrict__ out) {
  __shared__ float s_red[4];
  float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int p = threadIdx.x; p < nparts; p += 256)
#pragma unroll
    for (int q = 0; q < 7; ++q) v[q] += part[(size_t)p * 7 + q];
  float t[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) t[q] = block_sum(v[q], s_red);
  if (threadIdx.x == 0) {
    const float n = fmaxf(t[6], 1.0f);
#pragma unroll
    for (int q = 0; q < 6; ++q) out[q] = t[q] / n;
    out[6] = 1.0f / n;
  }
}

__global__ __launch_bounds__(256) void rel_scale_kernel(const float *__restrict__ dnum, const float *__restrict__ gout,
                                                        const float *__restrict__ out, long n, float *__restrict__ dpred) {
  const float inv = out[6], g0 = gout[0] * inv, g1 = gout[1] * inv, g2 = gout[2] * inv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % 9);
    dpred[i] = dnum[i] * (c < 3 ? g0 : (c < 6 ? g1 : g2));
  }
}

}  // namespace

extern "C" long spacap_rel_loss_nparts(int B, int K) { return (long)B * (((long)K * K + REL_THREADS - 1) / REL_THREADS); }

extern "C" int spacap_rel_loss_fwd_f32(const float *pred, const int64_t *assignment, const int64_t *box_mask_int,
                                       const int64_t *obj_label, const int64_t *x_label, const int64_t *y_label,
                                       const int64_t *z_label, int B, int K, int M, float *dnum, float *part, float *out,
                                       spacap_stream_t stream) {
  const char *what = "spacap_rel_loss_fwd_f32";
  SPACAP_REQUIRE(B >= 1 && K >= 1 && M >= 1 && B <= 65535, "%s: bad sizes", what);
  SPACAP_REQUIRE(pred && assignment && box_mask_int && obj_label && x_label && y_label && z_label && dnum && part && out,
                 "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
  const unsigned gx = (unsigned)(((long)K * K + REL_THREADS - 1) / REL_THREADS);
  hipLaunchKernelGGL(rel_loss_kernel, dim3(gx, B), dim3(REL_THREADS), 0, s, pred, assignment, box_mask_int, obj_label, x_label,
                     y_label, z_label, K, M, dnum, part);
  hipLaunchKernelGGL(rel_finalize_kernel, dim3(1), dim3(256), 0, s, part, (int)(gx * B), out);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_rel_loss_bwd_f32(const float *dnum, const float *grad_losses, const float *out, int B, int K,
                                       float *dpred, spacap_stream_t stream) {
  const char *what = "spacap_rel_loss_bwd_f32";
  SPACAP_REQUIRE(dnum && grad_losses && out && dpred && B >= 1 && K >= 1, "%s: bad arguments", what);
  const long n = (long)B * K * K * 9;
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(rel_scale_kernel, dim3((unsigned)g), dim3(256), 0, spacap::as_stream(stream), dnum, grad_losses, out, n, dpred);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_det_npart(void) { return NPARTL; }

extern "C" int spacap_det_losses_fwd_f32(
    const float *net, const float *center, const float *agg_xyz, const float *gt_center, const float *box_mask,
    const int64_t *heading_cls_label, const float *heading_res_label, const int64_t *size_cls_label,
    const float *size_res_label, const int64_t *sem_cls_label, const float *mean_size, const float *seed_xyz,
    const float *vote_xyz, const int32_t *seed_inds, const float *vote_label, const int64_t *vote_mask, int B, int K,
    int M, int NSEED, int N, int NH, int NS, int NC, float near_thr, float far_thr, float w0, float w1,
    int64_t *obj_label, float *obj_mask, int64_t *assignment, float *dnet_num, float *dcenter_num, float *dvote_num,
    float *part, float *losses, float *inv_den, spacap_stream_t stream) {
  const char *what = "spacap_det_losses_fwd_f32";
  SPACAP_REQUIRE(B >= 1 && K >= 1 && M >= 1 && M <= MAXM && NSEED >= 1 && NH >= 1 && NS >= 1 && NC >= 1 && B <= 65535,
                 "%s: bad sizes", what);
  SPACAP_REQUIRE(net && center && agg_xyz && gt_center && box_mask && heading_cls_label && heading_res_label &&
                     size_cls_label && size_res_label && sem_cls_label && mean_size && seed_xyz && vote_xyz && seed_inds &&
                     vote_label && vote_mask && obj_label && obj_mask && assignment && dnet_num && dcenter_num &&
                     dvote_num && part && losses && inv_den, "%s: null pointer", what);
  DetArgs A;
  A.net = net; A.center = center; A.agg_xyz = agg_xyz; A.gt_center = gt_center; A.box_mask = box_mask;
  A.heading_cls_label = heading_cls_label; A.size_cls_label = size_cls_label; A.sem_cls_label = sem_cls_label;
  A.heading_res_label = heading_res_label; A.size_res_label = size_res_label; A.mean_size = mean_size;
  A.B = B; A.K = K; A.M = M; A.NH = NH; A.NS = NS; A.NC = NC; A.CH = 5 + 2 * NH + 4 * NS + NC;
  A.near_thr = near_thr; A.far_thr = far_thr; A.w0 = w0; A.w1 = w1; A.heading_scale = (float)(M_PI / NH);
  A.obj_label = obj_label; A.assignment = assignment; A.obj_mask = obj_mask; A.dnet = dnet_num; A.dcenter = dcenter_num;
  A.part = part;
  hipStream_t s = spacap::as_stream(stream);
  // [M][3] gt | [K][3] centres | [M] nearest | [M] mask | 3 x [M] class labels | [M] + [M][3] residual labels | [NS][3] | [K][3] (+ pad)
  const size_t lds = sizeof(float) * (12 * (size_t)M + 6 * (size_t)K + 3 * (size_t)NS + 4);
  SPACAP_REQUIRE(lds <= 60000, "%s: K=%d too large", what, K);
  const size_t lds_staged = lds + sizeof(float) * (size_t)K * A.CH;
  if (lds_staged <= 150 * 1024) {
    static unsigned long long lds_ok = 0;
    SPACAP_CHECK_HIP(spacap::allow_dynamic_lds(reinterpret_cast<const void *>(&det_proposal_kernel<true>), 150 * 1024, lds_ok), what);
    hipLaunchKernelGGL(det_proposal_kernel<true>, dim3(B), dim3(DET_THREADS), lds_staged, s, A);
  } else {
    hipLaunchKernelGGL(det_proposal_kernel<false>, dim3(B), dim3(DET_THREADS), lds, s, A);
  }
  float *vpart = part + (size_t)B * NPARTL;
  hipLaunchKernelGGL(det_vote_kernel, dim3(B), dim3(256), 0, s, seed_xyz, vote_xyz, seed_inds, vote_label, vote_mask, NSEED, N,
                     dvote_num, vpart);
  hipLaunchKernelGGL(det_finalize_kernel, dim3(1), dim3(64), 0, s, part, vpart, B, losses, inv_den);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_det_losses_bwd_f32(const float *dnet_num, const float *dcenter_num, const float *dvote_num,
                                         const float *grad_losses, const float *inv_den, int B, int K, int NSEED, int NH,
                                         int NS, int NC, float *dnet, float *dcenter, float *dvote,
                                         spacap_stream_t stream) {
  const char *what = "spacap_det_losses_bwd_f32";
  SPACAP_REQUIRE(dnet_num && dcenter_num && dvote_num && grad_losses && inv_den && dnet && dcenter && dvote && B >= 1 && K >= 1,
                 "%s: bad arguments", what);
  const int CH = 5 + 2 * NH + 4 * NS + NC;
  const long rows = (long)B * K, nvote = (long)B * NSEED * 3, total = rows * CH + rows * 3 + nvote;
  long g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(det_scale_kernel, dim3((unsigned)g), dim3(256), 0, spacap::as_stream(stream), dnet_num, dcenter_num,
                     dvote_num, grad_losses, inv_den, rows, CH, NH, NS, nvote, dnet, dcenter, dvote);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ---- caption head: log-softmax over the vocabulary + masked cross entropy + accuracy ------------------------------------
// Replaces Generator.forward's F.log_softmax (models/transformer_captioner.py:93-99) and compute_cap_loss
// (lib/loss_helper.py:199-238: F.cross_entropy(ignore_index=0, reduction="none") on the log-probabilities -- a second,
// idempotent log-softmax -- masked by good_bbox_masks, summed and divided by (#good words + 1e-6); accuracy = arg-max hits
// over the non-pad words of good boxes): ~25 PyTorch launches forward + backward -> 2 + 1.
//   cap_rows_kernel   one workgroup per (scene, word) row of V logits: max, log-sum-exp, writes the log-probabilities
//                     (data_dict["lang_cap"]), per row (loss term, hit, valid);
//   cap_final_kernel  fixed-order sums over the rows -> out[0..3] = (cap_loss, cap_acc, 1 / (sum good + 1e-6), -)
//   cap_bwd_kernel    dlogits[r, v] = w_r (softmax - onehot) with w_r = g * good_r [target_r != 0] / (sum good + 1e-6)
//                     (+ the gradient that arrives on lang_cap itself, if any, through the log-softmax Jacobian: not used
//                     by the training loss, so it is not supported here and the Python side refuses it).
namespace {
constexpr int CAP_T = 256;
__global__ __launch_bounds__(CAP_T) void cap_rows_kernel(const float *__restrict__ logits, const int64_t *__restrict__ target,
                                                        const uint8_t *__restrict__ good, int W, int V, int tstride,
                                                        float *__restrict__ logp, float *__restrict__ rowstat) {
  __shared__ float s_f[CAP_T / 64];
  __shared__ int s_i[CAP_T / 64];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float *x = logits + (size_t)r * V;
  float m = -INFINITY;
  int am = 0;
  for (int v = tid; v < V; v += CAP_T) {
    const float a = x[v];
    if (a > m) m = a, am = v;        // first maximum per thread (ascending v)
  }
  // block arg-max with the lowest index among equal maxima (torch.argmax returns the first)
  float wm = spacap::wave_max_f32(m);
  int cand = (m == wm) ? am : 0x7fffffff;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
  if (lane == 0) s_f[wv] = wm, s_i[wv] = cand;
  __syncthreads();
  float M = s_f[0];
  for (int i = 1; i < CAP_T / 64; ++i) M = fmaxf(M, s_f[i]);
  int AM = 0x7fffffff;
  for (int i = 0; i < CAP_T / 64; ++i)
    if (s_f[i] == M) AM = min(AM, s_i[i]);
  __syncthreads();
  float s = 0.f;
  for (int v = tid; v < V; v += CAP_T) s += expf(x[v] - M);
  s = spacap::wave_sum_f32(s);
  if (lane == 0) s_f[wv] = s;
  __syncthreads();
  float S = 0.f;
  for (int i = 0; i < CAP_T / 64; ++i) S += s_f[i];
  const float lse = M + logf(S);
  for (int v = tid; v < V; v += CAP_T) logp[(size_t)r * V + v] = x[v] - lse;
  if (tid == 0) {
    const int b = r / W, wd = r % W;
    int64_t t = target[(size_t)b * tstride + wd];
    if ((uint64_t)t >= (uint64_t)V) t = 0;                         // out-of-range id: ignored like the pad id (never read out of bounds)
    const float gd = good[b] ? 1.f : 0.f;
    const float lt = (t != 0) ? -(x[t] - lse) : 0.f;              // ignore_index = 0
    rowstat[r * 4 + 0] = lt * gd;
    rowstat[r * 4 + 1] = gd;
    rowstat[r * 4 + 2] = (AM == (int)t) ? ((t != 0) ? gd : 0.f) : 0.f;   // hit * valid
    rowstat[r * 4 + 3] = (t != 0) ? gd : 0.f;                          // valid
  }
}
__global__ __launch_bounds__(64) void cap_final_kernel(const float *__restrict__ rowstat, int rows, float *__restrict__ out) {
  // one wavefront: lane l adds rows l, l + 64, ... in order, then a fixed shuffle tree (bitwise reproducible)
  float a = 0.f, g = 0.f, h = 0.f, v = 0.f;
  for (int r = threadIdx.x; r < rows; r += 64) a += rowstat[r * 4], g += rowstat[r * 4 + 1], h += rowstat[r * 4 + 2], v += rowstat[r * 4 + 3];
  a = spacap::wave_sum_f32(a), g = spacap::wave_sum_f32(g), h = spacap::wave_sum_f32(h), v = spacap::wave_sum_f32(v);
  if (threadIdx.x != 0) return;
  const float inv = 1.0f / (g + 1e-6f);
  out[0] = a * inv;
  out[1] = h / fmaxf(v, 1.0f);
  out[2] = inv;
  out[3] = g;
}
__global__ __launch_bounds__(CAP_T) void cap_bwd_kernel(const float *__restrict__ logp, const int64_t *__restrict__ target,
                                                       const uint8_t *__restrict__ good, const float *__restrict__ out,
                                                       const float *__restrict__ gloss, int W, int V, int tstride,
                                                       float *__restrict__ dlogits) {
  const int r = blockIdx.x, tid = threadIdx.x;
  const int b = r / W, wd = r % W;
  int64_t t = target[(size_t)b * tstride + wd];
  if ((uint64_t)t >= (uint64_t)V) t = 0;
  const float w = (good[b] && t != 0) ? gloss[0] * out[2] : 0.f;
  for (int v = tid; v < V; v += CAP_T) {
    const float p = expf(logp[(size_t)r * V + v]);
    dlogits[(size_t)r * V + v] = w * (p - (v == (int)t ? 1.f : 0.f));
  }
}
}  // namespace

extern "C" int spacap_cap_loss_fwd_f32(const float *logits, const int64_t *target, const uint8_t *good, int B, int W, int V,
                                       int tstride, float *logp, float *rowstat, float *out, spacap_stream_t stream) {
  const char *what = "spacap_cap_loss_fwd_f32";
  SPACAP_REQUIRE(B >= 0 && W >= 1 && V >= 2 && tstride >= W, "%s: bad sizes", what);
  if (B == 0) {   // no rows: loss 0, accuracy 0, 1 / (0 + 1e-6), 0 good boxes
    if (out) SPACAP_CHECK_HIP(hipMemsetAsync(out, 0, 4 * sizeof(float), spacap::as_stream(stream)), what);
    return SPACAP_OK;
  }
  SPACAP_REQUIRE(logits && target && good && logp && rowstat && out, "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
  hipLaunchKernelGGL(cap_rows_kernel, dim3(B * W), dim3(CAP_T), 0, s, logits, target, good, W, V, tstride, logp, rowstat);
  hipLaunchKernelGGL(cap_final_kernel, dim3(1), dim3(64), 0, s, rowstat, B * W, out);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
extern "C" int spacap_cap_loss_bwd_f32(const float *logp, const int64_t *target, const uint8_t *good, const float *out,
                                       const float *gloss, int B, int W, int V, int tstride, float *dlogits,
                                       spacap_stream_t stream) {
  const char *what = "spacap_cap_loss_bwd_f32";
  SPACAP_REQUIRE(B >= 0 && W >= 1 && V >= 2 && tstride >= W, "%s: bad sizes", what);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(logp && target && good && out && gloss && dlogits, "%s: null pointer", what);
  hipLaunchKernelGGL(cap_bwd_kernel, dim3(B * W), dim3(CAP_T), 0, spacap::as_stream(stream), logp, target, good, out, gloss, W, V,
                     tstride, dlogits);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ---- the tail of get_scene_cap_loss (lib/loss_helper.py:340-383): every derived scalar of a training step in ONE launch ----
//   det  f32 [8]  = (vote, objectness, center, heading_cls, heading_reg, size_cls, size_reg, sem_cls) loss
//   cap  f32 [4]  = (cap_loss, cap_acc, ., .)          rel f32 [7] = (x, y, z loss, x, y, z acc, .) or NULL (no relation head)
//   out  f32 [8]  = (box_loss, det_loss, relation_loss, loss, pos_ratio, neg_ratio, obj_acc, 0)
//       box = center + 0.1 heading_cls + heading_reg + 0.1 size_cls + size_reg;  det = vote + 0.5 objectness + box + 0.1 sem_cls;
//       relation = x + y + z;  loss = 10 det + cap + 0.1 relation   (:373-383);  the three ratios as :355-362.
// PyTorch composes these from ~25 scalar kernels forward and as many backward (stack / unbind / gemv / fill / add on 0-d tensors).
namespace {
__global__ __launch_bounds__(256) void loss_tail_fwd_kernel(const float *__restrict__ det, const float *__restrict__ cap,
                                                            const float *__restrict__ rel, const int64_t *__restrict__ obj_label,
                                                            const float *__restrict__ obj_mask, const int64_t *__restrict__ bbox_mask,
                                                            int n, float *__restrict__ out, float *__restrict__ loss) {
  __shared__ float s_red[4];
  float pos = 0.f, msk = 0.f, hit = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float m = obj_mask[i];
    pos += (float)obj_label[i];
    msk += m;
    hit += (bbox_mask[i] == obj_label[i]) ? m : 0.f;
  }
  pos = block_sum(pos, s_red), msk = block_sum(msk, s_red), hit = block_sum(hit, s_red);
  if (threadIdx.x == 0) {
    const float box = det[2] + 0.1f * det[3] + det[4] + 0.1f * det[5] + det[6];
    const float dl = det[0] + 0.5f * det[1] + box + 0.1f * det[7];
    const float rl = rel ? (rel[0] + rel[1]) + rel[2] : 0.f;
    out[0] = box, out[1] = dl, out[2] = rl, out[3] = 10.f * dl + cap[0] + 0.1f * rl;
    out[4] = pos / (float)n, out[5] = msk / (float)n - pos / (float)n, out[6] = hit / (msk + 1e-6f), out[7] = 0.f;
    *loss = out[3];
  }
}
// gradient of the total loss (g_loss[0]) w.r.t. the 8 + 1 + 3 terms
__global__ void loss_tail_bwd_kernel(const float *__restrict__ g_loss, float *__restrict__ gdet, float *__restrict__ gcap,
                                     float *__restrict__ grel) {
  const int t = threadIdx.x;
  const float gl = g_loss[0];
  const float d_det = 10.f * gl, d_box = d_det, d_rel = 0.1f * gl;
  if (t < 8) {
    const float wbox[8] = {0.f, 0.f, 1.f, 0.1f, 1.f, 0.1f, 1.f, 0.f}, wdet[8] = {1.f, 0.5f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.1f};
    gdet[t] = d_box * wbox[t] + d_det * wdet[t];
  } else if (t < 12) {
    gcap[t - 8] = t == 8 ? gl : 0.f;
  } else if (t < 19 && grel) {
    grel[t - 12] = t < 15 ? d_rel : 0.f;
  }
}
}  // namespace

extern "C" int spacap_loss_tail_fwd_f32(const float *det, const float *cap, const float *rel, const int64_t *obj_label,
                                        const float *obj_mask, const int64_t *bbox_mask, int n, float *out, float *loss,
                                        spacap_stream_t stream) {
  const char *what = "spacap_loss_tail_fwd_f32";
  SPACAP_REQUIRE(det && cap && obj_label && obj_mask && bbox_mask && out && loss && n >= 1, "%s: bad arguments", what);
  hipLaunchKernelGGL(loss_tail_fwd_kernel, dim3(1), dim3(256), 0, spacap::as_stream(stream), det, cap, rel, obj_label, obj_mask,
                     bbox_mask, n, out, loss);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
extern "C" int spacap_loss_tail_bwd_f32(const float *g_loss, float *g_det, float *g_cap, float *g_rel, spacap_stream_t stream) {
  const char *what = "spacap_loss_tail_bwd_f32";
  SPACAP_REQUIRE(g_loss && g_det && g_cap, "%s: null pointer", what);
  hipLaunchKernelGGL(loss_tail_bwd_kernel, dim3(1), dim3(64), 0, spacap::as_stream(stream), g_loss, g_det, g_cap, g_rel);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
