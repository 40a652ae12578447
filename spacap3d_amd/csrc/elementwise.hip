// Fused elementwise pieces of the Transformer sublayers for gfx950 (MI355X).
//
//   relu_dropout :  y = dropout(relu(x))            models/transformer_captioner.py:126  (feed-forward hidden layer)
//   dropout_add  :  out = res + dropout(y)          models/transformer_captioner.py:115-123 (SublayerConnection)
//
// PyTorch runs each as two launches forward (and keeps a byte mask for backward); here one launch each way.  The
// keep mask is a counter hash of (seed, element index) -- regenerated in the backward instead of stored -- with the
// same two-part seed as the attention dropout (host word per call + device-resident step counter, so a replayed
// hipGraph draws new masks every step).  Dropout semantics as torch.nn.Dropout: keep with probability 1 - p,
// scale kept values by 1 / (1 - p).
#include "common.hpp"

namespace {

using f32x4 = float __attribute__((ext_vector_type(4)));

struct DropSeed {
  unsigned lo, hi;
};

__device__ __forceinline__ DropSeed make_seed(unsigned long long seed, const unsigned long long *seed_dev) {
  const unsigned long long s = seed + (seed_dev ? *seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
  return DropSeed{(unsigned)s, (unsigned)(s >> 32)};
}

// 32-bit finaliser (murmur3 fmix32) over the element index mixed with both seed words
__device__ __forceinline__ unsigned hash32(unsigned long long idx, DropSeed s) {
  unsigned h = (unsigned)idx ^ s.lo;
  h += ((unsigned)(idx >> 32) ^ s.hi) * 0x9E3779B1u;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

template <int MODE>  // 0: y = drop(relu(x)); 1: out = res + drop(x); 2: dx = keep ? g * scale : 0
__global__ __launch_bounds__(256) void drop_kernel(const float *__restrict__ x, const float *__restrict__ res, long n,
                                                   unsigned thresh, float scale, unsigned long long seed,
                                                   const unsigned long long *__restrict__ seed_dev, float *__restrict__ out) {
  const DropSeed sd = make_seed(seed, seed_dev);
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v = *reinterpret_cast<const f32x4 *>(x + 4 * i);
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) r = *reinterpret_cast<const f32x4 *>(res + 4 * i);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool keep = thresh == 0u || hash32((unsigned long long)(4 * i + u), sd) >= thresh;
      float a = MODE == 0 ? fmaxf(v[u], 0.f) : v[u];
      a = keep ? a * scale : 0.f;
      v[u] = MODE == 1 ? r[u] + a : a;
    }
    *reinterpret_cast<f32x4 *>(out + 4 * i) = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail
    const long i = (n4 << 2) + threadIdx.x;
    const bool keep = thresh == 0u || hash32((unsigned long long)i, sd) >= thresh;
    float a = MODE == 0 ? fmaxf(x[i], 0.f) : x[i];
    a = keep ? a * scale : 0.f;
    out[i] = MODE == 1 ? res[i] + a : a;
  }
}

// dx = (y > 0) ? g * scale : 0      (y = the saved output of relu_dropout: positive exactly where kept and x > 0)
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const float *__restrict__ g, const float *__restrict__ y,
                                                               long n, float scale, float *__restrict__ dx) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + 4 * i), yv = *reinterpret_cast<const f32x4 *>(y + 4 * i);
    f32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = yv[u] > 0.f ? gv[u] * scale : 0.f;
    *reinterpret_cast<f32x4 *>(dx + 4 * i) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    dx[i] = y[i] > 0.f ? g[i] * scale : 0.f;
  }
}

inline unsigned grid_for(long n) {
  long g = ((n >> 2) + 255) / 256;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return (unsigned)g;
}

inline bool drop_params(float p, unsigned &thresh, float &scale) {
  if (!(p >= 0.f && p < 1.f)) return false;
  thresh = p > 0.f ? (unsigned)((double)p * 4294967296.0) : 0u;
  scale = 1.0f / (1.0f - p);
  return true;
}

}  // namespace

#define DROP_ENTRY(NAME, MODE, XARG, RESARG)                                                                         \
  unsigned thresh;                                                                                                   \
  float scale;                                                                                                       \
  SPACAP_REQUIRE(n >= 0 && drop_params(p, thresh, scale), NAME ": bad arguments (n=%ld, p=%f)", n, (double)p);      \
  if (n == 0) return SPACAP_OK;                                                                                      \
  SPACAP_REQUIRE(XARG && out && ((reinterpret_cast<uintptr_t>(XARG) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, \
                 NAME ": null or unaligned pointer");                                                               \
  hipLaunchKernelGGL((drop_kernel<MODE>), dim3(grid_for(n)), dim3(256), 0, spacap::as_stream(stream), XARG, RESARG,  \
                     n, thresh, scale, (unsigned long long)seed, (const unsigned long long *)seed_dev, out);         \
  SPACAP_CHECK_LAUNCH(NAME);                                                                                         \
  return SPACAP_OK;

extern "C" int spacap_relu_dropout_fwd_f32(const float *x, long n, float p, uint64_t seed, const uint64_t *seed_dev,
                                           float *out, spacap_stream_t stream) {
  DROP_ENTRY("spacap_relu_dropout_fwd_f32", 0, x, (const float *)nullptr)
}

extern "C" int spacap_relu_dropout_bwd_f32(const float *g, const float *y, long n, float p, float *dx,
                                           spacap_stream_t stream) {
  unsigned thresh;
  float scale;
  SPACAP_REQUIRE(n >= 0 && drop_params(p, thresh, scale), "spacap_relu_dropout_bwd_f32: bad arguments");
  if (n == 0) return SPACAP_OK;
  SPACAP_REQUIRE(g && y && dx, "spacap_relu_dropout_bwd_f32: null pointer");
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, spacap::as_stream(stream), g, y, n, scale, dx);
  SPACAP_CHECK_LAUNCH("spacap_relu_dropout_bwd_f32");
  return SPACAP_OK;
}

extern "C" int spacap_dropout_add_fwd_f32(const float *res, const float *y, long n, float p, uint64_t seed,
                                          const uint64_t *seed_dev, float *out, spacap_stream_t stream) {
  SPACAP_REQUIRE(res != nullptr || n == 0, "spacap_dropout_add_fwd_f32: null pointer");
  DROP_ENTRY("spacap_dropout_add_fwd_f32", 1, y, res)
}

extern "C" int spacap_dropout_add_bwd_f32(const float *g, long n, float p, uint64_t seed, const uint64_t *seed_dev,
                                          float *out, spacap_stream_t stream) {
  DROP_ENTRY("spacap_dropout_add_bwd_f32", 2, g, (const float *)nullptr)
}

// ---- Adam over one flat parameter buffer --------------------------------------------------------------------
// torch.optim.Adam(lr, betas, eps, weight_decay) as the reference builds it (scripts/train.py:262; L2 weight decay
// added to the gradient, bias-corrected moments) for ALL parameters in one launch: the parameters are views of one
// flat buffer, so are the moments; the step count lives on the device (the launch is captured in the step's
// hipGraph).  PyTorch's fused multi-tensor Adam needs 8 launches of ~43 us for the model's ~300 tensors.
namespace {
__global__ __launch_bounds__(256) void adam_flat_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v, long n, float lr,
                                                        float b1, float b2, float eps, float wd,
                                                        const float *__restrict__ step, float grad_scale,
                                                        const long long *__restrict__ skip_if_nonzero) {
  // (a gradient that a timed-out stream wait may have let through half-written is never applied: see wait_ge_kernel)
  if (skip_if_nonzero && __hip_atomic_load(skip_if_nonzero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  const float t = *step;
  const float c1 = 1.0f - powf(b1, t), c2 = 1.0f - powf(b2, t);
  const float step_size = lr / c1, rs2 = 1.0f / sqrtf(c2);
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 pv = *reinterpret_cast<f32x4 *>(p + 4 * i), mv = *reinterpret_cast<f32x4 *>(m + 4 * i),
          vv = *reinterpret_cast<f32x4 *>(v + 4 * i);
    const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + 4 * i);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float gg = gv[u] * grad_scale + wd * pv[u];
      mv[u] = b1 * mv[u] + (1.0f - b1) * gg;
      vv[u] = b2 * vv[u] + (1.0f - b2) * gg * gg;
      pv[u] -= step_size * (mv[u] / (sqrtf(vv[u]) * rs2 + eps));
    }
    *reinterpret_cast<f32x4 *>(p + 4 * i) = pv;
    *reinterpret_cast<f32x4 *>(m + 4 * i) = mv;
    *reinterpret_cast<f32x4 *>(v + 4 * i) = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float gg = g[i] * grad_scale + wd * p[i];
    m[i] = b1 * m[i] + (1.0f - b1) * gg;
    v[i] = b2 * v[i] + (1.0f - b2) * gg * gg;
    p[i] -= step_size * (m[i] / (sqrtf(v[i]) * rs2 + eps));
  }
}
}  // namespace

// p, g, m, v: f32 [n] (16-byte aligned); step: device f32 holding the 1-based step count of THIS update;
// grad_scale multiplies the gradient first (1 / world size after a summing all-reduce, else 1).
// skip_if_nonzero (nullable): a device int64; when it is non-zero at launch the update is skipped entirely (the sticky
// error word of spacap_stream_wait_ge: gradients behind a timed-out wait are never applied).
extern "C" int spacap_adam_flat_f32(float *p, const float *g, float *m, float *v, long n, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, const float *step, float grad_scale,
                                    const int64_t *skip_if_nonzero, spacap_stream_t stream) {
  SPACAP_REQUIRE(n >= 0, "spacap_adam_flat_f32: bad size");
  if (n == 0) return SPACAP_OK;
  SPACAP_REQUIRE(p && g && m && v && step, "spacap_adam_flat_f32: null pointer");
  SPACAP_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                   reinterpret_cast<uintptr_t>(v)) & 15) == 0, "spacap_adam_flat_f32: unaligned pointer");
  hipLaunchKernelGGL(adam_flat_kernel, dim3(grid_for(n)), dim3(256), 0, spacap::as_stream(stream), p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, step, grad_scale, reinterpret_cast<const long long *>(skip_if_nonzero));
  SPACAP_CHECK_LAUNCH("spacap_adam_flat_f32");
  return SPACAP_OK;
}

// ---- sum of per-slab partial results ----------------------------------------------------------------------------
// out[i] = sum_s part[s][i], s ascending (the fixed-order second stage of every split reduction in this library:
// weight-gradient slabs, partial bias / dU sums).  torch.sum(0) on these [<= 1024, 16 K .. 270 K] buffers takes ~9 us
// of mostly launch geometry; here 4 row groups x 64 float4 columns per workgroup, combined through LDS.
namespace {
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float *__restrict__ part, int nslab, long n4,
                                                        float *__restrict__ out) {
  __shared__ f32x4 s[4][64];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + c;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    const int per = (nslab + 3) / 4, s0 = grp * per, s1 = min(nslab, s0 + per);
#pragma unroll 8
    for (int k = s0; k < s1; ++k) a += *reinterpret_cast<const f32x4 *>(part + ((size_t)k * n4 + i) * 4);
  }
  s[grp][c] = a;
  __syncthreads();
  if (grp == 0 && i < n4) *reinterpret_cast<f32x4 *>(out + 4 * i) = (s[0][c] + s[1][c]) + (s[2][c] + s[3][c]);
}
}  // namespace

// Many slab sums in ONE launch: a training step produces ~70 of them (one per Linear / 1x1 convolution weight
// gradient), each a few microseconds of mostly launch latency.  The segment table travels by value in the kernel
// arguments (no device-side table to copy, so the launch can be captured in a hipGraph as it is); a workgroup finds
// its segment by binary search over the first-block prefix and then does exactly what sum_slabs_kernel does (same
// slab grouping, same order: identical values).
namespace {
constexpr int SLAB_SEG_MAX = 96;
struct SlabSeg {
  const float *part;
  float *out;
  long n4;
  int nslab, block0;
};
struct SlabTable {
  int nseg, pad;
  SlabSeg seg[SLAB_SEG_MAX];
};
__global__ __launch_bounds__(256) void sum_slabs_batched_kernel(const SlabTable T) {
  __shared__ f32x4 s[4][64];
  int lo = 0, hi = T.nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (T.seg[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SlabSeg sg = T.seg[lo];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long i = (long)((int)blockIdx.x - sg.block0) * 64 + c;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (i < sg.n4) {
    const int per = (sg.nslab + 3) / 4, s0 = grp * per, s1 = min(sg.nslab, s0 + per);
#pragma unroll 8
    for (int k = s0; k < s1; ++k) a += *reinterpret_cast<const f32x4 *>(sg.part + ((size_t)k * sg.n4 + i) * 4);
  }
  s[grp][c] = a;
  __syncthreads();
  if (grp == 0 && i < sg.n4) *reinterpret_cast<f32x4 *>(sg.out + 4 * i) = (s[0][c] + s[1][c]) + (s[2][c] + s[3][c]);
}
}  // namespace

// nseg independent slab sums: parts[i] f32 [nslabs[i]][n[i]] -> outs[i] f32 [n[i]] (each n a multiple of 4, pointers
// 16-byte aligned).  parts / outs / n / nslabs are HOST arrays, read before the call returns.
extern "C" int spacap_sum_slabs_batched_f32(const float *const *parts, float *const *outs, const long *n, const int *nslabs,
                                            int nseg, spacap_stream_t stream) {
  const char *what = "spacap_sum_slabs_batched_f32";
  SPACAP_REQUIRE(nseg >= 0 && (nseg == 0 || (parts && outs && n && nslabs)), "%s: bad arguments", what);
  hipStream_t st = spacap::as_stream(stream);
  for (int base = 0; base < nseg; base += SLAB_SEG_MAX) {
    SlabTable T;
    T.nseg = 0, T.pad = 0;
    long blocks = 0;
    for (int i = base; i < nseg && T.nseg < SLAB_SEG_MAX; ++i) {
      SPACAP_REQUIRE(nslabs[i] >= 1 && n[i] >= 0 && (n[i] & 3) == 0 && parts[i] && outs[i] &&
                         ((reinterpret_cast<uintptr_t>(parts[i]) | reinterpret_cast<uintptr_t>(outs[i])) & 15) == 0,
                     "%s: segment %d: bad size or unaligned pointer", what, i);
      if (n[i] == 0) continue;
      SlabSeg &sg = T.seg[T.nseg++];
      sg.part = parts[i], sg.out = outs[i], sg.n4 = n[i] >> 2, sg.nslab = nslabs[i], sg.block0 = (int)blocks;
      blocks += (sg.n4 + 63) / 64;
      SPACAP_REQUIRE(blocks < 2147483647L, "%s: too many blocks", what);
    }
    if (T.nseg == 0) continue;
    hipLaunchKernelGGL(sum_slabs_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, st, T);
  }
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// part f32 [nslab][n] dense (n a multiple of 4, 16-byte aligned) -> out f32 [n]
extern "C" int spacap_sum_slabs_f32(const float *part, int nslab, long n, float *out, spacap_stream_t stream) {
  SPACAP_REQUIRE(nslab >= 1 && n >= 0 && (n & 3) == 0, "spacap_sum_slabs_f32: bad sizes (nslab=%d, n=%ld)", nslab, n);
  if (n == 0) return SPACAP_OK;
  SPACAP_REQUIRE(part && out && ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                 "spacap_sum_slabs_f32: null or unaligned pointer");
  const long n4 = n >> 2;
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, spacap::as_stream(stream), part, nslab, n4,
                     out);
  SPACAP_CHECK_LAUNCH("spacap_sum_slabs_f32");
  return SPACAP_OK;
}

// ---- row-wise L2 normalisation: y = x / |x|  (models/SpaCapNet.py:66-67: vote features, no epsilon) ---------------------
// One wavefront per row (D <= 1024 floats, multiple of 4).  Backward: dx = (g - y (g . y)) / |x|.
// PyTorch: norm + div forward, ~8 launches backward.
namespace {
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float *__restrict__ x, long rows, int D, float *__restrict__ y,
                                                         float *__restrict__ inv_norm) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float *xr = x + (size_t)r * D;
  float s = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + c);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  s = spacap::wave_sum_f32(s);
  const float nrm = sqrtf(s);
  for (int c = lane * 4; c < D; c += 256) {
    f32x4 v = *reinterpret_cast<const f32x4 *>(xr + c);
    v[0] /= nrm, v[1] /= nrm, v[2] /= nrm, v[3] /= nrm;      // true division, as torch.div
    *reinterpret_cast<f32x4 *>(y + (size_t)r * D + c) = v;
  }
  if (lane == 0) inv_norm[r] = 1.0f / nrm;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float *__restrict__ g, const float *__restrict__ y,
                                                         const float *__restrict__ inv_norm, long rows, int D,
                                                         float *__restrict__ dx) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float *gr = g + (size_t)r * D, *yr = y + (size_t)r * D;
  float s = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(gr + c), b = *reinterpret_cast<const f32x4 *>(yr + c);
    s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  }
  s = spacap::wave_sum_f32(s);
  const float inv = inv_norm[r];
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(gr + c), b = *reinterpret_cast<const f32x4 *>(yr + c);
    f32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = (a[u] - b[u] * s) * inv;
    *reinterpret_cast<f32x4 *>(dx + (size_t)r * D + c) = o;
  }
}
}  // namespace

extern "C" int spacap_l2norm_rows_fwd_f32(const float *x, long rows, int D, float *y, float *inv_norm, spacap_stream_t stream) {
  SPACAP_REQUIRE(rows >= 0 && D >= 4 && D % 4 == 0 && D <= 4096, "spacap_l2norm_rows_fwd_f32: bad sizes rows=%ld D=%d", rows, D);
  if (rows == 0) return SPACAP_OK;
  SPACAP_REQUIRE(x && y && inv_norm, "spacap_l2norm_rows_fwd_f32: null pointer");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, spacap::as_stream(stream), x, rows, D, y, inv_norm);
  SPACAP_CHECK_LAUNCH("spacap_l2norm_rows_fwd_f32");
  return SPACAP_OK;
}
extern "C" int spacap_l2norm_rows_bwd_f32(const float *g, const float *y, const float *inv_norm, long rows, int D, float *dx,
                                          spacap_stream_t stream) {
  SPACAP_REQUIRE(rows >= 0 && D >= 4 && D % 4 == 0 && D <= 4096, "spacap_l2norm_rows_bwd_f32: bad sizes rows=%ld D=%d", rows, D);
  if (rows == 0) return SPACAP_OK;
  SPACAP_REQUIRE(g && y && inv_norm && dx, "spacap_l2norm_rows_bwd_f32: null pointer");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, spacap::as_stream(stream), g, y, inv_norm, rows, D, dx);
  SPACAP_CHECK_LAUNCH("spacap_l2norm_rows_bwd_f32");
  return SPACAP_OK;
}

// ---- many small device-to-device copies in one launch -----------------------------------------------------------------------
// The step copies the batch and the prefetched pyramid into its static buffers (~30 tensors of four dtypes) and packs ~170
// gradient tensors into the flat bucket; as multi-tensor library copies these are 4 + 4 launches of ~20 us each.  Job table by
// value in the kernel arguments (hipGraph-capturable as it is); a workgroup finds its job by binary search over the first-block
// prefix and moves 16 KB: 16-byte pieces when source, destination and size allow, bytes otherwise.
namespace {
constexpr int CPJ_MAX = 120;
constexpr long CPJ_BLOCK_BYTES = 16384;
struct CopyJob {
  const char *src;
  char *dst;
  long nbytes;
  int block0, vec;
};
struct CopyTable {
  int njobs, pad;
  CopyJob job[CPJ_MAX];
};
__global__ __launch_bounds__(256) void copy_batched_kernel(const CopyTable T) {
  int lo = 0, hi = T.njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (T.job[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const CopyJob J = T.job[lo];
  const long beg = (long)((int)blockIdx.x - J.block0) * CPJ_BLOCK_BYTES, end = min(J.nbytes, beg + CPJ_BLOCK_BYTES);
  if (J.vec) {
    for (long o = beg + 16 * threadIdx.x; o < end; o += 16 * 256)
      *reinterpret_cast<f32x4 *>(J.dst + o) = *reinterpret_cast<const f32x4 *>(J.src + o);
  } else {
    for (long o = beg + threadIdx.x; o < end; o += 256) J.dst[o] = J.src[o];
  }
}
}  // namespace
extern "C" int spacap_copy_batched(const void *const *src, void *const *dst, const long *nbytes, int njobs, spacap_stream_t stream) {
  const char *what = "spacap_copy_batched";
  SPACAP_REQUIRE(njobs >= 0 && (njobs == 0 || (src && dst && nbytes)), "%s: bad arguments", what);
  hipStream_t s = spacap::as_stream(stream);
  int i = 0;
  while (i < njobs) {
    CopyTable T;
    T.njobs = 0, T.pad = 0;
    long blocks = 0;
    for (; i < njobs && T.njobs < CPJ_MAX; ++i) {
      SPACAP_REQUIRE(nbytes[i] >= 0 && (nbytes[i] == 0 || (src[i] && dst[i])), "%s: job %d: null pointer or negative size", what, i);
      if (nbytes[i] == 0) continue;
      CopyJob &J = T.job[T.njobs++];
      J.src = static_cast<const char *>(src[i]), J.dst = static_cast<char *>(dst[i]), J.nbytes = nbytes[i], J.block0 = (int)blocks;
      J.vec = ((reinterpret_cast<uintptr_t>(src[i]) | reinterpret_cast<uintptr_t>(dst[i]) | (uintptr_t)nbytes[i]) & 15) == 0;
      blocks += (nbytes[i] + CPJ_BLOCK_BYTES - 1) / CPJ_BLOCK_BYTES;
      SPACAP_REQUIRE(blocks < 2147483647L, "%s: too many blocks", what);
    }
    if (T.njobs) hipLaunchKernelGGL(copy_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, s, T);
  }
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

// ---- a pause on a stream: one wave spins on the 100 MHz wall clock (engine.py: the side-stream pyramid must not be released
// by the same event as the step's graph) ------------------------------------------------------------------------------------
namespace {
__global__ void delay_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace
extern "C" int spacap_stream_delay(int microseconds, spacap_stream_t stream) {
  SPACAP_REQUIRE(microseconds >= 0 && microseconds <= 100000, "spacap_stream_delay: %d us out of range", microseconds);
  if (microseconds == 0) return SPACAP_OK;
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(1), 0, spacap::as_stream(stream), (unsigned long long)microseconds * 100ull);
  SPACAP_CHECK_LAUNCH("spacap_stream_delay");
  return SPACAP_OK;
}

// ---- a stream waits for a word in device memory: one wave spins until *flag >= value (engine.py: the gradient all-reduce of
// the captioner's slice starts on the communication stream as soon as the captured step, in the middle of its backward, has
// written the step number there -- a dependency from INSIDE a hipGraph to a stream outside it, which events cannot express).
// A wait that runs out of time does NOT fall through silently: it leaves `value` in the sticky word *err (first failure
// wins; never cleared by the library).  What is queued behind the wait on that stream still runs -- the device cannot
// un-queue it -- so the consumer of that work must be gated on *err: spacap_adam_flat_f32 skips its update when the word
// it is given is non-zero, and the host raises when it reads the word (engine.Trainer.check_health / the next step()).
namespace {
__global__ void wait_ge_kernel(const long long *flag, long long value, unsigned long long ticks, unsigned long long *err) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value) {
    if (wall_clock64() - t0 > ticks) {
      atomicCAS(err, 0ull, (unsigned long long)(value > 0 ? value : 1));
      __threadfence_system();
      return;
    }
    __builtin_amdgcn_s_sleep(16);
  }
}
__global__ void signal_set_kernel(long long *flag, const long long *value) {
  __hip_atomic_store(flag, *value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace
extern "C" int spacap_stream_wait_ge(const int64_t *flag, int64_t value, int timeout_ms, int64_t *err, spacap_stream_t stream) {
  SPACAP_REQUIRE(flag && err && timeout_ms >= 1 && timeout_ms <= 600000, "spacap_stream_wait_ge: bad arguments");
  hipLaunchKernelGGL(wait_ge_kernel, dim3(1), dim3(1), 0, spacap::as_stream(stream), reinterpret_cast<const long long *>(flag),
                     (long long)value, (unsigned long long)timeout_ms * 100000ull, reinterpret_cast<unsigned long long *>(err));
  SPACAP_CHECK_LAUNCH("spacap_stream_wait_ge");
  return SPACAP_OK;
}
/* *flag = *value (both device words), visible to spacap_stream_wait_ge on any stream once everything before it on this stream is done */
extern "C" int spacap_stream_signal(int64_t *flag, const int64_t *value, spacap_stream_t stream) {
  SPACAP_REQUIRE(flag && value, "spacap_stream_signal: null pointer");
  hipLaunchKernelGGL(signal_set_kernel, dim3(1), dim3(1), 0, spacap::as_stream(stream), reinterpret_cast<long long *>(flag),
                     reinterpret_cast<const long long *>(value));
  SPACAP_CHECK_LAUNCH("spacap_stream_signal");
  return SPACAP_OK;
}

// ---- lab: device timestamps inside a captured step (tools/lab/step_stamps.py) --------------------------------------------
namespace {
__global__ void stamp_kernel(unsigned long long *slot) { *slot = wall_clock64(); }   // s_memrealtime: 100 MHz
}  // namespace
extern "C" int spacap_lab_stamp(uint64_t *slot, spacap_stream_t stream) {
  SPACAP_REQUIRE(slot, "spacap_lab_stamp: null pointer");
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, spacap::as_stream(stream), reinterpret_cast<unsigned long long *>(slot));
  SPACAP_CHECK_LAUNCH("spacap_lab_stamp");
  return SPACAP_OK;
}
