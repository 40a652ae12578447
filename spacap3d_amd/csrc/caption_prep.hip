// Decoder input of the captioner's training step in one launch each way, for gfx950 (MI355X).
//
// Reference: models/transformer_captioner.py:350-367 (forward_train: the proposal nearest to the referred object's centre
// by squared distance, its feature as the object indicator, teacher-forcing tokens and their mask), :246-249
// (EncoderDecoder.decode: the indicator gets the encoder output of the same proposal added), :129-137, :150-161
// (Embeddings * sqrt(d_model) + sinusoidal positional encoding, dropout) and :193-199 (the indicator is prepended).
// As tensor operations these are ~25 launches on a few kilobytes.  Forward, one workgroup per scene:
//   idx[b]  = argmin_k |xyz[b,k] - ref[b]|^2 (first minimum), dist[b] = that minimum, good[b] = dist > -1
//   x0[b,0] = src[b,idx] + memory[b,idx];   x0[b,1+t] = dropout(E[tok[b,1+t]] * sqrt(D) + pe[t])      t = 0 .. L-2
//   mask[b,q,k] = tok[b,k] > 0 and k <= q                                                               q, k = 0 .. L-1
//   (L = T - 1 for T label tokens per scene: the decoder sees the indicator and tokens 1 .. T-2)
//   the last workgroup to finish adds up pred = sum_b dist[b] good[b] / max(1, sum_b good[b]) in scene order.
// Backward: d src = d memory = the indicator row's gradient at (b, idx[b]), zero elsewhere; d E[v] = sum over the positions
// holding token v, in (scene, position) order, of dropout'(g) * sqrt(D): one workgroup per vocabulary row, no atomics.
// Dropout is the library's counter hash (elementwise.hip), regenerated in the backward.
#include "common.hpp"

namespace {

struct DropSeed {
  unsigned lo, hi;
};
__device__ __forceinline__ DropSeed make_seed(unsigned long long seed, const unsigned long long *seed_dev) {
  const unsigned long long s = seed + (seed_dev ? *seed_dev * 0x9E3779B97F4A7C15ull : 0ull);
  return DropSeed{(unsigned)s, (unsigned)(s >> 32)};
}
__device__ __forceinline__ unsigned hash32(unsigned long long idx, DropSeed s) {   // murmur3 fmix32, as elementwise.hip
  unsigned h = (unsigned)idx ^ s.lo;
  h += ((unsigned)(idx >> 32) ^ s.hi) * 0x9E3779B1u;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
inline bool drop_params(float p, unsigned &thresh, float &scale) {
  if (!(p >= 0.f && p < 1.f)) return false;
  thresh = p > 0.f ? (unsigned)((double)p * 4294967296.0) : 0u;
  scale = 1.0f / (1.0f - p);
  return true;
}

struct PrepArgs {
  const float *xyz, *ref, *src, *memory, *emb, *pe;
  const long long *tok;
  int B, K, D, T, V;
  unsigned thresh;
  float scale, sqrt_d;
  unsigned long long seed;
  const unsigned long long *seed_dev;
  float *x0;
  unsigned char *mask, *good;
  long long *idx;
  float *dist, *pred;
  int *counter;
};

// grid (L, B): block (0, b) finds scene b's proposal and writes the indicator row, the mask and the per-scene scalars; block
// (row, b) one embedded token row.  Two dependent loads deep at most: the launch is a few microseconds of latency.
__global__ __launch_bounds__(256) void cap_prep_fwd_kernel(PrepArgs a) {
  __shared__ float s_d[256];
  __shared__ int s_k[256];
  __shared__ int s_last;
  const int row = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, L = a.T - 1, D = a.D;
  if (row > 0) {
    long long t = a.tok[(size_t)b * a.T + row];
    t = t < 0 ? 0 : (t >= a.V ? a.V - 1 : t);
    const DropSeed sd = make_seed(a.seed, a.seed_dev);
    for (int c = tid; c < D; c += 256) {
      float v = a.emb[(size_t)t * D + c] * a.sqrt_d + a.pe[(size_t)(row - 1) * D + c];
      const bool keep = a.thresh == 0u || hash32((unsigned long long)((size_t)b * L + row) * D + c, sd) >= a.thresh;
      a.x0[((size_t)b * L + row) * D + c] = keep ? v * a.scale : 0.f;
    }
    return;
  }
  // ---- nearest proposal: d = ((x - rx)^2 + (z - rz)^2) + (y - ry)^2, the order torch.sum(dim=-1) adds three terms in ----
  const float rx = a.ref[b * 3 + 0], ry = a.ref[b * 3 + 1], rz = a.ref[b * 3 + 2];
  float best = INFINITY;
  int bk = 0x7fffffff;
  for (int k = tid; k < a.K; k += 256) {
    const float *p = a.xyz + ((size_t)b * a.K + k) * 3;
    const float dx = p[0] - rx, dy = p[1] - ry, dz = p[2] - rz;
    const float d = (dx * dx + dz * dz) + dy * dy;
    if (d < best) best = d, bk = k;
  }
  s_d[tid] = best, s_k[tid] = bk;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const float d2 = s_d[tid + s];
      const int k2 = s_k[tid + s];
      if (d2 < s_d[tid] || (d2 == s_d[tid] && k2 < s_k[tid])) s_d[tid] = d2, s_k[tid] = k2;
    }
    __syncthreads();
  }
  const int kb = s_k[0] == 0x7fffffff ? 0 : s_k[0];
  if (tid == 0) {
    a.idx[b] = kb;
    a.dist[b] = s_d[0];
    a.good[b] = s_d[0] > -1.f;
  }
  for (int c = tid; c < D; c += 256) {
    float v = a.src[((size_t)b * a.K + kb) * D + c];
    if (a.memory) v += a.memory[((size_t)b * a.K + kb) * D + c];
    a.x0[((size_t)b * L) * D + c] = v;
  }
  for (int e = tid; e < L * L; e += 256) {
    const int q = e / L, k = e - q * L;
    a.mask[(size_t)b * L * L + e] = (a.tok[(size_t)b * a.T + k] > 0 && k <= q) ? 1 : 0;
  }
  // ---- the last scene block adds the scenes up in order ----
  __threadfence();
  if (tid == 0) s_last = atomicAdd(a.counter, 1) == a.B - 1;
  __syncthreads();
  if (s_last && tid == 0) {
    __threadfence();
    float num = 0.f, cnt = 0.f;
    for (int i = 0; i < a.B; ++i) {
      const float d = __builtin_nontemporal_load(a.dist + i);
      const bool g = d > -1.f;
      num += g ? d : 0.f;
      cnt += g ? 1.f : 0.f;
    }
    a.pred[0] = num / fmaxf(cnt, 1.f);
  }
}

struct PrepBwdArgs {
  const float *g;
  const long long *tok, *idx;
  int B, K, D, T, V, with_rows;
  unsigned thresh;
  float scale, sqrt_d;
  unsigned long long seed;
  const unsigned long long *seed_dev;
  float *d_rows, *d_emb;
};

__global__ __launch_bounds__(128) void cap_prep_bwd_kernel(PrepBwdArgs a) {
  const int tid = threadIdx.x, L = a.T - 1, D = a.D;
  if ((int)blockIdx.x >= a.V) {   // 2 048 elements of one scene's (K, D) slab of d src = d memory (was one block per scene: 39 us)
    const int per = (a.K * D + 2047) / 2048, q = blockIdx.x - a.V, b = q / per, e0 = (q - b * per) * 2048;
    const int kb = (int)a.idx[b];
    float *o = a.d_rows + (size_t)b * a.K * D;
    for (int e = e0 + tid; e < min(e0 + 2048, a.K * D); e += 128) {
      const int k = e / D, c = e - k * D;
      o[e] = k == kb ? a.g[((size_t)b * L) * D + c] : 0.f;
    }
    return;
  }
  // the positions that hold this row's token, found once by the block (B (L - 1) token ids, a few hundred), then summed in order
  const int v = blockIdx.x, NP = a.B * (L - 1);
  __shared__ int s_pos[1024];
  __shared__ int s_n;
  const DropSeed sd = make_seed(a.seed, a.seed_dev);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  // most rows of the table hold no token of this batch (a few hundred positions against V rows): one vote, then zeros
  bool mine = false;
  for (int i = tid; i < NP; i += 128) {
    const int b = i / (L - 1), row = 1 + i - b * (L - 1);
    long long t = a.tok[(size_t)b * a.T + row];
    t = t < 0 ? 0 : (t >= a.V ? a.V - 1 : t);
    mine |= t == v;
  }
  if (!__syncthreads_or(mine)) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (tid + 128 * j < D) a.d_emb[(size_t)v * D + tid + 128 * j] = 0.f;
    return;
  }
  for (int base = 0; base < NP; base += 1024) {
    __syncthreads();
    if (tid == 0) s_n = 0;
    __syncthreads();
    // ordered compaction: thread tid scans its 8 consecutive positions, ranks come from a running count per 128-wide pass
    for (int pass = 0; pass < 8; ++pass) {
      const int i = base + pass * 128 + tid;
      bool hit = false;
      if (i < NP) {
        const int b = i / (L - 1), row = 1 + i - b * (L - 1);
        long long t = a.tok[(size_t)b * a.T + row];
        t = t < 0 ? 0 : (t >= a.V ? a.V - 1 : t);
        hit = t == v;
      }
      const unsigned long long m = __ballot(hit);
      __shared__ int s_w[2];
      if ((tid & 63) == 0) s_w[tid >> 6] = __popcll(m);
      __syncthreads();
      if (hit) s_pos[s_n + (tid >= 64 ? s_w[0] : 0) + __popcll(m & ((1ull << (tid & 63)) - 1ull))] = i;
      __syncthreads();
      if (tid == 0) s_n += s_w[0] + s_w[1];
      __syncthreads();
    }
    const int n = s_n;
    for (int q = 0; q < n; ++q) {
      const int i = s_pos[q], b = i / (L - 1), row = 1 + i - b * (L - 1);
      const size_t e0 = ((size_t)b * L + row) * D;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = tid + 128 * j;
        if (c < D) {
          const bool keep = a.thresh == 0u || hash32((unsigned long long)(e0 + c), sd) >= a.thresh;
          if (keep) acc[j] += (a.g[e0 + c] * a.scale) * a.sqrt_d;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (tid + 128 * j < D) a.d_emb[(size_t)v * D + tid + 128 * j] = acc[j];
}

}  // namespace

extern "C" int spacap_caption_prep_fwd_f32(const float *xyz, const float *ref, const float *src, const float *memory,
                                           const int64_t *tok, const float *emb, const float *pe, int B, int K, int D, int T, int V,
                                           float p, uint64_t seed, const uint64_t *seed_dev, float *x0, uint8_t *mask, int64_t *idx,
                                           float *dist, uint8_t *good, float *pred, int32_t *counter, spacap_stream_t stream) {
  const char *what = "spacap_caption_prep_fwd_f32";
  PrepArgs a;
  SPACAP_REQUIRE(B >= 0 && K >= 1 && D >= 1 && T >= 2 && V >= 1 && drop_params(p, a.thresh, a.scale), "%s: bad arguments", what);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(xyz && ref && src && tok && emb && pe && x0 && mask && idx && dist && good && pred && counter, "%s: null pointer", what);
  a.xyz = xyz, a.ref = ref, a.src = src, a.memory = memory, a.emb = emb, a.pe = pe;
  a.tok = reinterpret_cast<const long long *>(tok);
  a.B = B, a.K = K, a.D = D, a.T = T, a.V = V;
  a.sqrt_d = (float)sqrt((double)D);
  a.seed = seed, a.seed_dev = reinterpret_cast<const unsigned long long *>(seed_dev);
  a.x0 = x0, a.mask = mask, a.good = good, a.idx = reinterpret_cast<long long *>(idx), a.dist = dist, a.pred = pred, a.counter = counter;
  SPACAP_REQUIRE(B <= 65535, "%s: B out of range", what);
  // the last-block ticket belongs to THIS call (the caller allocates it with the outputs): launches in flight on other
  // streams, or captured into other graphs, cannot disturb it
  SPACAP_CHECK_HIP(hipMemsetAsync(counter, 0, sizeof(int32_t), spacap::as_stream(stream)), what);
  hipLaunchKernelGGL(cap_prep_fwd_kernel, dim3(T - 1, B), dim3(256), 0, spacap::as_stream(stream), a);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_caption_prep_bwd_f32(const float *g, const int64_t *tok, const int64_t *idx, int B, int K, int D, int T, int V,
                                           float p, uint64_t seed, const uint64_t *seed_dev, float *d_rows, float *d_emb,
                                           spacap_stream_t stream) {
  const char *what = "spacap_caption_prep_bwd_f32";
  PrepBwdArgs a;
  SPACAP_REQUIRE(B >= 0 && K >= 1 && D >= 1 && T >= 2 && V >= 1 && drop_params(p, a.thresh, a.scale), "%s: bad arguments", what);
  if (B == 0) return SPACAP_OK;
  SPACAP_REQUIRE(g && tok && idx && d_emb, "%s: null pointer", what);
  SPACAP_REQUIRE(D <= 1024, "%s: D=%d unsupported (at most 1024)", what, D);
  a.g = g, a.tok = reinterpret_cast<const long long *>(tok), a.idx = reinterpret_cast<const long long *>(idx);
  a.B = B, a.K = K, a.D = D, a.T = T, a.V = V, a.with_rows = d_rows != nullptr;
  a.sqrt_d = (float)sqrt((double)D);
  a.seed = seed, a.seed_dev = reinterpret_cast<const unsigned long long *>(seed_dev);
  a.d_rows = d_rows, a.d_emb = d_emb;
  hipLaunchKernelGGL(cap_prep_bwd_kernel, dim3(V + (d_rows ? B * ((K * D + 2047) / 2048) : 0)), dim3(128), 0, spacap::as_stream(stream), a);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
