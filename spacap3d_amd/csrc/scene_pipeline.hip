// Device side of the input pipeline for gfx950 (MI355X): subsample + augmentation + vote labels of a batch of
// scenes that are resident in HBM.
//
// Replaces the per-item numpy work of ScannetReferenceDataset.__getitem__ (lib/dataset.py:291-531) that touches
// every point: the random subsample gather (:335-338, utils/pc_utils.py:32-40), the flips / three axis rotations /
// translation of the coordinates (:364-404) and the vote generation loop over instances (:415-428).  The labels of
// the <= 128 boxes per scene are tiny and stay in batched torch ops (spacap3d_amd/dataset.py).
//
// Arithmetic follows the reference so that results are reproducible against it: coordinates are float32, every
// rotation is evaluated in float64 and rounded to float32 (numpy assigns a float64 product into the float32 array),
// the translation likewise; instance boxes are min / max over the SAMPLED, AUGMENTED points in float32, centre =
// 0.5f * (min + max); an instance votes iff its first sampled point belongs to one of the 37 object classes.
// Min / max / first are order-independent reductions (integer atomics on order-preserving keys): deterministic.
#include "common.hpp"

namespace {

constexpr int AUG_DOUBLES = 32;  // per item: flip_x, flip_y, Rx[9], Ry[9], Rz[9], t[3]

// out[b, p, :] = augment(scene_b[choices[b, p], :]);  labels gathered alongside
__global__ __launch_bounds__(256) void scene_sample_augment_kernel(const float *const *__restrict__ scene_feat,
                                                                   const int32_t *const *__restrict__ scene_ins,
                                                                   const uint8_t *const *__restrict__ scene_isobj,
                                                                   const float *const *__restrict__ scene_color,
                                                                   const int32_t *__restrict__ choices,
                                                                   const double *__restrict__ aug, int P, int C, int augment,
                                                                   int C_out, const int *__restrict__ dst_off,
                                                                   float *__restrict__ pc, int32_t *__restrict__ ins_out,
                                                                   uint8_t *__restrict__ isobj_out, float *__restrict__ color_out) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int src = choices[(size_t)b * P + p];
  const float *row = scene_feat[b] + (size_t)src * C;
  float x = row[0], y = row[1], z = row[2];
  if (augment) {
    const double *a = aug + (size_t)b * AUG_DOUBLES;
    if (a[0] != 0.0) x = -1.0f * x;
    if (a[1] != 0.0) y = -1.0f * y;
#pragma unroll
    for (int r = 0; r < 3; ++r) {  // pc[:, 0:3] = np.dot(pc[:, 0:3], R^T): float64 products, float32 store
      const double *R = a + 2 + 9 * r;
      const double dx = x, dy = y, dz = z;
      const float nx = (float)(dx * R[0] + dy * R[1] + dz * R[2]);
      const float ny = (float)(dx * R[3] + dy * R[4] + dz * R[5]);
      const float nz = (float)(dx * R[6] + dy * R[7] + dz * R[8]);
      x = nx, y = ny, z = nz;
    }
    x = (float)((double)x + a[29]);
    y = (float)((double)y + a[30]);
    z = (float)((double)z + a[31]);
  }
  // output rows have C_out channels; source channel c >= 3 goes to dst_off[c] (identity without a map): colour and
  // multiview columns, when present, sit between them and are filled by scene_gather_rows_kernel
  float *o = pc + ((size_t)b * P + p) * C_out;
  o[0] = x, o[1] = y, o[2] = z;
  for (int c = 3; c < C; ++c) o[dst_off ? dst_off[c] : c] = row[c];
  ins_out[(size_t)b * P + p] = scene_ins[b][src];
  isobj_out[(size_t)b * P + p] = scene_isobj[b][src];
  if (color_out) {
    const float *c = scene_color[b] + (size_t)src * 3;
    float *co = color_out + ((size_t)b * P + p) * 3;
    co[0] = c[0], co[1] = c[1], co[2] = c[2];
  }
}

// out[b, p, off + j] = src_b[choices[b, p] * W + j], j < W: the per-vertex colour (W = 3, lib/dataset.py:312-315) and multiview
// (W = 128, :321-328) columns of the sampled points.  LPR lanes per row (1 for narrow rows, 32 for the 512-byte multiview rows:
// coalesced 128-byte segments both ways).
template <int LPR>
__global__ __launch_bounds__(256) void scene_gather_rows_kernel(const float *const *__restrict__ src,
                                                                const int32_t *__restrict__ choices, int P, int W,
                                                                float *__restrict__ out, int out_stride, int out_off) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int p = t / LPR, l = t % LPR;
  if (p >= P) return;
  const float *row = src[b] + (size_t)choices[(size_t)b * P + p] * W;
  float *o = out + ((size_t)b * P + p) * out_stride + out_off;
  for (int j = l; j < W; j += LPR) o[j] = row[j];
}

__device__ __forceinline__ int f2key(float f) {  // order-preserving float -> signed int
  const int b = __float_as_int(f);
  return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

// ws[b, inst, 0..2] = min keys, 3..5 = max keys, 6 = first sampled position
__global__ __launch_bounds__(256) void votes_init_kernel(int *__restrict__ ws, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int k = (int)(i % 7);
  ws[i] = k < 3 ? 0x7fffffff : (k < 6 ? (int)0x80000000 : 0x7fffffff);
}

__global__ __launch_bounds__(256) void votes_reduce_kernel(const float *__restrict__ pc, const int32_t *__restrict__ ins, int P,
                                                           int C, int max_inst, int *__restrict__ ws) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int inst = ins[(size_t)b * P + p];
  if (inst < 0 || inst >= max_inst) return;
  const float *x = pc + ((size_t)b * P + p) * C;
  int *w = ws + ((size_t)b * max_inst + inst) * 7;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int k = f2key(x[d]);
    atomicMin(&w[d], k);
    atomicMax(&w[3 + d], k);
  }
  atomicMin(&w[6], p);
}

__global__ __launch_bounds__(256) void votes_write_kernel(const float *__restrict__ pc, const int32_t *__restrict__ ins,
                                                          const uint8_t *__restrict__ isobj, int P, int C, int max_inst,
                                                          const int *__restrict__ ws, float *__restrict__ votes,
                                                          int64_t *__restrict__ vmask) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const size_t o = (size_t)b * P + p;
  const int inst = ins[o];
  float v[3] = {0.f, 0.f, 0.f};
  int64_t m = 0;
  if (inst >= 0 && inst < max_inst) {
    const int *w = ws + ((size_t)b * max_inst + inst) * 7;
    if (isobj[(size_t)b * P + w[6]]) {  // semantic class of the instance's first sampled point (lib/dataset.py:420)
      const float *x = pc + o * C;
#pragma unroll
      for (int d = 0; d < 3; ++d) v[d] = 0.5f * (key2f(w[d]) + key2f(w[3 + d])) - x[d];
      m = 1;
    }
  }
  float *vo = votes + o * 9;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int d = 0; d < 3; ++d) vo[3 * r + d] = v[d];  // np.tile(point_votes, (1, 3)): three identical votes
  vmask[o] = m;
}

}  // namespace

extern "C" int spacap_scene_aug_doubles(void) { return AUG_DOUBLES; }

extern "C" int spacap_scene_sample_augment_map_f32(const float *const *scene_feat, const int32_t *const *scene_ins,
                                                   const uint8_t *const *scene_isobj, const float *const *scene_color,
                                                   const int32_t *choices, const double *aug, int B, int P, int C,
                                                   int augment, int C_out, const int *dst_off, float *pc, int32_t *ins_out,
                                                   uint8_t *isobj_out, float *color_out, spacap_stream_t stream) {
  const char *what = "spacap_scene_sample_augment_f32";
  SPACAP_REQUIRE(B >= 0 && P >= 0 && C >= 3 && C_out >= C && B <= 65535, "%s: bad sizes", what);
  if (B == 0 || P == 0) return SPACAP_OK;
  SPACAP_REQUIRE(scene_feat && scene_ins && scene_isobj && choices && pc && ins_out && isobj_out && (aug || !augment) &&
                     (scene_color || !color_out) && (dst_off || C_out == C), "%s: null pointer", what);
  hipLaunchKernelGGL(scene_sample_augment_kernel, dim3((P + 255) / 256, B), dim3(256), 0, spacap::as_stream(stream), scene_feat,
                     scene_ins, scene_isobj, scene_color, choices, aug, P, C, augment, C_out, dst_off, pc, ins_out, isobj_out,
                     color_out);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" int spacap_scene_sample_augment_f32(const float *const *scene_feat, const int32_t *const *scene_ins,
                                               const uint8_t *const *scene_isobj, const float *const *scene_color,
                                               const int32_t *choices, const double *aug, int B, int P, int C,
                                               int augment, float *pc, int32_t *ins_out, uint8_t *isobj_out,
                                               float *color_out, spacap_stream_t stream) {
  return spacap_scene_sample_augment_map_f32(scene_feat, scene_ins, scene_isobj, scene_color, choices, aug, B, P, C, augment, C,
                                             nullptr, pc, ins_out, isobj_out, color_out, stream);
}

extern "C" int spacap_scene_gather_rows_f32(const float *const *src, const int32_t *choices, int B, int P, int W, float *out,
                                            int out_stride, int out_off, spacap_stream_t stream) {
  const char *what = "spacap_scene_gather_rows_f32";
  SPACAP_REQUIRE(B >= 0 && P >= 0 && W >= 1 && out_off >= 0 && out_stride >= out_off + W && B <= 65535, "%s: bad sizes", what);
  if (B == 0 || P == 0) return SPACAP_OK;
  SPACAP_REQUIRE(src && choices && out, "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
  if (W >= 32)
    hipLaunchKernelGGL(scene_gather_rows_kernel<32>, dim3((unsigned)(((long)P * 32 + 255) / 256), B), dim3(256), 0, s, src, choices, P, W,
                       out, out_stride, out_off);
  else
    hipLaunchKernelGGL(scene_gather_rows_kernel<1>, dim3((P + 255) / 256, B), dim3(256), 0, s, src, choices, P, W, out, out_stride,
                       out_off);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}

extern "C" size_t spacap_scene_votes_workspace_bytes(int B, int max_inst) {
  return B > 0 && max_inst > 0 ? (size_t)B * max_inst * 7 * sizeof(int) : 0;
}

extern "C" int spacap_scene_votes_f32(const float *pc, const int32_t *ins, const uint8_t *isobj, int B, int P, int C,
                                      int max_inst, void *workspace, float *votes, int64_t *vmask,
                                      spacap_stream_t stream) {
  const char *what = "spacap_scene_votes_f32";
  SPACAP_REQUIRE(B >= 0 && P >= 0 && C >= 3 && max_inst >= 1 && B <= 65535, "%s: bad sizes", what);
  if (B == 0 || P == 0) return SPACAP_OK;
  SPACAP_REQUIRE(pc && ins && isobj && workspace && votes && vmask, "%s: null pointer", what);
  hipStream_t s = spacap::as_stream(stream);
  int *ws = reinterpret_cast<int *>(workspace);
  const long n = (long)B * max_inst * 7;
  hipLaunchKernelGGL(votes_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ws, n);
  hipLaunchKernelGGL(votes_reduce_kernel, dim3((P + 255) / 256, B), dim3(256), 0, s, pc, ins, P, C, max_inst, ws);
  hipLaunchKernelGGL(votes_write_kernel, dim3((P + 255) / 256, B), dim3(256), 0, s, pc, ins, isobj, P, C, max_inst, ws, votes, vmask);
  SPACAP_CHECK_LAUNCH(what);
  return SPACAP_OK;
}
