/*
 * spacap_hip.h -- C ABI of libspacap_hip.so, the MI355X (gfx950) implementation of the
 * SpaCap3D hot path.
 *
 * Boundary B1 replaces the nine functions the reference registers in its pybind module
 * `pointnet2._ext` (lib/pointnet2/_ext_src/src/bindings.cpp:6-19).  Boundary B2 replaces the
 * Python function `attention()` (models/transformer_captioner.py:27-37) and the relation feature
 * built from its outputs (models/transformer_captioner.py:392-397).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer (hipMalloc'd / torch CUDA
 *    tensor storage) unless it is named host_*; tensors are dense row-major in the stated shape
 *    unless explicit strides are passed;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); kernels are enqueued,
 *    never synchronised (the reference enqueues on the current stream the same way,
 *    include/cuda_utils.h + at::cuda::getCurrentCUDAStream());
 *  - every entry point returns 0 on success and a negative SPACAP_E_* code otherwise (the
 *    reference prints and exit(-1)s on a launch failure, include/cuda_utils.h:30-39; a library
 *    must not), `spacap_last_error()` returns a thread-local message for the last failure;
 *  - no entry point allocates, frees or synchronises: all scratch is caller-provided
 *    (`*_workspace_bytes`), so every call is hipGraph-capturable;
 *  - outputs that the reference creates with torch::zeros and then scatters into
 *    (`*_grad`) are zero-filled by the entry point itself (a memset node on `stream`).
 */
#ifndef SPACAP_HIP_H
#define SPACAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPACAP_ABI_VERSION 4

#define SPACAP_OK 0
#define SPACAP_E_INVALID (-1)   /* bad argument (null pointer, negative size, unsupported shape) */
#define SPACAP_E_LAUNCH (-2)    /* hipGetLastError() after a launch / memset                    */
#define SPACAP_E_NO_DEVICE (-3) /* no HIP device visible                                         */

typedef void *spacap_stream_t; /* hipStream_t */

int spacap_abi_version(void);
const char *spacap_last_error(void);
/* number of HIP devices visible, or a negative SPACAP_E_* (used by the host shim to fail loudly) */
int spacap_device_count(void);

/* The reference's launch helper, restated for the host side: clamp(2^floor(log2 w), 1, 512)
 * computed through the same double-precision log quotient (include/cuda_utils.h:15-19).  The FPS
 * arg-max tie-break depends on it. */
int spacap_opt_n_threads(int work_size);

/* ---- sampling (replaces src/sampling.cpp) -------------------------------------------------- */

/* furthest_point_sampling(points f32[B,N,3], nsamples) -> i32[B,m]   (src/sampling.cpp:66-87,
 * kernel src/sampling_gpu.cu:69-173).  Bit-exact indices, including the reference's block-tree
 * tie-break and its |p|^2 <= 1e-3 skip.  `workspace` replaces the reference's `tmp` tensor
 * (B*N floats filled with 1e10): size it with spacap_fps_workspace_bytes; contents need no
 * initialisation. */
size_t spacap_fps_workspace_bytes(int B, int N);
int spacap_fps_f32(const float *xyz, int B, int N, int m, void *workspace, int32_t *idx,
                   spacap_stream_t stream);

/* gather_points(points f32[B,C,N], idx i32[B,m]) -> f32[B,C,m]   (src/sampling.cpp:15-38) */
int spacap_gather_points_f32(const float *points, const int32_t *idx, int B, int C, int N, int m,
                             float *out, spacap_stream_t stream);
/* gather_points_grad(grad_out f32[B,C,m], idx, n) -> f32[B,C,n]   (src/sampling.cpp:40-65);
 * grad_points is zero-filled here, then scatter-added. */
int spacap_gather_points_grad_f32(const float *grad_out, const int32_t *idx, int B, int C, int N,
                                  int m, float *grad_points, spacap_stream_t stream);

/* ---- ball query (replaces src/ball_query.cpp) ----------------------------------------------- */

/* ball_query(new_xyz f32[B,m,3], xyz f32[B,N,3], radius, nsample) -> i32[B,m,nsample]
 * (src/ball_query.cpp:8-32, kernel src/ball_query_gpu.cu:9-44): first `nsample` points in index
 * order with d2 < radius*radius, padded with the first hit, all-zero rows when there is none.
 * Every element of idx is written. */
int spacap_ball_query_f32(const float *new_xyz, const float *xyz, int B, int N, int m,
                          float radius, int nsample, int32_t *idx, spacap_stream_t stream);

/* Same contract and bit-identical output through a cell grid (points binned into cells >= radius wide, only the
 * 27 neighbouring cells of a centre are tested): for large clouds (SA1: 40 000 points).  workspace: device scratch of
 * spacap_ball_query_grid_workspace_bytes(B, N) bytes; N <= 131 000 (hit bitmap in LDS), radius > 0. */
size_t spacap_ball_query_grid_workspace_bytes(int B, int N);
int spacap_ball_query_grid_f32(const float *new_xyz, const float *xyz, int B, int N, int m, float radius, int nsample,
                               int32_t *idx, void *workspace, size_t workspace_bytes, spacap_stream_t stream);

/* ---- grouping (replaces src/group_points.cpp) ------------------------------------------------ */

/* group_points(points f32[B,C,N], idx i32[B,P,S]) -> f32[B,C,P,S]   (src/group_points.cpp:12-36) */
int spacap_group_points_f32(const float *points, const int32_t *idx, int B, int C, int N, int P,
                            int S, float *out, spacap_stream_t stream);
/* group_points_grad(grad_out f32[B,C,P,S], idx, n) -> f32[B,C,n]   (src/group_points.cpp:38-62).
 * With a workspace of spacap_group_points_grad_workspace_bytes() bytes (non-zero for C >= 16) the
 * index is inverted and the gradient gathered in ascending (centre, sample) order: no float atomics,
 * bitwise reproducible.  With workspace == NULL (or a zero size) grad_points is zero-filled and
 * scatter-added with atomics as the reference does.  Every element of grad_points is written. */
size_t spacap_group_points_grad_workspace_bytes(int B, int C, int N, int P, int S);
int spacap_group_points_grad_f32(const float *grad_out, const int32_t *idx, int B, int C, int N,
                                 int P, int S, float *grad_points, void *workspace,
                                 spacap_stream_t stream);

/* ---- max over the samples of a group (replaces F.max_pool2d(x, [1, nsample]),
 *      lib/pointnet2/pointnet2_modules.py:256-259) ------------------------------------------------ */

/* x f32 [rows, S] dense (rows = B*C*npoint, S = nsample <= 256) -> out f32 [rows], arg u8 [rows] (index of
 * the first maximum; a NaN wins, as in PyTorch's pooling). */
int spacap_group_max_f32(const float *x, long rows, int S, float *out, uint8_t *arg,
                         spacap_stream_t stream);
/* grad_in[row, s] = (s == arg[row]) ? grad_out[row] : 0; every element of grad_in f32 [rows, S] is written. */
int spacap_group_max_grad_f32(const float *grad_out, const uint8_t *arg, long rows, int S,
                              float *grad_in, spacap_stream_t stream);

/* ---- interpolation (replaces src/interpolate.cpp) -------------------------------------------- */

/* three_nn(unknown f32[B,n,3], known f32[B,m,3]) -> dist2 f32[B,n,3], idx i32[B,n,3]
 * (src/interpolate.cpp:14-40).  dist2 is SQUARED distance as in the reference. */
int spacap_three_nn_f32(const float *unknown, const float *known, int B, int n, int m,
                        float *dist2, int32_t *idx, spacap_stream_t stream);
/* The same search returning the feature-propagation modules' interpolation weights instead of the squared distances
 * (pointnet2_modules.py:399-405: dist = sqrt(dist2); r = 1 / (dist + 1e-8); weight = r / (r0 + r1 + r2)), weight f32 [B,n,3]. */
int spacap_three_nn_weights_f32(const float *unknown, const float *known, int B, int n, int m,
                                float *weight, int32_t *idx, spacap_stream_t stream);
/* out[b, j, :] = xyz[b, idx[b, j], :]: the sampled centres of a set-abstraction level (pointnet2_modules.py:232-239's
 * gather_operation on the flipped coordinates); xyz f32 [B,N,3], idx i32 [B,m] in [0,N), out f32 [B,m,3]. */
int spacap_gather_xyz_f32(const float *xyz, const int32_t *idx, int B, int N, int m, float *out, spacap_stream_t stream);
/* three_interpolate(points f32[B,C,m], idx i32[B,n,3], weight f32[B,n,3]) -> f32[B,C,n]
 * (src/interpolate.cpp:42-70) */
int spacap_three_interpolate_f32(const float *points, const int32_t *idx, const float *weight,
                                 int B, int C, int m, int n, float *out, spacap_stream_t stream);
/* three_interpolate_grad(grad_out f32[B,C,n], idx, weight, m) -> f32[B,C,m]
 * (src/interpolate.cpp:71-99); grad_points is zero-filled here, then scatter-added. */
int spacap_three_interpolate_grad_f32(const float *grad_out, const int32_t *idx,
                                      const float *weight, int B, int C, int n, int m,
                                      float *grad_points, spacap_stream_t stream);
/* The same gradient with POINT-MAJOR operands: grad_pm f32 [B,n,C] -> grad_points_pm f32 [B,m,C] (same summation
 * order and values as spacap_three_interpolate_grad_f32 on the transposed tensors; contiguous C-float rows instead
 * of one element per 4 KB row). */
int spacap_three_interpolate_grad_pm_f32(const float *grad_pm, const int32_t *idx, const float *weight, int B, int C,
                                         int n, int m, float *grad_points_pm, spacap_stream_t stream);
/* The input of a feature-propagation module's shared MLP in one launch each way (lib/pointnet2/pointnet2_modules.py:406-412:
   three_interpolate + torch.cat with the skip features): cat f32 [B,K1+K2,n] channel-major; known f32 [B,m,K1] (known_pm != 0:
   an SA module's point-major output) or [B,K1,m]; idx i32 [B,n,3], weight f32 [B,n,3]; skip f32 [B,n,K2] point-major; K1 % 32
   == 0.  Same values as three_interpolate (products added left to right).  _bwd: the gradient of cat split into its halves,
   both point-major: g1 f32 [B,n,K1] (feed spacap_three_interpolate_grad_pm_f32), g2 f32 [B,n,K2]. */
int spacap_fp_concat_fwd_f32(const float *known, int known_pm, const int32_t *idx, const float *weight, const float *skip, int B,
                             int K1, int K2, int m, int n, float *out, spacap_stream_t stream);
int spacap_fp_concat_bwd_f32(const float *g, int B, int K1, int K2, int n, float *g1, float *g2, spacap_stream_t stream);

/* ---- attention (replaces models/transformer_captioner.py:27-37) ------------------------------ */

/* One fused launch for  S = Q K^T * scale (+bias);  S[mask==0] = -1e9;  P = softmax(S);
 * P = dropout(P);  O = P V.   d_k must be 16, 32 or 64; Lk <= 512.
 *
 *  q,k,v   f32, logical shape [B,h,L,d_k], last dim contiguous, element strides (sb, sh, sl)
 *          (the reference passes transposed views of [B,L,h*d_k] projections,
 *           models/transformer_captioner.py:59-60)
 *  mask    uint8 or NULL, logical [B,Lq,Lk] with element strides (mask_sb, mask_sq), mask_sq = 0
 *          broadcasts one key mask over the queries (encoder), keys contiguous; 0 = masked
 *  bias    f32 or NULL, logical [B,h,Lq,Lk] added to the scaled logits before masking, element
 *          strides (bias_sb, bias_sh, bias_sq), keys contiguous.  The reference adds no bias
 *          (SURVEY.md fact 1); the argument exists for the spatial-relation bias of north_star.
 *  dropout_p in [0,1): keep-probability 1-p, kept entries scaled by 1/(1-p); the mask is a
 *          counter hash of (seed, b, h, q, k) so backward regenerates it.  seed_dev (nullable) points
 *          to a device-resident 64-bit word mixed into the seed at run time, so that a captured
 *          hipGraph draws a fresh mask on every replay (the host-side `seed` is frozen in the graph).
 *  out     f32 [B,Lq,h,d_k] dense (i.e. already in the layout of `x.transpose(1,2).contiguous()`,
 *          models/transformer_captioner.py:68)
 *  p_out   f32 [B,h,Lq,Lk] dense or NULL: the post-dropout attention matrix the reference returns
 *          as `p_attn` and stores as `self.attn` (models/transformer_captioner.py:63)
 *  stats   f32 [B,h,Lq,2] dense: (row max, row sum of exp) of the masked logits, consumed by backward
 *          (kept as a pair rather than one log-sum-exp: a fully masked row has max = -1e9, where
 *          max + log(sum) is not representable in fp32). */
int spacap_mha_fwd_f32(const float *q, const float *k, const float *v, long q_sb, long q_sh,
                       long q_sl, long k_sb, long k_sh, long k_sl, long v_sb, long v_sh, long v_sl,
                       const uint8_t *mask, long mask_sb, long mask_sq, const float *bias,
                       long bias_sb, long bias_sh, long bias_sq, int B, int h, int Lq, int Lk,
                       int d_k, float scale, float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                       float *out, float *p_out, float *stats, spacap_stream_t stream);

/* Backward of the above (two launches, no atomics, bitwise reproducible).  d_out f32 [B,Lq,h,d_k]
 * dense; d_p f32 [B,h,Lq,Lk] dense or NULL is the gradient w.r.t. the returned post-dropout p_attn
 * (non-NULL for the encoder layer that feeds the relation head,
 * models/transformer_captioner.py:392-394).  `workspace`: spacap_mha_bwd_workspace_bytes(B,h,Lq)
 * bytes of scratch.  Outputs dq f32 [B,Lq,h,d_k], dk,dv f32 [B,Lk,h,d_k] (the layout of the projections
 * before `.transpose(1,2)`); every element is written.  grad_row_stride: floats between consecutive rows
 * (b, position) of dq / dk / dv; 0 = dense (h*d_k).  A packed q|k|v projection (one GEMM) passes 3*h*d_k and
 * three pointers into one [B,L,3*h*d_k] buffer. */
size_t spacap_mha_bwd_workspace_bytes(int B, int h, int Lq);
int spacap_mha_bwd_f32(const float *q, const float *k, const float *v, long q_sb, long q_sh,
                       long q_sl, long k_sb, long k_sh, long k_sl, long v_sb, long v_sh, long v_sl,
                       const uint8_t *mask, long mask_sb, long mask_sq, const float *bias,
                       long bias_sb, long bias_sh, long bias_sq, int B, int h, int Lq, int Lk,
                       int d_k, float scale, float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                       const float *stats, const float *d_out, const float *d_p, void *workspace,
                       float *dq, float *dk, float *dv, long grad_row_stride, spacap_stream_t stream);

/* As spacap_mha_bwd_f32 with delta f32 [B,h,Lq] = sum_k p_attn d(p_attn) precomputed by the caller (without a gradient on
 * p_attn itself: sum_d out[b,q,head,d] d_out[b,q,head,d]; spacap_tf_rows_f32 emits it): the dQ and dK / dV halves are then
 * independent and run as ONE launch.  Self-attention shapes only (Lq, Lk both <= 64 or both > 64). */
int spacap_mha_bwd_delta_f32(const float *q, const float *k, const float *v, long q_sb, long q_sh,
                             long q_sl, long k_sb, long k_sh, long k_sl, long v_sb, long v_sh, long v_sl,
                             const uint8_t *mask, long mask_sb, long mask_sq, const float *bias,
                             long bias_sb, long bias_sh, long bias_sq, int B, int h, int Lq, int Lk,
                             int d_k, float scale, float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                             const float *stats, const float *d_out, const float *delta,
                             float *dq, float *dk, float *dv, long grad_row_stride, spacap_stream_t stream);

/* ---- train-mode BatchNorm + ReLU (+ max over the samples) of the shared MLPs ---------------------
 * (replaces BatchNorm2d -> ReLU [-> F.max_pool2d] of lib/pointnet2/pytorch_utils.py:11-36 and
 *  lib/pointnet2/pointnet2_modules.py:253-259, forward and backward).  z f32 [B,C,L] dense (L = npoint*nsample).
 * `workspace`: spacap_bn_workspace_bytes(C) bytes, shared by the forward-statistics and backward calls. */
size_t spacap_bn_workspace_bytes(int C);
/* process-wide: 1 (default) = small tensors (<= 32 768 elements per channel) take the one-launch kernels, 0 = always the
 * partial / final / apply form (identical results; for tests). */
int spacap_bn_set_single_launch(int enabled);
/* stats f32 [C,2] = (batch mean, 1/sqrt(biased var + eps)); running_mean/var (nullable pair) are updated with
 * `momentum` and the unbiased variance, as torch.nn.BatchNorm does. */
int spacap_bn_stats_f32(const float *z, int B, int C, long L, float eps, float momentum, float *running_mean,
                        float *running_var, float *stats, void *workspace, spacap_stream_t stream);
/* out[b,c,l] = relu((z - mean) * invstd * gamma + beta) */
/* spacap_bn_stats_f32 + spacap_bn_relu_apply_f32 in one call: one launch when a channel has <= 32 768 elements. */
int spacap_bn_relu_train_f32(const float *z, int B, int C, long L, float eps, float momentum, float *running_mean,
                             float *running_var, const float *gamma, const float *beta, float *stats, float *out,
                             void *workspace, spacap_stream_t stream);
int spacap_bn_relu_apply_f32(const float *z, const float *stats, const float *gamma, const float *beta, int B,
                             int C, long L, float *out, spacap_stream_t stream);
/* same followed by the max over the S samples of every (b,c,p): out f32 [B,C,P], arg u8 [B,C,P]; S in {16,32,64,128} */
int spacap_bn_relu_max_f32(const float *z, const float *stats, const float *gamma, const float *beta, int B, int C,
                           int P, int S, float *out, uint8_t *arg, spacap_stream_t stream);
/* backward of apply: dA f32 [B,C,L] -> dz f32 [B,C,L], dgamma, dbeta f32 [C] */
int spacap_bn_relu_bwd_f32(const float *z, const float *stats, const float *gamma, const float *beta,
                           const float *dA, int B, int C, long L, float *dz, float *dgamma, float *dbeta,
                           void *workspace, spacap_stream_t stream);
/* backward of apply+max: dP f32 [B,C,P], arg u8 [B,C,P] -> dz f32 [B,C,P*S], dgamma, dbeta f32 [C] */
int spacap_bn_relu_max_bwd_f32(const float *z, const float *stats, const float *gamma, const float *beta,
                               const float *dP, const uint8_t *arg, int B, int C, int P, int S, float *dz,
                               float *dgamma, float *dbeta, void *workspace, spacap_stream_t stream);

/* ---- pairwise relation feature (replaces models/transformer_captioner.py:393-396) ---------------
 * R[b,i,j,h*D+d] = P[b,h,i,j] * V[b,h,j,d].  P f32 [B,H,K,K] dense (the post-dropout attention matrix of the
 * last encoder layer), V f32 logical [B,H,K,D] with element strides (v_sb, v_sh, v_sl), last dim contiguous;
 * R f32 [B,K,K,H*D] dense.  D a power of two in 4..64, H*D <= 1024 with 256 % (H*D/4) == 0. */
int spacap_relation_feature_fwd_f32(const float *P, const float *V, long v_sb, long v_sh, long v_sl, int B,
                                    int H, int K, int D, float *R, spacap_stream_t stream);
/* dR f32 [B,K,K,H*D] -> dP f32 [B,H,K,K] dense, dV f32 [B,K,H,D] dense (the layout of the value projection
 * before `.transpose(1,2)`); every element written, fixed summation order. */
int spacap_relation_feature_bwd_f32(const float *dR, const float *P, const float *V, long v_sb, long v_sh,
                                    long v_sl, int B, int H, int K, int D, float *dP, float *dV,
                                    spacap_stream_t stream);

/* First layer of the relation MLP fused with the feature (models/transformer_captioner.py:319-326,393-397):
 *   H1[b,i,j,o] = relu(b1[o] + sum_h P[b,h,i,j] * U[b,j,h,o]),  U[b,j,h,o] = sum_d V[b,h,j,d] * W1[o,h*D+d]
 * (= relu(R W1^T + b1) without ever forming R).  P f32 [B,H,K,K], U f32 [B,K,H,C], b1 f32 [C], H1 f32 [B,K,K,C],
 * all dense; H in {4,8,16,32}, C in {16,32,64,128,256}. */
int spacap_relation_l1_fwd_f32(const float *P, const float *U, const float *b1, int B, int H, int K, int C,
                               float *H1, spacap_stream_t stream);
/* dH1, H1 f32 [B,K,K,C] -> dP f32 [B,H,K,K]; dU_part f32 [spacap_relation_l1_isplit(), B,K,H,C] and
 * db_part f32 [spacap_relation_l1_blocks(B,K,C), C] are partial sums (per query chunk / per workgroup) that the
 * caller adds up in order. */
int spacap_relation_l1_isplit(void);
int spacap_relation_l1_supported(int H, int K, int C); /* 1 when relation_l1_fwd/bwd have a kernel for this shape */
int spacap_relation_l1_blocks(int B, int K, int C);
int spacap_relation_l1_bwd_f32(const float *dH1, const float *H1, const float *P, const float *U, int B, int H,
                               int K, int C, float *dP, float *dU, float *db_part, spacap_stream_t stream);

/* The whole relation head in one launch each way (models/transformer_captioner.py:319-326 relation_proposal, :392-397 the
 * pair feature): hid1 = relu(b1 + sum_h P U) as above, hid2 = relu(hid1 W2^T + b2), pred = hid2 W3^T + b3, for H = 8,
 * C = 128, 9 outputs and K a multiple of 8 (spacap_relation_fused_supported).  hid1 is never stored; hid2 f32 [B,K,K,128]
 * is the backward's only large input.  W2 f32 [128,128] (out, in), W3 f32 [9,128], pred f32 [B,K,K,9].
 * Each launch is one workgroup per CU not reserved by spacap_sa_reserve_cus, every workgroup with a contiguous range of
 * (scene, key block, query block) tiles.
 * Backward: dpred -> dP f32 [B,H,K,K]; part f32 [nparts, spacap_relation_fused_part_floats()] per-workgroup partial sums laid
 * out dW2 [128*128] | dW3 [9*128] | db1 [128] | db2 [128] | db3 [16, 9 used]; dU f32 [zslots, B,K,H,C] partial sums (a key
 * block's query range can straddle workgroups; unused slots are written as zeros).  The caller adds both up in order.
 * nparts = spacap_relation_fused_nparts(B, K) (the grid; depends on the reserved CUs at the time of the call) and
 * zslots = spacap_relation_fused_zsplit(B, K, nparts) are passed back in so that buffers and launch agree. */
/* CUs the fused relation head's persistent grids leave free in addition to spacap_sa_reserve_cus (process-wide, 0 by default): for a
 * caller that runs the caption decoder on another stream beside the head.  0 <= n <= CUs / 2. */
int spacap_relation_fused_leave_cus(int n);
int spacap_relation_fused_supported(int H, int K, int C, int n_out);
int spacap_relation_fused_nparts(int B, int K);
int spacap_relation_fused_zsplit(int B, int K, int nparts);
int spacap_relation_fused_part_floats(void);
int spacap_relation_fused_fwd_f32(const float *P, const float *U, const float *b1, const float *W2, const float *b2,
                                  const float *W3, const float *b3, int B, int K, float *hid2, float *pred,
                                  spacap_stream_t stream);
int spacap_relation_fused_bwd_f32(const float *dpred, const float *hid2, const float *P, const float *U, const float *b1,
                                  const float *W2, const float *W3, int B, int K, int nparts, int zslots, float *dP,
                                  float *dU, float *part, spacap_stream_t stream);

/* ---- decoder input of the captioner's training step (replaces models/transformer_captioner.py:350-367, 246-249, 129-137,
 * 150-161, 193-199 as separate tensor operations) ---------------------------------------------------------------------------
 * xyz f32 [B,K,3] proposal centres, ref f32 [B,3] referred object's centre, src / memory f32 [B,K,D] proposal features and
 * encoder output (memory may be NULL), tok i64 [B,T] label tokens, emb f32 [V,D], pe f32 [>= T-2, D]; L = T - 1.
 *   idx i64 [B] = nearest proposal (squared distance, first minimum), dist f32 [B], good u8 [B] = dist > -1,
 *   pred f32 [1] = sum(dist * good) / max(1, sum good);
 *   x0 f32 [B,L,D]: row 0 = src[b,idx] + memory[b,idx], row 1+t = dropout_p(emb[tok[b,1+t]] * sqrt(D) + pe[t]);
 *   mask u8 [B,L,L] = tok[b,k] > 0 and k <= q.
 * counter i32 [1]: the call's own last-block ticket, any content (zeroed on the stream by the entry point).  Dropout: counter hash of (seed, *seed_dev, element). */
int spacap_caption_prep_fwd_f32(const float *xyz, const float *ref, const float *src, const float *memory,
                                const int64_t *tok, const float *emb, const float *pe, int B, int K, int D, int T, int V,
                                float p, uint64_t seed, const uint64_t *seed_dev, float *x0, uint8_t *mask, int64_t *idx,
                                float *dist, uint8_t *good, float *pred, int32_t *counter, spacap_stream_t stream);
/* g f32 [B,L,D] -> d_rows f32 [B,K,D] (the gradient of src and of memory: row idx[b] = g[b,0], zero elsewhere; may be NULL)
 * and d_emb f32 [V,D] (dense, summed in (scene, position) order). */
int spacap_caption_prep_bwd_f32(const float *g, const int64_t *tok, const int64_t *idx, int B, int K, int D, int T, int V,
                                float p, uint64_t seed, const uint64_t *seed_dev, float *d_rows, float *d_emb,
                                spacap_stream_t stream);

/* ---- LayerNorm of the Transformer (replaces models/transformer_captioner.py:102-113) ----------- */

/* y = a * (x - mean) / (std_unbiased + eps) + b over the last dimension; x,y f32 [rows, D] dense, a,b f32 [D];
 * stats f32 [rows, 2] = (mean, 1 / (std + eps)) is kept for backward. */
int spacap_layernorm_fwd_f32(const float *x, const float *a, const float *b, long rows, int D, float eps,
                             float *y, float *stats, spacap_stream_t stream);
/* dx f32 [rows, D], da, db f32 [D] (every element written; fixed summation order, no atomics). */
size_t spacap_layernorm_bwd_workspace_bytes(long rows, int D);
int spacap_layernorm_bwd_f32(const float *x, const float *a, const float *stats, const float *dy, long rows,
                             int D, float eps, float *dx, float *da, float *db, void *workspace,
                             spacap_stream_t stream);
/* (da and db may both be null: the caller then adds up the workgroup partials left in `workspace`, f32 [blocks][2*D]
 * with blocks = spacap_layernorm_bwd_workspace_bytes(rows, D) / (8*D), rows of [da | db] -- e.g. batched with other
 * slab sums.)
 * Same, plus an addend: dx = (LayerNorm backward of dy) + addend (f32, x's shape; may be null).  For the pre-norm
 * residual x + f(norm(x)) (models/transformer_captioner.py:115-123), whose output gradient reaches x twice. */
int spacap_layernorm_bwd_add_f32(const float *x, const float *a, const float *stats, const float *dy, const float *addend,
                                 long rows, int D, float eps, float *dx, float *da, float *db, void *workspace,
                                 spacap_stream_t stream);

/* ---- Shared MLP of a set-abstraction module, point-major layout, training mode --------------------------------
 * Replaces the per-module chain QueryAndGroup -> SharedMLP([Conv2d 1x1 -> BatchNorm2d -> ReLU] x 3) -> max_pool2d
 * (lib/pointnet2/pointnet2_modules.py:241-259, lib/pointnet2/pytorch_utils.py:11-36) for the training step.
 * Rows r = (b, centre n, sample s), R = B*N*S; activations z_k are f32 [R, C_k] dense; G = B*N groups.
 * stats f32 [C,4] = (mean, 1/sqrt(var+eps), gamma/sqrt(var+eps), beta); coef f32 [C,4] = (g, k0, k1, -) with
 * dz = g*dy + k0 - k1*z.  part: f64 [spacap_sa_nparts(), 2, C] partial sums (workspace, fully overwritten). */
int spacap_sa_nparts(void);
int spacap_sa_wgrad_slabs(long R, int CK, int CP, int pooled);
/* 1 when (C1, C2, C3) has kernels: (64,64,128), (128,128,128), (128,128,256). */
int spacap_sa_mlp_supported(int C1, int C2, int C3);
/* z1[r,:] = Y[b,idx[r],:] + W1[:,0:3] (xyz[b,idx[r]] - new_xyz[b,n]) / rdiv + W1[:,3] feat[b,idx[r]].
 * Y f32 [B,Np,C1] or NULL, feat f32 [B,Np] or NULL, xyz f32 [B,Np,3], new_xyz f32 [B,N,3], idx i32 [B,N,S],
 * W1 f32 rows of ldw floats; C1 in {64,128}. */
int spacap_sa_l1_fwd_f32(const float *Y, const float *feat, const float *xyz, const float *new_xyz,
                         const int32_t *idx, const float *W1, int ldw, float rdiv, int B, int Np, int N, int S,
                         int C1, float *z1, double *part, spacap_stream_t stream);
/* part -> stats; running_mean / running_var (may be NULL) get torch's momentum update (unbiased variance). */
int spacap_sa_bn_finalize_f32(const double *part, int C, long count, float eps, float momentum, const float *gamma,
                              const float *beta, float *running_mean, float *running_var, float *stats,
                              spacap_stream_t stream);
/* zout = relu(bn(zin)) W^T, W f32 [Cout, Cin]; part receives the sums of zout. */
int spacap_sa_mid_fwd_f32(const float *zin, const float *st_in, const float *W, long R, int Cin, int Cout,
                          float *zout, double *part, spacap_stream_t stream);
/* The same layer when its output is max-pooled over groups of S consecutive rows (the LAST layer of an SA module's shared
   MLP, pointnet2_modules.py:258-259): additionally leaves, per sub-group of min(S,32) rows and channel, the two best
   pooling candidates (cand_v f32, cand_i u8: [R / min(S,32)][Cout][2]); spacap_sa_pool_finalize_f32 turns them into
   (out, arg) once this layer's batch statistics are final, so that the pooling pass never reads z_out.
   gamma_out = BatchNorm weight of this layer's output.  _supported: 1 when there is a kernel for (Cin, Cout, S). */
/* out[R][Cout] = x[R][Cin] W[Cout][Cin]^T (fp32 in / out / accumulate) on the streaming split-bf16 kernel: the relation
   head's dhid1 = dz2 W2 (backward of models/transformer_captioner.py:319-326).  _supported: 1 when there is a kernel. */
int spacap_gemm_rows_supported(int Cin, int Cout);
int spacap_gemm_rows_f32(const float *x, const float *W, long R, int Cin, int Cout, float *out, spacap_stream_t stream);
int spacap_sa_mid_fwd_pool_supported(int Cin, int Cout, int S);
/* n CUs (0..64) are left free by the forward layer kernels' persistent grids: for callers that run other kernels (the next
   batch's sampling chain) beside the forward pass, whose workgroups would otherwise push the grid's last ones into a second
   round. */
int spacap_sa_reserve_cus(int n);
int spacap_sa_mid_fwd_pool_f32(const float *zin, const float *st_in, const float *W, const float *gamma_out, long R,
                               int Cin, int Cout, int S, float *zout, double *part, float *cand_v, uint8_t *cand_i,
                               spacap_stream_t stream);
/* (zout of spacap_sa_mid_fwd_pool_f32 may be NULL: the layer's output is then not stored at all.)  zmax (nullable) f32 [G,C]: the pre-activation of the arg-max row, what
   the pooled layer's BatchNorm backward needs of z when z is not stored. */
int spacap_sa_pool_finalize_f32(const float *cand_v, const uint8_t *cand_i, const float *stats, const float *gamma, long G,
                                int S, int C, float *out, uint8_t *arg, float *zmax, spacap_stream_t stream);
/* out[g,c] = max_s relu(bn(z[g*S+s,c])) (first maximum), arg u8 [G,C]. */
int spacap_sa_pool_fwd_f32(const float *z, const float *stats, long G, int S, int C, float *out, uint8_t *arg,
                           spacap_stream_t stream);
/* dym = (out > 0) ? dout : 0; part receives (sum dy, sum dy*xhat) of the pooled layer.  xhat at the arg-max rows comes from
   zmax [G,C] (spacap_sa_pool_finalize_f32) when given, else from z [G*S,C] (a 4-byte gather per element); with both, z serves
   only the channels whose BatchNorm weight is exactly 0 (every row ties there and zmax is not the arg-max row's value). */
int spacap_sa_pool_bwd_f32(const float *dout, const float *out, const uint8_t *arg, const float *z, const float *zmax,
                           const float *stats, long G, int S, int C, float *dym, double *part,
                           spacap_stream_t stream);
/* Weight gradient of a POOLED last layer of a shared MLP from its INPUT pre-activation z2 alone (lib/pointnet2/pytorch_utils.py:11-36
   Conv2d -> BatchNorm2d -> ReLU, lib/pointnet2/pointnet2_modules.py:256-259 max_pool2d; autograd backward).  z3 = a2 W3^T is linear
   in the layer's input a2 = relu(bn(z2)), so with dz3 = g d + k0 - k1 z3 (coef3 = (g, k0, k1) rows, spacap_sa_bwd_finalize_f32)
     dW3 = (g d)^T a2 + k0 (x) colsum(a2) - diag(k1) W3 (a2^T a2)
   -- z3 is not read (replaces spacap_sa_wgrad_f32 with arg != NULL, which streams z3 and z2).  spacap_sa_wgrad_pool_f32 leaves partial
   sums of (g d)^T a2, a2^T a2 and colsum(a2) per workgroup, partW f32 [_parts][spacap_sa_l3bwd_part_floats(C2, C3)];
   spacap_sa_l3bwd_dw_f32 combines them into dW3 f32 [C3,C2] (sums = double scratch of _part_floats entries).
   _supported: (C2, C3, S) in {(64,128,64), (128,128,32), (128,256,32)}.  arg 4-byte, z2 and dym 16-byte aligned. */
long spacap_sa_l3bwd_part_floats(int C2, int C3);
int spacap_sa_l3bwd_dw_f32(const float *partW, int nparts, const float *coef3, const float *W3, int C3, int C2, double *sums,
                           float *dW3, spacap_stream_t stream);
int spacap_sa_wgrad_pool_supported(int C2, int C3, int S);
int spacap_sa_wgrad_pool_parts(long R, int C2, int C3, int S);
int spacap_sa_wgrad_pool_f32(const float *dym, const uint8_t *arg, int S, const float *coef3, const float *z2, const float *st2,
                             long R, int C3, int C2, float *partW, spacap_stream_t stream);
/* part -> coef, dgamma, dbeta of a layer (stats = that layer's forward statistics). */
int spacap_sa_bwd_finalize_f32(const double *part, int C, long count, const float *stats, float *coef,
                               float *dgamma, float *dbeta, spacap_stream_t stream);
/* dy_prev = (dz_k W_k) * [relu'(bn(z_prev))], W_k f32 [CK, CP]; dy = dense [R,CK] gradient when arg == NULL, else
 * the masked pooled gradient [R/S, CK] with its arg-max map.  part receives the BN sums of dy_prev. */
int spacap_sa_dgrad_f32(const float *dy, const uint8_t *arg, int S, const float *zk, const float *coef,
                        const float *Wk, const float *zp, const float *st_p, long R, int CK, int CP, float *dyp,
                        double *part, spacap_stream_t stream);
/* SA1 form of the layer-2 data gradient (first layer = 3 relative coordinates + one inline feature, CK = CP = 64):
 * dy_prev is not written; part_l1 f32 [spacap_sa_nparts()][64*8+4] receives, per workgroup, S1[c,0:4] = sum dy_prev
 * in_d, S3[c,0:4] = sum z_prev in_d (8 floats per channel) and S2[0:4] = sum in_d, from which
 * dW1[c,d] = g_c S1[c,d] + k0_c S2[d] - k1_c S3[c,d] with the layer-1 coef row (g, k0, k1). */
int spacap_sa_dgrad_l1_f32(const float *dy, const float *zk, const float *coef, const float *Wk, const float *zp,
                           const float *st_p, const float *feat, const float *xyz, const float *new_xyz,
                           const int32_t *idx, float rdiv, int B, int Np, int N, int S, int CK, int CP, double *part,
                           float *part_l1, spacap_stream_t stream);
/* dW1 of spacap_sa_dgrad_l1[in]_f32's three sums: part_l1 f32 [nparts, C1*8+4], coef f32 [C1,4] (g, k0, k1 of layer 1) ->
 * dW1 f32 [C1, ldw] (ldw <= 4 columns: relative x, y, z, inline feature), partials added in double. */
int spacap_sa_l1_dw_f32(const float *part_l1, int nparts, const float *coef, int C1, int ldw, float *dW1, spacap_stream_t stream);

/* The SA1 family without z1 in HBM (first layer: 3 relative coordinates + at most one inline feature, no point features;
 * C1 = C2 = 64; lib/pointnet2/pointnet2_modules.py:241-259 with pytorch_utils.py:11-36).  spacap_sa_l1_stats_f32 is
 * spacap_sa_l1_fwd_f32 without the z1 store: the BatchNorm sums in `part` and rel4 f32 [R,4] = each grouped row's inputs
 * (x, y, z relative to the centre / rdiv, inline feature or 0).  The *_l1in_* entries are spacap_sa_mid_fwd_f32 /
 * spacap_sa_wgrad_f32 / spacap_sa_dgrad_l1_f32 with z_prev = W1 in rebuilt from rel4 (W1 f32 [64, ldw], has_feat: column 3
 * is used) with the statistics pass's own arithmetic, bit for bit. */
int spacap_sa_l1_stats_f32(const float *feat, const float *xyz, const float *new_xyz, const int32_t *idx, const float *W1,
                           int ldw, float rdiv, int B, int Np, int N, int S, int C1, float *rel4, double *part,
                           spacap_stream_t stream);
/* The same pass in closed form: z1 is linear in the row's four inputs (relative x, y, z, inline feature), so the first layer's
   batch statistics follow from the first and second moments of those inputs: mom f64 [spacap_sa_nparts()][16] (4 sums, 10
   products, 2 pads per workgroup) next to rel4; _finalize turns them into stats [C1,4] (mean, 1/std, gamma/std, beta) and
   updates the running statistics (either pointer NULL: not tracked), as spacap_sa_bn_finalize_f32 does from (sum, sum of squares). */
int spacap_sa_l1_moments_f32(const float *feat, const float *xyz, const float *new_xyz, const int32_t *idx, float rdiv, int B, int Np,
                             int N, int S, float *rel4, double *mom, spacap_stream_t stream);
int spacap_sa_l1_moments_finalize_f32(const double *mom, const float *W1, int ldw, int has_feat, int C1, long count, float eps,
                                      float momentum, const float *gamma, const float *beta, float *running_mean, float *running_var,
                                      float *stats, spacap_stream_t stream);
int spacap_sa_mid_fwd_l1in_f32(const float *rel4, const float *W1, int ldw, int has_feat, const float *st_in, const float *W,
                               long R, float *zout, double *part, spacap_stream_t stream);
int spacap_sa_wgrad_l1in_f32(const float *dy, const float *zk, const float *coef, const float *rel4, const float *W1, int ldw,
                             int has_feat, const float *st_p, long R, float *partW, spacap_stream_t stream);
int spacap_sa_dgrad_l1in_f32(const float *dy, const float *zk, const float *coef, const float *Wk, const float *rel4,
                             const float *W1, int ldw, int has_feat, const float *st_p, int B, int N, int S, double *part,
                             float *part_l1, spacap_stream_t stream);
/* spacap_sa_dgrad_l1in_f32 + the layer's weight-gradient partials from the same pass (what spacap_sa_wgrad_l1in_f32 computes from a
 * second read of dy and zk): partW f32 [spacap_sa_dgrad_wgrad_l1in_slabs(R)][64][64], summed by the caller in slab order. */
int spacap_sa_dgrad_wgrad_l1in_slabs(long R);
int spacap_sa_dgrad_wgrad_l1in_f32(const float *dy, const float *zk, const float *coef, const float *Wk, const float *rel4,
                                   const float *W1, int ldw, int has_feat, const float *st_p, int B, int N, int S, double *part,
                                   float *part_l1, float *partW, spacap_stream_t stream);
/* partW f32 [spacap_sa_wgrad_slabs(R,CK,CP,arg != NULL), CK, CP]: per-slab partial sums of dW_k = dz_k^T relu(bn(z_prev)). */
int spacap_sa_wgrad_f32(const float *dy, const uint8_t *arg, int S, const float *zk, const float *coef,
                        const float *zp, const float *st_p, long R, int CK, int CP, float *partW,
                        spacap_stream_t stream);
/* dy1 <- dz1 in place when write_dz != 0; partW f32 [spacap_sa_nparts(), C1, 4] partial dW1 (rel x, y, z, inline feature);
 * drel f32 [R,3] (d loss / d (xyz[idx] - new_xyz)) or NULL. */
int spacap_sa_l1_bwd_f32(float *dy1, const float *z1, const float *coef, const float *feat, const float *xyz,
                         const float *new_xyz, const int32_t *idx, const float *W1, int ldw, float rdiv, int B,
                         int Np, int N, int S, int C1, float *partW, float *drel, int write_dz,
                         spacap_stream_t stream);
/* out[b,p,:] = sum over rows r = (b,e), e < E, with idx[b,e] = p of dz[r,:] in ascending r (inverted index). */
size_t spacap_sa_rows_scatter_workspace_bytes(int B, int Np, long E);
int spacap_sa_rows_scatter_f32(const float *dz, const int32_t *idx, int B, int Np, long E, int C, float *out,
                               void *workspace, spacap_stream_t stream);
/* The two halves of spacap_sa_rows_scatter_f32: the inverted index depends on idx only (= on the input coordinates), so
 * it can be built ahead of the step; the gather then reads it from the same workspace. */
int spacap_sa_rows_index_f32(const int32_t *idx, int B, int Np, long E, void *workspace, spacap_stream_t stream);
int spacap_sa_rows_gather_f32(const float *dz, int B, int Np, long E, int C, const void *workspace, float *out,
                              spacap_stream_t stream);
/* Gradient of the grouped relative coordinates (lib/pointnet2/pointnet2_utils.py:350-355 QueryAndGroup: grouped_xyz -= new_xyz, /= radius;
   autograd backward) routed to both sources in one launch: drel f32 [B, N*S, 3] -> dxyz f32 [B, Np, 3] (sum over the rows
   referencing each source point, ascending row order, from the index spacap_sa_rows_index_f32 left in workspace) and
   dnew f32 [B, N, 3] = -sum over each group's S rows.  Either output may be NULL (workspace may be NULL without dxyz). */
int spacap_sa_drel_sums_f32(const float *drel, int B, int Np, int N, int S, const void *workspace, float *dxyz, float *dnew,
                            spacap_stream_t stream);
/* First-layer weight gradient dW1 f32 [C1, 3+Cf] of a set-abstraction module with point features from its two sets of partial
   results: pw1 f32 [n1][C1][4] (spacap_sa_l1_bwd_f32: relative coordinates) and pf f32 [nf][C1*Cf] (spacap_linear_wgrad_f32 of
   the feature product); same values as spacap_sum_slabs_f32 on each followed by a concatenation. */
int spacap_sa_dw1_assemble_f32(const float *pw1, int n1, const float *pf, int nf, int C1, int Cf, float *dW1, spacap_stream_t stream);

/* ---- Linear layers of the Transformer: weight + bias gradient in one launch ------------------------------------
 * dW[ck,cp] = sum_r g[r,ck] x[r,cp], db[ck] = sum_r g[r,ck]  (backward of torch.nn.Linear as used by
 * models/transformer_captioner.py:63-99,117-126).  g f32 [R,CK], x f32 [R,CP] dense, CK and CP multiples of 128.
 * part f32 [spacap_linear_wgrad_slabs(R,CK,CP)][CK*CP (+CK when with_bias)]: per-slab partial sums (weights, then
 * bias) that the caller adds up in slab order.  spacap_linear_wgrad_slabs returns 0 for shapes without a kernel. */
int spacap_linear_wgrad_slabs(long R, int CK, int CP);
int spacap_linear_wgrad_f32(const float *g, const float *x, long R, int CK, int CP, int with_bias, float *part,
                            spacap_stream_t stream);
/* ... with the number of row slabs chosen by the caller: part f32 [nslab][CK*CP (+ CK when with_bias)], 1 <= nslab <= ceil(R / 32). */
int spacap_linear_wgrad_nslab_f32(const float *g, const float *x, long R, int CK, int CP, int with_bias, int nslab, float *part,
                                  spacap_stream_t stream);
/* njobs independent weight gradients in ONE launch; part[i] receives nslabs[i] partial results (with nslabs[i] =
 * spacap_linear_wgrad_slabs(...) the values of spacap_linear_wgrad_f32; other counts regroup the rows).  All
 * arrays are HOST arrays read before the call returns; the job table is passed to the kernel by value (capturable in a
 * hipGraph).  For the end of a backward pass: only the optimizer reads these gradients. */
int spacap_linear_wgrad_slabs_batched(long R, int CK, int CP);   /* recommended slabs per job inside a batch */
int spacap_linear_wgrad_batched_f32(const float *const *g, const float *const *x, const long *R, const int *CK,
                                    const int *CP, const int *with_bias, const int *nslabs, float *const *part, int njobs,
                                    spacap_stream_t stream);

/* ---- fused elementwise pieces of the Transformer sublayers (csrc/elementwise.hip) -----------------------------
 * Dropout as torch.nn.Dropout (keep with probability 1-p, kept values scaled by 1/(1-p)); the keep mask is a
 * counter hash of (seed + *seed_dev, element index), regenerated in the backward.  All tensors dense f32 [n],
 * 16-byte aligned.
 *   relu_dropout:  out = dropout(relu(x))      (models/transformer_captioner.py:126)
 *                  dx  = (y > 0) ? g/(1-p) : 0 with y the saved forward output
 *   dropout_add:   out = res + dropout(y)      (models/transformer_captioner.py:115-123)
 *                  dy  = keep ? g/(1-p) : 0    (same seed words as the forward call) */
int spacap_relu_dropout_fwd_f32(const float *x, long n, float p, uint64_t seed, const uint64_t *seed_dev,
                                float *out, spacap_stream_t stream);
int spacap_relu_dropout_bwd_f32(const float *g, const float *y, long n, float p, float *dx, spacap_stream_t stream);
int spacap_dropout_add_fwd_f32(const float *res, const float *y, long n, float p, uint64_t seed,
                               const uint64_t *seed_dev, float *out, spacap_stream_t stream);
int spacap_dropout_add_bwd_f32(const float *g, long n, float p, uint64_t seed, const uint64_t *seed_dev,
                               float *out, spacap_stream_t stream);

/* ---- input pipeline: subsample + augmentation + vote labels of HBM-resident scenes (csrc/scene_pipeline.hip) -----
 * Replaces the per-point numpy work of ScannetReferenceDataset.__getitem__ (lib/dataset.py:335-338, 364-404, 415-428).
 * scene_feat / scene_ins / scene_isobj: DEVICE arrays of B device pointers (one per batch item) to that item's scene:
 * f32 [N_b, C] rows (xyz first), i32 [N_b] instance labels, u8 [N_b] "semantic label is one of the 37 object classes".
 * choices i32 [B,P]: sampled row per output point.  aug f64 [B, spacap_scene_aug_doubles()]: flip_x, flip_y (non-zero
 * = flip), Rx[9], Ry[9], Rz[9] (row-major rotation matrices applied as x' = x R^T in that order), t[3]; float64
 * arithmetic rounded to float32 after every step, as numpy does.  Outputs pc f32 [B,P,C], ins_out i32 [B,P],
 * isobj_out u8 [B,P]; optionally (scene_color: B pointers to f32 [N_b,3], color_out f32 [B,P,3]) the sampled colours. */
int spacap_scene_aug_doubles(void);
int spacap_scene_sample_augment_f32(const float *const *scene_feat, const int32_t *const *scene_ins,
                                    const uint8_t *const *scene_isobj, const float *const *scene_color,
                                    const int32_t *choices, const double *aug, int B, int P, int C, int augment,
                                    float *pc, int32_t *ins_out, uint8_t *isobj_out, float *color_out,
                                    spacap_stream_t stream);
/* Same with output rows of C_out >= C channels: source channel c >= 3 is written to channel dst_off[c] (device array of C
 * ints, NULL = identity).  The columns in between -- colour (lib/dataset.py:312-315) and multiview (:321-328) -- are filled
 * by spacap_scene_gather_rows_f32: out[b,p,out_off+j] = src[b][choices[b,p]*W + j], j < W, out rows of out_stride floats. */
int spacap_scene_sample_augment_map_f32(const float *const *scene_feat, const int32_t *const *scene_ins,
                                        const uint8_t *const *scene_isobj, const float *const *scene_color,
                                        const int32_t *choices, const double *aug, int B, int P, int C, int augment,
                                        int C_out, const int *dst_off, float *pc, int32_t *ins_out, uint8_t *isobj_out,
                                        float *color_out, spacap_stream_t stream);
int spacap_scene_gather_rows_f32(const float *const *src, const int32_t *choices, int B, int P, int W, float *out,
                                 int out_stride, int out_off, spacap_stream_t stream);
/* votes f32 [B,P,9] (three identical votes: centre of the instance's sampled points - point), vmask i64 [B,P]; an
 * instance votes iff isobj of its FIRST sampled point is set; instance labels outside [0, max_inst) never vote. */
size_t spacap_scene_votes_workspace_bytes(int B, int max_inst);
int spacap_scene_votes_f32(const float *pc, const int32_t *ins, const uint8_t *isobj, int B, int P, int C,
                           int max_inst, void *workspace, float *votes, int64_t *vmask, spacap_stream_t stream);

/* ---- optimizer: torch.optim.Adam (scripts/train.py:262) over one flat parameter buffer, one launch ---------------
 * g' = grad_scale*g + weight_decay*p;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
 * p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),  t = *step (device f32, 1-based).  All f32 [n], 16-byte aligned.
 * skip_if_nonzero: NULL, or a device int64 that turns the whole update off when it is non-zero at launch (the sticky error
 * word of spacap_stream_wait_ge: a gradient that a timed-out wait may have let through half-written is never applied). */
int spacap_adam_flat_f32(float *p, const float *g, float *m, float *v, long n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, const float *step, float grad_scale,
                         const int64_t *skip_if_nonzero, spacap_stream_t stream);

/* ---- detection losses of the training step (csrc/losses.hip) ----------------------------------------------------
 * Replaces compute_vote_loss / compute_objectness_loss / compute_box_and_sem_cls_loss (lib/loss_helper.py:35-197,
 * utils/nn_distance.py:32-62).  net f32 [B,K,CH] = the proposal head's output rows [objectness 2 | centre offset 3 |
 * heading scores NH | heading residuals NH | size scores NS | size residuals NS*3 | class scores NC]; center, agg_xyz
 * f32 [B,K,3]; gt_center f32 [B,M,3] (M <= 256); box_mask f32 [B,M]; *_label per ground-truth box; seed_xyz, vote_xyz
 * f32 [B,NSEED,3]; seed_inds i32 [B,NSEED]; vote_label f32 [B,N,9]; vote_mask i64 [B,N].
 * Outputs: obj_label i64 / obj_mask f32 / assignment i64 [B,K]; losses f32 [8] (vote, objectness, center, heading_cls,
 * heading_reg, size_cls, size_reg, sem_cls); gradient numerators dnet_num [B,K,CH], dcenter_num [B,K,6], dvote_num
 * [B,NSEED,3]; part f32 [B*(spacap_det_npart()+2)] scratch; inv_den f32 [4].  The backward turns the numerators and
 * the upstream gradient grad_losses f32 [8] into dnet [B,K,CH], dcenter [B,K,3], dvote [B,NSEED,3]. */
int spacap_det_npart(void);
int spacap_det_losses_fwd_f32(const float *net, const float *center, const float *agg_xyz, const float *gt_center,
                              const float *box_mask, const int64_t *heading_cls_label, const float *heading_res_label,
                              const int64_t *size_cls_label, const float *size_res_label, const int64_t *sem_cls_label,
                              const float *mean_size, const float *seed_xyz, const float *vote_xyz,
                              const int32_t *seed_inds, const float *vote_label, const int64_t *vote_mask, int B, int K,
                              int M, int NSEED, int N, int NH, int NS, int NC, float near_thr, float far_thr, float w0,
                              float w1, int64_t *obj_label, float *obj_mask, int64_t *assignment, float *dnet_num,
                              float *dcenter_num, float *dvote_num, float *part, float *losses, float *inv_den,
                              spacap_stream_t stream);
int spacap_det_losses_bwd_f32(const float *dnet_num, const float *dcenter_num, const float *dvote_num,
                              const float *grad_losses, const float *inv_den, int B, int K, int NSEED, int NH, int NS,
                              int NC, float *dnet, float *dcenter, float *dvote, spacap_stream_t stream);

/* Relation loss (lib/loss_helper.py:240-289).  pred f32 [B,K,K,9] = [x 3 | y 3 | z 3 logits]; assignment, obj_label
 * i64 [B,K]; box_mask_int i64 [B,M]; x/y/z_label i64 [B,M,M].  A pair (i,j) counts iff both proposals are positive and
 * assigned to valid boxes; its label is rel_a[b, assignment_i, assignment_j].  out f32 [7] = x,y,z loss, x,y,z accuracy,
 * 1/max(#pairs,1); dnum f32 [B,K,K,9] gradient numerators; part f32 [spacap_rel_loss_nparts(B,K)*7] scratch.  The
 * backward maps the upstream gradients of the three losses (grad_losses f32 [3]) to dpred. */
long spacap_rel_loss_nparts(int B, int K);
int spacap_rel_loss_fwd_f32(const float *pred, const int64_t *assignment, const int64_t *box_mask_int,
                            const int64_t *obj_label, const int64_t *x_label, const int64_t *y_label,
                            const int64_t *z_label, int B, int K, int M, float *dnum, float *part, float *out,
                            spacap_stream_t stream);
int spacap_rel_loss_bwd_f32(const float *dnum, const float *grad_losses, const float *out, int B, int K, float *dpred,
                            spacap_stream_t stream);

/* out[i] = sum_s part[s][i] in ascending s: the second stage of the split reductions (weight-gradient slabs, partial
 * bias sums).  part f32 [nslab, n] dense, n a multiple of 4, pointers 16-byte aligned. */
int spacap_sum_slabs_f32(const float *part, int nslab, long n, float *out, spacap_stream_t stream);
/* nseg independent slab sums in one launch (same values as nseg calls of spacap_sum_slabs_f32): parts[i] f32
 * [nslabs[i]][n[i]] -> outs[i] f32 [n[i]].  parts / outs / n / nslabs are HOST arrays (read before the call returns);
 * the segment table is passed to the kernel by value, so the launch is capturable in a hipGraph. */
int spacap_sum_slabs_batched_f32(const float *const *parts, float *const *outs, const long *n, const int *nslabs, int nseg,
                                 spacap_stream_t stream);

/* Feed-forward block, backward of w_2(dropout(relu(.))) w.r.t. the hidden pre-activation in one launch:
 * dx[r,n] = (y[r,n] > 0) ? scale * sum_k g[r,k] W[k,n] : 0 with g f32 [R,128] (gradient of the block output), W f32
 * [128,CP] = w_2.weight (CP a multiple of 128), y f32 [R,CP] the saved dropout(relu(.)) output, scale = 1/(1-p). */
int spacap_linear_dgrad_mask_f32(const float *g, const float *W, const float *y, float scale, long R, int CK, int CP,
                                 float *dx, spacap_stream_t stream);

/* Relation head, layers 2 and 3 (models/transformer_captioner.py:319-326, 392-397) on R = B*K*K pair rows in one pass:
 * hid2 = relu(hid1 W2^T + b2) f32 [R,128] (kept for the backward) and pred = hid2 W3^T + b3 f32 [R,9];
 * hid1 f32 [R,128] (the first layer's ReLU output), W2 f32 [128,128], b2 f32 [128], W3 f32 [9,128], b3 f32 [9]. */
int spacap_rel_tail_fwd_f32(const float *hid1, const float *W2, const float *b2, const float *W3, const float *b3, long R,
                            float *hid2, float *pred, spacap_stream_t stream);
/* First backward stage of the same two layers, one streaming pass: dz2 = (dpred W3) * (hid2 > 0) f32 [R,128] plus
 * part f32 [spacap_rel_tail_bwd_nparts(R)][9*128 + 128 + 16] = per-workgroup partial sums of dW3 = dpred^T hid2
 * ([9][128]), db2 = sum_r dz2 ([128]) and db3 = sum_r dpred ([9], padded to 16); add the rows in order. */
int spacap_rel_tail_bwd_nparts(long R);
int spacap_rel_tail_bwd_f32(const float *dpred, const float *W3, const float *hid2, long R, float *dz2, float *part,
                            spacap_stream_t stream);

/* Weight gradient of a 1x1 convolution on channel-major tensors (vote net: models/voting_module.py:33-60; feature
 * propagation MLPs: lib/pointnet2/pointnet2_modules.py:376-421): dW[co,ci] = sum_b sum_n g[b,co,n] x[b,ci,n] with
 * g f32 [B,CO,N], x f32 [B,CI,N] (CO, CI multiples of 128, N a multiple of 32).  part f32
 * [spacap_conv1x1_wgrad_slabs(B,CO,CI,N)][CO*CI] receives per-slab partial sums that the caller adds in slab order
 * (spacap_sum_slabs_f32); the slab count is 0 for shapes without a kernel. */
int spacap_conv1x1_wgrad_slabs(int B, int CO, int CI, int N);
int spacap_conv1x1_wgrad_f32(const float *g, const float *x, int B, int CO, int CI, int N, float *part,
                             spacap_stream_t stream);
/* Several of them in ONE launch (end of a backward pass); HOST arrays, job table by value, nslabs[i] =
 * spacap_conv1x1_wgrad_slabs_batched(...) (or any multiple of B[i] that divides the point tiles).  with_bias (may be NULL):
 * where with_bias[i] != 0 the partial row of job i is [CO*CI | CO rounded up to 4] and its tail receives the bias gradient
 * db[co] = sum over the slab's points of g[b,co,:] (padding zero). */
int spacap_conv1x1_wgrad_slabs_batched(int B, int CO, int CI, int N);
int spacap_conv1x1_wgrad_batched_f32(const float *const *g, const float *const *x, const int *B, const int *CO, const int *CI,
                                     const int *N, const int *nslabs, const int *with_bias, float *const *part, int njobs,
                                     spacap_stream_t stream);

/* Dense row products of any shape (csrc/dense_rows.hip):  out[r, n] = sum_k A[r, k] Wop[k, n] (+ bias[n]),
   Wop[k, n] = trans_w ? W[n, k] : W[k, n].  A: rows of K floats at stride lda; W: [CO, K] (trans_w) or [K, CO] rows at stride
   ldw (a column slice of a wider matrix is fine: no alignment required); out: rows of CO floats at stride ldo.  Any R >= 0,
   K >= 1, CO >= 1.  Two-level rows (a_grp / o_grp > 0): row r lives at (r / grp) * gstride + (r % grp + skip) * ld -- the
   caption head reads positions 1.. of every sequence in place (models/transformer_captioner.py:373-379: out[:, 1:, :]) and
   its data gradient is written straight into the padded layout (o_zero: the skipped leading rows of every group are zeroed).
   slices > 1 splits the reduction over K: `slices` partial results `slice_stride` floats apart (bias in slice 0), to be
   added in order by spacap_dense_sum_slices_f32; spacap_dense_rows_slices gives the split worth using for a shape.
   batch != 0: the `slices` are independent products instead (operand z at a + z a_zstride, W + z w_zstride, out + z
   slice_stride; bias shared).
   Replaces the rocBLAS GEMMs of: the set-abstraction modules' first-layer feature product F W1[:, 3:]^T and its data gradient
   (lib/pointnet2/pointnet2_modules.py:241-259, first Conv2d of the SharedMLP commuted with the grouping), the vocabulary
   projection (models/transformer_captioner.py:93-100), the relation head's value projection (:319-326). */
int spacap_dense_rows_slices(long R, int K, int CO);
int spacap_dense_rows_f32(const float *a, long lda, long a_grp, long a_gstride, long a_skip, const float *W, long ldw, int trans_w,
                          const float *bias, long R, int K, int CO, float *out, long ldo, long o_grp, long o_gstride, long o_skip,
                          int o_zero, int slices, long slice_stride, int batch, long a_zstride, long w_zstride,
                          spacap_stream_t stream);
int spacap_dense_sum_slices_f32(const float *parts, int S, long n, long stride, float *out, spacap_stream_t stream);
/* Weight + bias gradient of such a product with FEW rows and a wide output (the vocabulary projection: 248 rows, 3 001 x 128):
   dW f32 [M,N] = G^T X, db f32 [M] (nullable) = column sums of G; G rows of M floats at stride ldg, X rows of N floats at stride
   ldx (two-level rows as above), the R rows added in ascending order. */
int spacap_dense_wgrad_small_f32(const float *G, long ldg, const float *X, long ldx, long x_grp, long x_gstride, long x_skip, long R,
                                 int M, int N, float *dW, float *db, spacap_stream_t stream);
/* Block-diagonal weight gradient dW[o][z N + n] = sum_r G[r][z M + o] X[r][z N + n], z < Z (the relation head's first Linear
   applied per head, models/transformer_captioner.py:319-326): part f32 [nslab][M][Z N] per-row-slab partials in dW's layout,
   nslab = spacap_dense_wgrad_blocks_slabs(R) or any count >= 1; the caller adds the slabs in order. */
int spacap_dense_wgrad_blocks_slabs(long R);
int spacap_dense_wgrad_blocks_f32(const float *G, long ldg, const float *X, long ldx, long R, int Z, int M, int N, int nslab,
                                  float *part, spacap_stream_t stream);

/* Weight gradient of a row product with MANY rows and a narrow / odd-width result (the feature columns of an SA module's first
   layer at 7 or 132 input channels, lib/pointnet2/pytorch_utils.py:11-36): part f32 [nslab][M][N], per row slab
   dW[m][n] = sum_r G[r][m] X[r][n]; G rows of M floats at stride ldg, X rows of N floats at stride ldx, any sizes >= 1.
   nslab = spacap_dense_wgrad_tall_slabs(R, M, N) or any count >= 1; the caller adds the slabs in order
   (spacap_sum_slabs_f32, spacap_sa_dw1_assemble_f32). */
int spacap_dense_wgrad_tall_slabs(long R, int M, int N);
int spacap_dense_wgrad_tall_f32(const float *G, long ldg, const float *X, long ldx, long R, int M, int N, int nslab, float *part,
                                spacap_stream_t stream);

/* ---- wide layers on the matrix cores as split-bf16 ("bf16 x 3", fp32-equivalent) tiled products (csrc/gemm_bf3.hip) ----------
 * The relation head of the 512-wide / 32-head stress configuration (models/transformer_captioner.py:319-326, 392-398 at
 * d_model = 512) and the Transformer's Linear layers at that width.  K, N multiples of 128 (spacap_gemm_bf3_supported).
 *   spacap_gemm_bf3_split_w_f32   Wp bf16 [3][N][K] = the three bf16 pieces of W f32 [N][K] (trans == 0, row stride ldw) or of the
 *                                 transpose of W f32 [K][N] (trans != 0); 3 N K two-byte elements
 *   spacap_gemm_bf3_f32           out f32 [R][N] (stride ldo) = A f32 [R][K] (stride lda) W^T + bias (nullable), ReLU when relu != 0
 *   spacap_gemm_bf3_wgrad_f32     part f32 [nslab][N][K]: per row slab dW[n][k] = sum_r G[r][n] X[r][k]; the caller adds the slabs
 *                                 in order; nslab = spacap_gemm_bf3_wgrad_slabs(R, N, K) or any count >= 1 */
int spacap_gemm_bf3_supported(int K, int N);
int spacap_gemm_bf3_split_w_f32(const float *W, long ldw, int N, int K, int trans, void *Wp, spacap_stream_t stream);
int spacap_gemm_bf3_f32(const float *A, long lda, const void *Wp, const float *bias, long R, int K, int N, int relu, float *out, long ldo,
                        spacap_stream_t stream);
int spacap_gemm_bf3_wgrad_slabs(long R, int N, int K);
int spacap_gemm_bf3_wgrad_f32(const float *G, long ldg, const float *X, long ldx, long R, int N, int K, int nslab, float *part,
                              spacap_stream_t stream);
/* Tail of the relation head's backward at any width C (C / 4 divides 256): dz2 f32 [R,C] = (dpred f32 [R,9] W3 f32 [9,C]) where
 * hid2 f32 [R,C] > 0, and per-workgroup partial sums part f32 [nparts][9 C + C + 16] = dW3 [9][C] | db2 [C] | db3 [9 (+7 pad)]. */
int spacap_rel_wide_tail_supported(int C);
int spacap_rel_wide_tail_nparts(long R);
int spacap_rel_wide_tail_bwd_f32(const float *dpred, const float *W3, const float *hid2, long R, int C, int nparts, float *dz2, float *part,
                                 spacap_stream_t stream);
/* First layer of the relation head at wide C on the matrix cores (csrc/gemm_bf3.hip: one workgroup per key column, fp32 MFMA):
 * hid1 f32 [B,K,K,C] = relu(b1 + sum_h P[b,h,i,j] U[b,j,h,:]) from Pt f32 [B,K,H,K] = P transposed (spacap_rel_wide_transpose_f32
 * with to_t = 1: out[b,j,h,i] = in[b,h,i,j]; to_t = 0: the inverse), U f32 [B,K,H,C], b1 f32 [C].  Backward: dh1, hid1 -> dPt
 * f32 [B,K,H,K], dU f32 [B,K,H,C], db_part f32 [B K][C] (one row per key column, added in order by the caller); the ReLU mask
 * hid1 > 0 is applied here.  H in {8,16,32}, C in {128,256,512} (spacap_rel_wide_l1_supported). */
int spacap_rel_wide_l1_supported(int H, int K, int C);
int spacap_rel_wide_transpose_f32(const float *in, float *out, int B, int H, int K, int to_t, spacap_stream_t stream);
int spacap_rel_wide_l1_fwd_f32(const float *Pt, const float *U, const float *b1, int B, int H, int K, int C, float *hid1,
                               spacap_stream_t stream);
int spacap_rel_wide_l1_bwd_f32(const float *dh1, const float *hid1, const float *Pt, const float *U, int B, int H, int K, int C,
                               float *dPt, float *dU, float *db_part, spacap_stream_t stream);

/* Row-panel product of a d_model-sized projection (replaces nn.Linear's forward / data gradient where the BLAS
 * heuristics are poor, models/transformer_captioner.py:63-99): out[r,n] = sum_k a[r,k] Wop[k,n] (+ bias[n]) with
 * a f32 [R,K], Wop[k,n] = trans_w ? W[n,k] (W f32 [CO,K]: y = x W^T) : W[k,n] (W f32 [K,CO]: dx = g W), bias f32 [CO]
 * or null, out f32 [R,CO]; K in {128,256,384,512}, CO a multiple of 64 (spacap_linear_rows_supported says so). */
int spacap_linear_rows_supported(long R, int K, int CO);
int spacap_linear_rows_f32(const float *a, const float *W, const float *bias, long R, int K, int CO, int trans_w,
                           float *out, spacap_stream_t stream);

/* ---- row-wise L2 normalisation y = x / |x| (models/SpaCapNet.py:66-67, no epsilon); x, y f32 [rows, D], D % 4 == 0;
 * inv_norm f32 [rows] is kept for the backward dx = (g - y (g . y)) / |x|. */
int spacap_l2norm_rows_fwd_f32(const float *x, long rows, int D, float *y, float *inv_norm, spacap_stream_t stream);
int spacap_l2norm_rows_bwd_f32(const float *g, const float *y, const float *inv_norm, long rows, int D, float *dx,
                               spacap_stream_t stream);

/* ---- caption head loss: log-softmax over the vocabulary (models/transformer_captioner.py:93-99) + compute_cap_loss
 * (lib/loss_helper.py:199-238).  logits f32 [B*W, V]; target i64 [B, tstride] (word w of scene b at b*tstride + w, 0 = pad:
 * ignored); good u8 [B].  Outputs: logp f32 [B*W, V] (= data_dict["lang_cap"]), rowstat f32 [B*W, 4] (workspace),
 * out f32 [4] = (cap_loss, cap_acc, 1 / (sum good + 1e-6), sum good).  Backward: dlogits = gloss[0] * d cap_loss / d logits. */
int spacap_cap_loss_fwd_f32(const float *logits, const int64_t *target, const uint8_t *good, int B, int W, int V, int tstride,
                            float *logp, float *rowstat, float *out, spacap_stream_t stream);
int spacap_cap_loss_bwd_f32(const float *logp, const int64_t *target, const uint8_t *good, const float *out, const float *gloss,
                            int B, int W, int V, int tstride, float *dlogits, spacap_stream_t stream);

/* ---- score decoding of the proposal head (models/proposal_module.py:106-158 decode_scores, :81-104 decode_pred_box).
 * net f32 [B, CH, K] (the head's Conv1d output, CH = 5 + 2 NH + 4 NS + NC), agg_xyz f32 [B,K,3], mean_size f32 [NS,3] ->
 * nt f32 [B,K,CH] (net transposed: the rows every score slice is a view of), center f32 [B,K,3] = agg_xyz + nt[...,2:5],
 * heading_res f32 [B,K,NH] = nt[...,5+NH:5+2NH] * pi/NH, size_res f32 [B,K,NS,3] = normalised residuals * mean_size,
 * corners f64 [B,K,8,3] of the arg-max size class box (heading 0; mean_size_f64 f64 [NS,3] or NULL = the fp32 table widened), bbox_mask / sem_cls / size_cls i64 [B,K] (first maxima).
 * Backward: d_net f32 [B,CH,K] from the gradients of nt / center / heading_res / size_res (each nullable). */
int spacap_proposal_decode_fwd_f32(const float *net, const float *agg_xyz, const float *mean_size, const double *mean_size_f64,
                                   int B, int K, int NH, int NS, int NC, float *nt, float *center, float *heading_res, float *size_res, double *corners,
                                   int64_t *bbox_mask, int64_t *sem_cls, int64_t *size_cls, spacap_stream_t stream);
int spacap_proposal_decode_bwd_f32(const float *g_nt, const float *g_center, const float *g_heading_res, const float *g_size_res,
                                   const float *mean_size, int B, int K, int NH, int NS, int NC, float *d_net, spacap_stream_t stream);

/* ---- tail of get_scene_cap_loss (lib/loss_helper.py:340-383): det f32 [8] = (vote, objectness, center, heading_cls, heading_reg,
 * size_cls, size_reg, sem_cls), cap f32 [4] (cap[0] = caption loss), rel f32 [7] or NULL (rel[0..2] = x, y, z loss);
 * obj_label i64 [n], obj_mask f32 [n], bbox_mask i64 [n] -> out f32 [8] = (box_loss, det_loss, relation_loss, loss, pos_ratio,
 * neg_ratio, obj_acc, 0) and loss f32 [1] (= out[3], the differentiable output).  Backward: g_loss f32 [1] -> g_det [8],
 * g_cap [4], g_rel [7] (NULL when rel was). */
int spacap_loss_tail_fwd_f32(const float *det, const float *cap, const float *rel, const int64_t *obj_label, const float *obj_mask,
                             const int64_t *bbox_mask, int n, float *out, float *loss, spacap_stream_t stream);
int spacap_loss_tail_bwd_f32(const float *g_loss, float *g_det, float *g_cap, float *g_rel, spacap_stream_t stream);

/* ---- fused Transformer sub-layers, d_model = 128 (replaces everything BETWEEN two attention() calls of
 * models/transformer_captioner.py: SublayerConnection :115-127, LayerNorm :102-113, PositionwiseFeedForward :72-81, the
 * output projection and the packed q|k|v projection of MultiHeadedAttention :52-70).  One argument block, read before the
 * call returns; every tensor f32, dense, 16-byte aligned, rows of 128 unless stated.
 *
 *  mode 0 (forward), per row r:
 *      acc   = a1[r, 0:k1] w1^T + bias1          w1 [128, k1], k1 a multiple of 128        (skipped when a1 == NULL)
 *      x'    = res[r] + dropout(acc)             dropout(p, seed, seed_dev) as spacap_dropout_add_fwd_f32, element index
 *                                                r*128 + c; x' = res[r] when a1 == NULL;  stored to x_out when non-NULL
 *      n     = ln_a * (x' - mean) / (std_unbiased + eps) + ln_b      stored to n_out (nullable), (mean, 1/(std+eps)) to
 *                                                stats [R,2] (nullable)
 *      out2  = n w2^T + bias2                    w2 [n2, 128], n2 a multiple of 128, out2 [R, n2]   (skipped when n2 == 0)
 *  mode 1 (backward), per row r:
 *      dn    = a1[r, 0:k1] w1                    w1 [k1, 128]                              (dn = g[r] when a1 == NULL)
 *      dx    = LayerNorm'(dn; x_ln[r], stats[r], ln_a) + res[r]      (res nullable: the gradient arriving along the
 *                                                residual connection);  stored to x_out
 *      part[workgroup] = (sum_rows dn * xhat, sum_rows dn)           f32 [spacap_tf_rows_parts(R), 256], to be added in order
 *      dy    = dropout'(dx)                      the same mask as the forward call with the same (p, seed, seed_dev); stored
 *                                                to n_out (nullable)
 *      out2  = dy w2                             w2 [128, 128], n2 == 128, out2 [R,128]            (skipped when n2 == 0) */
typedef struct {
  int mode;
  long R;
  const float *a1;
  const float *w1;
  const float *bias1;
  int k1;
  float drop_p;
  float eps;
  uint64_t seed;
  const uint64_t *seed_dev;
  const float *res;
  float *x_out;
  const float *ln_a;
  const float *ln_b;
  float *n_out;
  float *stats;
  const float *x_ln;
  const float *g;
  float *part;
  const float *w2;
  const float *bias2;
  float *out2;
  int n2;
  int nparts; /* > 0: a1 is [nparts][R][128], partial sums of the first product (spacap_tf_gemm_f32), added in order */
  /* mode 1 with n2 == 128 (out2 = the gradient of the attention output): attn_out [R,128] = that output; delta_out
   * f32 [R/lq, 8, lq] (nullable) receives sum_d out2[r, 16 head + d] attn_out[r, 16 head + d], the per-(row, head) term
   * spacap_mha_bwd_delta_f32 takes as `delta` (8 heads of 16; rows r = b * lq + q) */
  const float *attn_out;
  float *delta_out;
  int lq;
} spacap_tf_rows_args;
int spacap_tf_rows_f32(const spacap_tf_rows_args *args, spacap_stream_t stream);
int spacap_tf_rows_parts(long R);
/* h = dropout(relu(x W^T + bias)): x [R,128], W [N,128], N a multiple of 128, h [R,N]; dropout element index r*N + c
 * (models/transformer_captioner.py:80).  The saved h is positive exactly where the unit was active and kept: the backward is
 * spacap_linear_dgrad_mask_f32. */
int spacap_tf_ffn1_f32(const float *x, const float *W, const float *bias, long R, int N, float drop_p, uint64_t seed,
                       const uint64_t *seed_dev, float *h, spacap_stream_t stream);
/* The feed-forward block as one launch per direction, chained through LDS (64-row x 128-hidden-unit tile per workgroup):
 *  mode 0: hid = dropout(relu(x Wa^T + bias)) [R,dff], Wa = w_1 [dff,128] (dropout element index r*dff + c);
 *          part[c][r][:] = hid[r, 128c:128c+128] Wb[:, 128c:128c+128]^T, Wb = w_2 [128,dff]; hid may be NULL (inference)
 *  mode 1: hid = (x Wa) * [y > 0] / (1 - drop_p), Wa = w_2 [128,dff], y = the forward hid;
 *          part[c][r][:] = hid[r, 128c:128c+128] Wb[128c:128c+128, :], Wb = w_1 [dff,128]
 * x [R,128]; part f32 [dff/128][R][128] is consumed by spacap_tf_rows_f32 (nparts = dff/128), which adds the slices in order. */
int spacap_tf_ffn_f32(int mode, const float *x, const float *Wa, const float *Wb, const float *bias, const float *y, long R, int dff,
                      float drop_p, uint64_t seed, const uint64_t *seed_dev, float *hid, float *part, spacap_stream_t stream);
/* The same block on split-bf16 products (three bf16 pieces per operand, six piece products: fp32-equivalent at 6/16 of the
 * fp32-MFMA time) for tall inputs (R > 512).  The weights are split ahead of time: spacap_tf_ffn_split_f32 writes, for each of
 * nlayers layers, four piece images (W1, W2, W2^T, W1^T; spacap_tf_ffn_pieces_elems(dff) bf16 elements per layer) in one launch
 * per 16 layers (host pointer arrays, read before the call returns); spacap_tf_ffn_bf3_f32 takes one layer's images. */
long spacap_tf_ffn_pieces_elems(int dff);
int spacap_tf_ffn_split_f32(const float *const *w1, const float *const *w2, void *const *pieces, int nlayers, int dff,
                            spacap_stream_t stream);
int spacap_tf_ffn_bf3_f32(int mode, const float *x, const void *pieces, const float *bias, const float *y, long R, int dff, float drop_p,
                          uint64_t seed, const uint64_t *seed_dev, float *hid, float *part, spacap_stream_t stream);
/* One greedy-decoding step of self-attention over a key / value cache (replaces the prefix recomputation of
 * models/transformer_captioner.py:435-438): qkv f32 [R, 3*128] = the packed projection of the NEW token of every sequence;
 * its k, v are appended at position t of kcache / vcache f32 [R, T, 128] (T <= 32) and its q attends over positions 0..t;
 * out f32 [R, 128] (heads concatenated).  h = 8, d_k = 16. */
int spacap_decode_attn_f32(const float *qkv, float *kcache, float *vcache, long R, int h, int d_k, int T, int t, float scale,
                           float *out, spacap_stream_t stream);
/* One greedy-decoding step's word choice without the logits in HBM (models/transformer_captioner.py:441-447 `torch.max(prob, dim=1)`
 * on the Generator of :93-100; the arg-max of the log-softmax is the arg-max of the logits): x f32 [R,128] = the decoder's output
 * rows, Wp bf16 [3][V][128] = the three split-bf16 pieces of the projection weight f32 [V,128] (spacap_gemm_bf3_split_w_f32; the
 * logits are fp32-equivalent split-bf16 products), bias f32 [V] -> ys i64 [R][ys_ld] column t_out = the arg-max word (first maximum) and x_next f32 [R,128] =
 * lut[word] * scale + pe_row, the next step's input rows (lut f32 [V,128], pe_row f32 [128]).  workspace: device memory of
 * spacap_decode_word_workspace_bytes(R, V) bytes. */
size_t spacap_decode_word_workspace_bytes(long R, int V);
int spacap_decode_word_f32(const float *x, const void *Wp, const float *bias, long R, int V, const float *lut, float scale,
                           const float *pe_row, int64_t *ys, int ys_ld, int t_out, float *x_next, void *workspace,
                           spacap_stream_t stream);
/* Split-K product for the skinny feed-forward products (K = d_ff, N = 128: w_2 forward, the data gradient through w_1):
 * out[s][r][n] = sum_{k in slice s} a[r][k] Wop[k][n], Wop[k][n] = trans_w ? W[n][k] (W [N,K]) : W[k][n] (W [K,N]);
 * a [R,K], K and N multiples of 128, nsplit a divisor of K / 128 (spacap_tf_gemm_splits suggests the one that fills the
 * chip); out [nsplit][R][N] is consumed by spacap_tf_rows_f32 (nparts), which adds the slices in order. */
int spacap_tf_gemm_splits(long R, int K, int N);
int spacap_tf_gemm_f32(const float *a, const float *W, long R, int K, int N, int trans_w, int nsplit, float *out,
                       spacap_stream_t stream);
/* out[r][n] = y[r][n] > 0 ? scale * sum_k g[r][k] W[k][n] : 0 -- g [R,128], W [128,N], y, out [R,N]: the gradient of
 * spacap_tf_ffn1_f32's pre-activation through the following Linear (y = its saved output). */
int spacap_tf_dgrad_mask_f32(const float *g, const float *W, const float *y, float scale, long R, int K, int N, float *out,
                             spacap_stream_t stream);

/* 1x1 convolutions on channel-major tensors (nn.Conv1d / nn.Conv2d with kernel size 1: models/voting_module.py:33-60,
 * models/proposal_module.py:41-55, lib/pointnet2/pytorch_utils.py:11-36, models/transformer_captioner.py:251-258), N = points
 * per scene, a multiple of 64 (spacap_conv1x1_cm_supported).  W f32 [CO,CI] dense.
 *   mode 0 (forward):        out[b,co,n] = sum_ci W[co,ci] in[b,ci,n] + bias[co]   in [B,CI,N], out [B,CO,N], bias f32 [CO] or NULL
 *   mode 1 (input gradient): out[b,ci,n] = sum_co W[co,ci] in[b,co,n]              in [B,CO,N], out [B,CI,N]
 *   mode 2: mode 0 with exact fp32 products (mode 0 multiplies split-bf16 pieces, fp32-equivalent to ~1e-6, unless SPACAP_SA_F32MFMA=1) */
int spacap_conv1x1_cm_supported(int CI, int CO, long N);
int spacap_conv1x1_cm_f32(int mode, const float *W, const float *in, const float *bias, int B, int CI, int CO, long N,
                          float *out, spacap_stream_t stream);

/* Votes from the voting module's last convolution (models/voting_module.py:49-60, vote_factor 1): net f32 [B, 3+C, N]
 * channel-major, seed_xyz f32 [B,N,3], seed_feat f32 [B,C,N] -> vote_xyz f32 [B,N,3] = seed_xyz + net[:, 0:3]^T and
 * vote_feat f32 [B,N,C] (point-major) = seed_feat^T + net[:, 3:]^T.  Backward: g_xyz [B,N,3] / g_feat [B,N,C] (either may
 * be NULL = zero) -> d_net f32 [B, 3+C, N] and d_seed f32 [B,C,N] (= d_net[:, 3:]; may be NULL). */
int spacap_vote_assemble_fwd_f32(const float *net, const float *seed_xyz, const float *seed_feat, int B, int C, int N,
                                 float *vote_xyz, float *vote_feat, spacap_stream_t stream);
int spacap_vote_assemble_bwd_f32(const float *g_xyz, const float *g_feat, int B, int C, int N, float *d_net, float *d_seed,
                                 spacap_stream_t stream);

/* njobs device-to-device copies (dst[i] <- src[i], nbytes[i] bytes, non-overlapping) in one launch per 120 jobs; the three
 * arrays are HOST arrays, read before the call returns. */
int spacap_copy_batched(const void *const *src, void *const *dst, const long *nbytes, int njobs, spacap_stream_t stream);

/* The stream idles for about `microseconds` (one wave spinning on the device's wall clock; 0 .. 100 000). */
int spacap_stream_delay(int microseconds, spacap_stream_t stream);
/* A dependency from inside a captured step to a stream outside it (events cannot express one): spacap_stream_signal writes
   *flag = *value (device words) in stream order; spacap_stream_wait_ge holds its stream (one spinning wave) until *flag >= value.
   A wait that lasts longer than timeout_ms (1 .. 600 000) gives up LOUDLY: it stores `value` into the sticky device word *err
   (compare-and-swap against 0: the first failure stays; the library never clears it).  Work queued behind the wait still
   runs, so its consumer must be gated on *err (spacap_adam_flat_f32's skip_if_nonzero) and the host must read the word.
   engine.py: the all-reduce of the captioner's gradient slice starts while the detector's backward runs. */
int spacap_stream_wait_ge(const int64_t *flag, int64_t value, int timeout_ms, int64_t *err, spacap_stream_t stream);
int spacap_stream_signal(int64_t *flag, const int64_t *value, spacap_stream_t stream);

/* Lab only (tools/lab/step_stamps.py): writes the device's 100 MHz wall clock into *slot when the stream reaches it. */
int spacap_lab_stamp(uint64_t *slot, spacap_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPACAP_HIP_H */
