// Shared helpers for the gfx950 kernels of libspacap_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/spacap_hip.h"

namespace spacap {

// thread-local last-error text (defined in capi.hip)
void set_error(const char *fmt, ...);

inline hipStream_t as_stream(spacap_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// CUs a caller asked the persistent grids to leave to side-stream work (spacap_sa_reserve_cus), and the device's CU count
// (both defined in sa_mlp.hip)
int sa_reserved_cus();
int device_cus();

// Raises a kernel's dynamic-LDS limit on the CURRENT device.  `done` is the call site's own bit mask of devices that already
// have it (a function attribute is per device: a process that moves to another GPU must set it there too).
inline hipError_t allow_dynamic_lds(const void *fn, int bytes, unsigned long long &done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done |= bit;
  return e;
}

#define SPACAP_REQUIRE(cond, ...)          \
  do {                                     \
    if (!(cond)) {                         \
      ::spacap::set_error(__VA_ARGS__);    \
      return SPACAP_E_INVALID;             \
    }                                      \
  } while (0)

#define SPACAP_CHECK_LAUNCH(what)                                              \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      ::spacap::set_error("%s: %s", what, hipGetErrorString(e__));             \
      return SPACAP_E_LAUNCH;                                                  \
    }                                                                          \
  } while (0)

#define SPACAP_CHECK_HIP(expr, what)                                           \
  do {                                                                         \
    hipError_t e__ = (expr);                                                   \
    if (e__ != hipSuccess) {                                                   \
      ::spacap::set_error("%s: %s", what, hipGetErrorString(e__));             \
      return SPACAP_E_LAUNCH;                                                  \
    }                                                                          \
  } while (0)

// ---- wave64 cross-lane helpers (DPP: no LDS round trip) ----------------------------------
// dpp_ctrl encodings (CDNA ISA): quad_perm = 0x00..0xFF, row_mirror 0x140, row_half_mirror 0x141.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

// max over the 64 lanes of a signed 32-bit value; result is wave-uniform (SGPR).
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, dpp_i32<0xB1>(v));   // quad_perm [1,0,3,2]  : lane ^ 1
  v = max(v, dpp_i32<0x4E>(v));   // quad_perm [2,3,0,1]  : lane ^ 2
  v = max(v, dpp_i32<0x141>(v));  // row_half_mirror      : folds 4-groups inside 8
  v = max(v, dpp_i32<0x140>(v));  // row_mirror           : folds 8-groups inside 16
  int r0 = __builtin_amdgcn_readlane(v, 0);
  int r1 = __builtin_amdgcn_readlane(v, 16);
  int r2 = __builtin_amdgcn_readlane(v, 32);
  int r3 = __builtin_amdgcn_readlane(v, 48);
  return max(max(r0, r1), max(r2, r3));
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = min(v, (unsigned)dpp_i32<0xB1>((int)v));
  v = min(v, (unsigned)dpp_i32<0x4E>((int)v));
  v = min(v, (unsigned)dpp_i32<0x141>((int)v));
  v = min(v, (unsigned)dpp_i32<0x140>((int)v));
  unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
  unsigned r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32);
  unsigned r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(r0, r1), min(r2, r3));
}

// Wave reductions with the DPP modifier ON the max / min itself (the helpers of common.hpp go through a v_mov_dpp: four
// instructions per step) and the two row_bcast steps instead of four readlanes: 6 steps, result from lane 63.  These sit
// on the serial chain of every sampling round.
__device__ __forceinline__ int wave_max_i32_fast(int v) {
  asm volatile("s_nop 1\n\t"
               "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
               "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
               : "+v"(v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32_fast(unsigned v) {
  asm volatile("s_nop 1\n\t"
               "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
               "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
               : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace spacap
