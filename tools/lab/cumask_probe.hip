// Lab: does hipExtStreamCreateWithCUMask partition the chip the way the side-stream sampling chain needs?
//  build:  hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/lab/cumask_probe.hip -o tools/lab/libcumask_probe.so
// Exports (ctypes): probe_create_stream(mask words, n) -> stream handle; probe_where(stream, grid, out[grid][2]) -> (xcc, hw_id) of each
// workgroup; probe_spin(stream, grid, microseconds).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
__global__ void where_kernel(uint32_t *out) {
  if (threadIdx.x == 0) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
  // stay resident long enough that the whole grid is placed before the first workgroup leaves
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) __builtin_amdgcn_s_sleep(8);   // 20 us at 100 MHz
}
__global__ void spin_kernel(uint64_t ticks) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

extern "C" void *probe_create_stream(const uint32_t *mask, int words) {
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) return nullptr;
  return s;
}
extern "C" int probe_get_mask(void *stream, uint32_t *mask, int words) {
  return (int)hipExtStreamGetCUMask((hipStream_t)stream, (uint32_t)words, mask);
}
extern "C" int probe_where(void *stream, int grid, int threads, int lds, uint32_t *out_dev) {
  hipLaunchKernelGGL(where_kernel, dim3(grid), dim3(threads), lds, (hipStream_t)stream, out_dev);
  return (int)hipGetLastError();
}
extern "C" int probe_spin(void *stream, int grid, int threads, int lds, int us) {
  hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(threads), lds, (hipStream_t)stream, (uint64_t)us * 100);
  return (int)hipGetLastError();
}
