// Lab: timing + bit-exactness check of the bucketed FPS kernel (spacap3d_amd/csrc/fps_bucket.inc) without torch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o fps_lab_bucket tools/lab/fps_bucket_lab.hip
//   ./fps_lab_bucket [N] [kind: 0 room surfaces | 1 uniform volume | 2 lattice (ties)]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../spacap3d_amd/csrc/common.hpp"
namespace spacap { void set_error(const char *, ...) {} }
#pragma clang fp contract(off)
using namespace spacap;
using f32x4 = float __attribute__((ext_vector_type(4)));

#include "../../spacap3d_amd/csrc/fps_bucket.inc"

static int opt_n_threads(int w) {
  int p = (int)(log((double)w) / log(2.0)), t = 1 << p;
  return t > 512 ? 512 : (t < 1 ? 1 : t);
}
static void fps_cpu(int n, int m, const float *d, int *idx) {
  const int bs = opt_n_threads(n);
  std::vector<float> temp(n, 1e10f), best(bs);
  std::vector<int> besti(bs);
  int old = 0;
  idx[0] = 0;
  for (int j = 1; j < m; ++j) {
    for (int t = 0; t < bs; ++t) { best[t] = -1.f; besti[t] = 0; }
    const float x1 = d[old * 3], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
    for (int k = 0; k < n; ++k) {
      const int t = k % bs;
      const float x2 = d[k * 3], y2 = d[k * 3 + 1], z2 = d[k * 3 + 2];
      const float mag = (x2 * x2) + (y2 * y2) + (z2 * z2);
      if ((double)mag <= 1e-3) continue;
      const float dd = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
      const float d2 = fminf(dd, temp[k]);
      temp[k] = d2;
      if (d2 > best[t]) { best[t] = d2; besti[t] = k; }
    }
    for (int s = bs / 2; s >= 1; s >>= 1)
      for (int t = 0; t < s; ++t)
        if (best[t + s] > best[t]) { best[t] = best[t + s]; besti[t] = besti[t + s]; }
    old = besti[0];
    idx[j] = old;
  }
}

int main(int argc, char **argv) {
  const int B = 8, m = 2048;
  const int N = argc > 1 ? atoi(argv[1]) : 40000;
  const int kind = argc > 2 ? atoi(argv[2]) : 0;
  std::vector<float> h((size_t)B * N * 3);
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX; };
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < N; ++k) {
      float *p = &h[((size_t)b * N + k) * 3];
      p[0] = rnd() * 6 - 3; p[1] = rnd() * 6 - 3; p[2] = rnd() * 3;
      if (kind == 0) { const int ax = rand() % 3; p[ax] = (rand() & 1) ? 3.f : (ax == 2 ? 0.f : -3.f); }
      if (kind == 2) { p[0] = (rand() % 12) * 0.5f - 3; p[1] = (rand() % 12) * 0.5f - 3; p[2] = (rand() % 6) * 0.5f; }
      if (k % 997 == 5) { p[0] = 0.01f; p[1] = 0.005f; p[2] = 0.f; }  // |p|^2 <= 1e-3: skipped
      if (k % 100 == 7 && k > 7) memcpy(p, p - 21, 12);                 // exact duplicates
    }
  float *xyz, *ws;
  int32_t *idx;
  const size_t wsb = fps_bucket_workspace_floats(N) * 4;
  hipMalloc(&xyz, h.size() * 4);
  hipMalloc(&ws, (size_t)B * wsb);
  hipMalloc(&idx, (size_t)B * m * 4);
  hipMemcpy(xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int bs = opt_n_threads(N);
  int lg = 0;
  while ((1 << lg) < bs) ++lg;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) launch_fps_bucket(xyz, ws, B, N, m, lg, idx, 0);
  hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) launch_fps_bucket(xyz, ws, B, N, m, lg, idx, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  printf("bucket FPS N=%d kind=%d: %8.3f ms  %6.3f us/round  (%s)\n", N, kind, ms, ms * 1e3 / (m - 1),
         hipGetErrorString(hipGetLastError()));
  std::vector<int> got((size_t)B * m), want(m);
  hipMemcpy(got.data(), idx, got.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < 2; ++b) {
    fps_cpu(N, m, &h[(size_t)b * N * 3], want.data());
    for (int j = 0; j < m; ++j)
      if (got[(size_t)b * m + j] != want[j]) {
        if (bad < 5) printf("  MISMATCH scene %d j=%d got %d want %d\n", b, j, got[(size_t)b * m + j], want[j]);
        ++bad;
      }
  }
  printf("  parity vs CPU (2 scenes): %s (%d mismatches)\n", bad ? "FAIL" : "bit-exact", bad);
  return bad != 0;
}
