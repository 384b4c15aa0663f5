// Development driver for the DCN kernels: times the C-ABI entry points with HIP events on random data and checks the
// new forward against the round-3 kernel (kept as fv2p_dcn_forward_v1 while both exist).
//   hipcc -O2 tools/ubench/dcn_bench.cpp -o tools/ubench/dcn_bench -Lfrom-voxel-to-point_amd/lib -lfv2p_ops -Wl,-rpath,'$ORIGIN/../../from-voxel-to-point_amd/lib'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>
#include "../../include/fv2p_ops.h"

#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1); } } while (0)
#define GEOM int, int, int, int, int, int, int, int, int, int, int, int, int, int, int, int
extern "C" int fv2p_dcn_forward_v1(const float*, const float*, const float*, const float*, const float*, GEOM, float*, void*);
extern "C" size_t fv2p_dcn_backward_ws_bytes_v1(int, int, int, int, int, int, int);
extern "C" int fv2p_dcn_backward_v1(const float*, const float*, const float*, const float*, const float*, GEOM, float*, float*, float*, float*, void*, size_t, void*);

static void compare(const char* name, const float* da, const float* db, size_t n) {
  std::vector<float> a(n), c(n);
  CK(hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), db, n * 4, hipMemcpyDeviceToHost));
  double maxd = 0, maxv = 0; size_t bad = 0, neq = 0;
  for (size_t i = 0; i < n; ++i) { double d = fabs((double)a[i] - c[i]); if (!(d <= 1e30)) ++bad; if (d > maxd) maxd = d; if (fabs(a[i]) > maxv) maxv = fabs(a[i]); if (memcmp(&a[i], &c[i], 4)) ++neq; }
  printf("    %-8s max abs diff %.3e (max |ref| %.3e, rel %.3e, non-finite %zu, differing words %zu of %zu)\n", name, maxd, maxv, maxd / (maxv > 0 ? maxv : 1), bad, neq, n);
}


template <typename F>
static float time_us(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  int B = 4, C = 128, H = 200, W = 176, dg = 1, reps = 10, CO = 0;
  float oscale = 0.5f;
  if (argc > 1) B = atoi(argv[1]);
  if (argc > 2) C = atoi(argv[2]);
  if (argc > 3) H = atoi(argv[3]);
  if (argc > 4) W = atoi(argv[4]);
  if (argc > 5) dg = atoi(argv[5]);
  if (argc > 6) reps = atoi(argv[6]);
  if (argc > 7) oscale = atof(argv[7]);
  if (argc > 8) CO = atoi(argv[8]);
  if (!CO) CO = C;
  const int K = 9;
  const size_t npix = (size_t)B * H * W;
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_real_distribution<float> ud(0.f, 1.f);
  std::vector<float> hx(npix * C), hw((size_t)K * C * CO), hwt((size_t)K * CO * C), hb(CO), hoff((size_t)B * dg * 2 * K * H * W), hm((size_t)B * dg * K * H * W);
  for (auto& v : hx) v = nd(rng);
  for (auto& v : hw) v = nd(rng) * 0.05f;
  for (int k = 0; k < K; ++k) for (int ci = 0; ci < C; ++ci) for (int co = 0; co < CO; ++co) hwt[((size_t)k * CO + co) * C + ci] = hw[((size_t)k * C + ci) * CO + co];
  for (auto& v : hb) v = nd(rng);
  for (auto& v : hoff) v = nd(rng) * oscale;
  for (auto& v : hm) v = ud(rng);
  float *x, *w, *wt, *b, *off, *m, *y1, *y2;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&wt, hwt.size() * 4)); CK(hipMalloc(&b, hb.size() * 4));
  CK(hipMalloc(&off, hoff.size() * 4)); CK(hipMalloc(&m, hm.size() * 4)); CK(hipMalloc(&y1, npix * CO * 4)); CK(hipMalloc(&y2, npix * CO * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wt, hwt.data(), hwt.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(off, hoff.data(), hoff.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(m, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(y1, 0xff, npix * CO * 4)); CK(hipMemset(y2, 0xff, npix * CO * 4));
  int rc = fv2p_dcn_forward_v1(x, w, b, off, m, B, H, W, C, CO, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg, y1, nullptr);
  if (rc) { printf("v1 rc %d: %s\n", rc, fv2p_last_error()); return 1; }
  rc = fv2p_dcn_forward(x, wt, b, off, m, B, H, W, C, CO, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg, y2, nullptr);
  if (rc) { printf("v2 rc %d: %s\n", rc, fv2p_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<float> a(npix * CO), c(npix * CO);
  CK(hipMemcpy(a.data(), y1, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), y2, c.size() * 4, hipMemcpyDeviceToHost));
  double maxd = 0, maxv = 0; size_t bad = 0;
  for (size_t i = 0; i < a.size(); ++i) { double d = fabs((double)a[i] - c[i]); if (!(d <= 1e30)) ++bad; if (d > maxd) maxd = d; if (fabs(a[i]) > maxv) maxv = fabs(a[i]); }
  printf("[%d,%d->%d,%d,%d] dg=%d: forward new vs v1: max abs diff %.3e (max |y| %.3e, rel %.3e, non-finite %zu)\n", B, C, CO, H, W, dg, maxd, maxv, maxd / maxv, bad);
  const double fl = 2.0 * npix * C * CO * K;
  float t1 = time_us([&] { fv2p_dcn_forward_v1(x, w, b, off, m, B, H, W, C, CO, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg, y1, nullptr); }, reps);
  float t2 = time_us([&] { fv2p_dcn_forward(x, wt, b, off, m, B, H, W, C, CO, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg, y2, nullptr); }, reps);
  printf("  forward v1 %9.1f us (%6.1f TF/s, %.3f of peak)   new %9.1f us (%6.1f TF/s, %.3f of peak)\n", t1, fl / t1 / 1e6, fl / t1 / 1e6 / 157.3, t2, fl / t2 / 1e6,
         fl / t2 / 1e6 / 157.3);
  // ---- backward
  const size_t nin = npix;
  float *dy, *dx1, *dx2, *dx3, *do1, *do2, *do3, *dm1, *dm2, *dm3, *dw1, *dw2, *dw3;
  std::vector<float> hdy(npix * CO);
  for (auto& v : hdy) v = nd(rng);
  CK(hipMalloc(&dy, hdy.size() * 4)); CK(hipMemcpy(dy, hdy.data(), hdy.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dx1, nin * C * 4)); CK(hipMalloc(&dx2, nin * C * 4)); CK(hipMalloc(&dx3, nin * C * 4));
  CK(hipMalloc(&do1, hoff.size() * 4)); CK(hipMalloc(&do2, hoff.size() * 4)); CK(hipMalloc(&do3, hoff.size() * 4));
  CK(hipMalloc(&dm1, hm.size() * 4)); CK(hipMalloc(&dm2, hm.size() * 4)); CK(hipMalloc(&dm3, hm.size() * 4));
  CK(hipMalloc(&dw1, hw.size() * 4)); CK(hipMalloc(&dw2, hw.size() * 4)); CK(hipMalloc(&dw3, hw.size() * 4));
  const size_t ws1b = fv2p_dcn_backward_ws_bytes_v1(B, H, W, C, CO, 3, 3), ws2b = fv2p_dcn_backward_ws_bytes(B, H, W, H, W, C, CO, 3, 3, dg);
  void *ws1, *ws2;
  CK(hipMalloc(&ws1, ws1b)); CK(hipMalloc(&ws2, ws2b));
  printf("  backward workspace: v1 %.1f MB, new %.1f MB\n", ws1b / 1e6, ws2b / 1e6);
  CK(hipMemset(dx1, 0, nin * C * 4));
  rc = fv2p_dcn_backward_v1(x, w, off, m, dy, B, H, W, C, CO, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg, dx1, do1, dm1, dw1, ws1, ws1b, nullptr);
  if (rc) { printf("bwd v1 rc %d: %s\n", rc, fv2p_last_error()); return 1; }
  CK(hipMemset(dx2, 0xff, nin * C * 4)); CK(hipMemset(dx3, 0xff, nin * C * 4));
  rc = fv2p_dcn_backward(x, w, off, m, dy, B, H, W, C, CO, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg, dx2, do2, dm2, dw2, ws2, ws2b, nullptr);
  if (rc) { printf("bwd new rc %d: %s\n", rc, fv2p_last_error()); return 1; }
  rc = fv2p_dcn_backward(x, w, off, m, dy, B, H, W, C, CO, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg, dx3, do3, dm3, dw3, ws2, ws2b, nullptr);
  CK(hipDeviceSynchronize());
  printf("  backward new vs v1:\n");
  compare("dx", dx1, dx2, nin * C); compare("doffset", do1, do2, hoff.size()); compare("dmask", dm1, dm2, hm.size()); compare("dW", dw1, dw2, hw.size());
  printf("  backward new, run 1 vs run 2 (must be bit-identical):\n");
  compare("dx", dx2, dx3, nin * C); compare("doffset", do2, do3, hoff.size()); compare("dmask", dm2, dm3, hm.size()); compare("dW", dw2, dw3, hw.size());
  float tb1 = time_us([&] { hipMemsetAsync(dx1, 0, nin * C * 4, 0); fv2p_dcn_backward_v1(x, w, off, m, dy, B, H, W, C, CO, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg, dx1, do1, dm1, dw1, ws1, ws1b, nullptr); }, reps);
  float tb2 = time_us([&] { fv2p_dcn_backward(x, w, off, m, dy, B, H, W, C, CO, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg, dx2, do2, dm2, dw2, ws2, ws2b, nullptr); }, reps);
  printf("  backward v1 %9.1f us (%6.1f TF/s, %.3f of peak)   new %9.1f us (%6.1f TF/s, %.3f of peak)\n", tb1, 2 * fl / tb1 / 1e6, 2 * fl / tb1 / 1e6 / 157.3, tb2,
         2 * fl / tb2 / 1e6, 2 * fl / tb2 / 1e6 / 157.3);
  return 0;
}
