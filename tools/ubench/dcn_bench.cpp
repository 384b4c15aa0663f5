// Driver for the DCN kernels without torch: times the C-ABI entry points with HIP events on random data and checks that two
// backward runs give the same bits.  (Values are checked by tests/test_dcn_gpu.py against the float64 oracle.)
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/dcn_bench.cpp -o tools/ubench/dcn_bench -Lfrom-voxel-to-point_amd/lib -lfv2p_ops \
//         -Wl,-rpath,'$ORIGIN/../../from-voxel-to-point_amd/lib'
//   tools/ubench/dcn_bench B C H W dg [reps] [offset scale] [Cout] [column-gradient cap per chunk, MB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>
#include "../../include/fv2p_ops.h"

#define CK(e) do { hipError_t ck_err_ = (e); if (ck_err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(ck_err_), __LINE__); exit(1); } } while (0)

static size_t differing(const float* da, const float* db, size_t n) {
  std::vector<float> a(n), c(n);
  CK(hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), db, n * 4, hipMemcpyDeviceToHost));
  size_t neq = 0;
  for (size_t i = 0; i < n; ++i) neq += memcmp(&a[i], &c[i], 4) != 0;
  return neq;
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
  if (argc > 9) fv2p_dcn_set_colg_cap((int64_t)(atof(argv[9]) * 1048576.0));   // cap on a chunk's column gradients in MB (0 = default)
  const int K = 9;
  const size_t npix = (size_t)B * H * W;
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_real_distribution<float> ud(0.f, 1.f);
  std::vector<float> hx(npix * C), hw((size_t)K * C * CO), hwt((size_t)K * CO * C), hb(CO), hoff((size_t)B * dg * 2 * K * H * W), hm((size_t)B * dg * K * H * W), hdy(npix * CO);
  for (auto& v : hx) v = nd(rng);
  for (auto& v : hw) v = nd(rng) * 0.05f;
  for (int k = 0; k < K; ++k) for (int ci = 0; ci < C; ++ci) for (int co = 0; co < CO; ++co) hwt[((size_t)k * CO + co) * C + ci] = hw[((size_t)k * C + ci) * CO + co];
  for (auto& v : hb) v = nd(rng);
  for (auto& v : hoff) v = nd(rng) * oscale;
  for (auto& v : hm) v = ud(rng);
  for (auto& v : hdy) v = nd(rng);
  float *x, *w, *wt, *b, *off, *m, *y, *dy, *dx[2], *dof[2], *dm[2], *dw[2];
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&wt, hwt.size() * 4)); CK(hipMalloc(&b, hb.size() * 4));
  CK(hipMalloc(&off, hoff.size() * 4)); CK(hipMalloc(&m, hm.size() * 4)); CK(hipMalloc(&y, npix * CO * 4)); CK(hipMalloc(&dy, hdy.size() * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wt, hwt.data(), hwt.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(off, hoff.data(), hoff.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(m, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dy, hdy.data(), hdy.size() * 4, hipMemcpyHostToDevice));
  for (int r = 0; r < 2; ++r) {
    CK(hipMalloc(&dx[r], npix * C * 4)); CK(hipMalloc(&dof[r], hoff.size() * 4)); CK(hipMalloc(&dm[r], hm.size() * 4)); CK(hipMalloc(&dw[r], hw.size() * 4));
    CK(hipMemset(dx[r], 0xff, npix * C * 4));
  }
  const size_t wsb = fv2p_dcn_backward_ws_bytes(B, H, W, H, W, C, CO, 3, 3, dg);
  void* ws;
  CK(hipMalloc(&ws, wsb));
  auto fwd = [&] { return fv2p_dcn_forward(x, wt, b, off, m, B, H, W, C, CO, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg, y, nullptr); };
  auto bwd = [&](int r) { return fv2p_dcn_backward(x, w, off, m, dy, B, H, W, C, CO, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg, dx[r], dof[r], dm[r], dw[r], ws, wsb, nullptr); };
  int rc = fwd();
  if (rc) { printf("forward rc %d: %s\n", rc, fv2p_last_error()); return 1; }
  for (int r = 0; r < 2; ++r) { rc = bwd(r); if (rc) { printf("backward rc %d: %s\n", rc, fv2p_last_error()); return 1; } }
  CK(hipDeviceSynchronize());
  const size_t d = differing(dx[0], dx[1], npix * C) + differing(dof[0], dof[1], hoff.size()) + differing(dm[0], dm[1], hm.size()) + differing(dw[0], dw[1], hw.size());
  const double fl = 2.0 * npix * C * CO * K;
  const float tf = time_us([&] { fwd(); }, reps), tb = time_us([&] { bwd(0); }, reps);
  printf("DCNv2 [%d,%d->%d,%d,%d] dg=%d: fwd %8.1f us (%5.1f TF/s, %.3f of the fp32-MFMA peak)  bwd %8.1f us (%5.1f TF/s, %.3f)  workspace %.0f MB  run-to-run differing words %zu\n",
         B, C, CO, H, W, dg, tf, fl / tf / 1e6, fl / tf / 1e6 / 157.3, tb, 2 * fl / tb / 1e6, 2 * fl / tb / 1e6 / 157.3, wsb / 1e6, d);
  return 0;
}
