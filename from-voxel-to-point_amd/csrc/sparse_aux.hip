// A7 — sparse max-pool and neighbour "group" over the rulebook tables.
// Replaces indice_maxpool_{fp32,half}(+backward) (pcdet/ops/spconv/include/spconv/pool_ops.h:25-94,
// src/maxpool_cuda.cu:28-454) and indice_group_fp32(+backward) (include/spconv/group_ops.h:29-291).
// Both are pure gathers once the rulebook is a table: one thread per (row, channel quad), no atomics.
#include "common.hpp"

namespace fv2p {

// out[o,c] = max(0, max_k in[tab[k][o], c])  — the reference starts from a zero-filled output and only
// overwrites when in > out (pool_ops.h:34, maxpool_cuda.cu:47-49), so negatives clamp to 0.
__global__ void maxpool_fwd(const float* __restrict__ in, int c, const int* __restrict__ tab, int kvol, int n_out, int flip,
                            float* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<int64_t>(n_out) * c) return;
  const int o = static_cast<int>(t / c), ch = static_cast<int>(t % c);
  float m = 0.f;
  for (int k = 0; k < kvol; ++k) {
    const int i = tab[static_cast<int64_t>(flip ? kvol - 1 - k : k) * n_out + o];
    if (i >= 0) {
      const float v = in[static_cast<int64_t>(i) * c + ch];
      if (v > m) m = v;
    }
  }
  out[t] = m;
}

// din[i,c] = sum_k [in[i,c] == out[tab_in[k][i], c]] * dout[tab_in[k][i], c]   (maxpool_cuda.cu:188-190)
__global__ void maxpool_bwd(const float* __restrict__ in, const float* __restrict__ out, const float* __restrict__ dout, int n_in,
                            int c, const int* __restrict__ tab_in, int kvol, float* __restrict__ din) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<int64_t>(n_in) * c) return;
  const int i = static_cast<int>(t / c), ch = static_cast<int>(t % c);
  const float v = in[t];
  float g = 0.f;
  for (int k = 0; k < kvol; ++k) {
    const int o = tab_in[static_cast<int64_t>(k) * n_in + i];
    if (o >= 0 && out[static_cast<int64_t>(o) * c + ch] == v) g += dout[static_cast<int64_t>(o) * c + ch];
  }
  din[t] = g;
}

// out[k, o, :] = in[tab[k][o], :] or 0
__global__ void group_fwd(const float* __restrict__ in, int c, const int* __restrict__ tab, int kvol, int n_out, int flip,
                          float* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t per_k = static_cast<int64_t>(n_out) * c;
  if (t >= per_k * kvol) return;
  const int k = static_cast<int>(t / per_k);
  const int64_t r = t % per_k;
  const int o = static_cast<int>(r / c), ch = static_cast<int>(r % c);
  const int i = tab[static_cast<int64_t>(flip ? kvol - 1 - k : k) * n_out + o];
  out[t] = i >= 0 ? in[static_cast<int64_t>(i) * c + ch] : 0.f;
}

// din[i,:] = sum_k g[k, tab[k][i], :]
__global__ void group_bwd(const float* __restrict__ g, int n_out, int c, const int* __restrict__ tab, int kvol, int n_in, int flip,
                          float* __restrict__ din) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<int64_t>(n_in) * c) return;
  const int i = static_cast<int>(t / c), ch = static_cast<int>(t % c);
  float s = 0.f;
  for (int k = 0; k < kvol; ++k) {
    const int o = tab[static_cast<int64_t>(flip ? kvol - 1 - k : k) * n_in + i];
    if (o >= 0) s += g[(static_cast<int64_t>(k) * n_out + o) * c + ch];
  }
  din[t] = s;
}

}  // namespace fv2p
using namespace fv2p;

#define FV2P_GRID1D(total) dim3(static_cast<unsigned>(ceil_div((total), 256))), dim3(256)

extern "C" int fv2p_sparse_maxpool_fwd(const float* in, int64_t n_in, int c, const int* tab, int kvol, int64_t n_out, int flip_k,
                                       float* out, fv2p_stream_t s) {
  FV2P_REQUIRE(c >= 1 && kvol >= 1 && n_out >= 0, FV2P_EINVAL, "maxpool_fwd: bad sizes");
  if (n_out == 0) return 0;
  FV2P_REQUIRE(in && tab && out, FV2P_EINVAL, "maxpool_fwd: null pointer");
  hipLaunchKernelGGL(maxpool_fwd, FV2P_GRID1D(n_out * c), 0, static_cast<hipStream_t>(s), in, c, tab, kvol, (int)n_out, flip_k, out);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_sparse_maxpool_bwd(const float* in, const float* out, const float* dout, int64_t n_in, int c, const int* tab_in,
                                       int kvol, float* din, fv2p_stream_t s) {
  FV2P_REQUIRE(c >= 1 && kvol >= 1 && n_in >= 0, FV2P_EINVAL, "maxpool_bwd: bad sizes");
  if (n_in == 0) return 0;
  FV2P_REQUIRE(in && out && dout && tab_in && din, FV2P_EINVAL, "maxpool_bwd: null pointer");
  hipLaunchKernelGGL(maxpool_bwd, FV2P_GRID1D(n_in * c), 0, static_cast<hipStream_t>(s), in, out, dout, (int)n_in, c, tab_in, kvol, din);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_sparse_group_fwd(const float* in, int64_t n_in, int c, const int* tab, int kvol, int64_t n_out, int flip_k,
                                     float* out, fv2p_stream_t s) {
  FV2P_REQUIRE(c >= 1 && kvol >= 1 && n_out >= 0, FV2P_EINVAL, "group_fwd: bad sizes");
  if (n_out == 0) return 0;
  FV2P_REQUIRE(in && tab && out, FV2P_EINVAL, "group_fwd: null pointer");
  hipLaunchKernelGGL(group_fwd, FV2P_GRID1D(n_out * c * kvol), 0, static_cast<hipStream_t>(s), in, c, tab, kvol, (int)n_out, flip_k, out);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_sparse_group_bwd(const float* grad, int64_t n_out, int c, const int* tab, int kvol, int64_t n_in, int flip_k,
                                     float* din, fv2p_stream_t s) {
  FV2P_REQUIRE(c >= 1 && kvol >= 1 && n_in >= 0, FV2P_EINVAL, "group_bwd: bad sizes");
  if (n_in == 0) return 0;
  FV2P_REQUIRE(grad && tab && din, FV2P_EINVAL, "group_bwd: null pointer");
  hipLaunchKernelGGL(group_bwd, FV2P_GRID1D(n_in * c), 0, static_cast<hipStream_t>(s), grad, (int)n_out, c, tab, kvol, (int)n_in, flip_k, din);
  FV2P_LAUNCH_CHECK();
  return 0;
}
