// BatchNorm1d (+ ReLU) over sparse-tensor features [N, C] — SURVEY §8(f).3.
//
// Every conv of the reference's sparse backbones is followed by BatchNorm1d(eps=1e-3, momentum=0.01) + ReLU
// (pcdet/models/backbones_3d/spconv_backbone.py:8-27 post_act_block, :75 conv_input).  On [N, C] with C = 16..128
// torch's channels-last BN kernels run 25-30 us per call at N ~ 5e4 (profiles/r01_bench_kernel_stats.csv), i.e.
// ~0.25 TB/s; here the layer is two launches forward and two backward, plain coalesced float4 streams:
//   forward : bn_reduce_k<FWD>  per-channel sum / sum of squares in fp64, one partial per workgroup (<= 64 of them);
//             bn_apply_fwd_k    every workgroup folds the partials in the same fixed order (deterministic; the kernel
//                               boundary replaces an agent-scope fence per producer, which on gfx950 writes the whole
//                               L2 back: measured 20 us per launch), workgroup 0 also stores mean / invstd and updates
//                               the running statistics; then y = relu((x - mean) * invstd * gamma + beta)
//   backward: bn_reduce_k<BWD>  dz = dy * [y > 0];  sum dz, sum dz * xhat
//             bn_apply_bwd_k    folds to dbeta, dgamma, c1, c2;  dx = gamma * invstd * (dz - c1 - xhat * c2)
// The ReLU mask is recomputed from x, so y is not needed by the backward pass.
#include "common.hpp"
#include "bn_fold.hpp"

namespace fv2p {

template <int V>
struct Vec;
template <>
struct Vec<4> {
  float v[4];
  __device__ __forceinline__ void load(const float* p, int, int) { const float4 q = *reinterpret_cast<const float4*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <>
struct Vec<1> {
  float v[1];
  __device__ __forceinline__ void load(const float* p, int, int) { v[0] = *p; }
  __device__ __forceinline__ void store(float* p) const { *p = v[0]; }
};

// Workgroup-level reduction of per-thread (a, b) channel sums.  partial: [nblk][2][c] doubles.
// mask_y (backward only, the residual form): the ReLU that follows is relu(bn(x) + identity), so its mask is read from the block's
// OUTPUT (mask_y > 0) instead of being recomputed from x.
template <int V, bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_k(const float* __restrict__ x, const float* __restrict__ dy, BnGeom g,
                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                   double* __restrict__ partial, const float* __restrict__ mask_y) {
  __shared__ double red[2][256 * V];
  const int tid = threadIdx.x;
  const int rl = tid / g.tcols, cq = tid % g.tcols;
  const int col = cq * V;
  const bool live = rl < g.rpp && col < g.c;
  double s1[V], s2[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s1[i] = s2[i] = 0.0;
  if (live) {
    float m[V] = {}, is[V] = {}, ga[V] = {}, be[V] = {};
    if (BWD) {
#pragma unroll
      for (int i = 0; i < V; ++i) { m[i] = mean[col + i]; is[i] = invstd[col + i]; ga[i] = gamma ? gamma[col + i] : 1.f; be[i] = beta ? beta[col + i] : 0.f; }
    }
    const long long r0 = static_cast<long long>(blockIdx.x) * g.rows_per_block;
    const long long r1 = min(r0 + g.rows_per_block, g.n);
    // U rows in flight per thread: the pass is latency bound (a few MB spread over the whole chip), so the loads of
    // a group are all issued before the first fp64 add
    constexpr int U = 8;
    auto accumulate = [&](const Vec<V>& xv, const Vec<V>& gv, const Vec<V>& mv) {
      if (!BWD) {
#pragma unroll
        for (int i = 0; i < V; ++i) { const double d = xv.v[i]; s1[i] += d; s2[i] += d * d; }
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float xhat = (xv.v[i] - m[i]) * is[i];
          const float y = mask_y ? mv.v[i] : xhat * ga[i] + be[i];
          const float dz = (relu && !(y > 0.f)) ? 0.f : gv.v[i];
          s1[i] += dz; s2[i] += static_cast<double>(dz) * xhat;
        }
      }
    };
    long long r = r0 + rl;
    for (; r + static_cast<long long>(U - 1) * g.rpp < r1; r += static_cast<long long>(U) * g.rpp) {
      Vec<V> xv[U], gv[U], mv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        xv[u].load(x + (r + static_cast<long long>(u) * g.rpp) * g.c + col, 0, 0);
        if (BWD) gv[u].load(dy + (r + static_cast<long long>(u) * g.rpp) * g.c + col, 0, 0);
        if (BWD && mask_y) mv[u].load(mask_y + (r + static_cast<long long>(u) * g.rpp) * g.c + col, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accumulate(xv[u], gv[u], mv[u]);
    }
    for (; r < r1; r += g.rpp) {
      Vec<V> xv, gv, mv;
      xv.load(x + r * g.c + col, 0, 0);
      if (BWD) gv.load(dy + r * g.c + col, 0, 0);
      if (BWD && mask_y) mv.load(mask_y + r * g.c + col, 0, 0);
      accumulate(xv, gv, mv);
    }
  }
  // fold the row lanes: red[.][rl * c + col]
  if (rl < g.rpp) {
#pragma unroll
    for (int i = 0; i < V; ++i) { red[0][(rl * g.tcols + cq) * V + i] = s1[i]; red[1][(rl * g.tcols + cq) * V + i] = s2[i]; }
  }
  __syncthreads();
  const int cpad = g.tcols * V;
  for (int e = tid; e < cpad; e += 256) {
    double a = 0.0, b = 0.0;
    for (int q = 0; q < g.rpp; ++q) { a += red[0][q * cpad + e]; b += red[1][q * cpad + e]; }
    if (e < g.c) {
      partial[(static_cast<long long>(blockIdx.x) * 2 + 0) * g.c + e] = a;
      partial[(static_cast<long long>(blockIdx.x) * 2 + 1) * g.c + e] = b;
    }
  }
}

constexpr int kBnMaxC = 1024;

// FOLD: batch statistics come from the partials (training); otherwise mean / invstd are read from memory (eval mode, or statistics
// the producing conv's last workgroup has finalised already: bn_fold.hpp).
// residual (optional): y = relu?((x - mean) * invstd * gamma + beta + residual) - the tail of a residual block
// (spconv_backbone.py:63-66: out.features += identity; relu) in the same pass.
template <int V, bool FOLD>
__global__ __launch_bounds__(256) void bn_apply_fwd_k(const float* __restrict__ x, long long units, BnGeom g, const double* __restrict__ partial,
                                                      BnFwdFin ff, const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                      float* __restrict__ y, double* __restrict__ zero_next, long long zero_count,
                                                      const float* __restrict__ residual) {
  __shared__ double red[2][256];
  // conv-epilogue statistics alternate between two slot buffers: this launch reads one and clears the other for the
  // next fused conv, which runs after it on the stream
  if (zero_next && blockIdx.x == 0)
    for (long long e = threadIdx.x; e < zero_count; e += 256) zero_next[e] = 0.0;
  __shared__ float sh_scale[kBnMaxC], sh_shift[kBnMaxC];   // y = x * scale + shift would change rounding: keep (x-mean)*invstd*gamma+beta
  __shared__ float sh_mean[kBnMaxC], sh_gamma[kBnMaxC];
  const int tid = threadIdx.x, c = g.c;
  const int cfold = c < 256 ? c : 256;
  for (int e0 = 0; e0 < c; e0 += cfold) {
    const int e = e0 + tid;
    float mu_f = 0.f, is_f = 0.f;
    if (FOLD) {
      double a, b;
      fold_chunk<false>(g.nblk, c, partial, e0, cfold, red, &a, &b);
      if (tid < cfold && e < c) bn_fwd_channel(a, b, g.n, ff, e, blockIdx.x == 0, &mu_f, &is_f);
    } else {
      mu_f = (tid < cfold && e < c) ? ff.mean[e] : 0.f;
      is_f = (tid < cfold && e < c) ? ff.invstd[e] : 0.f;
    }
    if (tid < cfold && e < c) {
      sh_mean[e] = mu_f;
      sh_scale[e] = is_f;
      sh_gamma[e] = gamma ? gamma[e] : 1.f;
      sh_shift[e] = beta ? beta[e] : 0.f;
    }
  }
  __syncthreads();
  if (FOLD && blockIdx.x == 0 && tid == 0 && ff.running_mean && ff.num_batches_tracked) *ff.num_batches_tracked += 1;
  const int cv = c / V;
  for (long long u = static_cast<long long>(blockIdx.x) * 256 + tid; u < units; u += static_cast<long long>(gridDim.x) * 256) {
    const int col = static_cast<int>(u % cv) * V;
    Vec<V> xv, rv, o;
    xv.load(x + u * V, 0, 0);
    if (residual) rv.load(residual + u * V, 0, 0);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float xhat = (xv.v[i] - sh_mean[col + i]) * sh_scale[col + i];
      float t = xhat * sh_gamma[col + i] + sh_shift[col + i];
      if (residual) t = t + rv.v[i];
      o.v[i] = (relu && t <= 0.f) ? 0.f : t;  // NaN passes through, like torch.relu
    }
    o.store(y + u * V);
  }
}

// FOLD: (sum dz, sum dz * xhat) come from the partials; otherwise their means c1, c2 are read from bf.coef (finalised, with dgamma and
// dbeta, by the last workgroup of the backward-data conv that took the sums).
// mask_y / dz_out (the residual form): the ReLU mask comes from the block's output, and dz = dy * [mask_y > 0] - the gradient of the
// identity branch - is written beside dx.
template <int V, bool FOLD>
__global__ __launch_bounds__(256) void bn_apply_bwd_k(const float* __restrict__ x, const float* __restrict__ dy, long long units, BnGeom g,
                                                      const double* __restrict__ partial, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int relu, BnBwdFin bf, float* __restrict__ dx,
                                                      double* __restrict__ zero_next, long long zero_count, const float* __restrict__ mask_y,
                                                      float* __restrict__ dz_out) {
  __shared__ double red[2][256];
  if (zero_next && blockIdx.x == 0)   // see bn_apply_fwd_k
    for (long long e = threadIdx.x; e < zero_count; e += 256) zero_next[e] = 0.0;
  __shared__ float sh_mean[kBnMaxC], sh_is[kBnMaxC], sh_gamma[kBnMaxC], sh_beta[kBnMaxC], sh_c1[kBnMaxC], sh_c2[kBnMaxC];
  const int tid = threadIdx.x, c = g.c;
  const int cfold = c < 256 ? c : 256;
  for (int e0 = 0; e0 < c; e0 += cfold) {
    const int e = e0 + tid;
    double a = 0.0, b = 0.0;
    if (FOLD) fold_chunk<false>(g.nblk, c, partial, e0, cfold, red, &a, &b);
    if (tid < cfold && e < c) {
      if (FOLD) {
        const double n = static_cast<double>(g.n);
        if (blockIdx.x == 0) {
          bf.dbeta[e] = static_cast<float>(a);
          bf.dgamma[e] = static_cast<float>(b);
        }
        sh_c1[e] = bf.batch_stats ? static_cast<float>(a / n) : 0.f;
        sh_c2[e] = bf.batch_stats ? static_cast<float>(b / n) : 0.f;
      } else {
        sh_c1[e] = bf.coef[e];
        sh_c2[e] = bf.coef[c + e];
      }
      sh_mean[e] = mean[e];
      sh_is[e] = invstd[e];
      sh_gamma[e] = gamma ? gamma[e] : 1.f;
      sh_beta[e] = beta ? beta[e] : 0.f;
    }
  }
  __syncthreads();
  const int cv = c / V;
  for (long long u = static_cast<long long>(blockIdx.x) * 256 + tid; u < units; u += static_cast<long long>(gridDim.x) * 256) {
    const int col = static_cast<int>(u % cv) * V;
    Vec<V> xv, gv, mv, o, z;
    xv.load(x + u * V, 0, 0);
    gv.load(dy + u * V, 0, 0);
    if (mask_y) mv.load(mask_y + u * V, 0, 0);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float is = sh_is[col + i], ga = sh_gamma[col + i];
      const float xhat = (xv.v[i] - sh_mean[col + i]) * is;
      const float t = mask_y ? mv.v[i] : xhat * ga + sh_beta[col + i];
      const float dz = (relu && !(t > 0.f)) ? 0.f : gv.v[i];
      z.v[i] = dz;
      o.v[i] = ga * is * (dz - sh_c1[col + i] - xhat * sh_c2[col + i]);
    }
    o.store(dx + u * V);
    if (dz_out) z.store(dz_out + u * V);
  }
}

constexpr int kBnPartials = 64;
static_assert(kStatSlots <= kBnPartials, "the fallback passes write their partials into the statistics slots");

static int bn_geom(int64_t n, int c, bool vec, BnGeom* g, int max_blocks = kBnPartials) {
  const int v = vec ? 4 : 1;
  g->n = n; g->c = c;
  g->tcols = static_cast<int>(ceil_div(c, v));
  if (g->tcols > 256 || c > kBnMaxC) return -1;
  g->rpp = 256 / g->tcols;
  // >= 8 rows per row lane per workgroup, at most kBnPartials workgroups (every apply workgroup re-folds the partials)
  int64_t nblk = ceil_div(n, static_cast<int64_t>(g->rpp) * 8);
  if (nblk > max_blocks) nblk = max_blocks;
  if (nblk < 1) nblk = 1;
  g->rows_per_block = ceil_div(n, nblk);
  g->nblk = static_cast<int>(ceil_div(n, g->rows_per_block));
  return 0;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static unsigned apply_blocks(long long units) {
  // grid-stride, ~4 units per thread, at most 2 workgroups per CU worth of fold prologues
  const int64_t b = ceil_div(units, 256 * 4);
  return static_cast<unsigned>(b > 512 ? 512 : (b < 1 ? 1 : b));
}

}  // namespace fv2p

using namespace fv2p;

extern "C" size_t fv2p_batchnorm_ws_bytes(int64_t n, int c) {
  (void)n;
  Sizer s;
  s.take<double>(static_cast<size_t>(kBnPartials) * 2 * (c > 0 ? c : 1));
  return s.bytes();
}

extern "C" int fv2p_batchnorm_forward(const float* x, int64_t n, int c, float eps, float momentum, const float* gamma, const float* beta,
                                      int relu, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean,
                                      float* invstd, float* y, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 1 && c >= 1, FV2P_EINVAL, "batchnorm_forward: n=%lld c=%d", static_cast<long long>(n), c);
  FV2P_REQUIRE(x && y && mean && invstd && ws, FV2P_EINVAL, "batchnorm_forward: null pointer");
  FV2P_REQUIRE((running_mean == nullptr) == (running_var == nullptr), FV2P_EINVAL, "batchnorm_forward: running_mean and running_var come together");
  FV2P_REQUIRE(ws_bytes >= fv2p_batchnorm_ws_bytes(n, c), FV2P_EWORKSPACE, "batchnorm_forward: workspace %lld < %lld",
               static_cast<long long>(ws_bytes), static_cast<long long>(fv2p_batchnorm_ws_bytes(n, c)));
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(y);
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  Carver cv(ws, ws_bytes);
  double* partial = cv.take<double>(static_cast<size_t>(kBnPartials) * 2 * c);
  BnFwdFin ff{mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), momentum, eps};
  const long long units = n * c / (vec ? 4 : 1);
  const unsigned blocks = apply_blocks(units);
  if (vec) {
    hipLaunchKernelGGL((bn_reduce_k<4, false>), dim3(g.nblk), dim3(256), 0, stream, x, nullptr, g, nullptr, nullptr, nullptr, nullptr, 0, partial, nullptr);
    hipLaunchKernelGGL((bn_apply_fwd_k<4, true>), dim3(blocks), dim3(256), 0, stream, x, units, g, partial, ff, gamma, beta, relu, y, nullptr, 0, nullptr);
  } else {
    hipLaunchKernelGGL((bn_reduce_k<1, false>), dim3(g.nblk), dim3(256), 0, stream, x, nullptr, g, nullptr, nullptr, nullptr, nullptr, 0, partial, nullptr);
    hipLaunchKernelGGL((bn_apply_fwd_k<1, true>), dim3(blocks), dim3(256), 0, stream, x, units, g, partial, ff, gamma, beta, relu, y, nullptr, 0, nullptr);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}

// column sums of x into the slot layout of the conv-epilogue statistics (slots beyond the reduce grid stay as they are: zero)
int fv2p::bn_column_sums(const float* x, int64_t n, int c, double* stats, hipStream_t stream) {
  const bool vec = (c % 4 == 0) && aligned16(x);
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g, kStatSlots) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  if (vec) hipLaunchKernelGGL((bn_reduce_k<4, false>), dim3(g.nblk), dim3(256), 0, stream, x, nullptr, g, nullptr, nullptr, nullptr, nullptr, 0, stats, nullptr);
  else hipLaunchKernelGGL((bn_reduce_k<1, false>), dim3(g.nblk), dim3(256), 0, stream, x, nullptr, g, nullptr, nullptr, nullptr, nullptr, 0, stats, nullptr);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// (sum dz, sum dz * xhat) into the same slot layout, for a backward-data conv that could not take them in its epilogue
int fv2p::bn_backward_sums(const float* x, const float* dy, int64_t n, int c, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, int relu, double* stats, hipStream_t stream) {
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(dy);
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g, kStatSlots) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  if (vec) hipLaunchKernelGGL((bn_reduce_k<4, true>), dim3(g.nblk), dim3(256), 0, stream, x, dy, g, mean, invstd, gamma, beta, relu, stats, nullptr);
  else hipLaunchKernelGGL((bn_reduce_k<1, true>), dim3(g.nblk), dim3(256), 0, stream, x, dy, g, mean, invstd, gamma, beta, relu, stats, nullptr);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_batchnorm_forward_stats(const float* x, int64_t n, int c, float eps, float momentum, const float* gamma, const float* beta,
                                            int relu, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean,
                                            float* invstd, float* y, const double* stats, double* zero_next, int64_t zero_count,
                                            fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 1 && c >= 1, FV2P_EINVAL, "batchnorm_forward_stats: n=%lld c=%d", static_cast<long long>(n), c);
  FV2P_REQUIRE(x && y && mean && invstd && stats, FV2P_EINVAL, "batchnorm_forward_stats: null pointer");
  FV2P_REQUIRE((running_mean == nullptr) == (running_var == nullptr), FV2P_EINVAL, "batchnorm_forward_stats: running_mean and running_var come together");
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(y);
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  g.nblk = kStatSlots;   // every slot is folded (untouched ones hold zeros)
  BnFwdFin ff{mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), momentum, eps};
  const long long units = n * c / (vec ? 4 : 1);
  const unsigned blocks = apply_blocks(units);
  if (vec) hipLaunchKernelGGL((bn_apply_fwd_k<4, true>), dim3(blocks), dim3(256), 0, stream, x, units, g, stats, ff, gamma, beta, relu, y, zero_next, static_cast<long long>(zero_count), nullptr);
  else hipLaunchKernelGGL((bn_apply_fwd_k<1, true>), dim3(blocks), dim3(256), 0, stream, x, units, g, stats, ff, gamma, beta, relu, y, zero_next, static_cast<long long>(zero_count), nullptr);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_batchnorm_backward_stats(const float* x, const float* dy, int64_t n, int c, const float* mean, const float* invstd,
                                             const float* gamma, const float* beta, int relu, int batch_stats, float* dx, float* dgamma,
                                             float* dbeta, const double* stats, double* zero_next, int64_t zero_count, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 1 && c >= 1, FV2P_EINVAL, "batchnorm_backward_stats: n=%lld c=%d", static_cast<long long>(n), c);
  FV2P_REQUIRE(x && dy && mean && invstd && dx && dgamma && dbeta && stats, FV2P_EINVAL, "batchnorm_backward_stats: null pointer");
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(dy) && aligned16(dx);
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  g.nblk = kStatSlots;
  BnBwdFin bf{dgamma, dbeta, nullptr, batch_stats};
  const long long units = n * c / (vec ? 4 : 1);
  const unsigned blocks = apply_blocks(units);
  if (vec) hipLaunchKernelGGL((bn_apply_bwd_k<4, true>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, stats, mean, invstd, gamma, beta, relu, bf, dx,
                              zero_next, static_cast<long long>(zero_count), nullptr, nullptr);
  else hipLaunchKernelGGL((bn_apply_bwd_k<1, true>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, stats, mean, invstd, gamma, beta, relu, bf, dx,
                          zero_next, static_cast<long long>(zero_count), nullptr, nullptr);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_batchnorm_apply(const float* x, int64_t n, int c, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, int relu, float* y, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 0 && c >= 1, FV2P_EINVAL, "batchnorm_apply: n=%lld c=%d", static_cast<long long>(n), c);
  if (n == 0) return 0;
  FV2P_REQUIRE(x && y && mean && invstd, FV2P_EINVAL, "batchnorm_apply: null pointer");
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(y);
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  BnFwdFin ff{const_cast<float*>(mean), const_cast<float*>(invstd), nullptr, nullptr, nullptr, 0.f, 0.f};
  const long long units = n * c / (vec ? 4 : 1);
  const unsigned blocks = apply_blocks(units);
  if (vec) hipLaunchKernelGGL((bn_apply_fwd_k<4, false>), dim3(blocks), dim3(256), 0, stream, x, units, g, nullptr, ff, gamma, beta, relu, y, nullptr, 0, nullptr);
  else hipLaunchKernelGGL((bn_apply_fwd_k<1, false>), dim3(blocks), dim3(256), 0, stream, x, units, g, nullptr, ff, gamma, beta, relu, y, nullptr, 0, nullptr);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_batchnorm_backward(const float* x, const float* dy, int64_t n, int c, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, int relu, int batch_stats, float* dx, float* dgamma,
                                       float* dbeta, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 1 && c >= 1, FV2P_EINVAL, "batchnorm_backward: n=%lld c=%d", static_cast<long long>(n), c);
  FV2P_REQUIRE(x && dy && mean && invstd && dx && dgamma && dbeta && ws, FV2P_EINVAL, "batchnorm_backward: null pointer");
  FV2P_REQUIRE(ws_bytes >= fv2p_batchnorm_ws_bytes(n, c), FV2P_EWORKSPACE, "batchnorm_backward: workspace %lld < %lld",
               static_cast<long long>(ws_bytes), static_cast<long long>(fv2p_batchnorm_ws_bytes(n, c)));
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(dy) && aligned16(dx);
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  Carver cv(ws, ws_bytes);
  double* partial = cv.take<double>(static_cast<size_t>(kBnPartials) * 2 * c);
  BnBwdFin bf{dgamma, dbeta, nullptr, batch_stats};
  const long long units = n * c / (vec ? 4 : 1);
  const unsigned blocks = apply_blocks(units);
  if (vec) {
    hipLaunchKernelGGL((bn_reduce_k<4, true>), dim3(g.nblk), dim3(256), 0, stream, x, dy, g, mean, invstd, gamma, beta, relu, partial, nullptr);
    hipLaunchKernelGGL((bn_apply_bwd_k<4, true>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, partial, mean, invstd, gamma, beta, relu, bf, dx, nullptr, 0, nullptr, nullptr);
  } else {
    hipLaunchKernelGGL((bn_reduce_k<1, true>), dim3(g.nblk), dim3(256), 0, stream, x, dy, g, mean, invstd, gamma, beta, relu, partial, nullptr);
    hipLaunchKernelGGL((bn_apply_bwd_k<1, true>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, partial, mean, invstd, gamma, beta, relu, bf, dx, nullptr, 0, nullptr, nullptr);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}

// One workgroup: the fold + finalisation the last workgroup of a conv launch does (conv_stats_done, sparse_conv.hip), for sums that a
// separate pass took (column counts the conv epilogues do not cover).  Leaves the slots zero like that one.
template <bool BWD>
__global__ __launch_bounds__(256) void bn_finalize_k(double* __restrict__ stats, long long n, int c, BnFwdFin ff, BnBwdFin bf) {
  __shared__ double red[2][256];
  const int tid = threadIdx.x, cfold = c < 256 ? c : 256;
  for (int e0 = 0; e0 < c; e0 += cfold) {
    const int e = e0 + tid;
    double a, b;
    fold_chunk<false>(kStatSlots, c, stats, e0, cfold, red, &a, &b);
    if (tid < cfold && e < c) {
      if (!BWD) {
        float mu, is;
        bn_fwd_channel(a, b, n, ff, e, true, &mu, &is);
      } else {
        const double nn = static_cast<double>(n);
        bf.dbeta[e] = static_cast<float>(a);
        bf.dgamma[e] = static_cast<float>(b);
        bf.coef[e] = bf.batch_stats ? static_cast<float>(a / nn) : 0.f;
        bf.coef[c + e] = bf.batch_stats ? static_cast<float>(b / nn) : 0.f;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < kStatSlots * 2 * c; i += 256) stats[i] = 0.0;
  if (!BWD && tid == 0 && ff.running_mean && ff.num_batches_tracked) *ff.num_batches_tracked += 1;
}
int fv2p::bn_finalize_forward(double* stats, int64_t n, int c, const BnFwdFin& ff, hipStream_t stream) {
  FV2P_REQUIRE(c <= kBnMaxC, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, kBnMaxC);
  hipLaunchKernelGGL((bn_finalize_k<false>), dim3(1), dim3(256), 0, stream, stats, static_cast<long long>(n), c, ff, BnBwdFin{nullptr, nullptr, nullptr, 1});
  FV2P_LAUNCH_CHECK();
  return 0;
}
int fv2p::bn_finalize_backward(double* stats, int64_t n, int c, const BnBwdFin& bf, hipStream_t stream) {
  FV2P_REQUIRE(c <= kBnMaxC, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, kBnMaxC);
  hipLaunchKernelGGL((bn_finalize_k<true>), dim3(1), dim3(256), 0, stream, stats, static_cast<long long>(n), c, BnFwdFin{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f}, bf);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// ---- round 6: statistics finalised by the producer, the residual tail, one-launch forms ------------------------------------------

// y = relu?((x - mean) * invstd * gamma + beta [+ residual]) with given mean / invstd: eval mode, or batch statistics the producing
// conv's last workgroup finalised (fv2p_sparse_conv_rows_bnfin).  One launch, no fold.
extern "C" int fv2p_batchnorm_apply_res(const float* x, int64_t n, int c, const float* mean, const float* invstd, const float* gamma,
                                        const float* beta, int relu, const float* residual, float* y, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 0 && c >= 1, FV2P_EINVAL, "batchnorm_apply_res: n=%lld c=%d", static_cast<long long>(n), c);
  if (n == 0) return 0;
  FV2P_REQUIRE(x && y && mean && invstd, FV2P_EINVAL, "batchnorm_apply_res: null pointer");
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(y) && (!residual || aligned16(residual));
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  BnFwdFin ff{const_cast<float*>(mean), const_cast<float*>(invstd), nullptr, nullptr, nullptr, 0.f, 0.f};
  const long long units = n * c / (vec ? 4 : 1);
  const unsigned blocks = apply_blocks(units);
  if (vec) hipLaunchKernelGGL((bn_apply_fwd_k<4, false>), dim3(blocks), dim3(256), 0, stream, x, units, g, nullptr, ff, gamma, beta, relu, y, nullptr, 0, residual);
  else hipLaunchKernelGGL((bn_apply_fwd_k<1, false>), dim3(blocks), dim3(256), 0, stream, x, units, g, nullptr, ff, gamma, beta, relu, y, nullptr, 0, residual);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// dx = gamma * invstd * (dz - c1 - xhat * c2) with c1 = coef[0][c], c2 = coef[1][c] finalised (with dgamma / dbeta) by the last
// workgroup of the backward-data conv that took the sums (fv2p_sparse_conv_rows_bnbwd_fin).  One launch, no fold.
extern "C" int fv2p_batchnorm_backward_fin(const float* x, const float* dy, int64_t n, int c, const float* mean, const float* invstd,
                                           const float* gamma, const float* beta, int relu, const float* coef, float* dx,
                                           fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 1 && c >= 1, FV2P_EINVAL, "batchnorm_backward_fin: n=%lld c=%d", static_cast<long long>(n), c);
  FV2P_REQUIRE(x && dy && mean && invstd && dx && coef, FV2P_EINVAL, "batchnorm_backward_fin: null pointer");
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(dy) && aligned16(dx);
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  BnBwdFin bf{nullptr, nullptr, const_cast<float*>(coef), 1};
  const long long units = n * c / (vec ? 4 : 1);
  const unsigned blocks = apply_blocks(units);
  if (vec) hipLaunchKernelGGL((bn_apply_bwd_k<4, false>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, nullptr, mean, invstd, gamma, beta, relu, bf, dx,
                              nullptr, 0, nullptr, nullptr);
  else hipLaunchKernelGGL((bn_apply_bwd_k<1, false>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, nullptr, mean, invstd, gamma, beta, relu, bf, dx,
                          nullptr, 0, nullptr, nullptr);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// Backward of out = relu(bn(x) + identity) (the tail of a residual block): dz = dout * [out > 0] is the identity branch's gradient
// (written to dz) and the BatchNorm's incoming gradient; dx, dgamma, dbeta as fv2p_batchnorm_backward.  Two launches
// (reduce with the mask read from `out`, apply) instead of torch's threshold_backward + the two of fv2p_batchnorm_backward.
extern "C" int fv2p_batchnorm_backward_res(const float* x, const float* out, const float* dout, int64_t n, int c, const float* mean,
                                           const float* invstd, const float* gamma, const float* beta, int batch_stats, float* dx, float* dz,
                                           float* dgamma, float* dbeta, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 1 && c >= 1, FV2P_EINVAL, "batchnorm_backward_res: n=%lld c=%d", static_cast<long long>(n), c);
  FV2P_REQUIRE(x && out && dout && mean && invstd && dx && dz && dgamma && dbeta && ws, FV2P_EINVAL, "batchnorm_backward_res: null pointer");
  FV2P_REQUIRE(ws_bytes >= fv2p_batchnorm_ws_bytes(n, c), FV2P_EWORKSPACE, "batchnorm_backward_res: workspace too small");
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(out) && aligned16(dout) && aligned16(dx) && aligned16(dz);
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  Carver cv(ws, ws_bytes);
  double* partial = cv.take<double>(static_cast<size_t>(kBnPartials) * 2 * c);
  BnBwdFin bf{dgamma, dbeta, nullptr, batch_stats};
  const long long units = n * c / (vec ? 4 : 1);
  const unsigned blocks = apply_blocks(units);
  if (vec) {
    hipLaunchKernelGGL((bn_reduce_k<4, true>), dim3(g.nblk), dim3(256), 0, stream, x, dout, g, mean, invstd, gamma, beta, 1, partial, out);
    hipLaunchKernelGGL((bn_apply_bwd_k<4, true>), dim3(blocks), dim3(256), 0, stream, x, dout, units, g, partial, mean, invstd, gamma, beta, 1, bf, dx, nullptr, 0, out, dz);
  } else {
    hipLaunchKernelGGL((bn_reduce_k<1, true>), dim3(g.nblk), dim3(256), 0, stream, x, dout, g, mean, invstd, gamma, beta, 1, partial, out);
    hipLaunchKernelGGL((bn_apply_bwd_k<1, true>), dim3(blocks), dim3(256), 0, stream, x, dout, units, g, partial, mean, invstd, gamma, beta, 1, bf, dx, nullptr, 0, out, dz);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}

// ---- one launch per BatchNorm pass: reduce, grid barrier, apply ----------------------------------------------------------------------
// The BatchNorms that do not sit behind one of this library's convs (the decoder's Linear -> BatchNorm1d -> ReLU rows, the residual
// tails, every layer whose incoming gradient is a sum of several) took two launches forward and two backward: a reduce pass on <= 64
// workgroups and the apply pass, with the launch gap of two dependent kernels between them (bn_reduce_k<BWD> alone: 25 us x 29 per FV2P
// step).  Here both phases are one kernel of G <= kOneGrid workgroups that are all resident at once (256 threads, 20 KB of LDS: a CU
// holds eight, the chip 2 048 - several such launches from different streams still fit side by side, and kernels that do not wait for
// anybody finish in between).  Phase 1: workgroup b reduces its rows and publishes partial[b] with agent-scope stores (write-through:
// the eight XCD L2s are not coherent with each other), waits for their acknowledgement and counts itself in.  Barrier: one thread per
// workgroup polls the counter with agent-scope loads.  Phase 2: every workgroup folds the G partials (fold_chunk: the same fixed order
// as the two-launch form, so the statistics are bit-identical to it) and applies its share of the rows.  The last workgroup to leave
// resets the two counters.  No fence: nothing but the partials and the counters crosses workgroups.
constexpr int kOneGrid = 128;
// Every workgroup folds ALL partials with loads that bypass its L2: G^2 x 2 c doubles cross the fabric per launch (33 MB at G = 128,
// c = 128: 46 against 33 us for the two-launch forward at 10 000 x 128).  Wide layers therefore run on 32 workgroups (2 MB); what that
// leaves of the chip is enough up to ~2 M elements (tools/bn_time.py), above that the caller keeps reduce + apply.
static int one_grid(int c) { return c >= 64 ? 32 : kOneGrid; }
extern "C" int fv2p_batchnorm_one_pays(int64_t n, int c, int backward) {
  // measured against the wide reduce + apply (tools/bn_time.py, profiles/r06_bn_time.txt): backward 64 against 73 us at 35 000 x 16 and
  // 59 against 70 at 39 000 x 32, but 81 against 72 at 22 000 x 64 and 82 against 69 at 10 000 x 128 (32 workgroups are too few for
  // the two passes over the tensor); forward equal at 16 columns, behind from 32 on
  const int64_t elems = n * c;
  if (elems > (2ll << 20)) return 0;
  return backward ? (c <= 32 ? 1 : 0) : (c <= 16 ? 1 : 0);
}

__device__ __forceinline__ void grid_arrive_and_wait(unsigned* counters, unsigned total) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this workgroup's partials have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
}
__device__ __forceinline__ void grid_leave(unsigned* counters, unsigned total) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (__hip_atomic_fetch_add(counters + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1u) {
      __hip_atomic_store(counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(counters + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// phase 1 of both directions: per-channel (a, b) sums of this workgroup's rows -> partial[blockIdx][2][c], published agent-wide
template <int V, bool BWD>
__device__ __forceinline__ void one_reduce(const float* __restrict__ x, const float* __restrict__ dy, const BnGeom& g, const float* mean,
                                           const float* invstd, const float* gamma, const float* beta, int relu, const float* mask_y,
                                           double* partial, double (*red)[256 * V]) {
  const int tid = threadIdx.x;
  const int rl = tid / g.tcols, cq = tid % g.tcols;
  const int col = cq * V;
  const bool live = rl < g.rpp && col < g.c;
  double s1[V], s2[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s1[i] = s2[i] = 0.0;
  if (live) {
    float m[V] = {}, is[V] = {}, ga[V] = {}, be[V] = {};
    if (BWD) {
#pragma unroll
      for (int i = 0; i < V; ++i) { m[i] = mean[col + i]; is[i] = invstd[col + i]; ga[i] = gamma ? gamma[col + i] : 1.f; be[i] = beta ? beta[col + i] : 0.f; }
    }
    const long long r0 = static_cast<long long>(blockIdx.x) * g.rows_per_block;
    const long long r1 = min(r0 + g.rows_per_block, g.n);
    constexpr int U = 8;
    auto accumulate = [&](const Vec<V>& xv, const Vec<V>& gv, const Vec<V>& mv) {
      if (!BWD) {
#pragma unroll
        for (int i = 0; i < V; ++i) { const double d = xv.v[i]; s1[i] += d; s2[i] += d * d; }
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float xhat = (xv.v[i] - m[i]) * is[i];
          const float y = mask_y ? mv.v[i] : xhat * ga[i] + be[i];
          const float dz = (relu && !(y > 0.f)) ? 0.f : gv.v[i];
          s1[i] += dz; s2[i] += static_cast<double>(dz) * xhat;
        }
      }
    };
    long long r = r0 + rl;
    for (; r + static_cast<long long>(U - 1) * g.rpp < r1; r += static_cast<long long>(U) * g.rpp) {
      Vec<V> xv[U], gv[U], mv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        xv[u].load(x + (r + static_cast<long long>(u) * g.rpp) * g.c + col, 0, 0);
        if (BWD) gv[u].load(dy + (r + static_cast<long long>(u) * g.rpp) * g.c + col, 0, 0);
        if (BWD && mask_y) mv[u].load(mask_y + (r + static_cast<long long>(u) * g.rpp) * g.c + col, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accumulate(xv[u], gv[u], mv[u]);
    }
    for (; r < r1; r += g.rpp) {
      Vec<V> xv, gv, mv;
      xv.load(x + r * g.c + col, 0, 0);
      if (BWD) gv.load(dy + r * g.c + col, 0, 0);
      if (BWD && mask_y) mv.load(mask_y + r * g.c + col, 0, 0);
      accumulate(xv, gv, mv);
    }
  }
  if (rl < g.rpp) {
#pragma unroll
    for (int i = 0; i < V; ++i) { red[0][(rl * g.tcols + cq) * V + i] = s1[i]; red[1][(rl * g.tcols + cq) * V + i] = s2[i]; }
  }
  __syncthreads();
  const int cpad = g.tcols * V;
  for (int e = tid; e < cpad; e += 256) {
    double a = 0.0, b = 0.0;
    for (int q = 0; q < g.rpp; ++q) { a += red[0][q * cpad + e]; b += red[1][q * cpad + e]; }
    if (e < g.c) {
      stat_publish(partial + (static_cast<long long>(blockIdx.x) * 2 + 0) * g.c + e, a);   // (not a store: see bn_fold.hpp)
      stat_publish(partial + (static_cast<long long>(blockIdx.x) * 2 + 1) * g.c + e, b);
    }
  }
}

template <int V>
__global__ __launch_bounds__(256) void bn_one_fwd_k(const float* __restrict__ x, long long units, BnGeom g, double* __restrict__ partial,
                                                    unsigned* __restrict__ counters, BnFwdFin ff, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int relu, float* __restrict__ y, const float* __restrict__ residual) {
  __shared__ double red1[2][256 * V];
  __shared__ float sh_is[kBnMaxC], sh_beta[kBnMaxC], sh_mean[kBnMaxC], sh_gamma[kBnMaxC];
  one_reduce<V, false>(x, nullptr, g, nullptr, nullptr, nullptr, nullptr, 0, nullptr, partial, red1);
  grid_arrive_and_wait(counters, gridDim.x);
  double (*red)[256] = reinterpret_cast<double (*)[256]>(&red1[0][0]);
  const int tid = threadIdx.x, c = g.c;
  const int cfold = c < 256 ? c : 256;
  for (int e0 = 0; e0 < c; e0 += cfold) {
    const int e = e0 + tid;
    double a, b;
    fold_chunk<true>(g.nblk, c, partial, e0, cfold, red, &a, &b);
    if (tid < cfold && e < c) {
      float mu_f, is_f;
      bn_fwd_channel(a, b, g.n, ff, e, blockIdx.x == 0, &mu_f, &is_f);
      sh_mean[e] = mu_f; sh_is[e] = is_f;
      sh_gamma[e] = gamma ? gamma[e] : 1.f;
      sh_beta[e] = beta ? beta[e] : 0.f;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0 && ff.running_mean && ff.num_batches_tracked) *ff.num_batches_tracked += 1;
  const int cv = c / V;
  for (long long u = static_cast<long long>(blockIdx.x) * 256 + tid; u < units; u += static_cast<long long>(gridDim.x) * 256) {
    const int col = static_cast<int>(u % cv) * V;
    Vec<V> xv, rv, o;
    xv.load(x + u * V, 0, 0);
    if (residual) rv.load(residual + u * V, 0, 0);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float xhat = (xv.v[i] - sh_mean[col + i]) * sh_is[col + i];
      float t = xhat * sh_gamma[col + i] + sh_beta[col + i];
      if (residual) t = t + rv.v[i];
      o.v[i] = (relu && t <= 0.f) ? 0.f : t;
    }
    o.store(y + u * V);
  }
  grid_leave(counters, gridDim.x);
}

template <int V>
__global__ __launch_bounds__(256) void bn_one_bwd_k(const float* __restrict__ x, const float* __restrict__ dy, long long units, BnGeom g,
                                                    double* __restrict__ partial, unsigned* __restrict__ counters, const float* __restrict__ mean,
                                                    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    int relu, BnBwdFin bf, float* __restrict__ dx, const float* __restrict__ mask_y,
                                                    float* __restrict__ dz_out) {
  __shared__ double red1[2][256 * V];
  __shared__ float sh_mean[kBnMaxC], sh_is[kBnMaxC], sh_gamma[kBnMaxC], sh_beta[kBnMaxC], sh_c1[kBnMaxC], sh_c2[kBnMaxC];
  one_reduce<V, true>(x, dy, g, mean, invstd, gamma, beta, relu, mask_y, partial, red1);
  grid_arrive_and_wait(counters, gridDim.x);
  double (*red)[256] = reinterpret_cast<double (*)[256]>(&red1[0][0]);
  const int tid = threadIdx.x, c = g.c;
  const int cfold = c < 256 ? c : 256;
  for (int e0 = 0; e0 < c; e0 += cfold) {
    const int e = e0 + tid;
    double a, b;
    fold_chunk<true>(g.nblk, c, partial, e0, cfold, red, &a, &b);
    if (tid < cfold && e < c) {
      const double n = static_cast<double>(g.n);
      if (blockIdx.x == 0) { bf.dbeta[e] = static_cast<float>(a); bf.dgamma[e] = static_cast<float>(b); }
      sh_c1[e] = bf.batch_stats ? static_cast<float>(a / n) : 0.f;
      sh_c2[e] = bf.batch_stats ? static_cast<float>(b / n) : 0.f;
      sh_mean[e] = mean[e]; sh_is[e] = invstd[e];
      sh_gamma[e] = gamma ? gamma[e] : 1.f;
      sh_beta[e] = beta ? beta[e] : 0.f;
    }
  }
  __syncthreads();
  const int cv = c / V;
  for (long long u = static_cast<long long>(blockIdx.x) * 256 + tid; u < units; u += static_cast<long long>(gridDim.x) * 256) {
    const int col = static_cast<int>(u % cv) * V;
    Vec<V> xv, gv, mv, o, z;
    xv.load(x + u * V, 0, 0);
    gv.load(dy + u * V, 0, 0);
    if (mask_y) mv.load(mask_y + u * V, 0, 0);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float is = sh_is[col + i], ga = sh_gamma[col + i];
      const float xhat = (xv.v[i] - sh_mean[col + i]) * is;
      const float t = mask_y ? mv.v[i] : xhat * ga + sh_beta[col + i];
      const float dz = (relu && !(t > 0.f)) ? 0.f : gv.v[i];
      z.v[i] = dz;
      o.v[i] = ga * is * (dz - sh_c1[col + i] - xhat * sh_c2[col + i]);
    }
    o.store(dx + u * V);
    if (dz_out) z.store(dz_out + u * V);
  }
  grid_leave(counters, gridDim.x);
}

extern "C" size_t fv2p_batchnorm_one_ws_bytes(int c) {
  Sizer s;
  s.take<double>(static_cast<size_t>(kOneGrid) * 2 * (c > 0 ? c : 1));
  return s.bytes();
}

// fv2p_batchnorm_forward (+ optional residual) in ONE launch.  counters: two zeroed device words the caller keeps per stream (zero again
// when the launch ends).  Same statistics, bit for bit, as the two-launch form on the same number of partials.
extern "C" int fv2p_batchnorm_forward_one(const float* x, int64_t n, int c, float eps, float momentum, const float* gamma, const float* beta,
                                          int relu, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean,
                                          float* invstd, const float* residual, float* y, void* ws, size_t ws_bytes, unsigned* counters,
                                          fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 1 && c >= 1, FV2P_EINVAL, "batchnorm_forward_one: n=%lld c=%d", static_cast<long long>(n), c);
  FV2P_REQUIRE(x && y && mean && invstd && ws && counters, FV2P_EINVAL, "batchnorm_forward_one: null pointer");
  FV2P_REQUIRE((running_mean == nullptr) == (running_var == nullptr), FV2P_EINVAL, "batchnorm_forward_one: running_mean and running_var come together");
  FV2P_REQUIRE(ws_bytes >= fv2p_batchnorm_one_ws_bytes(c), FV2P_EWORKSPACE, "batchnorm_forward_one: workspace too small");
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(y) && (!residual || aligned16(residual));
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g, one_grid(c)) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  Carver cv(ws, ws_bytes);
  double* partial = cv.take<double>(static_cast<size_t>(kOneGrid) * 2 * c);
  BnFwdFin ff{mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), momentum, eps};
  const long long units = n * c / (vec ? 4 : 1);
  if (vec) hipLaunchKernelGGL((bn_one_fwd_k<4>), dim3(g.nblk), dim3(256), 0, stream, x, units, g, partial, counters, ff, gamma, beta, relu, y, residual);
  else hipLaunchKernelGGL((bn_one_fwd_k<1>), dim3(g.nblk), dim3(256), 0, stream, x, units, g, partial, counters, ff, gamma, beta, relu, y, residual);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// fv2p_batchnorm_backward / _backward_res in ONE launch: mask_y (NULL = recompute the ReLU mask from x) and dz_out (NULL = not wanted) as
// in fv2p_batchnorm_backward_res.
extern "C" int fv2p_batchnorm_backward_one(const float* x, const float* dy, int64_t n, int c, const float* mean, const float* invstd,
                                           const float* gamma, const float* beta, int relu, int batch_stats, const float* mask_y, float* dx,
                                           float* dz_out, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, unsigned* counters,
                                           fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 1 && c >= 1, FV2P_EINVAL, "batchnorm_backward_one: n=%lld c=%d", static_cast<long long>(n), c);
  FV2P_REQUIRE(x && dy && mean && invstd && dx && dgamma && dbeta && ws && counters, FV2P_EINVAL, "batchnorm_backward_one: null pointer");
  FV2P_REQUIRE(ws_bytes >= fv2p_batchnorm_one_ws_bytes(c), FV2P_EWORKSPACE, "batchnorm_backward_one: workspace too small");
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(dy) && aligned16(dx) && (!mask_y || aligned16(mask_y)) && (!dz_out || aligned16(dz_out));
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g, one_grid(c)) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  Carver cv(ws, ws_bytes);
  double* partial = cv.take<double>(static_cast<size_t>(kOneGrid) * 2 * c);
  BnBwdFin bf{dgamma, dbeta, nullptr, batch_stats};
  const long long units = n * c / (vec ? 4 : 1);
  if (vec) hipLaunchKernelGGL((bn_one_bwd_k<4>), dim3(g.nblk), dim3(256), 0, stream, x, dy, units, g, partial, counters, mean, invstd, gamma, beta, relu, bf, dx, mask_y, dz_out);
  else hipLaunchKernelGGL((bn_one_bwd_k<1>), dim3(g.nblk), dim3(256), 0, stream, x, dy, units, g, partial, counters, mean, invstd, gamma, beta, relu, bf, dx, mask_y, dz_out);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// ---- two launches, the first one wide: reduce on up to 512 workgroups with the sums finalised by its last workgroups, then apply -------
// The decoder's Linear -> BatchNorm1d -> ReLU rows are 49 152 x 64 / 128 (12 - 25 MB): too large for the one-launch form, and their
// reduce pass ran on <= 64 workgroups because EVERY apply workgroup folded all the partials (36.5 us backward, 18.5 us forward:
// ~1.4 TB/s).  With the finalisation of bn_fold.hpp the partials are folded once, by the reduce launch's own last workgroups, so the
// reduce can be as wide as the tensor wants and the apply pass reads finished numbers (mean / invstd, or c1 / c2).
constexpr int kWideGrid = 512;
template <int V, bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_fin_k(const float* __restrict__ x, const float* __restrict__ dy, BnGeom g, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int relu, const float* __restrict__ mask_y, double* __restrict__ rows, double* __restrict__ gslots,
                                                       unsigned* __restrict__ counter, int groups, FinOut fo) {
  __shared__ double red1[2][256 * V];
  __shared__ unsigned flag;
  one_reduce<V, BWD>(x, dy, g, mean, invstd, gamma, beta, relu, mask_y, rows, red1);   // publishes row blockIdx.x of rows [nblk][2][c]
  __syncthreads();
  for (int e0 = 0; e0 < g.c; e0 += 128) {   // the protocol handles <= 128 columns a pass; every pass sees all workgroups (counters reset in between)
    FinOut f = fo;
    const int cc = min(128, g.c - e0);
    f.ff.mean = fo.ff.mean ? fo.ff.mean + e0 : nullptr; f.ff.invstd = fo.ff.invstd ? fo.ff.invstd + e0 : nullptr;
    f.ff.running_mean = fo.ff.running_mean ? fo.ff.running_mean + e0 : nullptr; f.ff.running_var = fo.ff.running_var ? fo.ff.running_var + e0 : nullptr;
    f.bf.dgamma = fo.bf.dgamma ? fo.bf.dgamma + e0 : nullptr; f.bf.dbeta = fo.bf.dbeta ? fo.bf.dbeta + e0 : nullptr; f.bf.coef = fo.bf.coef ? fo.bf.coef + e0 : nullptr;
    f.bump = fo.bump && (e0 + 128 >= g.c);
    fin_rows_done(rows + e0, gslots + e0, counter + (e0 / 128) * ((1 + kFinSubs) * kFinStride), static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x), 1, groups,
                  cc, g.c, &flag, reinterpret_cast<double (*)[256]>(&red1[0][0]), f);
    __syncthreads();
  }
}

extern "C" size_t fv2p_batchnorm_wide_ws_bytes(int c) {
  Sizer s;
  s.take<double>(static_cast<size_t>(kWideGrid + kFinSubs) * 2 * (c > 0 ? c : 1));
  s.take<float>(2 * static_cast<size_t>(c > 0 ? c : 1));   // c1, c2 of the backward pass
  return s.bytes();
}
extern "C" int fv2p_batchnorm_wide_counter_words(int c) { return static_cast<int>(((c > 0 ? c : 1) + 127) / 128 * (1 + kFinSubs) * kFinStride); }

// fv2p_batchnorm_forward (+ optional residual) with the wide reduce: two launches, nothing folded in the apply pass.  counters:
// fv2p_batchnorm_wide_counter_words(c) zeroed device words the caller keeps per stream (zero again when the reduce launch ends).
extern "C" int fv2p_batchnorm_forward_wide(const float* x, int64_t n, int c, float eps, float momentum, const float* gamma, const float* beta,
                                           int relu, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean,
                                           float* invstd, const float* residual, float* y, void* ws, size_t ws_bytes, unsigned* counters,
                                           fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 1 && c >= 1, FV2P_EINVAL, "batchnorm_forward_wide: n=%lld c=%d", static_cast<long long>(n), c);
  FV2P_REQUIRE(x && y && mean && invstd && ws && counters, FV2P_EINVAL, "batchnorm_forward_wide: null pointer");
  FV2P_REQUIRE((running_mean == nullptr) == (running_var == nullptr), FV2P_EINVAL, "batchnorm_forward_wide: running_mean and running_var come together");
  FV2P_REQUIRE(ws_bytes >= fv2p_batchnorm_wide_ws_bytes(c), FV2P_EWORKSPACE, "batchnorm_forward_wide: workspace too small");
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(y) && (!residual || aligned16(residual));
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g, kWideGrid) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  Carver cv(ws, ws_bytes);
  double* rows = cv.take<double>(static_cast<size_t>(kWideGrid + kFinSubs) * 2 * c);
  double* gslots = rows + static_cast<size_t>(kWideGrid) * 2 * c;
  FinOut fo;
  fo.bwd = 0; fo.n = n; fo.bump = 1; fo.coef_ld = c;
  fo.ff = BnFwdFin{mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), momentum, eps};
  fo.bf = BnBwdFin{nullptr, nullptr, nullptr, 1};
  const int groups = g.nblk > 256 ? 32 : 16;
  const long long units = n * c / (vec ? 4 : 1);
  const unsigned blocks = apply_blocks(units);
  BnFwdFin ffa{mean, invstd, nullptr, nullptr, nullptr, 0.f, 0.f};
  if (vec) {
    hipLaunchKernelGGL((bn_reduce_fin_k<4, false>), dim3(g.nblk), dim3(256), 0, stream, x, nullptr, g, nullptr, nullptr, nullptr, nullptr, 0, nullptr, rows, gslots, counters, groups, fo);
    hipLaunchKernelGGL((bn_apply_fwd_k<4, false>), dim3(blocks), dim3(256), 0, stream, x, units, g, nullptr, ffa, gamma, beta, relu, y, nullptr, 0, residual);
  } else {
    hipLaunchKernelGGL((bn_reduce_fin_k<1, false>), dim3(g.nblk), dim3(256), 0, stream, x, nullptr, g, nullptr, nullptr, nullptr, nullptr, 0, nullptr, rows, gslots, counters, groups, fo);
    hipLaunchKernelGGL((bn_apply_fwd_k<1, false>), dim3(blocks), dim3(256), 0, stream, x, units, g, nullptr, ffa, gamma, beta, relu, y, nullptr, 0, residual);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_batchnorm_backward_wide(const float* x, const float* dy, int64_t n, int c, const float* mean, const float* invstd,
                                            const float* gamma, const float* beta, int relu, int batch_stats, const float* mask_y, float* dx,
                                            float* dz_out, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, unsigned* counters,
                                            fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 1 && c >= 1, FV2P_EINVAL, "batchnorm_backward_wide: n=%lld c=%d", static_cast<long long>(n), c);
  FV2P_REQUIRE(x && dy && mean && invstd && dx && dgamma && dbeta && ws && counters, FV2P_EINVAL, "batchnorm_backward_wide: null pointer");
  FV2P_REQUIRE(ws_bytes >= fv2p_batchnorm_wide_ws_bytes(c), FV2P_EWORKSPACE, "batchnorm_backward_wide: workspace too small");
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(dy) && aligned16(dx) && (!mask_y || aligned16(mask_y)) && (!dz_out || aligned16(dz_out));
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g, kWideGrid) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  Carver cv(ws, ws_bytes);
  double* rows = cv.take<double>(static_cast<size_t>(kWideGrid + kFinSubs) * 2 * c);
  double* gslots = rows + static_cast<size_t>(kWideGrid) * 2 * c;
  float* coef = cv.take<float>(2 * static_cast<size_t>(c));
  FinOut fo;
  fo.bwd = 1; fo.n = n; fo.bump = 0; fo.coef_ld = c;
  fo.ff = BnFwdFin{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};
  fo.bf = BnBwdFin{dgamma, dbeta, coef, batch_stats};
  const int groups = g.nblk > 256 ? 32 : 16;
  const long long units = n * c / (vec ? 4 : 1);
  const unsigned blocks = apply_blocks(units);
  BnBwdFin bfa{nullptr, nullptr, coef, batch_stats};
  if (vec) {
    hipLaunchKernelGGL((bn_reduce_fin_k<4, true>), dim3(g.nblk), dim3(256), 0, stream, x, dy, g, mean, invstd, gamma, beta, relu, mask_y, rows, gslots, counters, groups, fo);
    hipLaunchKernelGGL((bn_apply_bwd_k<4, false>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, nullptr, mean, invstd, gamma, beta, relu, bfa, dx, nullptr, 0, mask_y, dz_out);
  } else {
    hipLaunchKernelGGL((bn_reduce_fin_k<1, true>), dim3(g.nblk), dim3(256), 0, stream, x, dy, g, mean, invstd, gamma, beta, relu, mask_y, rows, gslots, counters, groups, fo);
    hipLaunchKernelGGL((bn_apply_bwd_k<1, false>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, nullptr, mean, invstd, gamma, beta, relu, bfa, dx, nullptr, 0, mask_y, dz_out);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}
