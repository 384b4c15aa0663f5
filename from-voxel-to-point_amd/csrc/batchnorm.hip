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

namespace fv2p {

struct BnGeom {
  long long n;
  int c, tcols, rpp, nblk;
  long long rows_per_block;
};

struct BnFwdFin {   // outputs of the forward finalisation
  float* mean; float* invstd;
  float* running_mean; float* running_var; long long* num_batches_tracked;
  float momentum;   // < 0: cumulative moving average (momentum=None)
  float eps;
};
struct BnBwdFin {
  float* dgamma; float* dbeta; float* coef;  // coef[0][c] = mean(dz), coef[1][c] = mean(dz * xhat) (0 when running stats were used)
  int batch_stats;
};

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
template <int V, bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_k(const float* __restrict__ x, const float* __restrict__ dy, BnGeom g,
                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                   double* __restrict__ partial) {
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
    auto accumulate = [&](const Vec<V>& xv, const Vec<V>& gv) {
      if (!BWD) {
#pragma unroll
        for (int i = 0; i < V; ++i) { const double d = xv.v[i]; s1[i] += d; s2[i] += d * d; }
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float xhat = (xv.v[i] - m[i]) * is[i];
          const float y = xhat * ga[i] + be[i];
          const float dz = (relu && !(y > 0.f)) ? 0.f : gv.v[i];
          s1[i] += dz; s2[i] += static_cast<double>(dz) * xhat;
        }
      }
    };
    long long r = r0 + rl;
    for (; r + static_cast<long long>(U - 1) * g.rpp < r1; r += static_cast<long long>(U) * g.rpp) {
      Vec<V> xv[U], gv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        xv[u].load(x + (r + static_cast<long long>(u) * g.rpp) * g.c + col, 0, 0);
        if (BWD) gv[u].load(dy + (r + static_cast<long long>(u) * g.rpp) * g.c + col, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accumulate(xv[u], gv[u]);
    }
    for (; r < r1; r += g.rpp) {
      Vec<V> xv, gv;
      xv.load(x + r * g.c + col, 0, 0);
      if (BWD) gv.load(dy + r * g.c + col, 0, 0);
      accumulate(xv, gv);
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

// Folds the [nblk][2][c] partials of channels [e0, e0 + cfold) in a fixed order: L = 256 / cfold lanes per channel take
// interleaved slices, then an ordered LDS fold.  Returns the two sums of channel e0 + tid (valid for tid < cfold).
__device__ __forceinline__ void fold_chunk(const BnGeom& g, const double* __restrict__ partial, int e0, int cfold, double (*red)[256],
                                           double* a_out, double* b_out) {
  const int tid = threadIdx.x, L = 256 / cfold;
  const int e = e0 + tid % cfold, lane_q = tid / cfold;
  double a = 0.0, b = 0.0;
  if (lane_q < L && e < g.c) {
#pragma unroll 8
    for (int q = lane_q; q < g.nblk; q += L) {
      a += partial[(static_cast<long long>(q) * 2 + 0) * g.c + e];
      b += partial[(static_cast<long long>(q) * 2 + 1) * g.c + e];
    }
  }
  __syncthreads();
  red[0][tid] = a; red[1][tid] = b;
  __syncthreads();
  a = 0.0; b = 0.0;
  if (tid < cfold)
    for (int q = 0; q < L; ++q) { a += red[0][q * cfold + tid]; b += red[1][q * cfold + tid]; }
  *a_out = a; *b_out = b;
}

constexpr int kBnMaxC = 1024;

// FOLD: batch statistics come from the partials (training); otherwise mean / invstd are read from memory (eval).
template <int V, bool FOLD>
__global__ __launch_bounds__(256) void bn_apply_fwd_k(const float* __restrict__ x, long long units, BnGeom g, const double* __restrict__ partial,
                                                      BnFwdFin ff, const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                      float* __restrict__ y, double* __restrict__ zero_next, long long zero_count) {
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
    float mu_f, is_f;
    if (FOLD) {
      double a, b;
      fold_chunk(g, partial, e0, cfold, red, &a, &b);
      const double n = static_cast<double>(g.n);
      const double mu = a / n;
      double var = b / n - mu * mu;
      if (var < 0.0) var = 0.0;
      mu_f = static_cast<float>(mu);
      is_f = static_cast<float>(1.0 / sqrt(var + static_cast<double>(ff.eps)));
      if (blockIdx.x == 0 && tid < cfold && e < c) {
        ff.mean[e] = mu_f;
        ff.invstd[e] = is_f;
        if (ff.running_mean) {
          double f = ff.momentum;
          if (ff.momentum < 0.f) f = 1.0 / static_cast<double>(ff.num_batches_tracked ? (*ff.num_batches_tracked + 1) : 1);
          const double unbiased = g.n > 1 ? var * n / (n - 1.0) : var;
          ff.running_mean[e] = static_cast<float>((1.0 - f) * ff.running_mean[e] + f * mu);
          ff.running_var[e] = static_cast<float>((1.0 - f) * ff.running_var[e] + f * unbiased);
        }
      }
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
    Vec<V> xv, o;
    xv.load(x + u * V, 0, 0);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float xhat = (xv.v[i] - sh_mean[col + i]) * sh_scale[col + i];
      const float t = xhat * sh_gamma[col + i] + sh_shift[col + i];
      o.v[i] = (relu && t <= 0.f) ? 0.f : t;  // NaN passes through, like torch.relu
    }
    o.store(y + u * V);
  }
}

template <int V>
__global__ __launch_bounds__(256) void bn_apply_bwd_k(const float* __restrict__ x, const float* __restrict__ dy, long long units, BnGeom g,
                                                      const double* __restrict__ partial, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int relu, BnBwdFin bf, float* __restrict__ dx,
                                                      double* __restrict__ zero_next, long long zero_count) {
  __shared__ double red[2][256];
  if (zero_next && blockIdx.x == 0)   // see bn_apply_fwd_k
    for (long long e = threadIdx.x; e < zero_count; e += 256) zero_next[e] = 0.0;
  __shared__ float sh_mean[kBnMaxC], sh_is[kBnMaxC], sh_gamma[kBnMaxC], sh_beta[kBnMaxC], sh_c1[kBnMaxC], sh_c2[kBnMaxC];
  const int tid = threadIdx.x, c = g.c;
  const int cfold = c < 256 ? c : 256;
  for (int e0 = 0; e0 < c; e0 += cfold) {
    const int e = e0 + tid;
    double a, b;
    fold_chunk(g, partial, e0, cfold, red, &a, &b);
    if (tid < cfold && e < c) {
      const double n = static_cast<double>(g.n);
      if (blockIdx.x == 0) {
        bf.dbeta[e] = static_cast<float>(a);
        bf.dgamma[e] = static_cast<float>(b);
      }
      sh_c1[e] = bf.batch_stats ? static_cast<float>(a / n) : 0.f;
      sh_c2[e] = bf.batch_stats ? static_cast<float>(b / n) : 0.f;
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
    Vec<V> xv, gv, o;
    xv.load(x + u * V, 0, 0);
    gv.load(dy + u * V, 0, 0);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float is = sh_is[col + i], ga = sh_gamma[col + i];
      const float xhat = (xv.v[i] - sh_mean[col + i]) * is;
      const float t = xhat * ga + sh_beta[col + i];
      const float dz = (relu && !(t > 0.f)) ? 0.f : gv.v[i];
      o.v[i] = ga * is * (dz - sh_c1[col + i] - xhat * sh_c2[col + i]);
    }
    o.store(dx + u * V);
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
    hipLaunchKernelGGL((bn_reduce_k<4, false>), dim3(g.nblk), dim3(256), 0, stream, x, nullptr, g, nullptr, nullptr, nullptr, nullptr, 0, partial);
    hipLaunchKernelGGL((bn_apply_fwd_k<4, true>), dim3(blocks), dim3(256), 0, stream, x, units, g, partial, ff, gamma, beta, relu, y, nullptr, 0);
  } else {
    hipLaunchKernelGGL((bn_reduce_k<1, false>), dim3(g.nblk), dim3(256), 0, stream, x, nullptr, g, nullptr, nullptr, nullptr, nullptr, 0, partial);
    hipLaunchKernelGGL((bn_apply_fwd_k<1, true>), dim3(blocks), dim3(256), 0, stream, x, units, g, partial, ff, gamma, beta, relu, y, nullptr, 0);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}

// column sums of x into the slot layout of the conv-epilogue statistics (slots beyond the reduce grid stay as they are: zero)
int fv2p::bn_column_sums(const float* x, int64_t n, int c, double* stats, hipStream_t stream) {
  const bool vec = (c % 4 == 0) && aligned16(x);
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g, kStatSlots) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  if (vec) hipLaunchKernelGGL((bn_reduce_k<4, false>), dim3(g.nblk), dim3(256), 0, stream, x, nullptr, g, nullptr, nullptr, nullptr, nullptr, 0, stats);
  else hipLaunchKernelGGL((bn_reduce_k<1, false>), dim3(g.nblk), dim3(256), 0, stream, x, nullptr, g, nullptr, nullptr, nullptr, nullptr, 0, stats);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// (sum dz, sum dz * xhat) into the same slot layout, for a backward-data conv that could not take them in its epilogue
int fv2p::bn_backward_sums(const float* x, const float* dy, int64_t n, int c, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, int relu, double* stats, hipStream_t stream) {
  const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(dy);
  BnGeom g;
  FV2P_REQUIRE(bn_geom(n, c, vec, &g, kStatSlots) == 0, FV2P_ELIMIT, "batchnorm: c=%d exceeds %d", c, vec ? kBnMaxC : 256);
  if (vec) hipLaunchKernelGGL((bn_reduce_k<4, true>), dim3(g.nblk), dim3(256), 0, stream, x, dy, g, mean, invstd, gamma, beta, relu, stats);
  else hipLaunchKernelGGL((bn_reduce_k<1, true>), dim3(g.nblk), dim3(256), 0, stream, x, dy, g, mean, invstd, gamma, beta, relu, stats);
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
  if (vec) hipLaunchKernelGGL((bn_apply_fwd_k<4, true>), dim3(blocks), dim3(256), 0, stream, x, units, g, stats, ff, gamma, beta, relu, y, zero_next, static_cast<long long>(zero_count));
  else hipLaunchKernelGGL((bn_apply_fwd_k<1, true>), dim3(blocks), dim3(256), 0, stream, x, units, g, stats, ff, gamma, beta, relu, y, zero_next, static_cast<long long>(zero_count));
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
  if (vec) hipLaunchKernelGGL((bn_apply_bwd_k<4>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, stats, mean, invstd, gamma, beta, relu, bf, dx,
                              zero_next, static_cast<long long>(zero_count));
  else hipLaunchKernelGGL((bn_apply_bwd_k<1>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, stats, mean, invstd, gamma, beta, relu, bf, dx,
                          zero_next, static_cast<long long>(zero_count));
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
  if (vec) hipLaunchKernelGGL((bn_apply_fwd_k<4, false>), dim3(blocks), dim3(256), 0, stream, x, units, g, nullptr, ff, gamma, beta, relu, y, nullptr, 0);
  else hipLaunchKernelGGL((bn_apply_fwd_k<1, false>), dim3(blocks), dim3(256), 0, stream, x, units, g, nullptr, ff, gamma, beta, relu, y, nullptr, 0);
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
    hipLaunchKernelGGL((bn_reduce_k<4, true>), dim3(g.nblk), dim3(256), 0, stream, x, dy, g, mean, invstd, gamma, beta, relu, partial);
    hipLaunchKernelGGL((bn_apply_bwd_k<4>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, partial, mean, invstd, gamma, beta, relu, bf, dx, nullptr, 0);
  } else {
    hipLaunchKernelGGL((bn_reduce_k<1, true>), dim3(g.nblk), dim3(256), 0, stream, x, dy, g, mean, invstd, gamma, beta, relu, partial);
    hipLaunchKernelGGL((bn_apply_bwd_k<1>), dim3(blocks), dim3(256), 0, stream, x, dy, units, g, partial, mean, invstd, gamma, beta, relu, bf, dx, nullptr, 0);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}
