// Bilinear gather of BEV features at key points — SURVEY §8(f).2.
//
// Replaces bilinear_interpolate_torch / BEVGridPooling.interpolate_from_bev_features
// (pcdet/models/backbones_3d/pfe/bev_grid_pooling.py:11-45, 68-83): per sample the reference permutes the [C, H, W] map
// to [H, W, C] (a copy of the whole map), indexes four corner rows per point (four [N, C] temporaries), forms four
// weight vectors and sums four transposed products.  Here: one tiled transpose of the batch (only when the map comes
// channel-first) and one gather launch — a wave per point, float4 per lane along C, the four corner rows read once,
// nothing materialised.  Arithmetic and its quirks are the reference's: corners are floor / floor + 1 clamped to the
// map, the weights use the CLAMPED corners (so a point outside the map gets weights that do not sum to 1, exactly as
// there), products summed in the order a, b, c, d without contraction.
// Backward (gradient of the map): the transposed scatter with float atomics (torch's index backward is an unordered
// accumulation as well), then the transpose back.
#include "common.hpp"

namespace fv2p {

struct Corners {
  long long a, b, c, d;   // row offsets (in rows of C floats) of (y0,x0) (y1,x0) (y0,x1) (y1,x1)
  float wa, wb, wc, wd;
};
__device__ __forceinline__ Corners bev_corners(float x, float y, int h, int w) {
  // floor in fp32, then the integer arithmetic of the reference (torch.floor(x).long(), +1, clamp)
  long long x0 = static_cast<long long>(floorf(x)), y0 = static_cast<long long>(floorf(y));
  long long x1 = x0 + 1, y1 = y0 + 1;
  x0 = x0 < 0 ? 0 : (x0 > w - 1 ? w - 1 : x0);
  x1 = x1 < 0 ? 0 : (x1 > w - 1 ? w - 1 : x1);
  y0 = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0);
  y1 = y1 < 0 ? 0 : (y1 > h - 1 ? h - 1 : y1);
  const float fx0 = static_cast<float>(x0), fx1 = static_cast<float>(x1), fy0 = static_cast<float>(y0), fy1 = static_cast<float>(y1);
  Corners k;
  k.a = y0 * w + x0; k.b = y1 * w + x0; k.c = y0 * w + x1; k.d = y1 * w + x1;
  k.wa = (fx1 - x) * (fy1 - y);
  k.wb = (fx1 - x) * (y - fy0);
  k.wc = (x - fx0) * (fy1 - y);
  k.wd = (x - fx0) * (y - fy0);
  return k;
}

// im [B][H*W][C], x / y [B][N] -> out [B][N][C]; one wave per point, lanes stride the channels
template <int V>
__global__ __launch_bounds__(256) void bev_gather_k(const float* __restrict__ im, const float* __restrict__ xs, const float* __restrict__ ys,
                                                    int batch, long long n, int h, int w, int c, float* __restrict__ out) {
  const long long p = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (p >= static_cast<long long>(batch) * n) return;
  const int lane = threadIdx.x & 63;
  const long long b = p / n;
  const Corners k = bev_corners(xs[p], ys[p], h, w);
  const float* base = im + b * static_cast<long long>(h) * w * c;
  float* o = out + p * c;
  for (int ch = lane * V; ch < c; ch += 64 * V) {
    float va[V], vb[V], vc[V], vd[V], r[V];
    if constexpr (V == 4) {
      const float4 qa = *reinterpret_cast<const float4*>(base + k.a * c + ch), qb = *reinterpret_cast<const float4*>(base + k.b * c + ch);
      const float4 qc = *reinterpret_cast<const float4*>(base + k.c * c + ch), qd = *reinterpret_cast<const float4*>(base + k.d * c + ch);
      va[0] = qa.x; va[1] = qa.y; va[2] = qa.z; va[3] = qa.w; vb[0] = qb.x; vb[1] = qb.y; vb[2] = qb.z; vb[3] = qb.w;
      vc[0] = qc.x; vc[1] = qc.y; vc[2] = qc.z; vc[3] = qc.w; vd[0] = qd.x; vd[1] = qd.y; vd[2] = qd.z; vd[3] = qd.w;
    } else {
      va[0] = base[k.a * c + ch]; vb[0] = base[k.b * c + ch]; vc[0] = base[k.c * c + ch]; vd[0] = base[k.d * c + ch];
    }
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = ((va[i] * k.wa + vb[i] * k.wb) + vc[i] * k.wc) + vd[i] * k.wd;
    if constexpr (V == 4) *reinterpret_cast<float4*>(o + ch) = make_float4(r[0], r[1], r[2], r[3]);
    else o[ch] = r[0];
  }
}

// gim [B][H*W][C] += weights * gout [B][N][C].  A wave takes kBevRun consecutive points; neighbours with the SAME position (the RoI head's
// grid points come z-fastest: the six points of a column share x and y, iouguided_roi_head.py:243-255) are summed first, in point order,
// and reach the map as ONE set of atomics - six times fewer of them on that input, nothing lost on any other.
constexpr int kBevRun = 8;
__global__ __launch_bounds__(256) void bev_scatter_k(const float* __restrict__ gout, const float* __restrict__ xs, const float* __restrict__ ys,
                                                     int batch, long long n, int h, int w, int c, float* __restrict__ gim) {
  const long long total = static_cast<long long>(batch) * n;
  const long long p0 = (static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * kBevRun;
  if (p0 >= total) return;
  const int lane = threadIdx.x & 63;
  float x[kBevRun], y[kBevRun];
  bool last[kBevRun];   // point u closes a group: the next point is elsewhere (or in another sample, or past the end)
#pragma unroll
  for (int u = 0; u < kBevRun; ++u) {
    const long long q = p0 + u < total ? p0 + u : total - 1;
    x[u] = xs[q]; y[u] = ys[q];
  }
#pragma unroll
  for (int u = 0; u < kBevRun; ++u)
    last[u] = u == kBevRun - 1 || p0 + u + 1 >= total || x[u + 1 < kBevRun ? u + 1 : u] != x[u] || y[u + 1 < kBevRun ? u + 1 : u] != y[u] ||
              (p0 + u + 1) / n != (p0 + u) / n;
  for (int ch = lane; ch < c; ch += 64) {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < kBevRun; ++u) {
      if (p0 + u < total) {   // uniform
        acc += gout[(p0 + u) * c + ch];
        if (last[u]) {        // uniform
          const Corners k = bev_corners(x[u], y[u], h, w);
          float* base = gim + ((p0 + u) / n) * static_cast<long long>(h) * w * c;
          atomicAdd(base + k.a * c + ch, acc * k.wa);
          atomicAdd(base + k.b * c + ch, acc * k.wb);
          atomicAdd(base + k.c * c + ch, acc * k.wc);
          atomicAdd(base + k.d * c + ch, acc * k.wd);
          acc = 0.f;
        }
      }
    }
  }
}

// [B][R][S] -> [B][S][R] (R = C, S = H*W: channel-first -> channel-last, and back with the roles swapped) through a 64 x 65 LDS tile:
// 16-byte global accesses on both sides (a row of the tile is 256 contiguous bytes in the source, a column 256 contiguous bytes in the
// destination), scalar LDS accesses with the odd pitch (two lanes per bank at most).  torch's .permute().contiguous() moves the same
// bytes with a generic strided-copy kernel at a fraction of the bandwidth (the DCN layers' layout copies).
__global__ __launch_bounds__(256) void transpose_k(const float* __restrict__ in, int rows, long long cols, float* __restrict__ out) {
  __shared__ float tile[64][65];
  const long long b = blockIdx.z;
  const float* src = in + b * rows * cols;
  float* dst = out + b * rows * cols;
  const long long c0 = static_cast<long long>(blockIdx.x) * 64;
  const int r0 = blockIdx.y * 64;
  const int q = threadIdx.x & 15, t = threadIdx.x >> 4;   // 16 quads x 16
  const bool vec_in = (cols & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
  const bool vec_out = (rows & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
#pragma unroll
  for (int j = 0; j < 64; j += 16) {
    const int r = r0 + t + j;
    const long long cc = c0 + 4 * q;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      const float* p = src + static_cast<long long>(r) * cols + cc;
      if (vec_in && cc + 3 < cols) { const float4 f = *reinterpret_cast<const float4*>(p); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (cc + e < cols) v[e] = p[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[t + j][4 * q + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 64; j += 16) {
    const long long cc = c0 + t + j;
    const int r = r0 + 4 * q;
    if (cc >= cols) continue;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = tile[4 * q + e][t + j];
    float* p = dst + cc * rows + r;
    if (vec_out && r + 3 < rows) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (r + e < rows) p[e] = v[e];
    }
  }
}

static int launch_transpose(const float* in, int batch, int rows, long long cols, float* out, hipStream_t stream) {
  FV2P_REQUIRE(batch <= 65535 && ceil_div(rows, 64) <= 65535, FV2P_ELIMIT, "transpose: batch or row count too large");
  const dim3 grid(static_cast<unsigned>(ceil_div(cols, 64)), static_cast<unsigned>(ceil_div(rows, 64)), static_cast<unsigned>(batch));
  hipLaunchKernelGGL(transpose_k, grid, dim3(256), 0, stream, in, rows, cols, out);
  return 0;
}

}  // namespace fv2p

using namespace fv2p;

extern "C" size_t fv2p_bev_interp_ws_bytes(int batch, int c, int h, int w, int channels_first) {
  return channels_first ? static_cast<size_t>(batch) * c * h * w * sizeof(float) : 0;
}

extern "C" int fv2p_bev_interp_fwd(const float* bev, int batch, int c, int h, int w, int channels_first, const float* x, const float* y,
                                   int64_t n, float* out, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(batch >= 1 && c >= 1 && h >= 1 && w >= 1 && n >= 0, FV2P_EINVAL, "bev_interp_fwd: bad sizes");
  if (n == 0) return 0;
  FV2P_REQUIRE(bev && x && y && out, FV2P_EINVAL, "bev_interp_fwd: null pointer");
  const float* im = bev;
  if (channels_first) {
    FV2P_REQUIRE(ws && ws_bytes >= fv2p_bev_interp_ws_bytes(batch, c, h, w, 1), FV2P_EWORKSPACE, "bev_interp_fwd: workspace too small");
    if (int rc = launch_transpose(bev, batch, c, static_cast<long long>(h) * w, static_cast<float*>(ws), stream)) return rc;
    im = static_cast<const float*>(ws);
  }
  const long long pts = static_cast<long long>(batch) * n;
  FV2P_REQUIRE(ceil_div(pts, 4) < (1ll << 31), FV2P_ELIMIT, "bev_interp_fwd: too many points");
  const bool vec = (c % 4 == 0) && (reinterpret_cast<uintptr_t>(im) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (vec) hipLaunchKernelGGL((bev_gather_k<4>), dim3(static_cast<unsigned>(ceil_div(pts, 4))), dim3(256), 0, stream, im, x, y, batch,
                              static_cast<long long>(n), h, w, c, out);
  else hipLaunchKernelGGL((bev_gather_k<1>), dim3(static_cast<unsigned>(ceil_div(pts, 4))), dim3(256), 0, stream, im, x, y, batch,
                          static_cast<long long>(n), h, w, c, out);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_bev_interp_bwd(const float* grad_out, int batch, int c, int h, int w, int channels_first, const float* x, const float* y,
                                   int64_t n, float* grad_bev, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(batch >= 1 && c >= 1 && h >= 1 && w >= 1 && n >= 0 && grad_bev, FV2P_EINVAL, "bev_interp_bwd: bad arguments");
  const size_t bytes = static_cast<size_t>(batch) * c * h * w * sizeof(float);
  float* gim = grad_bev;
  if (channels_first) {
    FV2P_REQUIRE(ws && ws_bytes >= bytes, FV2P_EWORKSPACE, "bev_interp_bwd: workspace too small");
    gim = static_cast<float*>(ws);
  }
  FV2P_HIP(hipMemsetAsync(gim, 0, bytes, stream));
  if (n > 0) {
    FV2P_REQUIRE(grad_out && x && y, FV2P_EINVAL, "bev_interp_bwd: null pointer");
    const long long pts = static_cast<long long>(batch) * n;
    hipLaunchKernelGGL(bev_scatter_k, dim3(static_cast<unsigned>(ceil_div(pts, 4 * kBevRun))), dim3(256), 0, stream, grad_out, x, y, batch,
                       static_cast<long long>(n), h, w, c, gim);
  }
  if (channels_first)
    if (int rc = launch_transpose(gim, batch, h * w, c, grad_bev, stream)) return rc;   // [B][HW][C] -> [B][C][HW]
  FV2P_LAUNCH_CHECK();
  return 0;
}

// in [batch][rows][cols] -> out [batch][cols][rows]: the NCHW <-> NHWC copies around the NHWC entry points (rows = C, cols = H*W one way,
// rows = H*W, cols = C the other)
extern "C" int fv2p_transpose_batched(const float* in, int batch, int64_t rows, int64_t cols, float* out, fv2p_stream_t stream_) {
  FV2P_REQUIRE(batch >= 0 && rows >= 0 && cols >= 0, FV2P_EINVAL, "transpose_batched: bad sizes");
  if (batch == 0 || rows == 0 || cols == 0) return 0;
  FV2P_REQUIRE(in && out && in != out, FV2P_EINVAL, "transpose_batched: null or aliased pointers");
  FV2P_REQUIRE(rows < (1ll << 31) && ceil_div(cols, 64) < (1ll << 31), FV2P_ELIMIT, "transpose_batched: too large");
  if (int rc = launch_transpose(in, batch, static_cast<int>(rows), cols, out, static_cast<hipStream_t>(stream_))) return rc;
  FV2P_LAUNCH_CHECK();
  return 0;
}
