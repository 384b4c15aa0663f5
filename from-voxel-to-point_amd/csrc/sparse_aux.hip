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

// ---- SparseConvTensor.dense() --------------------------------------------------------------------
// The reference builds the dense tensor as zeros [B, *spatial, C] -> index scatter -> permute(0, C, ...).contiguous()
// (structure.py:57-66, used by HeightCompression, height_compression.py:10-26): three passes over the dense volume.
// Here: one fill + one scatter straight into the requested layout.  vol = D*H*W cells per sample.
//   channels_first: out[b][ch][cell]   (thread = (ch, row): rows sorted by cell -> neighbouring threads write neighbouring x)
//   channels_last : out[b][cell][ch]   (thread = (row, ch): whole rows contiguous)
__global__ __launch_bounds__(256) void dense_scatter(const float* __restrict__ feat, const int* __restrict__ ind, int n, int c, int ndim,
                                                     int d1, int d2, int64_t vol, int channels_first, float* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= static_cast<int64_t>(n) * c) return;
  const int row = channels_first ? static_cast<int>(t % n) : static_cast<int>(t / c);
  const int ch = channels_first ? static_cast<int>(t / n) : static_cast<int>(t % c);
  const int* p = ind + static_cast<int64_t>(row) * (ndim + 1);
  const int64_t cell = ndim == 3 ? (static_cast<int64_t>(p[1]) * d1 + p[2]) * d2 + p[3] : static_cast<int64_t>(p[1]) * d2 + p[2];
  const int64_t b = p[0];
  const float v = feat[static_cast<int64_t>(row) * c + ch];
  if (channels_first) out[(b * c + ch) * vol + cell] = v;
  else out[(b * vol + cell) * c + ch] = v;
}
// gradient of dense(): drows[row][ch] = ddense at the row's cell
__global__ __launch_bounds__(256) void dense_gather(const float* __restrict__ dense, const int* __restrict__ ind, int n, int c, int ndim,
                                                    int d1, int d2, int64_t vol, int channels_first, float* __restrict__ rows) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= static_cast<int64_t>(n) * c) return;
  const int row = channels_first ? static_cast<int>(t % n) : static_cast<int>(t / c);
  const int ch = channels_first ? static_cast<int>(t / n) : static_cast<int>(t % c);
  const int* p = ind + static_cast<int64_t>(row) * (ndim + 1);
  const int64_t cell = ndim == 3 ? (static_cast<int64_t>(p[1]) * d1 + p[2]) * d2 + p[3] : static_cast<int64_t>(p[1]) * d2 + p[2];
  const int64_t b = p[0];
  rows[static_cast<int64_t>(row) * c + ch] = channels_first ? dense[(b * c + ch) * vol + cell] : dense[(b * vol + cell) * c + ch];
}

// ---- MeanVFE + collate of one voxelised cloud -------------------------------------------------------
// mean over the first num_points[v] points of voxel v (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31: sum over the
// zero-padded slots / clamp_min(num, 1)) and (batch_idx, z, y, x) coordinates (dataset.collate_batch,
// pcdet/datasets/dataset.py:165-171), written at this cloud's offset of the batch tensors; the voxel count is read on
// the device, so the launch needs no host round trip of its own.
__global__ __launch_bounds__(256) void voxel_mean_collate(const float* __restrict__ voxels, const int* __restrict__ coors,
                                                          const int* __restrict__ num, const int* __restrict__ count, int max_voxels,
                                                          int max_pts, int ndim, int batch_idx, float* __restrict__ feats,
                                                          int* __restrict__ coords) {
  const int m = min(*count, max_voxels);
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= static_cast<int64_t>(m) * ndim) return;
  const int v = static_cast<int>(t / ndim), d = static_cast<int>(t % ndim);
  float s = 0.f;
  for (int p = 0; p < max_pts; ++p) s += voxels[(static_cast<int64_t>(v) * max_pts + p) * ndim + d];
  const int k = num[v];
  feats[static_cast<int64_t>(v) * ndim + d] = s / static_cast<float>(k < 1 ? 1 : k);
  if (d < 4) coords[static_cast<int64_t>(v) * 4 + d] = d == 0 ? batch_idx : coors[static_cast<int64_t>(v) * 3 + d - 1];
}

}  // namespace fv2p
using namespace fv2p;

#define FV2P_GRID1D(total) dim3(static_cast<unsigned>(ceil_div((total), 256))), dim3(256)

extern "C" int fv2p_sparse_maxpool_fwd(const float* in, int64_t n_in, int c, const int* tab, int kvol, int64_t n_out, int flip_k,
                                       float* out, fv2p_stream_t s) {
  FV2P_REQUIRE(c >= 1 && kvol >= 1 && n_out >= 0, FV2P_EINVAL, "maxpool_fwd: bad sizes");
  if (n_out == 0) return 0;
  FV2P_REQUIRE(in && tab && out, FV2P_EINVAL, "maxpool_fwd: null pointer");
  hipLaunchKernelGGL(maxpool_fwd, FV2P_GRID1D(n_out * c), 0, static_cast<hipStream_t>(s), in, c, tab, kvol, (int)n_out, flip_k & 1, out);
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
  hipLaunchKernelGGL(group_fwd, FV2P_GRID1D(n_out * c * kvol), 0, static_cast<hipStream_t>(s), in, c, tab, kvol, (int)n_out, flip_k & 1, out);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_sparse_group_bwd(const float* grad, int64_t n_out, int c, const int* tab, int kvol, int64_t n_in, int flip_k,
                                     float* din, fv2p_stream_t s) {
  FV2P_REQUIRE(c >= 1 && kvol >= 1 && n_in >= 0, FV2P_EINVAL, "group_bwd: bad sizes");
  if (n_in == 0) return 0;
  FV2P_REQUIRE(grad && tab && din, FV2P_EINVAL, "group_bwd: null pointer");
  hipLaunchKernelGGL(group_bwd, FV2P_GRID1D(n_in * c), 0, static_cast<hipStream_t>(s), grad, (int)n_out, c, tab, kvol, (int)n_in, flip_k & 1, din);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_sparse_to_dense(const float* features, const int* indices, int64_t n, int c, int ndim, int batch,
                                    const int spatial[3], int channels_first, float* dense, fv2p_stream_t s_) {
  hipStream_t s = static_cast<hipStream_t>(s_);
  FV2P_REQUIRE(c >= 1 && n >= 0 && batch >= 1 && (ndim == 2 || ndim == 3), FV2P_EINVAL, "sparse_to_dense: bad sizes");
  FV2P_REQUIRE(dense && spatial, FV2P_EINVAL, "sparse_to_dense: null pointer");
  const int64_t vol = ndim == 3 ? static_cast<int64_t>(spatial[0]) * spatial[1] * spatial[2] : static_cast<int64_t>(spatial[0]) * spatial[1];
  FillJobs fill;
  fill.add(dense, sizeof(float) * static_cast<size_t>(batch) * c * vol, 0u);
  if (int rc = multi_fill(fill, s)) return rc;
  if (n == 0) return 0;
  FV2P_REQUIRE(features && indices, FV2P_EINVAL, "sparse_to_dense: null pointer");
  hipLaunchKernelGGL(dense_scatter, FV2P_GRID1D(n * c), 0, s, features, indices, (int)n, c, ndim, ndim == 3 ? spatial[1] : 0,
                     ndim == 3 ? spatial[2] : spatial[1], vol, channels_first, dense);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_dense_to_sparse(const float* dense, const int* indices, int64_t n, int c, int ndim, int batch, const int spatial[3],
                                    int channels_first, float* rows, fv2p_stream_t s_) {
  hipStream_t s = static_cast<hipStream_t>(s_);
  FV2P_REQUIRE(c >= 1 && n >= 0 && batch >= 1 && (ndim == 2 || ndim == 3), FV2P_EINVAL, "dense_to_sparse: bad sizes");
  if (n == 0) return 0;
  FV2P_REQUIRE(dense && indices && rows && spatial, FV2P_EINVAL, "dense_to_sparse: null pointer");
  const int64_t vol = ndim == 3 ? static_cast<int64_t>(spatial[0]) * spatial[1] * spatial[2] : static_cast<int64_t>(spatial[0]) * spatial[1];
  hipLaunchKernelGGL(dense_gather, FV2P_GRID1D(n * c), 0, s, dense, indices, (int)n, c, ndim, ndim == 3 ? spatial[1] : 0,
                     ndim == 3 ? spatial[2] : spatial[1], vol, channels_first, rows);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_voxel_mean_collate(const float* voxels, const int* coors, const int* num_points, const int* num_voxels,
                                       int max_voxels, int max_points, int ndim, int batch_idx, float* feats, int* coords,
                                       fv2p_stream_t s_) {
  hipStream_t s = static_cast<hipStream_t>(s_);
  FV2P_REQUIRE(max_voxels >= 0 && max_points >= 1 && ndim >= 3, FV2P_EINVAL, "voxel_mean_collate: bad sizes");
  if (max_voxels == 0) return 0;
  FV2P_REQUIRE(voxels && coors && num_points && num_voxels && feats && coords, FV2P_EINVAL, "voxel_mean_collate: null pointer");
  hipLaunchKernelGGL(voxel_mean_collate, FV2P_GRID1D(static_cast<int64_t>(max_voxels) * ndim), 0, s, voxels, coors, num_points, num_voxels,
                     max_voxels, max_points, ndim, batch_idx, feats, coords);
  FV2P_LAUNCH_CHECK();
  return 0;
}
