// Deformable position-sensitive RoI pooling — SURVEY §8 A14 (second half).
//
// Replaces DCN.deform_psroi_pooling_forward / _backward
// (pcdet/ops/DeformableConvolutionV2PyTorch/src/vision.cpp:11-12 -> src/deform_psroi_pooling.h ->
// src/cuda/deform_psroi_pooling_cuda.cu:264-418, kernels :59-147 forward, :149-262 backward).
// Arithmetic follows the reference's expression by expression: RoI corners rounded half away from zero, scaled, moved by -0.5;
// a bin is `sample_per_part`^2 bilinear samples of ONE channel plane (floor / ceil corners of the clamped position), samples
// outside [-0.5, size - 0.5] are skipped and the bin is the mean of the rest; the learned shift of a bin is
// trans[n, 2*class + {0,1}, part_h, part_w] * trans_std * RoI size.
// The reference writes several of these with double literals in float code (`round(x) * scale - 0.5`, `max(d, 0.1)`, `w < -0.5`,
// `min(max(w, 0.), width - 1.)`, :88-96, :131-136), which C++ evaluates in double and rounds once on assignment.  Each of them is a
// SINGLE operation on float operands whose double result is then rounded to float — (double)(a*b) - 0.5, a comparison against an
// exactly representable bound, a min / max that returns one of its arguments, and max(d, 0.1): no float lies in [0.1, 0.1f) — so the
// float operation written here gives the same float (IEEE: one correctly rounded operation).  What is NOT pinned is contraction: nvcc
// may fuse `wstart + iw * sub_bin` and the bilinear blend into FMAs (its default), this file is built with -ffp-contract=off; a
// sample within one ulp of a bin border can then take the other corner.  tests/test_psroi_gpu.py::test_samples_on_bin_borders_and_on_the_map_limits holds the exact cases.
// group_size: the reference's Python asserts channels == output_dim (modules/deform_psroi_pooling.py), which leaves only
// group_size == 1 usable; ps_check accepts the kernel's own condition (channels == output_dim * group_size^2) and the Python layer
// repeats the reference's assert.
// Layout stays the reference's NCHW: every bin samples one plane, and neighbouring lanes are neighbouring bins of the
// same plane, so their taps fall into the same few cache lines.
// What differs from the reference's launch: the grid covers every bin once (no 4096-block cap with a grid-stride loop),
// an RoI whose batch index is outside the batch produces zeros instead of reading beyond the map, and the backward pass
// sums a bin's shift gradient over its samples in registers (2 atomics per bin instead of 2 per sample).
#include "common.hpp"

namespace fv2p {

struct PsGeom {
  int batch, channels, height, width, rois, no_trans, out_dim, group, pooled, part, spp, classes, per_class;
  float scale, trans_std;
};

struct PsBin {
  bool live;
  int n, cls, part_h, part_w;
  long long plane;       // offset of the sampled channel plane in the map
  long long trans_x, trans_y;   // offsets of the two shift entries in trans
  float wstart, hstart, sub_w, sub_h, roi_w, roi_h;
};

__device__ __forceinline__ PsBin ps_bin(const PsGeom& g, const float* __restrict__ rois, const float* __restrict__ trans, long long index) {
  PsBin b;
  const int pw = static_cast<int>(index % g.pooled);
  const int ph = static_cast<int>((index / g.pooled) % g.pooled);
  const int ctop = static_cast<int>((index / g.pooled / g.pooled) % g.out_dim);
  b.n = static_cast<int>(index / g.pooled / g.pooled / g.out_dim);
  const float* r = rois + static_cast<long long>(b.n) * 5;
  const int bi = static_cast<int>(r[0]);
  b.live = bi >= 0 && bi < g.batch;
  const float start_w = roundf(r[1]) * g.scale - 0.5f;
  const float start_h = roundf(r[2]) * g.scale - 0.5f;
  const float end_w = (roundf(r[3]) + 1.f) * g.scale - 0.5f;
  const float end_h = (roundf(r[4]) + 1.f) * g.scale - 0.5f;
  b.roi_w = fmaxf(end_w - start_w, 0.1f);
  b.roi_h = fmaxf(end_h - start_h, 0.1f);
  const float bin_h = b.roi_h / static_cast<float>(g.pooled);
  const float bin_w = b.roi_w / static_cast<float>(g.pooled);
  b.sub_h = bin_h / static_cast<float>(g.spp);
  b.sub_w = bin_w / static_cast<float>(g.spp);
  b.part_h = static_cast<int>(floorf(static_cast<float>(ph) / static_cast<float>(g.pooled) * static_cast<float>(g.part)));
  b.part_w = static_cast<int>(floorf(static_cast<float>(pw) / static_cast<float>(g.pooled) * static_cast<float>(g.part)));
  b.cls = ctop / g.per_class;
  b.trans_x = ((static_cast<long long>(b.n) * g.classes + b.cls) * 2 * g.part + b.part_h) * g.part + b.part_w;
  b.trans_y = (((static_cast<long long>(b.n) * g.classes + b.cls) * 2 + 1) * g.part + b.part_h) * g.part + b.part_w;
  const float tx = g.no_trans ? 0.f : trans[b.trans_x] * g.trans_std;
  const float ty = g.no_trans ? 0.f : trans[b.trans_y] * g.trans_std;
  b.wstart = static_cast<float>(pw) * bin_w + start_w;
  b.wstart += tx * b.roi_w;
  b.hstart = static_cast<float>(ph) * bin_h + start_h;
  b.hstart += ty * b.roi_h;
  int gw = static_cast<int>(floorf(static_cast<float>(pw) * static_cast<float>(g.group) / static_cast<float>(g.pooled)));
  int gh = static_cast<int>(floorf(static_cast<float>(ph) * static_cast<float>(g.group) / static_cast<float>(g.pooled)));
  gw = min(max(gw, 0), g.group - 1);
  gh = min(max(gh, 0), g.group - 1);
  const int c = (ctop * g.group + gh) * g.group + gw;
  b.plane = (static_cast<long long>(b.live ? bi : 0) * g.channels + c) * g.height * g.width;
  return b;
}

// sample position -> inside?; clamps w / h to the map when it is
__device__ __forceinline__ bool ps_sample(const PsGeom& g, const PsBin& b, int ih, int iw, float& w, float& h) {
  w = b.wstart + static_cast<float>(iw) * b.sub_w;
  h = b.hstart + static_cast<float>(ih) * b.sub_h;
  if (w < -0.5f || w > static_cast<float>(g.width) - 0.5f || h < -0.5f || h > static_cast<float>(g.height) - 0.5f) return false;
  w = fminf(fmaxf(w, 0.f), static_cast<float>(g.width) - 1.f);
  h = fminf(fmaxf(h, 0.f), static_cast<float>(g.height) - 1.f);
  return true;
}

__global__ __launch_bounds__(256) void psroi_fwd_k(PsGeom g, long long count, const float* __restrict__ data, const float* __restrict__ rois,
                                                   const float* __restrict__ trans, float* __restrict__ out, float* __restrict__ top_count) {
  const long long index = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (index >= count) return;
  const PsBin b = ps_bin(g, rois, trans, index);
  float sum = 0.f;
  int cnt = 0;
  if (b.live) {
    const float* plane = data + b.plane;
    for (int ih = 0; ih < g.spp; ++ih)
      for (int iw = 0; iw < g.spp; ++iw) {
        float w, h;
        if (!ps_sample(g, b, ih, iw, w, h)) continue;
        const int x1 = static_cast<int>(floorf(w)), x2 = static_cast<int>(ceilf(w));
        const int y1 = static_cast<int>(floorf(h)), y2 = static_cast<int>(ceilf(h));
        const float dx = w - static_cast<float>(x1), dy = h - static_cast<float>(y1);
        const float v11 = plane[y1 * g.width + x1], v12 = plane[y2 * g.width + x1];
        const float v21 = plane[y1 * g.width + x2], v22 = plane[y2 * g.width + x2];
        const float val = (1.f - dx) * (1.f - dy) * v11 + (1.f - dx) * dy * v12 + dx * (1.f - dy) * v21 + dx * dy * v22;
        sum += val;
        ++cnt;
      }
  }
  out[index] = cnt == 0 ? 0.f : sum / static_cast<float>(cnt);
  top_count[index] = static_cast<float>(cnt);
}

__global__ __launch_bounds__(256) void psroi_bwd_k(PsGeom g, long long count, const float* __restrict__ grad_out, const float* __restrict__ top_count,
                                                   const float* __restrict__ data, const float* __restrict__ rois, const float* __restrict__ trans,
                                                   float* __restrict__ grad_data, float* __restrict__ grad_trans) {
  const long long index = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (index >= count) return;
  const float cnt = top_count[index];
  if (cnt <= 0.f) return;
  const PsBin b = ps_bin(g, rois, trans, index);
  if (!b.live) return;
  const float diff = grad_out[index] / cnt;
  const float* plane = data + b.plane;
  float* gplane = grad_data + b.plane;
  float gx = 0.f, gy = 0.f;
  for (int ih = 0; ih < g.spp; ++ih)
    for (int iw = 0; iw < g.spp; ++iw) {
      float w, h;
      if (!ps_sample(g, b, ih, iw, w, h)) continue;
      const int x0 = static_cast<int>(floorf(w)), x1 = static_cast<int>(ceilf(w));
      const int y0 = static_cast<int>(floorf(h)), y1 = static_cast<int>(ceilf(h));
      const float dx = w - static_cast<float>(x0), dy = h - static_cast<float>(y0);
      atomicAdd(gplane + y0 * g.width + x0, (1.f - dx) * (1.f - dy) * diff);
      atomicAdd(gplane + y1 * g.width + x0, (1.f - dx) * dy * diff);
      atomicAdd(gplane + y0 * g.width + x1, dx * (1.f - dy) * diff);
      atomicAdd(gplane + y1 * g.width + x1, dx * dy * diff);
      if (g.no_trans) continue;
      const float u00 = plane[y0 * g.width + x0], u01 = plane[y1 * g.width + x0];
      const float u10 = plane[y0 * g.width + x1], u11 = plane[y1 * g.width + x1];
      float sx = (u11 * dy + u10 * (1.f - dy) - u01 * dy - u00 * (1.f - dy)) * g.trans_std * diff;
      sx *= b.roi_w;
      float sy = (u11 * dx + u01 * (1.f - dx) - u10 * dx - u00 * (1.f - dx)) * g.trans_std * diff;
      sy *= b.roi_h;
      gx += sx;
      gy += sy;
    }
  if (!g.no_trans) {
    atomicAdd(grad_trans + b.trans_x, gx);
    atomicAdd(grad_trans + b.trans_y, gy);
  }
}

static int ps_check(const PsGeom& g, const char* who) {
  FV2P_REQUIRE(g.batch >= 0 && g.channels > 0 && g.height > 0 && g.width > 0 && g.rois >= 0, FV2P_EINVAL, "%s: bad map / RoI sizes", who);
  FV2P_REQUIRE(g.out_dim > 0 && g.group > 0 && g.pooled > 0 && g.part > 0 && g.spp > 0, FV2P_EINVAL, "%s: output_dim, group_size, pooled_size, part_size and sample_per_part must be positive", who);
  FV2P_REQUIRE(g.channels == g.out_dim * g.group * g.group, FV2P_EINVAL,
               "%s: a bin of output channel c reads input channel (c * group_size + gh) * group_size + gw, so the map needs output_dim * group_size^2 = %d channels, got %d",
               who, g.out_dim * g.group * g.group, g.channels);
  FV2P_REQUIRE(g.classes > 0 && g.out_dim % g.classes == 0, FV2P_EINVAL, "%s: output_dim (%d) must be a multiple of the number of shift classes (%d)", who, g.out_dim, g.classes);
  return 0;
}

}  // namespace fv2p
using namespace fv2p;

#define PS_GEOM_ARGS int batch, int channels, int height, int width, int num_rois, int no_trans, float spatial_scale, int output_dim, \
                     int group_size, int pooled_size, int part_size, int sample_per_part, float trans_std, int num_classes
#define PS_GEOM_INIT                                                                                                          \
  PsGeom g{batch, channels, height, width, num_rois, no_trans ? 1 : 0, output_dim, group_size, pooled_size, part_size, sample_per_part, \
           no_trans ? 1 : num_classes, 0, spatial_scale, trans_std};                                                           \
  g.per_class = g.classes > 0 ? g.out_dim / g.classes : 0

extern "C" int fv2p_deform_psroi_pool_forward(const float* data, const float* rois, const float* trans, PS_GEOM_ARGS, float* out,
                                              float* top_count, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PS_GEOM_INIT;
  if (int rc = ps_check(g, "deform_psroi_pool_forward")) return rc;
  const long long count = static_cast<long long>(g.rois) * g.out_dim * g.pooled * g.pooled;
  if (count == 0) return 0;
  FV2P_REQUIRE(data && rois && out && top_count && (g.no_trans || trans), FV2P_EINVAL, "deform_psroi_pool_forward: null pointer");
  FV2P_REQUIRE(g.batch > 0, FV2P_EINVAL, "deform_psroi_pool_forward: RoIs over an empty batch");
  hipLaunchKernelGGL(psroi_fwd_k, dim3(static_cast<unsigned>(ceil_div(count, 256))), dim3(256), 0, stream, g, count, data, rois, trans, out, top_count);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// grad_data [B,C,H,W] and grad_trans (same shape as trans) must be zeroed by the caller: both are accumulated with atomics.
extern "C" int fv2p_deform_psroi_pool_backward(const float* grad_out, const float* data, const float* rois, const float* trans,
                                               const float* top_count, PS_GEOM_ARGS, float* grad_data, float* grad_trans,
                                               fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PS_GEOM_INIT;
  if (int rc = ps_check(g, "deform_psroi_pool_backward")) return rc;
  const long long count = static_cast<long long>(g.rois) * g.out_dim * g.pooled * g.pooled;
  if (count == 0 || g.batch == 0) return 0;
  FV2P_REQUIRE(grad_out && data && rois && top_count && grad_data && (g.no_trans || (trans && grad_trans)), FV2P_EINVAL,
               "deform_psroi_pool_backward: null pointer");
  hipLaunchKernelGGL(psroi_bwd_k, dim3(static_cast<unsigned>(ceil_div(count, 256))), dim3(256), 0, stream, g, count, grad_out, top_count, data, rois,
                     trans, grad_data, grad_trans);
  FV2P_LAUNCH_CHECK();
  return 0;
}
