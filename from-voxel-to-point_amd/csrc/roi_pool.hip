// A15 / A18 — point-in-rotated-box tests, RoI-aware voxel pooling and RoI point pooling.
//
// Replaces roiaware_pool3d_cuda.{forward, backward, points_in_boxes_gpu, points_in_boxes_cpu}
// (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:29-177, roiaware_pool3d_kernel.cu:16-359) and
// roipoint_pool3d_cuda.forward (pcdet/ops/roipoint_pool3d/src/roipoint_pool3d.cpp, roipoint_pool3d_kernel.cu:16-165).
//
// The reference materialises a boxes x points int matrix (cudaMalloc per call) and then lets ONE thread per box
// walk all points serially to keep index order.  Here a wavefront owns a box and walks the points 64 at a time:
// a wave64 ballot + popcount gives every inside point its position in index order, so the order-dependent
// outputs ("first 512 points inside", "first max_pts points of each voxel") are reproduced exactly with no
// temporary matrix and no serial thread.
#include "common.hpp"
#include "../../include/fv2p_math.h"

namespace fv2p {

// check_pt_in_box3d (roiaware_pool3d_kernel.cu:23-36): z test and the margin comparison are carried out in
// double exactly as the reference's mixed float/double expression does; the rotation is fp32.
FV2P_HD int pt_in_box3d(const float* pt, const float* box, float margin, float* local_x, float* local_y) {
  const float x = pt[0], y = pt[1], z = pt[2];
  const float cx = box[0], cy = box[1], cz = box[2];
  const float dx = box[3], dy = box[4], dz = box[5], rz = box[6];
  if (static_cast<double>(fabsf(z - cz)) > static_cast<double>(dz) / 2.0) return 0;
  const float sx = x - cx, sy = y - cy;
  const float cosa = fv2p_cosf(-rz), sina = fv2p_sinf(-rz);
  const float lx = sx * cosa + sy * (-sina);
  const float ly = sx * sina + sy * cosa;
  *local_x = lx;
  *local_y = ly;
  const bool in_x = static_cast<double>(fabsf(lx)) < static_cast<double>(dx) / 2.0 + static_cast<double>(margin);
  const bool in_y = static_cast<double>(fabsf(ly)) < static_cast<double>(dy) / 2.0 + static_cast<double>(margin);
  return in_x && in_y;
}

constexpr float kMarginGpu = 1e-5f;  // roiaware_pool3d_kernel.cu:27, roipoint_pool3d_kernel.cu:26
constexpr float kMarginCpu = 1e-2f;  // roiaware_pool3d.cpp:131

// (B, npoints) <- index of the first box containing the point, or -1 (roiaware_pool3d_kernel.cu:313-336)
__global__ __launch_bounds__(256) void points_in_boxes_k(int boxes_num, int pts_num, const float* __restrict__ boxes,
                                                         const float* __restrict__ pts, int* __restrict__ out) {
  extern __shared__ float sbox[];  // boxes of this sample
  const int b = blockIdx.y;
  for (int e = threadIdx.x; e < boxes_num * 7; e += 256) sbox[e] = boxes[static_cast<int64_t>(b) * boxes_num * 7 + e];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= pts_num) return;
  const float* p = pts + (static_cast<int64_t>(b) * pts_num + i) * 3;
  const float pt[3] = {p[0], p[1], p[2]};
  int idx = -1;
  float lx, ly;
  for (int k = 0; k < boxes_num; ++k)
    if (pt_in_box3d(pt, sbox + k * 7, kMarginGpu, &lx, &ly)) { idx = k; break; }
  out[static_cast<int64_t>(b) * pts_num + i] = idx;
}

// ---- RoI point pooling: one workgroup of 16 waves per (sample, box) -----------------------------------
// (16 waves: at 384 boxes a 4-wave workgroup per box left six waves on a CU, and both phases wait on memory)
constexpr int kRoiPoolWaves = 16;
__global__ __launch_bounds__(kRoiPoolWaves * 64) void roipoint_pool_k(int pts_num, int boxes_num, int feat_len, int sampled, const float* __restrict__ xyz,
                                                       const float* __restrict__ boxes3d, const float* __restrict__ feats,
                                                       float* __restrict__ pooled, int* __restrict__ empty_flag) {
  extern __shared__ int sidx[];  // [sampled]
  __shared__ int wave_cnt[kRoiPoolWaves];
  constexpr int T = kRoiPoolWaves * 64;
  const int m = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float box[7];
  for (int j = 0; j < 7; ++j) box[j] = boxes3d[(static_cast<int64_t>(b) * boxes_num + m) * 7 + j];
  int total = 0;
  for (int base = 0; base < pts_num && total < sampled; base += T) {
    const int i = base + threadIdx.x;
    int in = 0;
    if (i < pts_num) {
      const float* p = xyz + (static_cast<int64_t>(b) * pts_num + i) * 3;
      const float pt[3] = {p[0], p[1], p[2]};
      float lx, ly;
      in = pt_in_box3d(pt, box, kMarginGpu, &lx, &ly);
    }
    const uint64_t vote = __ballot(in);
    if (lane == 0) wave_cnt[w] = __popcll(vote);
    __syncthreads();
    int pos = total + __popcll(vote & lanemask_lt()), all = 0;
#pragma unroll
    for (int ww = 0; ww < kRoiPoolWaves; ++ww) { const int c = wave_cnt[ww]; if (ww < w) pos += c; all += c; }
    if (in && pos < sampled) sidx[pos] = i;
    total += all;
    __syncthreads();
  }
  const int cnt = total < sampled ? total : sampled;
  if (threadIdx.x == 0) empty_flag[static_cast<int64_t>(b) * boxes_num + m] = (cnt == 0) ? 1 : 0;
  const int row = 3 + feat_len;
  float* dst = pooled + (static_cast<int64_t>(b) * boxes_num + m) * sampled * row;
  if (cnt == 0) {   // an empty box pools zeros (the reference pre-zeroes the whole output, roipoint_pool3d_utils.py:54; here only these rows)
    for (int e = threadIdx.x; e < sampled * row; e += T) dst[e] = 0.f;
    return;
  }
  // wrap-around duplication (roipoint_pool3d_kernel.cu:90-98): slot k takes slot k % cnt.  A wave copies four slots at a time, a lane the
  // elements lane, lane + 64, ... of each row: the loads of the four rows are independent (one slot per iteration and thread-flat indexing,
  // the first form, left every 4-byte load waiting for the one before it: 170 us at 384 boxes x 512 slots x 133 floats).
  const float* fx = xyz + static_cast<int64_t>(b) * pts_num * 3;
  const float* ff = feats + static_cast<int64_t>(b) * pts_num * feat_len;
  for (int s0 = 4 * w; s0 < sampled; s0 += 4 * kRoiPoolWaves) {
    int src[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int sl = s0 + u < sampled ? s0 + u : sampled - 1; src[u] = sidx[sl < cnt ? sl : sl % cnt]; }
    for (int j0 = 0; j0 < row; j0 += 128) {
      float v[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int j = j0 + 64 * q + lane;
          v[u][q] = j < 3 ? fx[static_cast<int64_t>(src[u]) * 3 + j] : (j < row ? ff[static_cast<int64_t>(src[u]) * feat_len + (j - 3)] : 0.f);
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int j = j0 + 64 * q + lane;
          if (j < row && s0 + u < sampled) dst[static_cast<int64_t>(s0 + u) * row + j] = v[u][q];
        }
    }
  }
}

// ---- RoI-aware pooling ------------------------------------------------------------------------------
// One wave per box: ordered binning of the inside points into the box's out_x*out_y*out_z voxels
// (generate_pts_mask_for_box3d + collect_inside_pts_for_box3d, roiaware_pool3d_kernel.cu:39-108).
__global__ __launch_bounds__(64) void roiaware_collect_k(int pts_num, int max_pts, int out_x, int out_y, int out_z,
                                                         const float* __restrict__ rois, const float* __restrict__ pts,
                                                         int* __restrict__ pts_idx_of_voxels) {
  extern __shared__ int s_cnt[];  // per-voxel fill count of this box (LDS: same-wave ds ops are ordered)
  const int box_idx = blockIdx.x, lane = threadIdx.x;
  const int nvox = out_x * out_y * out_z;
  for (int e = lane; e < nvox; e += 64) s_cnt[e] = 0;
  __builtin_amdgcn_wave_barrier();
  float box[7];
  for (int j = 0; j < 7; ++j) box[j] = rois[box_idx * 7 + j];
  int* vox = pts_idx_of_voxels + static_cast<int64_t>(box_idx) * nvox * max_pts;
  const int max_num = max_pts - 1;  // slot 0 is the counter
  const float dx = box[3], dy = box[4], dz = box[5];
  const float x_res = dx / out_x, y_res = dy / out_y, z_res = dz / out_z;
  for (int base = 0; base < pts_num; base += 64) {
    const int i = base + lane;
    int in = 0, v = -1;
    if (i < pts_num) {
      const float pt[3] = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
      float lx = 0.f, ly = 0.f;
      in = pt_in_box3d(pt, box, kMarginGpu, &lx, &ly);
      if (in) {
        const float lz = pt[2] - box[2];
        int xi = static_cast<int>((lx + dx / 2) / x_res);
        int yi = static_cast<int>((ly + dy / 2) / y_res);
        int zi = static_cast<int>((lz + dz / 2) / z_res);
        xi = min(max(xi, 0), out_x - 1);
        yi = min(max(yi, 0), out_y - 1);
        zi = min(max(zi, 0), out_z - 1);
        v = (xi * out_y + yi) * out_z + zi;
      }
    }
    uint64_t todo = __ballot(in);
    while (todo) {  // one round per distinct voxel among this chunk's inside points, in lane (= index) order
      const int leader = __ffsll(static_cast<long long>(todo)) - 1;
      const int lv = __shfl(v, leader, 64);
      const uint64_t same = __ballot(in && v == lv);
      const int cnt0 = s_cnt[lv];
      __builtin_amdgcn_wave_barrier();
      if (in && v == lv) {
        const int r = cnt0 + __popcll(same & lanemask_lt());
        if (r < max_num) vox[static_cast<int64_t>(lv) * max_pts + r + 1] = i;
      }
      if (lane == leader) {
        const int nc = cnt0 + __popcll(same);
        s_cnt[lv] = nc < max_num ? nc : max_num;
      }
      __builtin_amdgcn_wave_barrier();
      todo &= ~same;
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (int e = lane; e < nvox; e += 64) vox[static_cast<int64_t>(e) * max_pts] = s_cnt[e];
}

template <int METHOD>  // 0 max, 1 avg
__global__ void roiaware_pool_k(int64_t total, int channels, int max_pts, const float* __restrict__ feat,
                                const int* __restrict__ pts_idx_of_voxels, float* __restrict__ pooled, int* __restrict__ argmax) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t voxel = t / channels;
  const int c = static_cast<int>(t % channels);
  const int* list = pts_idx_of_voxels + voxel * max_pts;
  const int n = list[0];
  if (METHOD == 0) {
    int am = -1;
    float mv = -INFINITY;  // reference: float max_val = -1e50 (== -inf in float), :137
    for (int k = 1; k <= n; ++k) {
      const float f = feat[static_cast<int64_t>(list[k]) * channels + c];
      if (f > mv) { mv = f; am = list[k]; }
    }
    if (am != -1) pooled[t] = mv;
    argmax[t] = am;
  } else {
    float s = 0.f;
    for (int k = 1; k <= n; ++k) s += feat[static_cast<int64_t>(list[k]) * channels + c];
    if (n > 0) pooled[t] = s / n;
  }
}

template <int METHOD>
__global__ void roiaware_pool_bwd_k(int64_t total, int channels, int max_pts, const int* __restrict__ pts_idx_of_voxels,
                                    const int* __restrict__ argmax, const float* __restrict__ grad_out, float* __restrict__ grad_in) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t voxel = t / channels;
  const int c = static_cast<int>(t % channels);
  if (METHOD == 0) {
    const int am = argmax[t];
    if (am != -1) atomicAdd(&grad_in[static_cast<int64_t>(am) * channels + c], grad_out[t]);
  } else {
    const int* list = pts_idx_of_voxels + voxel * max_pts;
    const int n = list[0];
    const float g = grad_out[t] * (1.0f / fmaxf(static_cast<float>(n), 1.0f));
    for (int k = 1; k <= n; ++k) atomicAdd(&grad_in[static_cast<int64_t>(list[k]) * channels + c], g);
  }
}

}  // namespace fv2p
using namespace fv2p;

extern "C" int fv2p_points_in_boxes(const float* boxes, const float* pts, int batch, int boxes_num, int pts_num, int* box_idx_of_points,
                                    fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(batch >= 0 && boxes_num >= 0 && pts_num >= 0, FV2P_EINVAL, "points_in_boxes: negative size");
  if (batch == 0 || pts_num == 0) return 0;
  FV2P_REQUIRE(pts && box_idx_of_points && (boxes || boxes_num == 0), FV2P_EINVAL, "points_in_boxes: null pointer");
  FV2P_REQUIRE(static_cast<size_t>(boxes_num) * 7 * sizeof(float) <= 60000, FV2P_ELIMIT, "points_in_boxes: more than 2142 boxes per sample");
  hipLaunchKernelGGL(points_in_boxes_k, dim3((unsigned)ceil_div(pts_num, 256), batch), dim3(256), boxes_num * 7 * sizeof(float), stream,
                     boxes_num, pts_num, boxes, pts, box_idx_of_points);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// host entry point (roiaware_pool3d.cpp:143-168): [boxes_num, pts_num] 0/1 matrix, MARGIN 1e-2
extern "C" int fv2p_points_in_boxes_cpu(const float* boxes, const float* pts, int boxes_num, int pts_num, int* pts_indices) {
  FV2P_REQUIRE(boxes_num >= 0 && pts_num >= 0, FV2P_EINVAL, "points_in_boxes_cpu: negative size");
  if (boxes_num == 0 || pts_num == 0) return 0;
  FV2P_REQUIRE(boxes && pts && pts_indices, FV2P_EINVAL, "points_in_boxes_cpu: null pointer");
  float lx, ly;
  for (int i = 0; i < boxes_num; ++i)
    for (int j = 0; j < pts_num; ++j)
      pts_indices[static_cast<int64_t>(i) * pts_num + j] = pt_in_box3d(pts + j * 3, boxes + i * 7, kMarginCpu, &lx, &ly);
  return 0;
}

extern "C" int fv2p_roipoint_pool3d(const float* xyz, const float* boxes3d, const float* pts_feature, int batch, int pts_num, int boxes_num,
                                    int feature_len, int sampled_pts_num, float* pooled_features, int* pooled_empty_flag,
                                    fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(batch >= 0 && pts_num >= 0 && boxes_num >= 0 && feature_len >= 0 && sampled_pts_num >= 1, FV2P_EINVAL,
               "roipoint_pool3d: bad sizes");
  if (batch == 0 || boxes_num == 0) return 0;
  FV2P_REQUIRE(boxes3d && pooled_features && pooled_empty_flag && (xyz || pts_num == 0) && (pts_feature || feature_len == 0 || pts_num == 0),
               FV2P_EINVAL, "roipoint_pool3d: null pointer");
  FV2P_REQUIRE(static_cast<size_t>(sampled_pts_num) * 4 <= 60000, FV2P_ELIMIT, "roipoint_pool3d: sampled_pts_num > 15000");
  hipLaunchKernelGGL(roipoint_pool_k, dim3(boxes_num, batch), dim3(kRoiPoolWaves * 64), sampled_pts_num * sizeof(int), stream, pts_num, boxes_num,
                     feature_len, sampled_pts_num, xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_roiaware_pool3d_fwd(const float* rois, const float* pts, const float* pts_feature, int boxes_num, int pts_num,
                                        int channels, int max_pts_each_voxel, int out_x, int out_y, int out_z, int pool_method,
                                        int* argmax, int* pts_idx_of_voxels, float* pooled_features, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(boxes_num >= 0 && pts_num >= 0 && channels >= 1 && max_pts_each_voxel >= 2, FV2P_EINVAL, "roiaware_pool3d: bad sizes");
  FV2P_REQUIRE(out_x >= 1 && out_y >= 1 && out_z >= 1 && out_x < 256 && out_y < 256 && out_z < 256, FV2P_EINVAL,
               "roiaware_pool3d: out size must be in [1,255] (roiaware_pool3d.cpp:53)");
  FV2P_REQUIRE(pool_method == 0 || pool_method == 1, FV2P_EINVAL, "roiaware_pool3d: pool_method must be 0 (max) or 1 (avg)");
  FV2P_REQUIRE(static_cast<int64_t>(out_x) * out_y * out_z * 4 <= 64000, FV2P_ELIMIT, "roiaware_pool3d: more than 16000 voxels per RoI");
  if (boxes_num == 0) return 0;
  FV2P_REQUIRE(rois && argmax && pts_idx_of_voxels && pooled_features && (pts || pts_num == 0), FV2P_EINVAL, "roiaware_pool3d: null pointer");
  const int64_t voxels = static_cast<int64_t>(boxes_num) * out_x * out_y * out_z;
  FV2P_HIP(hipMemsetAsync(pts_idx_of_voxels, 0, sizeof(int) * voxels * max_pts_each_voxel, stream));
  FV2P_HIP(hipMemsetAsync(pooled_features, 0, sizeof(float) * voxels * channels, stream));
  FV2P_HIP(hipMemsetAsync(argmax, 0, sizeof(int) * voxels * channels, stream));
  if (pts_num > 0)
    hipLaunchKernelGGL(roiaware_collect_k, dim3(boxes_num), dim3(64), sizeof(int) * out_x * out_y * out_z, stream, pts_num, max_pts_each_voxel, out_x, out_y, out_z, rois, pts,
                       pts_idx_of_voxels);
  const int64_t total = voxels * channels;
  const dim3 grid(static_cast<unsigned>(ceil_div(total, 256))), block(256);
  if (pool_method == 0)
    hipLaunchKernelGGL(roiaware_pool_k<0>, grid, block, 0, stream, total, channels, max_pts_each_voxel, pts_feature, pts_idx_of_voxels,
                       pooled_features, argmax);
  else
    hipLaunchKernelGGL(roiaware_pool_k<1>, grid, block, 0, stream, total, channels, max_pts_each_voxel, pts_feature, pts_idx_of_voxels,
                       pooled_features, argmax);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_roiaware_pool3d_bwd(const int* pts_idx_of_voxels, const int* argmax, const float* grad_out, int boxes_num, int out_x,
                                        int out_y, int out_z, int channels, int max_pts_each_voxel, int pool_method, float* grad_in,
                                        fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(boxes_num >= 0 && channels >= 1 && (pool_method == 0 || pool_method == 1), FV2P_EINVAL, "roiaware_pool3d_bwd: bad arguments");
  if (boxes_num == 0) return 0;
  FV2P_REQUIRE(pts_idx_of_voxels && argmax && grad_out && grad_in, FV2P_EINVAL, "roiaware_pool3d_bwd: null pointer");
  const int64_t total = static_cast<int64_t>(boxes_num) * out_x * out_y * out_z * channels;
  const dim3 grid(static_cast<unsigned>(ceil_div(total, 256))), block(256);
  if (pool_method == 0)
    hipLaunchKernelGGL(roiaware_pool_bwd_k<0>, grid, block, 0, stream, total, channels, max_pts_each_voxel, pts_idx_of_voxels, argmax, grad_out, grad_in);
  else
    hipLaunchKernelGGL(roiaware_pool_bwd_k<1>, grid, block, 0, stream, total, channels, max_pts_each_voxel, pts_idx_of_voxels, argmax, grad_out, grad_in);
  FV2P_LAUNCH_CHECK();
  return 0;
}
