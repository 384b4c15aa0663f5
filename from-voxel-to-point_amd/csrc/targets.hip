// (f).4 — second-stage target sampling as one launch per batch.
//
// Replaces the per-sample Python of ProposalTargetLayer.sample_rois_for_rcnn / subsample_rois
// (pcdet/models/roi_heads/target_assigner/proposal_target_layer.py:92-217): per sample the best ground-truth box of every
// RoI, the foreground / hard-background / easy-background sets, a random permutation of the foreground set, sampling
// with replacement from the two background sets, and the gather of the sampled RoIs with their boxes.  The reference
// draws from torch's generator on the host side of boolean-mask indexing (`nonzero`, `.item()`); here the randomness
// is a caller-provided tensor of uniforms in [0, 1) (u[0..R) orders the foreground set, u[R..R+n) picks with
// replacement), so the launch is a pure function of its inputs: the replay harness's tensor formulation
// (fv2p_harness/fv2p_model.py: sample_targets_tensor_ops) and this kernel return the same rows bit for bit.
#include "common.hpp"

namespace fv2p {

// one workgroup per sample, one thread per RoI (R <= 1024)
__global__ __launch_bounds__(1024) void roi_sample_targets_k(int r, int g, int n, int gt_w, const float* __restrict__ iou,
                                                             const float* __restrict__ rois, const float* __restrict__ gt,
                                                             const float* __restrict__ uniforms, float fg_thresh, float bg_lo, float reg_fg,
                                                             int fg_quota, float hard_ratio, float* __restrict__ s_rois,
                                                             float* __restrict__ s_gt, float* __restrict__ s_iou, int* __restrict__ s_index) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_u = reinterpret_cast<float*>(smem);                 // [r] permutation keys
  float* s_ov = s_u + r;                                       // [r] best overlap
  int* s_as = reinterpret_cast<int*>(s_ov + r);                // [r] its ground-truth box
  unsigned char* s_cls = reinterpret_cast<unsigned char*>(s_as + r);   // [r] 1 fg, 2 hard, 4 easy (fg and hard may overlap)
  int* s_fg = reinterpret_cast<int*>(s_cls + ((r + 3) & ~3));  // [r] foreground RoIs in key order
  int* s_hard = s_fg + r;                                      // [r] hard background in index order
  int* s_easy = s_hard + r;                                    // [r] easy background in index order
  __shared__ int s_n[3];
  const int b = blockIdx.x, i = threadIdx.x;
  if (i < 3) s_n[i] = 0;
  float ov = 0.f;
  int as = 0, cls = 0;
  if (i < r) {
    const float* row = iou + (static_cast<long long>(b) * r + i) * g;
    ov = row[0];
    for (int j = 1; j < g; ++j) {     // first maximum, as torch.max over the last dimension
      const float v = row[j];
      if (v > ov) { ov = v; as = j; }
    }
    cls = (ov >= fg_thresh ? 1 : 0) | ((ov < reg_fg && ov >= bg_lo) ? 2 : 0) | (ov < bg_lo ? 4 : 0);
    s_u[i] = uniforms[static_cast<long long>(b) * (r + n) + i];
    s_ov[i] = ov;
    s_as[i] = as;
    s_cls[i] = static_cast<unsigned char>(cls);
  }
  __syncthreads();
  if (i < r) {
    // ranks: foreground by (key, index) ascending = stable sort by key; the background sets by index
    const float u = s_u[i];
    int rank_fg = 0, rank_hard = 0, rank_easy = 0;
    for (int j = 0; j < r; ++j) {
      const int c = s_cls[j];
      const float uj = s_u[j];
      rank_fg += ((c & 1) && (uj < u || (uj == u && j < i))) ? 1 : 0;
      rank_hard += ((c & 2) && j < i) ? 1 : 0;
      rank_easy += ((c & 4) && j < i) ? 1 : 0;
    }
    if (cls & 1) { s_fg[rank_fg] = i; atomicAdd(&s_n[0], 1); }
    if (cls & 2) { s_hard[rank_hard] = i; atomicAdd(&s_n[1], 1); }
    if (cls & 4) { s_easy[rank_easy] = i; atomicAdd(&s_n[2], 1); }
  }
  __syncthreads();
  if (i >= n) return;
  const int n_fg = s_n[0], n_hard = s_n[1], n_easy = s_n[2], n_bg = n_hard + n_easy;
  const int fg_take = n_bg > 0 ? min(n_fg, fg_quota) : (n_fg > 0 ? n : 0);
  const int bg_take = n - fg_take;
  int hard_take = n_easy > 0 ? min(static_cast<int>(static_cast<float>(bg_take) * hard_ratio), n_hard) : bg_take;
  if (n_hard == 0) hard_take = 0;
  const float up = uniforms[static_cast<long long>(b) * (r + n) + r + i];
  // floor(u * count) clamped to count - 1; an empty set answers RoI 0 (the head of "members first, then the rest")
  auto pick = [&](const int* lst, int count) {
    int j = static_cast<int>(up * static_cast<float>(count));
    j = min(j, max(count - 1, 0));
    return count > 0 ? lst[j] : 0;
  };
  int sel;
  if (i < fg_take) sel = n_bg == 0 ? pick(s_fg, n_fg) : s_fg[i];
  else if (i < fg_take + hard_take) sel = pick(s_hard, n_hard);
  else sel = pick(s_easy, n_easy);
  const long long o = static_cast<long long>(b) * n + i;
  s_index[o] = sel;
  s_iou[o] = s_ov[sel];
  const float* rs = rois + (static_cast<long long>(b) * r + sel) * 7;
  for (int c = 0; c < 7; ++c) s_rois[o * 7 + c] = rs[c];
  const float* gs = gt + (static_cast<long long>(b) * g + s_as[sel]) * gt_w;
  for (int c = 0; c < gt_w; ++c) s_gt[o * gt_w + c] = gs[c];
}

}  // namespace fv2p
using namespace fv2p;

extern "C" int fv2p_roi_sample_targets(const float* iou, const float* rois, const float* gt, const float* uniforms, int batch, int r, int g,
                                       int n, int gt_w, float fg_thresh, float bg_lo, float reg_fg, int fg_quota, float hard_ratio,
                                       float* s_rois, float* s_gt, float* s_iou, int* s_index, fv2p_stream_t stream_) {
  FV2P_REQUIRE(batch >= 0 && r >= 1 && r <= 1024 && g >= 1 && n >= 1 && n <= r && gt_w >= 7, FV2P_EINVAL,
               "roi_sample_targets: need 1 <= n <= rois <= 1024, at least one (padded) ground-truth box of >= 7 values");
  if (batch == 0) return 0;
  FV2P_REQUIRE(iou && rois && gt && uniforms && s_rois && s_gt && s_iou && s_index, FV2P_EINVAL, "roi_sample_targets: null pointer");
  const size_t lds = static_cast<size_t>(r) * (3 * sizeof(float) + 3 * sizeof(int)) + ((r + 3) & ~3);
  const int threads = ((r + 63) / 64) * 64;
  hipLaunchKernelGGL(roi_sample_targets_k, dim3(batch), dim3(threads), lds, static_cast<hipStream_t>(stream_), r, g, n, gt_w, iou, rois, gt,
                     uniforms, fg_thresh, bg_lo, reg_fg, fg_quota, hard_ratio, s_rois, s_gt, s_iou, s_index);
  FV2P_LAUNCH_CHECK();
  return 0;
}
