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

// ---- first-stage (anchor) target assignment ---------------------------------------------------------------------------------------
// AxisAlignedTargetAssigner.assign_targets_single (pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:66-210)
// for the whole batch in two launches: nearest-BEV IoU of every anchor with every ground-truth box, per-box maximum (pass 1),
// then per anchor: best box (first maximum), label = box class when the overlap reaches matched_thr or the anchor attains a box's
// (non-zero) maximum, 0 below unmatched_thr, -1 (ignored) in between, and the ResidualCoder regression target of positive anchors.
// Every float operation follows the tensor formulation of the replay harness (AnchorHead.assign_tensor_ops) in its order.
namespace fv2p {

__device__ __forceinline__ float aligned_iou(const float4 a, float area_a, const float4 g) {
  const float lx = fmaxf(a.x, g.x), ly = fmaxf(a.y, g.y), hx = fminf(a.z, g.z), hy = fminf(a.w, g.w);
  const float inter = fmaxf(hx - lx, 0.f) * fmaxf(hy - ly, 0.f);
  const float area_g = (g.z - g.x) * (g.w - g.y);
  return inter / fmaxf(area_a + area_g - inter, 1e-6f);
}

__global__ __launch_bounds__(256) void anchor_box_max_k(int n_anchor, int g, const float4* __restrict__ anchor_bev, const float4* __restrict__ gt_bev,
                                                        unsigned* __restrict__ g_max_bits) {
  extern __shared__ float4 s_gt[];
  const int b = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = threadIdx.x; j < g; j += blockDim.x) s_gt[j] = gt_bev[b * g + j];
  __syncthreads();
  if (a >= n_anchor) return;
  const float4 ab = anchor_bev[a];
  const float area_a = (ab.z - ab.x) * (ab.w - ab.y);
  for (int j = 0; j < g; ++j) {
    const float iou = aligned_iou(ab, area_a, s_gt[j]);
    if (iou > 0.f) atomicMax(&g_max_bits[b * g + j], __float_as_uint(iou));   // overlaps are >= 0: unsigned order = float order
  }
}

__global__ __launch_bounds__(256) void anchor_label_k(int n_anchor, int g, int gt_w, const float4* __restrict__ anchor_bev,
                                                      const float* __restrict__ anchors, const float4* __restrict__ gt_bev,
                                                      const float* __restrict__ gt, const unsigned* __restrict__ g_max_bits, float matched_thr,
                                                      float unmatched_thr, int* __restrict__ labels, float* __restrict__ reg) {
  extern __shared__ float4 s_gt[];
  float* s_max = reinterpret_cast<float*>(s_gt + g);
  const int b = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = threadIdx.x; j < g; j += blockDim.x) {
    s_gt[j] = gt_bev[b * g + j];
    const float m = __uint_as_float(g_max_bits[b * g + j]);
    s_max[j] = m == 0.f ? -1.f : m;   // a box nothing overlaps forces no anchor
  }
  __syncthreads();
  if (a >= n_anchor) return;
  const float4 ab = anchor_bev[a];
  const float area_a = (ab.z - ab.x) * (ab.w - ab.y);
  float best = aligned_iou(ab, area_a, s_gt[0]);
  int arg = 0;
  bool forced = best == s_max[0];
  for (int j = 1; j < g; ++j) {
    const float iou = aligned_iou(ab, area_a, s_gt[j]);
    if (iou > best) { best = iou; arg = j; }
    forced = forced || iou == s_max[j];
  }
  const float* gb = gt + (static_cast<long long>(b) * g + arg) * gt_w;
  int label = -1;
  if (best < unmatched_thr) label = 0;
  if (forced || best >= matched_thr) label = static_cast<int>(gb[7]);
  const long long o = static_cast<long long>(b) * n_anchor + a;
  labels[o] = label;
  float r[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (label > 0) {   // ResidualCoder.encode_torch (box_coder_utils.py:13-46)
    const float* an = anchors + static_cast<long long>(a) * 7;
    const float ax = fmaxf(an[3], 1e-5f), ay = fmaxf(an[4], 1e-5f), az = fmaxf(an[5], 1e-5f);
    const float bx = fmaxf(gb[3], 1e-5f), by = fmaxf(gb[4], 1e-5f), bz = fmaxf(gb[5], 1e-5f);
    const float diag = sqrtf(ax * ax + ay * ay);
    r[0] = (gb[0] - an[0]) / diag;
    r[1] = (gb[1] - an[1]) / diag;
    r[2] = (gb[2] - an[2]) / az;
    r[3] = logf(bx / ax);
    r[4] = logf(by / ay);
    r[5] = logf(bz / az);
    r[6] = gb[6] - an[6];
  }
#pragma unroll
  for (int c = 0; c < 7; ++c) reg[o * 7 + c] = r[c];
}

}  // namespace fv2p

extern "C" size_t fv2p_anchor_assign_ws_bytes(int batch, int g) { return static_cast<size_t>(batch > 0 ? batch : 1) * (g > 0 ? g : 1) * sizeof(unsigned); }

extern "C" int fv2p_anchor_assign(const float* anchor_bev, const float* anchors, int n_anchor, const float* gt_bev, const float* gt, int batch, int g,
                                  int gt_w, float matched_thr, float unmatched_thr, int* labels, float* reg, void* ws, size_t ws_bytes,
                                  fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(batch >= 0 && batch <= 65535 && n_anchor >= 1 && g >= 1 && g <= 1024 && gt_w >= 8, FV2P_EINVAL,
               "anchor_assign: need at least one anchor, 1..1024 (padded) boxes of >= 8 values (class id at [7])");
  if (batch == 0) return 0;
  FV2P_REQUIRE(anchor_bev && anchors && gt_bev && gt && labels && reg && ws, FV2P_EINVAL, "anchor_assign: null pointer");
  FV2P_REQUIRE(ws_bytes >= fv2p_anchor_assign_ws_bytes(batch, g), FV2P_EINVAL, "anchor_assign: workspace too small");
  FV2P_REQUIRE((reinterpret_cast<uintptr_t>(anchor_bev) & 15) == 0 && (reinterpret_cast<uintptr_t>(gt_bev) & 15) == 0, FV2P_EINVAL,
               "anchor_assign: the (x0, y0, x1, y1) footprints must be 16-byte aligned");
  unsigned* g_max = static_cast<unsigned*>(ws);
  FV2P_HIP(hipMemsetAsync(g_max, 0, static_cast<size_t>(batch) * g * sizeof(unsigned), stream));
  const dim3 grid(static_cast<unsigned>((n_anchor + 255) / 256), batch), block(256);
  hipLaunchKernelGGL(anchor_box_max_k, grid, block, g * sizeof(float4), stream, n_anchor, g, reinterpret_cast<const float4*>(anchor_bev),
                     reinterpret_cast<const float4*>(gt_bev), g_max);
  hipLaunchKernelGGL(anchor_label_k, grid, block, g * (sizeof(float4) + sizeof(float)), stream, n_anchor, g, gt_w,
                     reinterpret_cast<const float4*>(anchor_bev), anchors, reinterpret_cast<const float4*>(gt_bev), gt, g_max, matched_thr,
                     unmatched_thr, labels, reg);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// ---- first-stage losses in one pass -------------------------------------------------------------------------------------------------
// AnchorHeadTemplate.get_cls_layer_loss / get_box_reg_layer_loss (pcdet/models/dense_heads/anchor_head_template.py:98-206) with
// SigmoidFocalClassificationLoss (alpha, gamma = 2), WeightedSmoothL1Loss (beta) on the sin-difference encoded residuals and the
// direction-bin cross entropy (two bins), normalised per sample by max(#positive anchors, 1) and divided by the batch size: the
// reference composes ~45 element-wise torch ops over [B, A, .] tensors (and autograd ~70 more in the backward pass); here one
// kernel leaves the three weighted loss sums and the gradients with respect to the three logit tensors.
namespace fv2p {

__global__ __launch_bounds__(256) void anchor_pos_count_k(int n_anchor, const int* __restrict__ labels, int* __restrict__ pos) {
  const int b = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
  const bool t = a < n_anchor && labels[static_cast<long long>(b) * n_anchor + a] > 0;
  const int c = __popcll(__ballot(t));
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&pos[b], c);
}

struct AnchorLossArgs {
  int n_anchor, batch;
  const float *cls, *box, *dirs, *reg_t, *anchor_rot;
  const int *labels, *pos;
  float alpha, beta, dir_offset, w_cls, w_loc, w_dir;
  float *dcls, *dbox, *ddirs;
  double* partial;   // [blocks.x * batch][3]
};

__global__ __launch_bounds__(256) void anchor_loss_k(AnchorLossArgs p) {
  const int b = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
  const float norm = fmaxf(static_cast<float>(p.pos[b]), 1.f);
  double l_cls = 0.0, l_loc = 0.0, l_dir = 0.0;
  if (a < p.n_anchor) {
    const long long o = static_cast<long long>(b) * p.n_anchor + a;
    const int label = p.labels[o];
    const bool t = label > 0;
    const float inv_b = 1.f / static_cast<float>(p.batch);
    // classification: focal loss on the sigmoid, weight 1 / norm for positive and negative anchors, 0 for ignored ones
    const float x = p.cls[o];
    const float w = (label == 0 || t) ? 1.f / norm : 0.f;
    const float pr = 1.f / (1.f + expf(-x));
    const float aw = t ? p.alpha : 1.f - p.alpha;
    const float pt = t ? 1.f - pr : pr;
    const float bce = fmaxf(x, 0.f) - (t ? x : 0.f) + log1pf(expf(-fabsf(x)));
    l_cls = static_cast<double>(aw * pt * pt * bce * w);
    const float dpt = t ? -pr * (1.f - pr) : pr * (1.f - pr);
    p.dcls[o] = aw * w * (2.f * pt * dpt * bce + pt * pt * (pr - (t ? 1.f : 0.f))) * inv_b * p.w_cls;
    // localisation and direction: positive anchors only
    float db[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dd[2] = {0.f, 0.f};
    if (t) {
      const float* bx = p.box + o * 7;
      const float* rt = p.reg_t + o * 7;
      const float rw = 1.f / norm;
      float loc = 0.f;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        float d, chain = 1.f;
        if (j < 6) d = bx[j] - rt[j];
        else {
          const float sb = sinf(bx[6]), cb = cosf(bx[6]), sr = sinf(rt[6]), cr = cosf(rt[6]);
          d = sb * cr - cb * sr;
          chain = cb * cr + sb * sr;
        }
        const float n = fabsf(d);
        loc += n < p.beta ? 0.5f * n * n / p.beta : n - 0.5f * p.beta;
        const float g = n < p.beta ? d / p.beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        db[j] = g * chain * rw * inv_b * p.w_loc;
      }
      l_loc = static_cast<double>(loc * rw);
      // direction bin of the target heading (two bins over [0, 2 pi) from dir_offset)
      const float two_pi = 6.283185307179586f;
      const float v = rt[6] + p.anchor_rot[a] - p.dir_offset;
      const float lp = v - floorf(v / two_pi + 0.f) * two_pi;
      int bin = static_cast<int>(floorf(lp / (two_pi / 2.f)));
      bin = bin < 0 ? 0 : (bin > 1 ? 1 : bin);
      const float z0 = p.dirs[o * 2], z1 = p.dirs[o * 2 + 1];
      const float mz = fmaxf(z0, z1);
      const float e0 = expf(z0 - mz), e1 = expf(z1 - mz), se = e0 + e1;
      const float lse = mz + logf(se);
      l_dir = static_cast<double>((lse - (bin ? z1 : z0)) * rw);
      dd[0] = (e0 / se - (bin == 0 ? 1.f : 0.f)) * rw * inv_b * p.w_dir;
      dd[1] = (e1 / se - (bin == 1 ? 1.f : 0.f)) * rw * inv_b * p.w_dir;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) p.dbox[o * 7 + j] = db[j];
    p.ddirs[o * 2] = dd[0];
    p.ddirs[o * 2 + 1] = dd[1];
  }
  // block sums in double, one partial row per block (summed in block order by the reduce launch: deterministic)
  __shared__ double s_red[3][4];
  double v3[3] = {l_cls, l_loc, l_dir};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double v = v3[c];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) s_red[c][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double v = s_red[threadIdx.x][0] + s_red[threadIdx.x][1] + s_red[threadIdx.x][2] + s_red[threadIdx.x][3];
    p.partial[(static_cast<long long>(b) * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(256) void anchor_loss_reduce_k(int rows, int batch, const double* __restrict__ partial, float w_cls, float w_loc, float w_dir,
                                                            float* __restrict__ loss) {
  __shared__ double s[3][256];
  double acc[3] = {0.0, 0.0, 0.0};
  for (int r = threadIdx.x; r < rows; r += 256)
    for (int c = 0; c < 3; ++c) acc[c] += partial[static_cast<long long>(r) * 3 + c];
  for (int c = 0; c < 3; ++c) s[c][threadIdx.x] = acc[c];
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if (threadIdx.x < d)
      for (int c = 0; c < 3; ++c) s[c][threadIdx.x] += s[c][threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] = static_cast<float>((s[0][0] * w_cls + s[1][0] * w_loc + s[2][0] * w_dir) / batch);
    loss[1] = static_cast<float>(s[0][0] / batch * w_cls);
    loss[2] = static_cast<float>(s[1][0] / batch * w_loc);
    loss[3] = static_cast<float>(s[2][0] / batch * w_dir);
  }
}

}  // namespace fv2p

extern "C" size_t fv2p_anchor_loss_ws_bytes(int batch, int n_anchor) {
  const size_t blocks = static_cast<size_t>((n_anchor + 255) / 256) * (batch > 0 ? batch : 1);
  return blocks * 3 * sizeof(double) + 64 * sizeof(int);
}

extern "C" int fv2p_anchor_loss(const float* cls, const float* box, const float* dirs, const int* labels, const float* reg_t,
                                const float* anchor_rot, int batch, int n_anchor, float alpha, float beta, float dir_offset, float w_cls,
                                float w_loc, float w_dir, float* loss4, float* dcls, float* dbox, float* ddirs, void* ws, size_t ws_bytes,
                                fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(batch >= 1 && batch <= 64 && n_anchor >= 1, FV2P_EINVAL, "anchor_loss: 1..64 samples, at least one anchor");
  FV2P_REQUIRE(cls && box && dirs && labels && reg_t && anchor_rot && loss4 && dcls && dbox && ddirs && ws, FV2P_EINVAL, "anchor_loss: null pointer");
  FV2P_REQUIRE(ws_bytes >= fv2p_anchor_loss_ws_bytes(batch, n_anchor), FV2P_EINVAL, "anchor_loss: workspace too small");
  const unsigned bx = static_cast<unsigned>((n_anchor + 255) / 256);
  int* pos = static_cast<int*>(ws);
  double* partial = reinterpret_cast<double*>(static_cast<char*>(ws) + 64 * sizeof(int));
  FV2P_HIP(hipMemsetAsync(pos, 0, 64 * sizeof(int), stream));
  hipLaunchKernelGGL(anchor_pos_count_k, dim3(bx, batch), dim3(256), 0, stream, n_anchor, labels, pos);
  AnchorLossArgs a{n_anchor, batch, cls, box, dirs, reg_t, anchor_rot, labels, pos, alpha, beta, dir_offset, w_cls, w_loc, w_dir, dcls, dbox, ddirs, partial};
  hipLaunchKernelGGL(anchor_loss_k, dim3(bx, batch), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(anchor_loss_reduce_k, dim3(1), dim3(256), 0, stream, static_cast<int>(bx) * batch, batch, partial, w_cls, w_loc, w_dir, loss4);
  FV2P_LAUNCH_CHECK();
  return 0;
}
