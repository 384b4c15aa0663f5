// A16 — rotated BEV overlap / IoU and rotated / axis-aligned NMS.
//
// Replaces iou3d_nms_cuda.{boxes_overlap_bev_gpu, boxes_iou_bev_gpu, nms_gpu, nms_normal_gpu, boxes_iou_bev_cpu}
// (pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17, iou3d_nms.cpp:49-187, iou3d_nms_kernel.cu:104-372,
// iou3d_cpu.cpp:232-252).  The polygon-clipping arithmetic follows box_overlap() of the reference step by
// step in fp32 (rotate corners, 16 edge intersections, corner containment with MARGIN 1e-2, bubble sort by
// atan2 around the centroid, shoelace); sin/cos/atan2 come from include/fv2p_math.h so that host and device
// agree bit for bit.
//
// NMS: the 64x64 tile of the suppression bit-matrix is exactly one wave64 (one box row per lane); only the
// upper-triangular tiles are computed (the greedy pass never reads the others, iou3d_nms.cpp:121-135).
// The greedy pass itself runs on the device (single workgroup: in-register scan of each 64-box diagonal
// tile + parallel OR of the kept rows), which removes the reference's N*N/8-byte mask D2H copy
// (10.2 MB at N=9000) and its host loop.
#include "common.hpp"
#include "../../include/fv2p_math.h"

namespace fv2p {

struct P2 { float x, y; };

FV2P_HD float cross2(const P2& a, const P2& b) { return a.x * b.y - a.y * b.x; }
FV2P_HD float cross3(const P2& p1, const P2& p2, const P2& p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
FV2P_HD float fmin2(float a, float b) { return a < b ? a : b; }
FV2P_HD float fmax2(float a, float b) { return a > b ? a : b; }

FV2P_HD int rect_cross(const P2& p1, const P2& p2, const P2& q1, const P2& q2) {
  return fmin2(p1.x, p2.x) <= fmax2(q1.x, q2.x) && fmin2(q1.x, q2.x) <= fmax2(p1.x, p2.x) &&
         fmin2(p1.y, p2.y) <= fmax2(q1.y, q2.y) && fmin2(q1.y, q2.y) <= fmax2(p1.y, p2.y);
}

FV2P_HD int in_box2d(const float* box, const P2& p) {  // iou3d_nms_kernel.cu:51-62
  const float MARGIN = 1e-2f;
  const float cx = box[0], cy = box[1];
  const float ac = fv2p_cosf(-box[6]), as = fv2p_sinf(-box[6]);
  const float rx = (p.x - cx) * ac + (p.y - cy) * (-as);
  const float ry = (p.x - cx) * as + (p.y - cy) * ac;
  return (fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN);
}

FV2P_HD int seg_intersection(const P2& p1, const P2& p0, const P2& q1, const P2& q0, P2* ans) {  // :64-95
  const float EPS = 1e-8f;
  if (rect_cross(p0, p1, q0, q1) == 0) return 0;
  const float s1 = cross3(q0, p1, p0);
  const float s2 = cross3(p1, q1, p0);
  const float s3 = cross3(p0, q1, q0);
  const float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

FV2P_HD void rot_center(const P2& c, float ac, float as, P2* p) {
  const float nx = (p->x - c.x) * ac + (p->y - c.y) * (-as) + c.x;
  const float ny = (p->x - c.x) * as + (p->y - c.y) * ac + c.y;
  p->x = nx; p->y = ny;
}

FV2P_HD float box_overlap(const float* a, const float* b) {  // :104-225
  // Early out, result-preserving: if the centres are farther apart than the two half-diagonals plus the 1e-2 corner
  // margin, the reference code finds no edge crossing and no contained corner and returns exactly 0.
  // (ra + rb + m)^2 <= 3 (ra^2 + rb^2 + m^2), m = 0.1, keeps the test sqrt-free and conservative.
  {
    const float ddx = a[0] - b[0], ddy = a[1] - b[1];
    const float ra2 = 0.25f * (a[3] * a[3] + a[4] * a[4]), rb2 = 0.25f * (b[3] * b[3] + b[4] * b[4]);
    if (ddx * ddx + ddy * ddy > 3.0f * (ra2 + rb2 + 0.01f)) return 0.f;
  }
  const float a_angle = a[6], b_angle = b[6];
  const float a_dx = a[3] / 2, b_dx = b[3] / 2, a_dy = a[4] / 2, b_dy = b[4] / 2;
  const float ax1 = a[0] - a_dx, ay1 = a[1] - a_dy, ax2 = a[0] + a_dx, ay2 = a[1] + a_dy;
  const float bx1 = b[0] - b_dx, by1 = b[1] - b_dy, bx2 = b[0] + b_dx, by2 = b[1] + b_dy;
  const P2 ca = {a[0], a[1]}, cb = {b[0], b[1]};
  P2 A[5] = {{ax1, ay1}, {ax2, ay1}, {ax2, ay2}, {ax1, ay2}, {0, 0}};
  P2 B[5] = {{bx1, by1}, {bx2, by1}, {bx2, by2}, {bx1, by2}, {0, 0}};
  const float acs = fv2p_cosf(a_angle), asn = fv2p_sinf(a_angle);
  const float bcs = fv2p_cosf(b_angle), bsn = fv2p_sinf(b_angle);
  for (int k = 0; k < 4; ++k) {
    rot_center(ca, acs, asn, &A[k]);
    rot_center(cb, bcs, bsn, &B[k]);
  }
  A[4] = A[0];
  B[4] = B[0];
  P2 pts[16];
  P2 center = {0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (seg_intersection(A[i + 1], A[i], B[j + 1], B[j], &pts[cnt])) {
        center.x = center.x + pts[cnt].x;
        center.y = center.y + pts[cnt].y;
        ++cnt;
      }
  for (int k = 0; k < 4; ++k) {
    if (in_box2d(a, B[k])) {
      center.x = center.x + B[k].x; center.y = center.y + B[k].y;
      pts[cnt++] = B[k];
    }
    if (in_box2d(b, A[k])) {
      center.x = center.x + A[k].x; center.y = center.y + A[k].y;
      pts[cnt++] = A[k];
    }
  }
  if (cnt == 0) return 0.f;  // reference divides by zero here and then sums an empty polygon: 0
  center.x /= cnt;
  center.y /= cnt;
  float ang[16];
  for (int i = 0; i < cnt; ++i) ang[i] = fv2p_atan2f(pts[i].y - center.y, pts[i].x - center.x);
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        const P2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
        const float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    const P2 u = {pts[k].x - pts[0].x, pts[k].y - pts[0].y};
    const P2 v = {pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y};
    area += cross2(u, v);
  }
  return fabsf(area) / 2.0f;
}

FV2P_HD float iou_bev(const float* a, const float* b) {  // :227-234
  const float sa = a[3] * a[4], sb = b[3] * b[4];
  const float s = box_overlap(a, b);
  return s / fmax2(sa + sb - s, 1e-8f);
}

FV2P_HD float iou_normal(const float* a, const float* b) {  // :314-325
  const float left = fmax2(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fmin2(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmax2(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fmin2(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float width = fmax2(right - left, 0.f), height = fmax2(bottom - top, 0.f);
  const float inter = width * height;
  return inter / fmax2(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

template <int MODE>  // 0 overlap, 1 iou
__global__ void pairwise_bev(int na, const float* __restrict__ A, int nb, const float* __restrict__ B, float* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<int64_t>(na) * nb) return;
  const int ia = static_cast<int>(t / nb), ib = static_cast<int>(t % nb);
  float a[7], b[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) { a[j] = A[ia * 7 + j]; b[j] = B[ib * 7 + j]; }
  out[t] = MODE == 0 ? box_overlap(a, b) : iou_bev(a, b);
}

// 3-D IoU of every box of A with every box of B, per sample of a batch (boxes_iou3d_gpu, iou3d_nms_utils.py:454-491, whose
// torch composition this follows operation for operation: BEV overlap x height overlap / clamped union, result clamped to [0, 1])
__global__ void pairwise_iou3d_batch(int na, const float* __restrict__ A, int nb, const float* __restrict__ B, int b_stride,
                                     float* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<int64_t>(na) * nb) return;
  const int s = blockIdx.y, ia = static_cast<int>(t / nb), ib = static_cast<int>(t % nb);
  float a[7], b[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    a[j] = A[(static_cast<int64_t>(s) * na + ia) * 7 + j];
    b[j] = B[(static_cast<int64_t>(s) * nb + ib) * b_stride + j];
  }
  const float a_max = a[2] + a[5] / 2, a_min = a[2] - a[5] / 2, b_max = b[2] + b[5] / 2, b_min = b[2] - b[5] / 2;
  const float oh = fmaxf(fminf(a_max, b_max) - fmaxf(a_min, b_min), 0.f);
  const float o3 = box_overlap(a, b) * oh;
  const float vol_a = a[3] * a[4] * a[5], vol_b = b[3] * b[4] * b[5];
  float iou = o3 / fmaxf(vol_a + vol_b - o3, 1e-6f);
  iou = iou < 0.f ? 0.f : iou;
  iou = iou > 1.f ? 1.f : iou;
  out[static_cast<int64_t>(s) * na * nb + t] = iou;
}

// Rotated boxes: a pair whose bounding circles are apart cannot overlap, and at proposal-layer densities ~97 % of the pairs of a
// 64 x 64 tile are such pairs.  A lane's row box meets the tile's column boxes one by one, so skipping per lane would not help (some
// lane always has a near partner); instead every lane first collects its near columns as a bit mask (a few flops per pair), the
// tile's candidates are numbered by a wave prefix sum, and the ~1000-instruction overlap is evaluated for 64 candidates at a time,
// whatever rows they belong to.  The suppression bits are exactly those of the plain loop (the pre-test only drops pairs whose
// overlap is zero).
__device__ __forceinline__ bool circles_apart(const float* a, const float* b) {
  const float dx = a[0] - b[0], dy = a[1] - b[1];
  const float ra = 0.5f * sqrtf(a[3] * a[3] + a[4] * a[4]), rb = 0.5f * sqrtf(b[3] * b[3] + b[4] * b[4]);
  const float reach = (ra + rb) * 1.001f + 1e-4f;   // margin: a pair is dropped only when it is clearly apart
  return dx * dx + dy * dy > reach * reach;
}

// suppression words of one 64 x 64 tile: word of row box `lane` = the column boxes it suppresses (bit i = column cb * 64 + i)
template <int NORMAL>
__device__ __forceinline__ unsigned long long nms_tile(int n, float thresh, const float* __restrict__ boxes, int rb, int cb) {
  __shared__ float cbox[64 * 7];
  __shared__ float rbox[64 * 7];
  __shared__ unsigned long long s_cand[64], s_bits[64];
  __shared__ int s_ex[65];
  const int lane = threadIdx.x;
  const int col_size = min(n - cb * 64, 64), row_size = min(n - rb * 64, 64);
  if (lane < col_size)
    for (int j = 0; j < 7; ++j) cbox[lane * 7 + j] = boxes[(cb * 64 + lane) * 7 + j];
  if (lane < row_size)
    for (int j = 0; j < 7; ++j) rbox[lane * 7 + j] = boxes[(rb * 64 + lane) * 7 + j];
  __syncthreads();
  if (NORMAL) {   // axis-aligned overlap is a handful of flops: no pre-test
    unsigned long long bits = 0ull;
    if (lane < row_size) {
      const int start = (rb == cb) ? lane + 1 : 0;
      for (int i = start; i < col_size; ++i)
        if (iou_normal(rbox + lane * 7, cbox + i * 7) > thresh) bits |= 1ull << i;
    }
    return bits;
  }
  unsigned long long cand = 0ull;
  if (lane < row_size) {
    const int start = (rb == cb) ? lane + 1 : 0;
    for (int i = start; i < col_size; ++i)
      if (!circles_apart(rbox + lane * 7, cbox + i * 7)) cand |= 1ull << i;
  }
  // exclusive prefix sum of the candidate counts over the 64 lanes
  const int cnt = __popcll(cand);
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  s_ex[lane] = incl - cnt;
  if (lane == 63) s_ex[64] = incl;
  s_cand[lane] = cand;
  s_bits[lane] = 0ull;
  __syncthreads();
  const int total = s_ex[64];
  for (int base = 0; base < total; base += 64) {
    const int k = base + lane;
    if (k < total) {
      int r = 0;   // the row whose candidates contain number k: largest r with s_ex[r] <= k
#pragma unroll
      for (int step = 32; step >= 1; step >>= 1)
        if (r + step < 64 && s_ex[r + step] <= k) r += step;
      int j = k - s_ex[r];
      unsigned long long m = s_cand[r];
      for (; j > 0; --j) m &= m - 1;   // drop the j lowest candidates
      const int c = __builtin_ctzll(m);
      if (iou_bev(rbox + r * 7, cbox + c * 7) > thresh) atomicOr(&s_bits[r], 1ull << c);
    }
  }
  __syncthreads();
  return s_bits[lane];
}

// one wave per upper-triangular 64x64 tile
template <int NORMAL>
__global__ __launch_bounds__(64) void nms_mask(int n, float thresh, const float* __restrict__ boxes, int col_blocks,
                                               unsigned long long* __restrict__ mask) {
  // linear tile id -> (row_blk, col_blk) with col_blk >= row_blk
  int t = blockIdx.x, rb = 0, rem = col_blocks;
  while (t >= rem) { t -= rem; --rem; ++rb; }
  const int cb = rb + t;
  const unsigned long long bits = nms_tile<NORMAL>(n, thresh, boxes, rb, cb);
  const int row = rb * 64 + static_cast<int>(threadIdx.x);
  if (row < n) mask[static_cast<int64_t>(row) * col_blocks + cb] = bits;
}

// Greedy pass of iou3d_nms.cpp:121-135 on the device: one workgroup, remv[] in LDS.
// Per 64-box block: (1) wave 0 resolves the block's own 64 candidates on the scalar unit — the diagonal mask words are
// pulled out of the lanes with v_readlane (compile-time lane index, fully unrolled), so the inherently serial
// "keep t unless an earlier kept box suppresses it" chain costs a few SALU ops per box and no LDS round trips;
// (2) all 256 threads OR the kept rows' mask words into remv[] with independent loads + LDS atomic OR.
__device__ __forceinline__ unsigned long long lane_word(unsigned long long v, int t) {
  const unsigned lo = __builtin_amdgcn_readlane(static_cast<int>(v & 0xffffffffull), t);
  const unsigned hi = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), t);
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

__global__ __launch_bounds__(256) void nms_greedy(int n, int col_blocks, const unsigned long long* __restrict__ mask,
                                                  long long* __restrict__ keep, int* __restrict__ num_keep) {
  extern __shared__ unsigned long long remv[];  // [col_blocks]
  __shared__ unsigned long long s_kept;
  __shared__ int s_count;
  __shared__ int s_list[64];  // lanes (rows of this block) that survived, in order
  for (int j = threadIdx.x; j < col_blocks; j += 256) remv[j] = 0ull;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  for (int b = 0; b < col_blocks; ++b) {
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      const int row = b * 64 + lane;
      const unsigned long long diag = (row < n) ? mask[static_cast<int64_t>(row) * col_blocks + b] : 0ull;
      const int lim = min(64, n - b * 64);
      unsigned long long cur = remv[b];
      if (lim < 64) cur |= ~0ull << lim;  // rows past the end are never kept
      unsigned long long kept = 0ull;
#pragma unroll
      for (int t = 0; t < 64; ++t) {
        const unsigned long long d = lane_word(diag, t);
        if (!((cur >> t) & 1ull)) { kept |= 1ull << t; cur |= d; }
      }
      if (lane == 0) s_kept = kept;
      const int base = s_count;
      if ((kept >> lane) & 1ull) {
        const int rank = __popcll(kept & ((1ull << lane) - 1ull));
        keep[base + rank] = row;
        s_list[rank] = lane;
      }
    }
    __syncthreads();
    const unsigned long long kept = s_kept;
    if (threadIdx.x == 0) s_count += __popcll(kept);
    const int ncols = col_blocks - (b + 1);
    if (kept && ncols > 0) {
      // flatten (kept row, column) pairs over the workgroup: every load is independent of the others
      const int nk = __popcll(kept);
      for (int e = threadIdx.x; e < nk * ncols; e += 256) {
        const int ki = e / ncols, j = b + 1 + e % ncols;
        const int t = s_list[ki];
        const unsigned long long v = mask[static_cast<int64_t>(b * 64 + t) * col_blocks + j];
        if (v) atomicOr(&remv[j], v);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *num_keep = s_count;
}

// ---- batched, truncated NMS (proposal layers keep only the first `max_keep` survivors) --------------------------------
// The first K survivors of the greedy pass depend only on the rows up to the K-th survivor, so the suppression mask is
// produced in growing chunks of row blocks and each chunk's kernels return at once when their sample already holds K
// survivors (flag in the workspace): no host round trip, and with a high threshold (few suppressions) a 9000-box
// proposal set touches 16 of its 141 row blocks.  grid (col_blocks, chunk row blocks, samples).
template <int NORMAL>
__global__ __launch_bounds__(64) void nms_mask_rows(int n, float thresh, const float* __restrict__ boxes_all, int col_blocks, int rb0,
                                                    unsigned long long* __restrict__ mask_all, int chunk_rows, const int* __restrict__ done) {
  const int s = blockIdx.z;
  if (done[s]) return;
  const int cb = blockIdx.x, rb = rb0 + blockIdx.y;
  if (cb < rb || rb >= col_blocks) return;
  const float* boxes = boxes_all + static_cast<int64_t>(s) * n * 7;
  unsigned long long* mask = mask_all + static_cast<int64_t>(s) * chunk_rows * 64 * col_blocks;
  const unsigned long long bits = nms_tile<NORMAL>(n, thresh, boxes, rb, cb);
  const int lane = threadIdx.x;
  if (rb * 64 + lane < n) mask[static_cast<int64_t>(blockIdx.y * 64 + lane) * col_blocks + cb] = bits;
}

// greedy pass over the row blocks [rb0, rb0 + nrb) of one sample per workgroup; remv[] and the survivor count persist in
// the workspace between chunks.  Same per-block scheme as nms_greedy.
__global__ __launch_bounds__(256) void nms_greedy_rows(int n, int col_blocks, int rb0, int nrb, int chunk_rows, int max_keep,
                                                       const unsigned long long* __restrict__ mask_all, unsigned long long* __restrict__ remv_all,
                                                       long long* __restrict__ keep_all, int keep_stride, int* __restrict__ num_keep, int* __restrict__ done) {
  const int s = blockIdx.x;
  if (done[s]) return;
  extern __shared__ unsigned long long remv[];
  __shared__ unsigned long long s_kept;
  __shared__ int s_count;
  __shared__ int s_list[64];
  const unsigned long long* mask = mask_all + static_cast<int64_t>(s) * chunk_rows * 64 * col_blocks;
  unsigned long long* remv_g = remv_all + static_cast<int64_t>(s) * col_blocks;
  long long* keep = keep_all + static_cast<int64_t>(s) * keep_stride;
  for (int j = threadIdx.x; j < col_blocks; j += 256) remv[j] = rb0 ? remv_g[j] : 0ull;
  if (threadIdx.x == 0) s_count = rb0 ? num_keep[s] : 0;
  __syncthreads();
  const int rb1 = min(rb0 + nrb, col_blocks);
  bool full = false;
  for (int b = rb0; b < rb1 && !full; ++b) {
    const int lb = b - rb0;
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      const int row = b * 64 + lane;
      const unsigned long long diag = (row < n) ? mask[static_cast<int64_t>(lb * 64 + lane) * col_blocks + b] : 0ull;
      const int lim = min(64, n - b * 64);
      unsigned long long cur = remv[b];
      if (lim < 64) cur |= ~0ull << lim;
      unsigned long long kept = 0ull;
#pragma unroll
      for (int t = 0; t < 64; ++t) {
        const unsigned long long d = lane_word(diag, t);
        if (!((cur >> t) & 1ull)) { kept |= 1ull << t; cur |= d; }
      }
      if (lane == 0) s_kept = kept;
      const int base = s_count;
      if ((kept >> lane) & 1ull) {
        const int rank = __popcll(kept & ((1ull << lane) - 1ull));
        if (base + rank < keep_stride) keep[base + rank] = row;
        s_list[rank] = lane;
      }
    }
    __syncthreads();
    const unsigned long long kept = s_kept;
    const int total = s_count + __popcll(kept);
    full = max_keep > 0 && total >= max_keep;   // uniform: every thread reads the same LDS words
    const int ncols = col_blocks - (b + 1);
    if (kept && ncols > 0 && !full) {
      const int nk = __popcll(kept);
      for (int e = threadIdx.x; e < nk * ncols; e += 256) {
        const int ki = e / ncols, j = b + 1 + e % ncols;
        const unsigned long long v = mask[static_cast<int64_t>(lb * 64 + s_list[ki]) * col_blocks + j];
        if (v) atomicOr(&remv[j], v);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_count = total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int c = s_count;
    num_keep[s] = (max_keep > 0 && c > max_keep) ? max_keep : c;
    if (full || rb1 >= col_blocks) done[s] = 1;
  }
  if (!full && rb1 < col_blocks)
    for (int j = threadIdx.x; j < col_blocks; j += 256) remv_g[j] = remv[j];
}

}  // namespace fv2p
using namespace fv2p;

static int pairwise(int mode, const float* a, int na, const float* b, int nb, float* out, fv2p_stream_t s) {
  FV2P_REQUIRE(na >= 0 && nb >= 0, FV2P_EINVAL, "boxes_bev: negative box count");
  if (na == 0 || nb == 0) return 0;
  FV2P_REQUIRE(a && b && out, FV2P_EINVAL, "boxes_bev: null pointer");
  const int64_t total = static_cast<int64_t>(na) * nb;
  const dim3 grid(static_cast<unsigned>(ceil_div(total, 256))), block(256);
  if (mode == 0) hipLaunchKernelGGL(pairwise_bev<0>, grid, block, 0, static_cast<hipStream_t>(s), na, a, nb, b, out);
  else hipLaunchKernelGGL(pairwise_bev<1>, grid, block, 0, static_cast<hipStream_t>(s), na, a, nb, b, out);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_boxes_iou3d_batch(const float* boxes_a, int batch, int num_a, const float* boxes_b, int num_b, int b_stride,
                                      float* ans_iou, fv2p_stream_t stream) {
  FV2P_REQUIRE(batch >= 0 && num_a >= 0 && num_b >= 0 && b_stride >= 7 && batch <= 65535, FV2P_EINVAL, "boxes_iou3d_batch: bad sizes");
  if (batch == 0 || num_a == 0 || num_b == 0) return 0;
  FV2P_REQUIRE(boxes_a && boxes_b && ans_iou, FV2P_EINVAL, "boxes_iou3d_batch: null pointer");
  const int64_t total = static_cast<int64_t>(num_a) * num_b;
  hipLaunchKernelGGL(pairwise_iou3d_batch, dim3(static_cast<unsigned>(ceil_div(total, 256)), batch), dim3(256), 0, static_cast<hipStream_t>(stream),
                     num_a, boxes_a, num_b, boxes_b, b_stride, ans_iou);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_boxes_overlap_bev(const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* ans_overlap,
                                      fv2p_stream_t stream) {
  return pairwise(0, boxes_a, num_a, boxes_b, num_b, ans_overlap, stream);
}
extern "C" int fv2p_boxes_iou_bev(const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* ans_iou,
                                  fv2p_stream_t stream) {
  return pairwise(1, boxes_a, num_a, boxes_b, num_b, ans_iou, stream);
}

extern "C" size_t fv2p_nms_ws_bytes(int n) {
  const int64_t cb = ceil_div(n > 0 ? n : 1, 64);
  Sizer s;
  s.take<unsigned long long>(static_cast<size_t>(n > 0 ? n : 1) * cb);
  return s.bytes();
}

extern "C" int fv2p_nms(const float* boxes, int n, float thresh, int normal, int64_t* keep, int* num_keep, void* ws,
                        size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 0 && num_keep, FV2P_EINVAL, "nms: bad arguments");
  FV2P_HIP(hipMemsetAsync(num_keep, 0, sizeof(int), stream));
  if (n == 0) return 0;
  FV2P_REQUIRE(boxes && keep, FV2P_EINVAL, "nms: null pointer");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_nms_ws_bytes(n), FV2P_EWORKSPACE, "nms: workspace too small");
  const int cb = static_cast<int>(ceil_div(n, 64));
  FV2P_REQUIRE(static_cast<size_t>(cb) * 8 <= 60000, FV2P_ELIMIT, "nms: more than 480000 boxes");
  Carver c(ws, ws_bytes);
  unsigned long long* mask = c.take<unsigned long long>(static_cast<size_t>(n) * cb);
  const unsigned tiles = static_cast<unsigned>(static_cast<int64_t>(cb) * (cb + 1) / 2);
  if (normal) hipLaunchKernelGGL(nms_mask<1>, dim3(tiles), dim3(64), 0, stream, n, thresh, boxes, cb, mask);
  else hipLaunchKernelGGL(nms_mask<0>, dim3(tiles), dim3(64), 0, stream, n, thresh, boxes, cb, mask);
  hipLaunchKernelGGL(nms_greedy, dim3(1), dim3(256), cb * sizeof(unsigned long long), stream, n, cb, mask,
                     reinterpret_cast<long long*>(keep), num_keep);
  FV2P_LAUNCH_CHECK();
  return 0;
}

static int nms_chunk_rows(int cb, int max_keep) {   // row blocks whose mask is resident at a time
  if (max_keep <= 0) return cb;
  const int first = (2 * max_keep + 63) / 64;
  int longest = first, covered = first, cur = first;
  while (covered < cb) { cur *= 4; covered += cur; longest = cur; }
  return longest < cb ? longest : cb;
}
extern "C" size_t fv2p_nms_batch_ws_bytes(int batch, int n, int max_keep) {
  const int64_t nn = n > 0 ? n : 1, cb = ceil_div(nn, 64);
  const int64_t bb = batch > 0 ? batch : 1;
  Sizer s;
  s.take<unsigned long long>(static_cast<size_t>(bb) * nms_chunk_rows(static_cast<int>(cb), max_keep) * 64 * cb);
  s.take<unsigned long long>(static_cast<size_t>(bb) * cb);
  s.take<int>(static_cast<size_t>(bb));
  return s.bytes();
}
extern "C" int fv2p_nms_batch(const float* boxes, int batch, int n, float thresh, int normal, int max_keep, int64_t* keep, int keep_stride,
                              int* num_keep, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(batch >= 0 && n >= 0 && (num_keep || !batch), FV2P_EINVAL, "nms_batch: bad arguments");
  if (batch == 0) return 0;
  FV2P_HIP(hipMemsetAsync(num_keep, 0, sizeof(int) * batch, stream));
  if (n == 0) return 0;
  FV2P_REQUIRE(boxes && keep && keep_stride >= (max_keep > 0 ? (max_keep < n ? max_keep : n) : n), FV2P_EINVAL, "nms_batch: keep buffer too short");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_nms_batch_ws_bytes(batch, n, max_keep), FV2P_EWORKSPACE, "nms_batch: workspace too small");
  const int cb = static_cast<int>(ceil_div(n, 64));
  FV2P_REQUIRE(static_cast<size_t>(cb) * 8 <= 60000 && batch <= 65535, FV2P_ELIMIT, "nms_batch: too many boxes or samples");
  const int chunk_rows = nms_chunk_rows(cb, max_keep);
  Carver c(ws, ws_bytes);
  unsigned long long* mask = c.take<unsigned long long>(static_cast<size_t>(batch) * chunk_rows * 64 * cb);
  unsigned long long* remv = c.take<unsigned long long>(static_cast<size_t>(batch) * cb);
  int* done = c.take<int>(static_cast<size_t>(batch));
  FV2P_HIP(hipMemsetAsync(done, 0, sizeof(int) * batch, stream));
  int rb0 = 0, span = max_keep > 0 ? (2 * max_keep + 63) / 64 : cb;
  while (rb0 < cb) {
    const int nrb = span < cb - rb0 ? span : cb - rb0;
    const dim3 grid(cb, nrb, batch);
    if (normal) hipLaunchKernelGGL(nms_mask_rows<1>, grid, dim3(64), 0, stream, n, thresh, boxes, cb, rb0, mask, chunk_rows, done);
    else hipLaunchKernelGGL(nms_mask_rows<0>, grid, dim3(64), 0, stream, n, thresh, boxes, cb, rb0, mask, chunk_rows, done);
    hipLaunchKernelGGL(nms_greedy_rows, dim3(batch), dim3(256), cb * sizeof(unsigned long long), stream, n, cb, rb0, nrb, chunk_rows, max_keep,
                       mask, remv, reinterpret_cast<long long*>(keep), keep_stride, num_keep, done);
    rb0 += nrb;
    span *= 4;
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}

// Host entry point of the reference extension (boxes_iou_bev_cpu, iou3d_cpu.cpp:232-252): plain host pointers.
extern "C" int fv2p_boxes_iou_bev_cpu(const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* ans_iou) {
  FV2P_REQUIRE(num_a >= 0 && num_b >= 0 && (boxes_a || !num_a) && (boxes_b || !num_b) && (ans_iou || !num_a || !num_b), FV2P_EINVAL,
               "boxes_iou_bev_cpu: bad arguments");
  for (int i = 0; i < num_a; ++i)
    for (int j = 0; j < num_b; ++j) ans_iou[static_cast<int64_t>(i) * num_b + j] = iou_bev(boxes_a + i * 7, boxes_b + j * 7);
  return 0;
}
