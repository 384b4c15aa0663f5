// A1 — hashed first-come voxeliser (replaces the numba kernel
// pcdet/datasets/processor/voxel_generator.py:136-207 of the reference).
//
// The reference is a sequential loop over points with a dense [Z,Y,X] int map (360 MB per
// call at KITTI resolution).  Here the map is an open-addressing hash table of packed
// {flat voxel key : 40, first point index : 24} words:
//   1. insert   : atomicMin on the packed word keeps, per voxel, the smallest point index;
//   2. order    : a point is "first" iff it is that minimum; an exclusive scan over the
//                 first-flags gives every voxel its reference id (= order of first touch);
//   3. cut-off  : the first-flagged point whose id == max_voxels is where the reference
//                 loop `break`s (voxel_generator.py:198-199); later points are dropped;
//   4. slots    : kept points are keyed (voxel id, point index) and radix-sorted on the
//                 voxel-id bits (stable, so input order inside a voxel survives); the slot
//                 of a point is its sorted position minus the voxel's start offset.
// All integer outputs are therefore bit-identical to the sequential loop.
#include "common.hpp"
#include <cmath>
#include <cstring>
#include <vector>
#include <type_traits>

namespace fv2p {

struct VoxGeom {
  float vs[3];   // voxel size x,y,z
  float lo[3];   // range min x,y,z
  int grid[3];   // cells x,y,z
};

// scalars[0] = distinct voxels D, [1] = cut-off point index i*, [2] = kept points, [3] = M
__global__ void vox_insert(const float* __restrict__ pts, int64_t n, int ndim, VoxGeom g,
                           uint64_t* __restrict__ table, uint32_t mask, int* __restrict__ slot,
                           int* __restrict__ scalars) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i == 0) { scalars[1] = static_cast<int>(n); }
  if (i >= n) return;
  const float* p = pts + i * ndim;
  int c[3];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    // fp32 subtract then fp32 divide then floor, as voxel_generator.py:188 (no reciprocal, no fma)
    float f = floorf((p[j] - g.lo[j]) / g.vs[j]);
    if (!(f >= 0.0f) || f >= static_cast<float>(g.grid[j])) ok = false;  // NaN -> dropped
    c[j] = static_cast<int>(f);
  }
  if (!ok) { slot[i] = -1; return; }
  const uint64_t key = (static_cast<uint64_t>(c[2]) * g.grid[1] + c[1]) * g.grid[0] + c[0];
  const uint64_t word = slot_pack(key, static_cast<uint32_t>(i));
  uint32_t h = hash_u64(key, mask);
  while (true) {
    unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[h]),
                                       static_cast<unsigned long long>(kEmptySlot),
                                       static_cast<unsigned long long>(word));
    if (old == kEmptySlot) break;
    if (slot_key(old) == key) {
      atomicMin(reinterpret_cast<unsigned long long*>(&table[h]), static_cast<unsigned long long>(word));
      break;
    }
    h = (h + 1) & mask;
  }
  slot[i] = static_cast<int>(h);
}

__global__ void vox_mark_first(int64_t n, const uint64_t* __restrict__ table, const int* __restrict__ slot,
                               int* __restrict__ first) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = slot[i];
  first[i] = (s >= 0 && slot_val(table[s]) == static_cast<uint32_t>(i)) ? 1 : 0;
}

// rank[] holds the exclusive scan of the first-flags.
__global__ void vox_assign(int64_t n, const uint64_t* __restrict__ table, const int* __restrict__ slot,
                           const int* __restrict__ rank, VoxGeom g, int max_voxels, int* __restrict__ coors,
                           int* __restrict__ scalars) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = slot[i];
  if (s < 0) return;
  const uint64_t w = table[s];
  if (slot_val(w) != static_cast<uint32_t>(i)) return;  // not a first point
  const int v = rank[i];
  if (v == max_voxels) scalars[1] = static_cast<int>(i);  // unique writer: the reference's break point
  if (v >= max_voxels) return;
  uint64_t key = slot_key(w);
  const int x = static_cast<int>(key % g.grid[0]); key /= g.grid[0];
  const int y = static_cast<int>(key % g.grid[1]);
  const int z = static_cast<int>(key / g.grid[1]);
  coors[v * 3 + 0] = z; coors[v * 3 + 1] = y; coors[v * 3 + 2] = x;
}

__global__ void vox_words(int64_t n, const uint64_t* __restrict__ table, const int* __restrict__ slot,
                          const int* __restrict__ rank, const int* __restrict__ scalars,
                          uint64_t* __restrict__ words, int* __restrict__ count) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = slot[i];
  uint64_t w = ~0ull;
  if (s >= 0 && i < scalars[1]) {
    const int f = static_cast<int>(slot_val(table[s]));
    const int v = rank[f];
    atomicAdd(&count[v], 1);
    w = slot_pack(static_cast<uint64_t>(v), static_cast<uint32_t>(i));
  }
  words[i] = w;
}

__global__ void vox_finalize_scalars(int max_voxels, int* __restrict__ scalars, int* __restrict__ num_voxels) {
  const int d = scalars[0];
  const int m = d < max_voxels ? d : max_voxels;
  scalars[3] = m;
  *num_voxels = m;
}

// start[] = exclusive scan of count[]; scalars[2] = kept points.
__global__ void vox_fill(int64_t n, const float* __restrict__ pts, int ndim, const uint64_t* __restrict__ words,
                         const int* __restrict__ start, const int* __restrict__ scalars, int max_points,
                         float* __restrict__ voxels, int* __restrict__ num_per_voxel) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n || p >= scalars[2]) return;
  const uint64_t w = words[p];
  const int v = static_cast<int>(slot_key(w));
  const int64_t i = slot_val(w);
  const int pos = static_cast<int>(p) - start[v];
  if (pos == 0) {
    const int cnt = start[v + 1] - start[v];
    num_per_voxel[v] = cnt < max_points ? cnt : max_points;
  }
  if (pos < max_points) {
    float* dst = voxels + (static_cast<int64_t>(v) * max_points + pos) * ndim;
    const float* src = pts + i * ndim;
    for (int j = 0; j < ndim; ++j) dst[j] = src[j];
  }
}

struct VoxWs {
  uint64_t* table; uint32_t cap;
  int* slot; int* rank; int* count; int* scalars;
  uint64_t* words; uint64_t* tmp;
  char* aux; size_t aux_bytes;
};

template <typename C>
static void vox_carve(C& c, int64_t n, int max_voxels, VoxWs* w) {
  const uint32_t cap = next_pow2(static_cast<uint64_t>(n > 512 ? n : 512) * 2);
  size_t aux = radix_sort_ws_bytes(n);
  size_t a2 = scan_ws_bytes(n > max_voxels + 1 ? n : max_voxels + 1);
  if (a2 > aux) aux = a2;
  if constexpr (std::is_same<C, Carver>::value) {
    w->cap = cap;
    w->table = c.template take<uint64_t>(cap);
    w->slot = c.template take<int>(n);
    w->rank = c.template take<int>(n);
    w->count = c.template take<int>(max_voxels + 2);
    w->scalars = c.template take<int>(8);
    w->words = c.template take<uint64_t>(n);
    w->tmp = c.template take<uint64_t>(n);
    w->aux = c.template take<char>(aux);
    w->aux_bytes = aux;
  } else {
    c.template take<uint64_t>(cap);
    c.template take<int>(n);
    c.template take<int>(n);
    c.template take<int>(max_voxels + 2);
    c.template take<int>(8);
    c.template take<uint64_t>(n);
    c.template take<uint64_t>(n);
    c.template take<char>(aux);
  }
}

}  // namespace fv2p

using namespace fv2p;

extern "C" size_t fv2p_points_to_voxel_ws_bytes(int64_t n_points, int max_voxels) {
  if (n_points < 1) n_points = 1;
  if (max_voxels < 1) max_voxels = 1;
  Sizer s;
  vox_carve(s, n_points, max_voxels, static_cast<VoxWs*>(nullptr));
  return s.bytes();
}

extern "C" int fv2p_points_to_voxel(const float* points, int64_t n, int ndim, const float voxel_size[3],
                                    const float range_lo[3], const int grid[3], int max_points, int max_voxels,
                                    float* voxels, int* coors, int* num_points_per_voxel, int* num_voxels,
                                    void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 0 && ndim >= 3 && max_points >= 1 && max_voxels >= 1, FV2P_EINVAL,
               "points_to_voxel: bad sizes n=%lld ndim=%d max_points=%d max_voxels=%d", (long long)n, ndim,
               max_points, max_voxels);
  FV2P_REQUIRE(voxels && coors && num_points_per_voxel && num_voxels && (points || n == 0), FV2P_EINVAL,
               "points_to_voxel: null pointer");
  FV2P_REQUIRE(grid[0] > 0 && grid[1] > 0 && grid[2] > 0, FV2P_EINVAL, "points_to_voxel: empty grid");
  FV2P_REQUIRE(n <= kMaxRows, FV2P_ELIMIT, "points_to_voxel: more than %lld points", (long long)kMaxRows);
  FV2P_REQUIRE(static_cast<int64_t>(grid[0]) * grid[1] * grid[2] <= kMaxKey, FV2P_ELIMIT,
               "points_to_voxel: grid volume exceeds 2^40");
  FV2P_REQUIRE(max_voxels <= kMaxRows, FV2P_ELIMIT, "points_to_voxel: max_voxels too large");
  FillJobs fill;
  fill.add(voxels, sizeof(float) * max_voxels * (size_t)max_points * ndim, 0u);
  fill.add(coors, sizeof(int) * 3 * (size_t)max_voxels, 0u);
  fill.add(num_points_per_voxel, sizeof(int) * (size_t)max_voxels, 0u);
  fill.add(num_voxels, sizeof(int), 0u);
  if (n == 0) return multi_fill(fill, stream);
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_points_to_voxel_ws_bytes(n, max_voxels), FV2P_EWORKSPACE,
               "points_to_voxel: workspace too small");
  Carver c(ws, ws_bytes);
  VoxWs w;
  vox_carve(c, n, max_voxels, &w);
  VoxGeom g;
  for (int j = 0; j < 3; ++j) { g.vs[j] = voxel_size[j]; g.lo[j] = range_lo[j]; g.grid[j] = grid[j]; }

  fill.add(w.table, sizeof(uint64_t) * w.cap, 0xFFFFFFFFu);
  fill.add(w.count, sizeof(int) * (max_voxels + 2), 0u);
  fill.add(w.scalars, sizeof(int) * 8, 0u);
  if (int rc = multi_fill(fill, stream)) return rc;
  const int T = 256;
  const dim3 gridN(static_cast<unsigned>(ceil_div(n, T)));
  hipLaunchKernelGGL(vox_insert, gridN, dim3(T), 0, stream, points, n, ndim, g, w.table, w.cap - 1, w.slot, w.scalars);
  hipLaunchKernelGGL(vox_mark_first, gridN, dim3(T), 0, stream, n, w.table, w.slot, w.rank);
  int rc = exclusive_scan_i32(w.rank, w.rank, n, w.scalars + 0, w.aux, w.aux_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(vox_assign, gridN, dim3(T), 0, stream, n, w.table, w.slot, w.rank, g, max_voxels, coors, w.scalars);
  hipLaunchKernelGGL(vox_finalize_scalars, dim3(1), dim3(1), 0, stream, max_voxels, w.scalars, num_voxels);
  hipLaunchKernelGGL(vox_words, gridN, dim3(T), 0, stream, n, w.table, w.slot, w.rank, w.scalars, w.words, w.count);
  const int vb = bits_for(static_cast<uint64_t>(max_voxels));
  rc = radix_sort_u64(w.words, w.tmp, n, kValBits, kValBits + vb, w.aux, w.aux_bytes, stream);
  if (rc) return rc;
  rc = exclusive_scan_i32(w.count, w.count, max_voxels + 1, w.scalars + 2, w.aux, w.aux_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(vox_fill, gridN, dim3(T), 0, stream, n, points, ndim, w.words, w.count, w.scalars, max_points,
                     voxels, num_points_per_voxel);
  FV2P_LAUNCH_CHECK();
  return 0;
}


// ---- host entry point: the reference's own call site of A1 ---------------------------------------------------------------------------
// VoxelGenerator.generate runs inside forked DataLoader worker processes on numpy arrays (data_processor.py:43-81 -> voxel_generator.py:
// 75-207), where HIP cannot be initialised.  Like fv2p_points_in_boxes_cpu and fv2p_boxes_iou_bev_cpu this is the reference's CPU
// entry point served by the library on the calling thread (host pointers, no HIP call): the same sequential first-come scan, with an
// open-addressing table over the <= max_voxels voxels in place of the reference's dense coor_to_voxelidx grid (360 MB filled per call at
// the KITTI grid, :114).  Device inputs never come here (the Python layer sends CUDA tensors to fv2p_points_to_voxel).
extern "C" int fv2p_points_to_voxel_host(const float* points, int64_t n, int ndim, const float voxel_size[3], const float range_lo[3],
                                         const int grid[3], int max_points, int max_voxels, float* voxels, int* coors,
                                         int* num_points_per_voxel, int* num_voxels) {
  FV2P_REQUIRE(n >= 0 && ndim >= 3 && max_points >= 1 && max_voxels >= 0, FV2P_EINVAL, "points_to_voxel_host: bad sizes");
  FV2P_REQUIRE(num_voxels && (n == 0 || points) && (max_voxels == 0 || (voxels && coors && num_points_per_voxel)), FV2P_EINVAL,
               "points_to_voxel_host: null pointer");
  FV2P_REQUIRE(grid[0] > 0 && grid[1] > 0 && grid[2] > 0, FV2P_EINVAL, "points_to_voxel_host: empty grid");
  std::memset(voxels, 0, sizeof(float) * static_cast<size_t>(max_voxels) * max_points * ndim);
  std::memset(num_points_per_voxel, 0, sizeof(int) * static_cast<size_t>(max_voxels));
  size_t cap = 16;
  while (cap < 2 * static_cast<size_t>(max_voxels) + 2) cap <<= 1;
  std::vector<long long> keys(cap, -1);
  std::vector<int> vals(cap, 0);
  int count = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float* p = points + i * ndim;
    int c[3];
    bool inside = true;
    for (int j = 0; j < 3; ++j) {
      // fp32 subtract, fp32 divide, floor (voxel_generator.py:188); the file is built with -ffp-contract=off
      const float q = std::floor((p[j] - range_lo[j]) / voxel_size[j]);
      if (!(q >= 0.0f) || !(q < static_cast<float>(grid[j]))) { inside = false; break; }   // (:189-191; NaN coordinates are dropped too)
      c[j] = static_cast<int>(q);
    }
    if (!inside) continue;
    const long long key = (static_cast<long long>(c[2]) * grid[1] + c[1]) * grid[0] + c[0];
    size_t h = static_cast<size_t>(static_cast<unsigned long long>(key) * 0x9E3779B97F4A7C15ull) & (cap - 1);
    while (keys[h] != -1 && keys[h] != key) h = (h + 1) & (cap - 1);
    int idx;
    if (keys[h] == key) idx = vals[h];
    else {
      if (count >= max_voxels) break;   // (:198-199) the whole scan stops at the first point that would open voxel max_voxels + 1
      idx = count++;
      keys[h] = key; vals[h] = idx;
      coors[3 * idx + 0] = c[2]; coors[3 * idx + 1] = c[1]; coors[3 * idx + 2] = c[0];   // stored (z, y, x) (:192)
    }
    const int k = num_points_per_voxel[idx];
    if (k < max_points) {   // (:204-206)
      std::memcpy(voxels + (static_cast<size_t>(idx) * max_points + k) * ndim, p, sizeof(float) * ndim);
      num_points_per_voxel[idx] = k + 1;
    }
  }
  *num_voxels = count;
  return 0;
}
