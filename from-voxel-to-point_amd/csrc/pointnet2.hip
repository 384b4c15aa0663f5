// A8-A12 — pointnet2 stack: furthest point sampling, ball / voxel query, grouping, gather, 3-NN, interpolation.
//
// Replaces pointnet2_batch_cuda.* (pcdet/ops/pointnet2/pointnet2_batch/src/{ball_query,group_points,sampling,
// interpolate}_gpu.cu) and pointnet2_stack_cuda.* (pcdet/ops/pointnet2/pointnet2_stack/src/{ball_query,group_points,
// sampling,interpolate,voxel_query}_gpu.cu).  Index outputs reproduce the reference's sequential scan order and
// tie-breaks; distances are fp32 sums in the reference's operand order without fused multiply-add.
//
// Where the reference lets every thread stream the whole point set from global memory, the kernels here stage the
// scanned set through LDS once per workgroup (all queries of a workgroup belong to one sample), and FPS keeps each
// thread's points and running distances in registers for the whole 16 k-round loop (the reference re-reads and
// re-writes `temp` in global memory every round).
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace fv2p {

__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
  // (a-b)^2 summed x, y, z left to right (ball_query_gpu.cu:39, interpolate_gpu.cu:42, sampling_gpu.cu:140)
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return dx * dx + dy * dy + dz * dz;
}

// sample of a stacked row given per-sample counts (the reference's per-thread linear search, ball_query_gpu.cu:27-35)
__device__ __forceinline__ void stack_locate(int row, int B, const int* __restrict__ cnt, int* bs, int* start) {
  int b = 0, acc = cnt[0], st = 0;
  for (int k = 1; k < B; ++k) {
    if (row < acc) break;
    st = acc;
    acc += cnt[k];
    b = k;
  }
  *bs = b;
  *start = st;
}
__device__ __forceinline__ int stack_start(int bs, const int* __restrict__ cnt) {
  int s = 0;
  for (int k = 0; k < bs; ++k) s += cnt[k];
  return s;
}

// ------------------------------------------------------------------ ball query (batch) ------------
constexpr int kTile = 1024;  // points staged per LDS tile (12 KB)

// grid (ceil(M / kBqQueries), B); idx (B,M,nsample) pre-zeroed by the caller (pointnet2_utils.py:218).  A WAVE per query, 64 candidate points
// per step: a ballot gives every hit its position in index order, so "the first nsample points inside the ball, padded with the first
// one" (ball_query_gpu.cu:38-40) comes out exactly, with one coalesced store per step instead of a scattered store per hit - the thread-
// per-query form (a serial scan and up to 2 nsample scattered 4-byte stores per thread) took 79 us at the RoI head's 384 x 216 queries.
constexpr int kBqPerWave = 8;                    // queries a wave takes, one after the other, against every staged tile
constexpr int kBqQueries = 4 * kBqPerWave;       // per workgroup
__global__ __launch_bounds__(256) void ball_query_batch_k(int n, int m, float radius, int nsample, const float* __restrict__ new_xyz,
                                                          const float* __restrict__ xyz, int* __restrict__ idx) {
  __shared__ float tile[kTile * 3];
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q0 = blockIdx.x * kBqQueries + w * kBqPerWave;
  const float r2 = radius * radius;
  float qx[kBqPerWave], qy[kBqPerWave], qz[kBqPerWave];
  int cnt[kBqPerWave], first[kBqPerWave];
#pragma unroll
  for (int u = 0; u < kBqPerWave; ++u) {
    const int q = q0 + u < m ? q0 + u : m - 1;
    const float* p = new_xyz + (static_cast<int64_t>(b) * m + q) * 3;
    qx[u] = p[0]; qy[u] = p[1]; qz[u] = p[2];
    cnt[u] = q0 + u < m ? 0 : nsample;   // queries past the end are "full" from the start
    first[u] = -1;
  }
  for (int base = 0; base < n; base += kTile) {
    const int len = min(kTile, n - base);
    bool open = false;
#pragma unroll
    for (int u = 0; u < kBqPerWave; ++u) open = open || cnt[u] < nsample;
    if (__syncthreads_and(!open)) break;   // (also the barrier in front of the tile's re-use)
    for (int e = threadIdx.x; e < len * 3; e += 256) tile[e] = xyz[(static_cast<int64_t>(b) * n + base) * 3 + e];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kBqPerWave; ++u) {
      int* out = idx + (static_cast<int64_t>(b) * m + q0 + u) * nsample;
      for (int k0 = 0; k0 < len && cnt[u] < nsample; k0 += 64) {   // uniform
        const int k = k0 + lane;
        const bool hit = k < len && sqdist(qx[u], qy[u], qz[u], tile[k * 3], tile[k * 3 + 1], tile[k * 3 + 2]) < r2;
        const uint64_t vote = __ballot(hit);
        if (vote) {
          const int pos = cnt[u] + __popcll(vote & lanemask_lt());
          if (hit && pos < nsample) out[pos] = base + k;
          if (first[u] < 0) first[u] = base + k0 + __ffsll(static_cast<long long>(vote)) - 1;
          cnt[u] += __popcll(vote);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kBqPerWave; ++u)
    if (q0 + u < m && first[u] >= 0 && cnt[u] < nsample) {   // fewer than nsample points in the ball: the rest repeats the first
      int* out = idx + (static_cast<int64_t>(b) * m + q0 + u) * nsample;
      for (int l = cnt[u] + lane; l < nsample; l += 64) out[l] = first[u];
    }
}

// stacked variant: one workgroup handles 256 consecutive queries; queries of different samples in one workgroup are
// handled by staging, per query group, the sample of the workgroup's first query and looping over samples.
__global__ __launch_bounds__(256) void ball_query_stack_k(int B, int M, float radius, int nsample, const float* __restrict__ new_xyz,
                                                          const int* __restrict__ new_cnt, const float* __restrict__ xyz,
                                                          const int* __restrict__ xyz_cnt, int* __restrict__ idx) {
  __shared__ float tile[kTile * 3];
  const int q = blockIdx.x * 256 + threadIdx.x;
  const bool live = q < M;
  int my_bs = -1, tmp;
  float qx = 0, qy = 0, qz = 0;
  if (live) {
    stack_locate(q, B, new_cnt, &my_bs, &tmp);
    qx = new_xyz[q * 3]; qy = new_xyz[q * 3 + 1]; qz = new_xyz[q * 3 + 2];
  }
  int first_bs, last_bs;
  stack_locate(blockIdx.x * 256, B, new_cnt, &first_bs, &tmp);
  stack_locate(min(blockIdx.x * 256 + 255, M - 1), B, new_cnt, &last_bs, &tmp);
  int* out = idx + static_cast<int64_t>(live ? q : 0) * nsample;
  const float r2 = radius * radius;
  int cnt = 0;
  for (int bs = first_bs; bs <= last_bs; ++bs) {
    const int start = stack_start(bs, xyz_cnt), n = xyz_cnt[bs];
    const bool mine = live && my_bs == bs;
    bool done = !mine;
    for (int base = 0; base < n; base += kTile) {
      const int len = min(kTile, n - base);
      __syncthreads();
      for (int e = threadIdx.x; e < len * 3; e += 256) tile[e] = xyz[(static_cast<int64_t>(start) + base) * 3 + e];
      __syncthreads();
      if (__syncthreads_and(done)) break;
      if (!done)
        for (int k = 0; k < len; ++k) {
          const float d2 = sqdist(qx, qy, qz, tile[k * 3], tile[k * 3 + 1], tile[k * 3 + 2]);
          if (d2 < r2) {
            if (cnt == 0)
              for (int l = 0; l < nsample; ++l) out[l] = base + k;
            out[cnt] = base + k;
            if (++cnt >= nsample) { done = true; break; }
          }
        }
    }
  }
  if (live && cnt == 0) out[0] = -1;  // ball_query_gpu.cu:65
}

// voxel_query_gpu.cu:10-89: scan the (2r+1)^3 neighbourhood of the query's voxel in a dense (B,Z,Y,X) index volume
__global__ void voxel_query_stack_k(int M, int R1, int R2, int R3, int nsample, float radius, int z_range, int y_range, int x_range,
                                    const float* __restrict__ new_xyz, const float* __restrict__ xyz, const int* __restrict__ new_coords,
                                    const int* __restrict__ point_indices, int* __restrict__ idx) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= M) return;
  const float qx = new_xyz[q * 3], qy = new_xyz[q * 3 + 1], qz = new_xyz[q * 3 + 2];
  const int bi = new_coords[q * 4], cz = new_coords[q * 4 + 1], cy = new_coords[q * 4 + 2], cx = new_coords[q * 4 + 3];
  int* out = idx + static_cast<int64_t>(q) * nsample;
  const float r2 = radius * radius;
  int cnt = 0;
  for (int dz = -z_range; dz <= z_range; ++dz) {
    const int z = cz + dz;
    if (z < 0 || z >= R1) continue;
    for (int dy = -y_range; dy <= y_range; ++dy) {
      const int y = cy + dy;
      if (y < 0 || y >= R2) continue;
      for (int dx = -x_range; dx <= x_range; ++dx) {
        const int x = cx + dx;
        if (x < 0 || x >= R3) continue;
        const int nb = point_indices[((static_cast<int64_t>(bi) * R1 + z) * R2 + y) * R3 + x];
        if (nb < 0) continue;
        const float d2 = sqdist(xyz[nb * 3], xyz[nb * 3 + 1], xyz[nb * 3 + 2], qx, qy, qz);
        if (d2 > r2) continue;  // accepts equality, unlike ball query (voxel_query_gpu.cu:65)
        if (cnt < nsample) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) out[l] = nb;
          out[cnt++] = nb;
        }
      }
    }
  }
  if (cnt == 0) out[0] = -1;
}

// ------------------------------------------------------------------ grouping / gather -------------
// batch: points (B,C,N), idx (B,npoints,nsample) -> out (B,C,npoints,nsample)
__global__ void group_points_batch_k(int64_t total, int c, int n, int npoints, int nsample, const float* __restrict__ points,
                                     const int* __restrict__ idx, float* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t per = static_cast<int64_t>(npoints) * nsample;
  const int64_t bc = t / per, rem = t % per;
  const int b = static_cast<int>(bc / c);
  out[t] = points[bc * n + idx[b * per + rem]];
}
__global__ void group_points_batch_grad_k(int64_t total, int c, int n, int npoints, int nsample, const float* __restrict__ grad_out,
                                          const int* __restrict__ idx, float* __restrict__ grad_points) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t per = static_cast<int64_t>(npoints) * nsample;
  const int64_t bc = t / per, rem = t % per;
  const int b = static_cast<int>(bc / c);
  atomicAdd(&grad_points[bc * n + idx[b * per + rem]], grad_out[t]);
}
// Same gradient with the scatter kept on chip: a wave owns one (b, c) row of grad_points, accumulates the row's
// npoints*nsample contributions into its LDS slice with ds_add_f32 (neighbouring queries share points: collisions are
// resolved by the LDS atomic unit instead of L2 round trips) and adds the slice to the row once.  grad_out is read
// exactly once, coalesced; the sample's index list comes from L1/L2 (the WAVES rows of a workgroup share it).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void group_points_batch_grad_lds_k(int64_t rows, int c, int n, int per, const float* __restrict__ grad_out,
                                                                            const int* __restrict__ idx, float* __restrict__ grad_points) {
  extern __shared__ float acc_lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * WAVES + w;
  if (row >= rows) return;   // no workgroup barrier below: waves are independent
  float* a = acc_lds + w * n;
  for (int i = lane; i < n; i += 64) a[i] = 0.f;
  const float* g = grad_out + row * per;
  const int* id = idx + (row / c) * per;
  int e = lane;
  for (; e + 192 < per; e += 256) {   // four independent loads in flight per lane
    const int i0 = id[e], i1 = id[e + 64], i2 = id[e + 128], i3 = id[e + 192];
    const float g0 = g[e], g1 = g[e + 64], g2 = g[e + 128], g3 = g[e + 192];
    atomicAdd(&a[i0], g0); atomicAdd(&a[i1], g1); atomicAdd(&a[i2], g2); atomicAdd(&a[i3], g3);
  }
  for (; e < per; e += 64) atomicAdd(&a[id[e]], g[e]);
  float* out = grad_points + row * n;
  for (int i = lane; i < n; i += 64) out[i] += a[i];
}
// gather: points (B,C,N), idx (B,M) -> out (B,C,M)
__global__ void gather_points_k(int64_t total, int c, int n, int m, const float* __restrict__ points, const int* __restrict__ idx,
                                float* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t bc = t / m;
  const int b = static_cast<int>(bc / c);
  out[t] = points[bc * n + idx[static_cast<int64_t>(b) * m + t % m]];
}
__global__ void gather_points_grad_k(int64_t total, int c, int n, int m, const float* __restrict__ grad_out, const int* __restrict__ idx,
                                     float* __restrict__ grad_points) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t bc = t / m;
  const int b = static_cast<int>(bc / c);
  atomicAdd(&grad_points[bc * n + idx[static_cast<int64_t>(b) * m + t % m]], grad_out[t]);
}
// stack: features (N,C), idx (M,nsample) local to the sample -> out (M,C,nsample)
__global__ void group_points_stack_k(int B, int M, int C, int nsample, const float* __restrict__ features, const int* __restrict__ feat_cnt,
                                     const int* __restrict__ idx, const int* __restrict__ idx_cnt, float* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<int64_t>(M) * C * nsample) return;
  const int s = static_cast<int>(t % nsample), ch = static_cast<int>((t / nsample) % C), pt = static_cast<int>(t / nsample / C);
  int bs, tmp;
  stack_locate(pt, B, idx_cnt, &bs, &tmp);
  const int start = stack_start(bs, feat_cnt);
  out[t] = features[(static_cast<int64_t>(start) + idx[static_cast<int64_t>(pt) * nsample + s]) * C + ch];
}
__global__ void group_points_stack_grad_k(int B, int M, int C, int nsample, const float* __restrict__ grad_out, const int* __restrict__ idx,
                                          const int* __restrict__ idx_cnt, const int* __restrict__ feat_cnt, float* __restrict__ grad_features) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<int64_t>(M) * C * nsample) return;
  const int s = static_cast<int>(t % nsample), ch = static_cast<int>((t / nsample) % C), pt = static_cast<int>(t / nsample / C);
  int bs, tmp;
  stack_locate(pt, B, idx_cnt, &bs, &tmp);
  const int start = stack_start(bs, feat_cnt);
  atomicAdd(&grad_features[(static_cast<int64_t>(start) + idx[static_cast<int64_t>(pt) * nsample + s]) * C + ch], grad_out[t]);
}

// ---- wave64 maximum of a 64-bit key without LDS ------------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 (an LDS-crossbar round trip per step; a 64-bit butterfly is a chain of twelve).
// DPP moves data between lanes inside the VALU: xor-1 / xor-2 (quad_perm), row_half_mirror, row_mirror make every lane of a
// 16-lane row hold the row's maximum (max is idempotent, so mirrored partners are as good as xor partners), row_bcast15 /
// row_bcast31 carry it across the rows into lane 63, v_readlane broadcasts it.
// workgroup barrier that orders LDS traffic only (__syncthreads also waits for outstanding global stores: vmcnt(0))
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint64_t dpp_max_step(uint64_t v, uint64_t o) { return o > v ? o : v; }
#define FV2P_DPP_U64(v, ctrl, rmask)                                                                                    \
  ((static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>((v) >> 32), static_cast<int>((v) >> 32), (ctrl), (rmask), 0xf, false))) << 32) | \
   static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), (ctrl), (rmask), 0xf, false)))
__device__ __forceinline__ uint64_t row_max_u64(uint64_t v) {   // every lane: maximum over its 16-lane row
  v = dpp_max_step(v, FV2P_DPP_U64(v, 0xB1, 0xf));    // quad_perm [1,0,3,2]
  v = dpp_max_step(v, FV2P_DPP_U64(v, 0x4E, 0xf));    // quad_perm [2,3,0,1]
  v = dpp_max_step(v, FV2P_DPP_U64(v, 0x141, 0xf));   // row_half_mirror
  v = dpp_max_step(v, FV2P_DPP_U64(v, 0x140, 0xf));   // row_mirror
  return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {  // wave-uniform maximum over the 64 lanes
  v = row_max_u64(v);
  v = dpp_max_step(v, FV2P_DPP_U64(v, 0x142, 0xa));   // row_bcast15 into rows 1 and 3
  v = dpp_max_step(v, FV2P_DPP_U64(v, 0x143, 0xc));   // row_bcast31 into rows 2 and 3
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), 63), lo = __builtin_amdgcn_readlane(static_cast<int>(v), 63);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// ------------------------------------------------------------------ furthest point sampling -------
// One workgroup per sample.  Ownership and tie-break follow sampling_gpu.cu:100-216 exactly: with
// bs = 2^floor(log2 n) (<= 1024) reference threads, thread t owns points t, t+bs, ...; a thread keeps its FIRST
// maximum (strict >); the block reduction's tie-break is reproduced below.  REG: points + running distances live
// in registers (n <= THREADS*PPT).
template <int THREADS, int PPT, bool REG>
__global__ __launch_bounds__(THREADS) void fps_k(int n, int m, int bs, const float* __restrict__ dataset, float* __restrict__ temp,
                                                 int* __restrict__ idxs) {
  if (m <= 0) return;
  constexpr int NW = THREADS / 64;
  __shared__ uint64_t s_key[2][NW];
  __shared__ int s_idx[2][NW];
  int log2bs = 0;
  while ((1 << (log2bs + 1)) <= bs) ++log2bs;
  if (log2bs == 0) log2bs = 1;  // bs == 1: a single owner, priority irrelevant (shift by 31 stays defined)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  dataset += static_cast<int64_t>(b) * n * 3;
  temp += static_cast<int64_t>(b) * n;
  idxs += static_cast<int64_t>(b) * m;
  float px[PPT], py[PPT], pz[PPT], pt[PPT];
  if (REG) {
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int k = tid + j * bs;
      const bool ok = tid < bs && k < n;
      px[j] = ok ? dataset[k * 3] : 0.f;
      py[j] = ok ? dataset[k * 3 + 1] : 0.f;
      pz[j] = ok ? dataset[k * 3 + 2] : 0.f;
      pt[j] = ok ? temp[k] : -2.f;  // never selected: min(d, -2) = -2 < best init -1
    }
  }
  int old = 0;
  if (tid == 0) idxs[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = dataset[old * 3], y1 = dataset[old * 3 + 1], z1 = dataset[old * 3 + 2];
    float best = -1.f;
    int besti = 0;
    if (REG) {
#pragma unroll
      for (int q = 0; q < PPT; ++q) {
        const float d = sqdist(px[q], py[q], pz[q], x1, y1, z1);
        const float d2 = fminf(d, pt[q]);
        pt[q] = d2;
        if (d2 > best) { best = d2; besti = tid + q * bs; }
      }
    } else if (tid < bs) {
      for (int k = tid; k < n; k += bs) {
        const float d = sqdist(dataset[k * 3], dataset[k * 3 + 1], dataset[k * 3 + 2], x1, y1, z1);
        const float d2 = fminf(d, temp[k]);
        temp[k] = d2;
        if (d2 > best) { best = d2; besti = k; }
      }
    }
    // Block argmax with the reference's tie-break.  Its shared-memory tree (sampling_gpu.cu:93-98,150-207) merges slot
    // t+s into slot t and keeps slot t unless the other value is strictly larger, so among equal maxima the survivor
    // is the thread whose index is smallest in BIT-REVERSED order (the s = 1 step prefers even slots, s = 2 then
    // prefers slots = 0 mod 4, ...).  Encode (value, ~bitrev(tid)) in one 64-bit key and take the maximum.
    const uint32_t prio = (tid < bs) ? (__brev(static_cast<uint32_t>(tid)) >> (32 - log2bs)) : 0x7fffffffu;
    const uint32_t vbits = best >= 0.f ? __float_as_uint(best) : 0u;
    uint64_t key = (static_cast<uint64_t>(vbits) << 32) | static_cast<uint64_t>(0xffffffffu - prio);
    const uint64_t wkey = wave_max_u64(key);   // DPP, no LDS round trips
    const int leader = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(__ballot(key == wkey))) - 1);
    const int widx = __builtin_amdgcn_readlane(besti, leader);
    const int buf = j & 1;
    if (lane == 0) { s_key[buf][w] = wkey; s_idx[buf][w] = widx; }
    lds_barrier();   // waves talk through LDS only: the idxs[] store of the previous round need not be drained
    // NW (<= 16) wave keys: one per lane, row reduction, lowest wave holding the maximum wins (= ascending scan with '>')
    const uint64_t mine = lane < NW ? s_key[buf][lane] : 0ull;
    const uint64_t rmax = row_max_u64(mine);
    const uint32_t ghi = __builtin_amdgcn_readfirstlane(static_cast<int>(rmax >> 32)), glo = __builtin_amdgcn_readfirstlane(static_cast<int>(rmax));
    const uint64_t gkey = (static_cast<uint64_t>(ghi) << 32) | glo;
    const int gw = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(__ballot(lane < NW && mine == gkey))) - 1);
    old = s_idx[buf][gw];
    if (tid == 0) idxs[j] = old;
  }
  if (REG) {  // the reference leaves the final running distances in `temp` (caller-visible buffer)
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int k = tid + j * bs;
      if (tid < bs && k < n) temp[k] = pt[j];
    }
  }
}

// ---- bucketed (lazy) furthest point sampling -------------------------------------------------------------------------
// Same result as fps_k, bit for bit, with far less arithmetic per round.  The points of a sample are put in Morton order
// once (radix sort), so that the 64 points a wave holds in one register slot form a compact BUCKET with a bounding box.
// A selected point can lower the running distance of a bucket's points only if its distance to the box is below the
// bucket's current maximum: lb = gap_x^2 + gap_y^2 + gap_z^2 is evaluated with the same IEEE operations, in the same order,
// as sqdist, every one of them monotone, so lb <= sqdist(p, new) for every p in the box and skipping a bucket with
// lb >= max(pt) leaves every min(d, pt) unchanged — the lazy version is exact, not approximate.  Per round a wave
// evaluates lb for its PPT buckets in one 16-lane pass, touches only the buckets that fail the test, re-derives its
// per-lane / per-wave best only when something changed, and the workgroup merges 16 cached wave keys.
// The arg-max key is per POINT: (value bits, priority of the original index k under the reference's ownership
// k -> thread k % bs, slot k / bs and its bit-reversed tree tie-break), so any partition of the points gives the
// reference's winner (sampling_gpu.cu:100-216).
__global__ __launch_bounds__(256) void fps_bbox_k(int n, const float* __restrict__ pts, float* __restrict__ bbox) {
  __shared__ float red[6][256];
  const float* p = pts + static_cast<int64_t>(blockIdx.x) * n * 3;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = threadIdx.x; k < n; k += 256)
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float v = p[k * 3 + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
#pragma unroll
  for (int d = 0; d < 3; ++d) { red[d][threadIdx.x] = lo[d]; red[3 + d][threadIdx.x] = hi[d]; }
  __syncthreads();
  for (int sft = 128; sft > 0; sft >>= 1) {
    if (threadIdx.x < sft)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        red[d][threadIdx.x] = fminf(red[d][threadIdx.x], red[d][threadIdx.x + sft]);
        red[3 + d][threadIdx.x] = fmaxf(red[3 + d][threadIdx.x], red[3 + d][threadIdx.x + sft]);
      }
    __syncthreads();
  }
  if (threadIdx.x < 6) bbox[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}
__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 8 bits -> every third bit
  v &= 0xffu;
  v = (v | (v << 8)) & 0x00f00fu;
  v = (v | (v << 4)) & 0x0c30c3u;
  v = (v | (v << 2)) & 0x249249u;
  return v;
}
// key = {sample : high bits, curve position (24 bits) : bits [24, 48), point index : bits [0, 24)}
// The curve decides how compact the buckets of 64 (256) consecutive points are, i.e. how many of them a new point touches per round
// (the picks do not depend on it: skipping is exact).  Round 1-3: 3-D Morton code with 8 bits per axis, each axis scaled to ITS OWN
// extent - for a LiDAR scene (70 x 80 x 4 m) the z bits are 20 x finer than the x bits and a bucket is a thin, long box; 10 % of the
// buckets of a 16 384-point KITTI cloud had a diagonal above 15 m (the median is 2.5 m) and were touched in almost every round.
// Now: a 2-D Hilbert curve over (x, y) with square cells for clouds flatter than 1 : 4 (no jumps: consecutive cells are neighbours),
// an isotropic 3-D Morton code otherwise.  Simulated on the bench's clouds: 6.7 -> 4.6 touched buckets per round, 1.72 -> 1.34 on the
// busiest of the eight waves (the round's critical path); measured: DESIGN 3.3.
__device__ __forceinline__ uint32_t hilbert2(uint32_t x, uint32_t y) {   // 12-bit cell coordinates -> position on the 4096 x 4096 curve
  uint32_t d = 0;
#pragma unroll
  for (uint32_t s = 2048u; s > 0u; s >>= 1) {
    const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
    d += s * s * ((3u * rx) ^ ry);
    if (ry == 0u) {
      if (rx == 1u) { x = 4095u - x; y = 4095u - y; }
      const uint32_t t = x; x = y; y = t;
    }
  }
  return d;
}
__global__ void fps_keys_k(int b, int n, const float* __restrict__ pts, const float* __restrict__ bbox, uint64_t* __restrict__ keys) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<int64_t>(b) * n) return;
  const int s = static_cast<int>(t / n), k = static_cast<int>(t % n);
  const float* bb = bbox + s * 6;
  const float ex = bb[3] - bb[0], ey = bb[4] - bb[1], ez = bb[5] - bb[2];
  const float big = fmaxf(fmaxf(ex, ey), ez);
  uint64_t curve;
  if (4.f * fminf(fminf(ex, ey), ez) <= big && big > 0.f) {
    // flat: curve over the two long axes (the short one only separates points of one cell, which the bucket box handles)
    const int a0 = (ex <= ey && ex <= ez) ? 1 : 0, a1 = (ez <= ex && ez <= ey) ? (a0 == 0 ? 1 : 2) : 2;   // the two axes that are not the shortest
    const float cell = fmaxf(bb[3 + a0] - bb[a0], bb[3 + a1] - bb[a1]) / 4096.f;
    const int q0 = static_cast<int>((pts[t * 3 + a0] - bb[a0]) / cell), q1 = static_cast<int>((pts[t * 3 + a1] - bb[a1]) / cell);
    curve = hilbert2(static_cast<uint32_t>(q0 < 0 ? 0 : (q0 > 4095 ? 4095 : q0)), static_cast<uint32_t>(q1 < 0 ? 0 : (q1 > 4095 ? 4095 : q1)));
  } else {
    uint32_t q[3];
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) {
      const float f = big > 0.f ? (pts[t * 3 + dd] - bb[dd]) / big : 0.f;   // one cell size for the three axes
      const int v = static_cast<int>(f * 256.f);
      q[dd] = static_cast<uint32_t>(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
    curve = spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2);
  }
  keys[t] = (static_cast<uint64_t>(s) << 48) | ((curve & 0xffffffull) << 24) | static_cast<uint64_t>(k);
}

template <int PPT>
__global__ __launch_bounds__(1024) void fps_bucket_k(int n, int m, int bs, const float* __restrict__ dataset, const uint64_t* __restrict__ keys,
                                                     float* __restrict__ temp, int* __restrict__ idxs) {
  if (m <= 0) return;
  constexpr int NW = 16;
  __shared__ uint64_t s_key[2][NW];
  __shared__ __attribute__((aligned(16))) float s_pt[2][NW][4];   // candidate of each wave: original index (int bits), x, y, z
  int log2bs = 0;
  while ((1 << (log2bs + 1)) <= bs) ++log2bs;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  dataset += static_cast<int64_t>(b) * n * 3;
  keys += static_cast<int64_t>(b) * n;
  temp += static_cast<int64_t>(b) * n;
  idxs += static_cast<int64_t>(b) * m;
  // reference priority of original index k (smaller wins among equal distances): owner thread k % bs in bit-reversed
  // order, then the owner's slot k / bs.  It is a bijection of k, so a slot keeps the priority and the winner's index is
  // recovered from it.
  auto prio_of = [&](int k) -> uint32_t {
    const uint32_t owner = static_cast<uint32_t>(k) & static_cast<uint32_t>(bs - 1);
    return ((__brev(owner) >> (32 - log2bs)) << 16) | static_cast<uint32_t>(k >> log2bs);
  };
  auto index_of = [&](uint32_t pr) -> int {
    return static_cast<int>(((pr & 0xffffu) << log2bs) | (__brev(pr >> 16) >> (32 - log2bs)));
  };
  // A BUCKET is the PPT consecutive Morton-sorted points one thread holds: sorted positions tid * PPT + q.  Box, running
  // maximum and arg-max of a bucket are per-thread values: no cross-lane work on the skip path at all.
  float px[PPT], py[PPT], pz[PPT], pt[PPT];
  uint32_t pr[PPT];   // 0xffffffff: empty slot
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int q = 0; q < PPT; ++q) {
    const int pos = tid * PPT + q;
    const bool ok = pos < n;
    const int k = ok ? static_cast<int>(keys[pos] & 0xffffffull) : 0;
    pr[q] = ok ? prio_of(k) : 0xffffffffu;
    px[q] = ok ? dataset[k * 3] : 0.f;
    py[q] = ok ? dataset[k * 3 + 1] : 0.f;
    pz[q] = ok ? dataset[k * 3 + 2] : 0.f;
    pt[q] = ok ? temp[k] : -2.f;
    if (ok) {
      lo[0] = fminf(lo[0], px[q]); hi[0] = fmaxf(hi[0], px[q]);
      lo[1] = fminf(lo[1], py[q]); hi[1] = fmaxf(hi[1], py[q]);
      lo[2] = fminf(lo[2], pz[q]); hi[2] = fmaxf(hi[2], pz[q]);
    }
  }
  // thread's best point: (running distance, priority, slot)
  float lbest = -1.f;
  uint32_t lprio = 0xffffffffu;
  int lslot = 0;
  // branch-free on purpose ('|' and '&' on the predicates): with '||' / '&&' hipcc emits a divergent branch pair per slot
  auto consider = [&](int q) {   // empty slots hold -2: never better than the initial -1
    const bool better = (pt[q] > lbest) | ((pt[q] == lbest) & (pr[q] < lprio));
    lbest = better ? pt[q] : lbest; lprio = better ? pr[q] : lprio; lslot = better ? q : lslot;
  };
  auto rescan = [&]() {
    lbest = -1.f; lprio = 0xffffffffu; lslot = 0;
#pragma unroll
    for (int q = 0; q < PPT; ++q) consider(q);
  };
  uint64_t wkey = 0;
  float cidx = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;   // wave candidate (lane 0 publishes it)
  auto wave_best = [&]() {
    const uint32_t vbits = lbest >= 0.f ? __float_as_uint(lbest) : 0u;
    const uint64_t key = (static_cast<uint64_t>(vbits) << 32) | static_cast<uint64_t>(0xffffffffu - lprio);
    wkey = wave_max_u64(key);
    const int leader = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(__ballot(key == wkey))) - 1);
    const int slot = __builtin_amdgcn_readlane(lslot, leader);
    cidx = __int_as_float(index_of(0xffffffffu - static_cast<uint32_t>(wkey)));
#pragma unroll
    for (int q = 0; q < PPT; ++q)
      if (q == slot) {   // wave-uniform: one case runs
        cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px[q]), leader));
        cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, py[q]), leader));
        cz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pz[q]), leader));
      }
  };
  rescan();
  wave_best();
  // the first sample is index 0: its coordinates come from memory once
  float x1 = dataset[0], y1 = dataset[1], z1 = dataset[2];
  if (tid == 0) idxs[0] = 0;
  for (int j = 1; j < m; ++j) {
    // can the new point lower any running distance of my bucket?  lb uses sqdist's operations in sqdist's order
    const float gx = fmaxf(fmaxf(lo[0] - x1, x1 - hi[0]), 0.f);
    const float gy = fmaxf(fmaxf(lo[1] - y1, y1 - hi[1]), 0.f);
    const float gz = fmaxf(fmaxf(lo[2] - z1, z1 - hi[2]), 0.f);
    const float lb = gx * gx + gy * gy + gz * gz;
    const bool touch = lb < lbest;   // lbest == max of the bucket's running distances (-1 for an empty bucket)
    if (__ballot(touch)) {
      if (touch) {
        lbest = -1.f; lprio = 0xffffffffu; lslot = 0;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
          const float d = sqdist(px[q], py[q], pz[q], x1, y1, z1);
          pt[q] = fminf(d, pt[q]);
          consider(q);
        }
      }
      wave_best();
    }
    const int buf = j & 1;
    if (lane == 0) {
      s_key[buf][w] = wkey;
      *reinterpret_cast<float4*>(&s_pt[buf][w][0]) = make_float4(cidx, cx, cy, cz);
    }
    lds_barrier();   // the waves talk through LDS only: no need to drain the idxs[] store of the previous round
    // 16 wave keys: lanes 0..15 take one each, a row reduction finds the maximum, the lowest wave holding it wins
    const uint64_t mine = lane < NW ? s_key[buf][lane] : 0ull;
    const uint64_t rmax = row_max_u64(mine);
    const uint32_t ghi = __builtin_amdgcn_readfirstlane(static_cast<int>(rmax >> 32)), glo = __builtin_amdgcn_readfirstlane(static_cast<int>(rmax));
    const uint64_t gkey = (static_cast<uint64_t>(ghi) << 32) | glo;
    const int gw = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(__ballot(lane < NW && mine == gkey))) - 1);
    const float4 c = *reinterpret_cast<const float4*>(&s_pt[buf][gw][0]);
    x1 = c.y; y1 = c.z; z1 = c.w;
    if (tid == 0) idxs[j] = __float_as_int(c.x);
  }
#pragma unroll
  for (int q = 0; q < PPT; ++q)
    if (pr[q] != 0xffffffffu) temp[index_of(pr[q])] = pt[q];
}

// ---- wave-bucket furthest point sampling -------------------------------------------------------------------------------
// Third form of the same sampler (bit-identical picks).  fps_bucket_k gives every THREAD a bucket of PPT Morton-neighbours:
// a touched bucket is a serial PPT-point loop in one lane while 63 lanes idle, and 16 waves repeat the per-round
// bookkeeping on four SIMDs.  Here a bucket is the 64 points one WAVE holds in one register slot (lane = point), eight
// waves (two per SIMD) hold S slots each:
//   * skip test: lanes 0..S-1 of a wave evaluate the box bound of the wave's S buckets at once (one ballot);
//   * a touched bucket updates its 64 points with one pass of VALU ops and re-derives its maximum with a DPP reduction;
//   * arg-max = (float maximum, then smallest reference priority among the points that hold it): two 32-bit reductions,
//     exact for ties (lattice inputs, exhausted clouds) without a 64-bit compare chain;
//   * per round one LDS exchange of 8 wave candidates and one barrier (double buffered).
// Points, running distances and priorities stay in registers for all m rounds (5*S VGPRs: S <= 40, n <= 20480).
// Wave reductions as explicit DPP instruction chains: v_max/min_*_dpp take the permuted operand directly (one instruction per
// step; hipcc's update_dpp + fmaxf lowers to mov, mov_dpp, canonicalise, max).  quad_perm xor-1, xor-2, row_half_mirror and
// row_mirror leave every lane of a 16-lane row with the row's result; row_bcast:15 / :31 carry it into lane 63.  The two wait
// states a DPP read needs after the VALU write of its source are the s_nop 1 in front of every step.
#ifndef FV2P_FPS_PICK_LANES
#define FV2P_FPS_PICK_LANES 0
#endif
#define FV2P_ROW_CHAIN(op)                                                        \
  "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
  "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
  "s_nop 1\n\t" op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
  "s_nop 1\n\t" op " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
#define FV2P_WAVE_CHAIN(op)                                                      \
  FV2P_ROW_CHAIN(op)                                                             \
  "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"       \
  "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"       \
  "s_nop 1"
__device__ __forceinline__ float row_max_f32(float v) { asm volatile(FV2P_ROW_CHAIN("v_max_f32_dpp") "s_nop 1" : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t row_min_u32(uint32_t v) { asm volatile(FV2P_ROW_CHAIN("v_min_u32_dpp") "s_nop 1" : "+v"(v)); return v; }
__device__ __forceinline__ float wave_max_f32(float v) {   // wave-uniform
  asm volatile(FV2P_WAVE_CHAIN("v_max_f32_dpp") : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_min_f32(float v) {
  asm volatile(FV2P_WAVE_CHAIN("v_min_f32_dpp") : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  asm volatile(FV2P_WAVE_CHAIN("v_min_u32_dpp") : "+v"(v));
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}
__device__ __forceinline__ float lane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

constexpr int kFpsWaves = 8;
constexpr bool kFpsPickLanes = FV2P_FPS_PICK_LANES;
constexpr int kFpsIdxDefault = 1;   // touched buckets through run-time register indices (fps_wave_k<S, TRACE, IDX>)
// compile-time loop: f(integral_constant<int, Q>) for Q in [A, B) — register arrays are only ever indexed by constants
template <int A, int B, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (A < B) {
    f(std::integral_constant<int, A>{});
    static_for<A + 1, B>(f);
  }
}
// S register slots of a wave (one float per lane and slot), readable and writable by a RUN-TIME, wave-uniform slot number: hipcc lowers a
// dynamic index into an 8 / 16 / 32-wide vector to s_set_gpr_idx_on + v_mov_b32 (no scratch, no branches); slots 32..47 are a second vector
template <int S>
struct SlotRegs {
  static constexpr int LO = S <= 8 ? 8 : S <= 16 ? 16 : 32;
  static constexpr int HI = S > 32 ? 16 : 2;
  typedef float VLo __attribute__((ext_vector_type(LO)));
  typedef float VHi __attribute__((ext_vector_type(HI)));
  VLo lo;
  VHi hi;
  __device__ __forceinline__ float get(int q) const {
    if constexpr (S > 32) { if (q >= 32) return hi[q - 32]; }
    return lo[q];
  }
  __device__ __forceinline__ void set(int q, float v) {
    if constexpr (S > 32) { if (q >= 32) { hi[q - 32] = v; return; } }
    lo[q] = v;
  }
};
// IDX: the touched buckets one after another through ONE copy of the update code, their registers addressed by the run-time slot number
// (SlotRegs), instead of the straight-line scan of compile-time slots (S / 8 + 8 uniform tests per round and S copies of the update)
template <int S, bool TRACE, bool IDX = false>
__global__ __launch_bounds__(kFpsWaves * 64) void fps_wave_k(int n, int m, int bs, const float* __restrict__ dataset,
                                                             const uint64_t* __restrict__ keys, float* __restrict__ temp, int* __restrict__ idxs,
                                                             unsigned long long* __restrict__ trace_) {
  if (m <= 0) return;
  unsigned long long* const trace = TRACE ? trace_ : nullptr;   // compile-time off in the production instance (the runtime test cost 10 %)
  unsigned long long t_test = 0, t_touch = 0, t_best = 0, t_barrier = 0, t_pick = 0, t_mark = 0, n_touched = 0;   // test hook (fv2p_fps_set_trace)
  extern __shared__ uint32_t s_prio[];   // [S][kFpsWaves * 64]: reference priority of every point (read only when its bucket is touched)
  __shared__ __attribute__((aligned(16))) uint32_t s_wave[2][kFpsWaves][4];   // the point of every wave's candidate: x, y, z, -
  // the round's winner: maximum over the waves' keys (distance bits, kPrioTop - priority, wave) by ONE LDS atomic per wave; three slots in
  // rotation - round j fills slot j % 3 before the barrier, everybody reads it after, wave 0 clears slot (j + 1) % 3 (last read in round j - 2)
  __shared__ unsigned long long s_key[3];
  int log2bs = 0;
  while ((1 << (log2bs + 1)) <= bs) ++log2bs;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  dataset += static_cast<int64_t>(b) * n * 3;
  keys += static_cast<int64_t>(b) * n;
  temp += static_cast<int64_t>(b) * n;
  idxs += static_cast<int64_t>(b) * m;
  auto prio_of = [&](int k) -> uint32_t {   // reference tie order of original index k (see fps_bucket_k): smaller wins
    const uint32_t owner = static_cast<uint32_t>(k) & static_cast<uint32_t>(bs - 1);
    return ((__brev(owner) >> (32 - log2bs)) << 16) | static_cast<uint32_t>(k >> log2bs);
  };
  auto index_of = [&](uint32_t pr) -> int { return static_cast<int>(((pr & 0xffffu) << log2bs) | (__brev(pr >> 16) >> (32 - log2bs))); };
  SlotRegs<S> px, py, pz, pt;
  // lane s < S of this wave keeps the state of the wave's bucket s: box, running maximum, priority of the point holding it
  float lo0 = 0.f, lo1 = 0.f, lo2 = 0.f, hi0 = 0.f, hi1 = 0.f, hi2 = 0.f, bmax = -2.f, bcx = 0.f, bcy = 0.f, bcz = 0.f;
  uint32_t bprio = 0xffffffffu;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int pos = (s * kFpsWaves + w) * 64 + lane;   // neighbouring buckets of the Morton order go to different waves
    const bool ok = pos < n;
    const int k = static_cast<int>(keys[ok ? pos : n - 1] & 0xffffffull);   // clamped: no divergent load
    const float x = dataset[k * 3], y = dataset[k * 3 + 1], z = dataset[k * 3 + 2], t = temp[k];
    s_prio[s * (kFpsWaves * 64) + tid] = ok ? prio_of(k) : 0xffffffffu;   // only this thread ever reads it back
    px.set(s, ok ? x : 0.f);
    py.set(s, ok ? y : 0.f);
    pz.set(s, ok ? z : 0.f);
    pt.set(s, ok ? t : -2.f);   // an empty slot never holds a maximum (running distances are >= 0)
    // box of the bucket, kept by lane s
    const float bx0 = wave_min_f32(ok ? x : INFINITY), bx1 = wave_max_f32(ok ? x : -INFINITY);
    const float by0 = wave_min_f32(ok ? y : INFINITY), by1 = wave_max_f32(ok ? y : -INFINITY);
    const float bz0 = wave_min_f32(ok ? z : INFINITY), bz1 = wave_max_f32(ok ? z : -INFINITY);
    if (lane == s) { lo0 = bx0; hi0 = bx1; lo1 = by0; hi1 = by1; lo2 = bz0; hi2 = bz1; }
    __builtin_amdgcn_sched_barrier(0);   // one slot at a time: the S loads are not all hoisted to the top (register pressure)
  }
  // bucket s: recompute (max, priority, candidate coordinates) from the registers; wave-uniform control flow.  The priority
  // reduction only runs when several points share the maximum.
  auto refresh_slot = [&](int s, float x, float y, float z, float t) __attribute__((always_inline)) {   // s wave-uniform
    const uint32_t myp = s_prio[s * (kFpsWaves * 64) + tid];   // issued first: its latency hides behind the max reduction
    const float mx = wave_max_f32(t);
    const uint64_t holders = __ballot(t == mx);
    int leader = __builtin_amdgcn_readfirstlane(__builtin_ctzll(holders))   /* never empty: some lane holds the maximum */;
    uint32_t pm = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(myp), leader));
    if (holders & (holders - 1)) {
      pm = wave_min_u32(t == mx ? myp : 0xffffffffu);
      leader = __builtin_amdgcn_readfirstlane(__builtin_ctzll(__ballot(myp == pm && t == mx)));
    }
    const float cx = lane_f(x, leader), cy = lane_f(y, leader), cz = lane_f(z, leader);
    if (lane == s) { bmax = mx; bprio = pm; bcx = cx; bcy = cy; bcz = cz; }   // the bucket's candidate point travels with its state
  };
  auto refresh = [&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    refresh_slot(s, px.get(s), py.get(s), pz.get(s), pt.get(s));
  };
  static_for<0, S>(refresh);
  // wave candidate from the S bucket states (every stored maximum is exact: a touched bucket is refreshed at once; keeping
  // stale upper bounds and refreshing only the buckets that reach the top was measured slower, 1.44 vs 1.02 us per round)
  uint32_t wbits = 0, wprio = 0xffffffffu, wslot = 0;
  float wcx = 0.f, wcy = 0.f, wcz = 0.f;
  auto wave_best = [&]() __attribute__((always_inline)) {
    const float v = lane < S ? bmax : -3.f;
    const float mx = wave_max_f32(v);
    const uint64_t top = __builtin_amdgcn_ballot_w64(v == mx)   /* lanes >= S hold -3 < mx */;
    int slot = __builtin_amdgcn_readfirstlane(__builtin_ctzll(top));
    uint32_t pm = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(bprio), slot));
    if (top & (top - 1)) {
      pm = wave_min_u32((lane < S && v == mx) ? bprio : 0xffffffffu);
      slot = __builtin_amdgcn_readfirstlane(__builtin_ctzll(__ballot(v == mx && bprio == pm)));
    }
    wslot = static_cast<uint32_t>(slot);
    wbits = __float_as_uint(fmaxf(mx, 0.f));   // every real maximum is >= 0: unsigned order == float order; empty wave -> 0 with priority ~0
    wprio = pm;
    wcx = lane_f(bcx, slot); wcy = lane_f(bcy, slot); wcz = lane_f(bcz, slot);
  };
  wave_best();
  float x1 = dataset[0], y1 = dataset[1], z1 = dataset[2];
  if (tid == 0) { idxs[0] = 0; s_key[0] = 0ull; s_key[1] = 0ull; s_key[2] = 0ull; }
  __syncthreads();
  constexpr uint32_t kPrioTop = 0x03ffffffu;   // priorities are 26 bits (prio_of: 10 bits of owner order above 16 bits of slot)
  const uint32_t key_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&s_key[0]));   // LDS byte address (low half of the flat one)
  uint32_t ks = 8, ks_next = 16;                // byte offsets of slots j % 3 and (j + 1) % 3
  for (int j = 1; j < m; ++j) {
    if (trace) t_mark = __builtin_readcyclecounter();
    // which of my wave's buckets can the new point still lower?  (box bound in sqdist's operation order: exact skip)
    const float gx = fmaxf(fmaxf(lo0 - x1, x1 - hi0), 0.f);
    const float gy = fmaxf(fmaxf(lo1 - y1, y1 - hi1), 0.f);
    const float gz = fmaxf(fmaxf(lo2 - z1, z1 - hi2), 0.f);
    const float lb = gx * gx + gy * gy + gz * gz;
    const uint64_t touch = __builtin_amdgcn_ballot_w64(lb < bmax);   // lanes >= S hold bmax = -2 and lb >= 0: never set
    if (trace) { const unsigned long long t = __builtin_readcyclecounter(); t_test += t - t_mark; t_mark = t; n_touched += __popcll(touch); }
    if (touch) {
      // straight-line, wave-uniform tests (groups of eight first): every slot's update is a plain diamond, so the register
      // arrays are updated in place (a switch over the slot made hipcc keep two copies of pt[] and hoist the distance passes)
      if constexpr (IDX) {
        uint64_t rest = touch;
        do {   // uniform
          const int q = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rest));
          rest &= rest - 1;
          const float x = px.get(q), y = py.get(q), z = pz.get(q);
          const float t = fminf(sqdist(x, y, z, x1, y1, z1), pt.get(q));
          pt.set(q, t);
          refresh_slot(q, x, y, z, t);
        } while (rest);
      } else
      static_for<0, (S + 7) / 8>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value * 8;
        if ((touch >> g) & 0xffull)
          static_for<g, (g + 8 < S ? g + 8 : S)>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            if ((touch >> q) & 1ull) {
              float ax = x1, ay = y1, az = z1;
              asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az));   // pins the distance pass inside its branch
              pt.set(q, fminf(sqdist(px.get(q), py.get(q), pz.get(q), ax, ay, az), pt.get(q)));
              refresh(qc);
            }
          });
      });
      if (trace) { const unsigned long long t = __builtin_readcyclecounter(); t_touch += t - t_mark; t_mark = t; }
      // running distances only fall: while the bucket holding the wave's candidate is untouched, the candidate stands
      // (tried: skipping the refresh of a touched bucket whose maximum holders did not come closer - one readlane + one ballot instead of
      // the reduction - 0.755 -> 0.883 us per round: hipcc then carries a second copy of the distance registers through the diamond)
      if ((touch >> wslot) & 1ull) wave_best();
      if (trace) { const unsigned long long t = __builtin_readcyclecounter(); t_best += t - t_mark; t_mark = t; }
    }
    const int buf = j & 1;
    if (lane == 0) {
      *reinterpret_cast<uint4*>(&s_wave[buf][w][0]) = make_uint4(__float_as_uint(wcx), __float_as_uint(wcy), __float_as_uint(wcz), 0u);
      // larger distance first (every real maximum is >= 0: unsigned order == float order), then the SMALLER reference priority; the wave
      // number below the priority never decides (priorities are unique) and tells where the winner's point lies
      const uint32_t low = ((kPrioTop - (wprio < kPrioTop ? wprio : kPrioTop)) << 3) | static_cast<uint32_t>(w);
      // (written as the instruction: hipcc turns the builtin atomic into a loop over the active lanes with a cross-lane read each)
      const unsigned long long key = (static_cast<unsigned long long>(wbits) << 32) | low;
      asm volatile("ds_max_u64 %0, %1" :: "v"(key_base + ks), "v"(key) : "memory");
      if (w == 0) asm volatile("ds_write_b64 %0, %1" :: "v"(key_base + ks_next), "v"(0ull) : "memory");
    }
    lds_barrier();
    if (trace) { const unsigned long long t = __builtin_readcyclecounter(); t_barrier += t - t_mark; t_mark = t; }
    // every wave reads the same winner: two dependent LDS reads.  (Rounds 3 - 5: the eight candidates one per lane, a DPP row maximum, two
    // ballots and five cross-lane reads - 533 clocks of the round's ~1 900.  Also tried: every lane reads the eight keys and folds them with
    // seven 64-bit maxima - 810 clocks, 0.87 us per round.)
    unsigned long long gkey;
    uint32_t glow;
    if constexpr (kFpsPickLanes) {
      // the eight candidate points one per lane (lane & 7) beside the key: one LDS round trip, then three cross-lane reads
      uint4 c;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(c), "=&v"(gkey)
                   : "v"(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&s_wave[buf][lane & 7][0]))), "v"(key_base + ks) : "memory");
      glow = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(gkey))));
      const int gw = static_cast<int>(glow & 7u);
      x1 = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(c.x), gw)));
      y1 = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(c.y), gw)));
      z1 = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(c.z), gw)));
    } else {
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(gkey) : "v"(key_base + ks) : "memory");
      glow = static_cast<uint32_t>(gkey);
      const uint4 c = *reinterpret_cast<const uint4*>(&s_wave[buf][glow & 7u][0]);
      x1 = __uint_as_float(c.x);
      y1 = __uint_as_float(c.y);
      z1 = __uint_as_float(c.z);
    }
    if (tid == 0) idxs[j] = index_of(kPrioTop - (glow >> 3));
    ks = ks_next;
    ks_next = ks_next == 16 ? 0 : ks_next + 8;
    if (trace) { asm volatile("" : "+v"(x1), "+v"(y1), "+v"(z1)); t_pick += __builtin_readcyclecounter() - t_mark; }
  }
  if (trace && lane == 0 && b == 0) {
    unsigned long long* o = trace + w * 8;
    o[0] = t_test; o[1] = 0; o[2] = t_touch; o[3] = 0; o[4] = t_best; o[5] = t_barrier; o[6] = t_pick; o[7] = n_touched;
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const uint32_t myp = s_prio[s * (kFpsWaves * 64) + tid];
    if (myp != 0xffffffffu) temp[index_of(myp)] = pt.get(s);
  }
}

// ---- streaming furthest point sampling (clouds too large for one CU's registers: n > 24576, e.g. Waymo's ~180 k) ----------
// Same picks again, with the points left in memory: fps_wave_k with its register slots replaced by a Morton-sorted SoA copy
// (x, y, z, running distance, reference priority; every sample padded to whole buckets with points that never hold a maximum)
// that stays in L2.  A bucket is kStreamBucket = 256 consecutive points of that order (<= 1024 buckets: one per thread of the 16
// waves), four per lane: five 16-byte loads fetch it.  Bucket `bk` belongs to lane bk / 16 of wave bk % 16 — Morton neighbours
// go to different waves, so the cluster of buckets a new point touches spreads over all of them.  The owner LANE keeps the
// bucket's box, exact running maximum, the priority and the coordinates of the point that holds it in registers; the owner WAVE
// is the only one that ever reads or writes the bucket's distances, so a round needs no list of touched buckets, no LDS state
// and ONE barrier (the exchange of the 16 wave candidates, double buffered):
//   box test per lane (exact skip: sqdist's operations in sqdist's order) -> ballot -> the wave walks its touched buckets three
//   at a time (all their loads in flight before the first distance pass) -> bucket state back to its lane -> arg-max over the 64
//   lane states -> wave candidate to LDS -> barrier -> every wave derives the same winner.
// Round 2's form compacted the touched buckets into an LDS list between two extra barriers and kept the states in LDS
// (2.1 us/round at n = 180 000); the kernel also claims its CU (>= 97 VGPRs at four waves per SIMD, v127 below): beside a
// training step other workgroups shared the CU and stretched the sampler from 34.8 to 45.7 ms (profiles/README.md, round 2).
__global__ void fps_stream_prep_k(int64_t total, int n, int ns, const float* __restrict__ dataset, const float* __restrict__ temp,
                                  const uint64_t* __restrict__ keys, int bs, float* __restrict__ sx, float* __restrict__ sy, float* __restrict__ sz,
                                  float* __restrict__ sd, uint32_t* __restrict__ sp) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;   // over the padded copies: ns = n rounded up to whole buckets
  if (t >= total) return;
  const int64_t smp = t / ns;
  const int pos = static_cast<int>(t - smp * ns);
  if (pos >= n) {   // padding of the last bucket: a point that never holds a maximum (running distances are >= 0) and never moves
    sx[t] = 0.f; sy[t] = 0.f; sz[t] = 0.f; sd[t] = -2.f; sp[t] = 0xffffffffu;
    return;
  }
  int log2bs = 0;
  while ((1 << (log2bs + 1)) <= bs) ++log2bs;
  const int k = static_cast<int>(keys[smp * n + pos] & 0xffffffull);
  const float* p = dataset + (smp * n + k) * 3;
  sx[t] = p[0]; sy[t] = p[1]; sz[t] = p[2];
  sd[t] = temp[smp * n + k];
  const uint32_t owner = static_cast<uint32_t>(k) & static_cast<uint32_t>(bs - 1);
  sp[t] = ((__brev(owner) >> (32 - log2bs)) << 16) | static_cast<uint32_t>(k >> log2bs);
}
__global__ void fps_stream_post_k(int64_t total, int n, int ns, const uint64_t* __restrict__ keys, const float* __restrict__ sd, float* __restrict__ temp) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t smp = t / n;
  temp[smp * n + static_cast<int>(keys[t] & 0xffffffull)] = sd[smp * ns + (t - smp * n)];
}

constexpr int kStreamWaves = 16;
constexpr int kStreamBucket = 256;                  // points per bucket: four consecutive points per lane, one 16-byte load per array
struct StreamRows { float4 x, y, z, d; uint4 p; };
__global__ __launch_bounds__(kStreamWaves * 64) void fps_stream_k(int n, int m, int bs, const float* __restrict__ dataset,
                                                                  const float* __restrict__ sx_, const float* __restrict__ sy_,
                                                                  const float* __restrict__ sz_, float* __restrict__ sd_,
                                                                  const uint32_t* __restrict__ sp_, int* __restrict__ idxs,
                                                                  unsigned long long* __restrict__ trace) {
  if (m <= 0) return;
  unsigned long long t_test = 0, t_fetch = 0, t_first = 0, t_buckets = 0, t_best = 0, t_barrier = 0, t_pick = 0, t_mark = 0, n_touched = 0;   // test hook: clocks per phase (fv2p_fps_set_trace)
  asm volatile("v_mov_b32 v127, 0" ::: "v127");   // 128 VGPRs per wave x 4 waves per SIMD = the SIMD's register file: no other workgroup joins this CU
  __shared__ __attribute__((aligned(16))) uint32_t s_wave[2][kStreamWaves][4];   // the point of every wave's candidate: x, y, z, -
  __shared__ unsigned long long s_key[3];   // the round's winner by one LDS atomic maximum per wave, three slots in rotation (see fps_wave_k)
  int log2bs = 0;
  while ((1 << (log2bs + 1)) <= bs) ++log2bs;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nb = (n + kStreamBucket - 1) / kStreamBucket;
  const int64_t base = static_cast<int64_t>(b) * nb * kStreamBucket;   // the sorted copies are padded to whole buckets (fps_stream_prep_k)
  const float *sx = sx_ + base, *sy = sy_ + base, *sz = sz_ + base;
  float* sd = sd_ + base;
  const uint32_t* sp = sp_ + base;
  dataset += static_cast<int64_t>(b) * n * 3;
  idxs += static_cast<int64_t>(b) * m;
  auto index_of = [&](uint32_t pr) -> int { return static_cast<int>(((pr & 0xffffu) << log2bs) | (__brev(pr >> 16) >> (32 - log2bs))); };
  // my bucket (lane * 16 + w): box, running maximum (-2: no such bucket), priority and coordinates of the point holding it
  float lo0 = 0.f, lo1 = 0.f, lo2 = 0.f, hi0 = 0.f, hi1 = 0.f, hi2 = 0.f, bmax = -2.f, bcx = 0.f, bcy = 0.f, bcz = 0.f;
  uint32_t bprio = 0xffffffffu;
  // the bucket of lane `l` (wave-uniform): five 16-byte loads per lane, 1 KB per wave instruction
  auto fetch = [&](int l, StreamRows& r) __attribute__((always_inline)) {
    const int p0 = (l * kStreamWaves + w) * kStreamBucket + lane * 4;
    r.x = *reinterpret_cast<const float4*>(sx + p0); r.y = *reinterpret_cast<const float4*>(sy + p0); r.z = *reinterpret_cast<const float4*>(sz + p0);
    r.d = *reinterpret_cast<const float4*>(sd + p0); r.p = *reinterpret_cast<const uint4*>(sp + p0);
  };
  // distance pass (when `update`) and the bucket's exact (max, priority, coordinates) back into lane l's registers
  auto fold = [&](int l, StreamRows& r, bool update, bool boxes, float x1, float y1, float z1) __attribute__((always_inline)) {
    const int p0 = (l * kStreamWaves + w) * kStreamBucket + lane * 4;
    const float xs[4] = {r.x.x, r.x.y, r.x.z, r.x.w}, ys[4] = {r.y.x, r.y.y, r.y.z, r.y.w}, zs[4] = {r.z.x, r.z.y, r.z.z, r.z.w};
    const uint32_t ps[4] = {r.p.x, r.p.y, r.p.z, r.p.w};
    float ds[4] = {r.d.x, r.d.y, r.d.z, r.d.w};
    if (update) {
#pragma unroll
      for (int q = 0; q < 4; ++q) ds[q] = fminf(sqdist(xs[q], ys[q], zs[q], x1, y1, z1), ds[q]);   // padding: min(., -2) stays -2
      *reinterpret_cast<float4*>(sd + p0) = make_float4(ds[0], ds[1], ds[2], ds[3]);
    }
    float bv = -3.f, bx = 0.f, by = 0.f, bz = 0.f;
    uint32_t bp = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool better = (ds[q] > bv) | ((ds[q] == bv) & (ps[q] < bp));
      bv = better ? ds[q] : bv; bp = better ? ps[q] : bp; bx = better ? xs[q] : bx; by = better ? ys[q] : by; bz = better ? zs[q] : bz;
    }
    const float mx = wave_max_f32(bv);
    const uint64_t holders = __ballot(bv == mx);
    int leader = __builtin_amdgcn_readfirstlane(__builtin_ctzll(holders));
    uint32_t pm = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(bp), leader));
    if (holders & (holders - 1)) {
      pm = wave_min_u32(bv == mx ? bp : 0xffffffffu);
      leader = __builtin_amdgcn_readfirstlane(__builtin_ctzll(__ballot(bv == mx && bp == pm)));
    }
    const float cx = lane_f(bx, leader), cy = lane_f(by, leader), cz = lane_f(bz, leader);
    if (lane == l) { bmax = mx; bprio = pm; bcx = cx; bcy = cy; bcz = cz; }
    if (boxes) {
      float a0 = INFINITY, a1 = INFINITY, a2 = INFINITY, c0 = -INFINITY, c1 = -INFINITY, c2 = -INFINITY;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (p0 + q < n) {
          a0 = fminf(a0, xs[q]); c0 = fmaxf(c0, xs[q]); a1 = fminf(a1, ys[q]); c1 = fmaxf(c1, ys[q]); a2 = fminf(a2, zs[q]); c2 = fmaxf(c2, zs[q]);
        }
      a0 = wave_min_f32(a0); a1 = wave_min_f32(a1); a2 = wave_min_f32(a2);
      c0 = wave_max_f32(c0); c1 = wave_max_f32(c1); c2 = wave_max_f32(c2);
      if (lane == l) { lo0 = a0; lo1 = a1; lo2 = a2; hi0 = c0; hi1 = c1; hi2 = c2; }
    }
  };
  for (int l = 0; l * kStreamWaves + w < nb; ++l) {
    StreamRows r;
    fetch(l, r);
    fold(l, r, false, true, 0.f, 0.f, 0.f);
  }
  // wave candidate from the 64 lane states
  uint32_t wbits = 0, wprio = 0xffffffffu;
  int wlane = 0;
  float wcx = 0.f, wcy = 0.f, wcz = 0.f;
  auto wave_best = [&]() __attribute__((always_inline)) {
    const float mx = wave_max_f32(bmax);
    const uint64_t top = __ballot(bmax == mx);
    int l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(top));
    uint32_t pm = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(bprio), l));
    if (top & (top - 1)) {
      pm = wave_min_u32(bmax == mx ? bprio : 0xffffffffu);
      l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(__ballot(bmax == mx && bprio == pm)));
    }
    wlane = l;
    wbits = __float_as_uint(fmaxf(mx, 0.f));   // every real maximum is >= 0: unsigned order == float order; a wave without buckets -> 0 with priority ~0
    wprio = pm;
    wcx = lane_f(bcx, l); wcy = lane_f(bcy, l); wcz = lane_f(bcz, l);
  };
  wave_best();
  float x1 = dataset[0], y1 = dataset[1], z1 = dataset[2];
  if (tid == 0) { idxs[0] = 0; s_key[0] = 0ull; s_key[1] = 0ull; s_key[2] = 0ull; }
  __syncthreads();
  constexpr uint32_t kPrioTop = 0x03ffffffu;   // priorities are 26 bits
  const uint32_t key_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&s_key[0]));
  uint32_t ks = 8, ks_next = 16;                // byte offsets of slots j % 3 and (j + 1) % 3
  for (int j = 1; j < m; ++j) {
    if (trace) t_mark = __builtin_readcyclecounter();
    const float gx = fmaxf(fmaxf(lo0 - x1, x1 - hi0), 0.f);
    const float gy = fmaxf(fmaxf(lo1 - y1, y1 - hi1), 0.f);
    const float gz = fmaxf(fmaxf(lo2 - z1, z1 - hi2), 0.f);
    const float lb = gx * gx + gy * gy + gz * gz;
    uint64_t touch = __ballot(lb < bmax);   // bmax == -2 where the lane has no bucket
    if (trace) { const unsigned long long t = __builtin_readcyclecounter(); t_test += t - t_mark; t_mark = t; n_touched += __popcll(touch); }
    if (touch) {
      const bool mine = (touch >> wlane) & 1ull;
      while (touch) {   // up to three buckets per pass: all their loads in flight before the first distance pass
        const int l0 = __builtin_ctzll(touch);
        touch &= touch - 1;
        const bool two = touch != 0;
        const int l1 = two ? __builtin_ctzll(touch) : l0;
        touch &= touch - 1;   // (0 & ~0 stays 0)
        const bool three = touch != 0;
        const int l2 = three ? __builtin_ctzll(touch) : l0;
        touch &= touch - 1;
        StreamRows ra, rb, rc;
        fetch(l0, ra);
        if (two) fetch(l1, rb);
        if (three) fetch(l2, rc);
        if (trace) { const unsigned long long t = __builtin_readcyclecounter(); t_fetch += t - t_mark; t_mark = t; }
        fold(l0, ra, true, false, x1, y1, z1);
        if (trace) { const unsigned long long t = __builtin_readcyclecounter(); t_first += t - t_mark; t_mark = t; }
        if (two) fold(l1, rb, true, false, x1, y1, z1);
        if (three) fold(l2, rc, true, false, x1, y1, z1);
      }
      if (trace) { const unsigned long long t = __builtin_readcyclecounter(); t_buckets += t - t_mark; t_mark = t; }
      // running distances only fall: while the bucket holding the wave's candidate is untouched, the candidate stands
      if (mine) wave_best();
      if (trace) { const unsigned long long t = __builtin_readcyclecounter(); t_best += t - t_mark; t_mark = t; }
    }
    const int buf = j & 1;
    if (lane == 0) {
      *reinterpret_cast<uint4*>(&s_wave[buf][w][0]) = make_uint4(__float_as_uint(wcx), __float_as_uint(wcy), __float_as_uint(wcz), 0u);
      const uint32_t low = ((kPrioTop - (wprio < kPrioTop ? wprio : kPrioTop)) << 4) | static_cast<uint32_t>(w);
      const unsigned long long key = (static_cast<unsigned long long>(wbits) << 32) | low;
      asm volatile("ds_max_u64 %0, %1" :: "v"(key_base + ks), "v"(key) : "memory");
      if (w == 0) asm volatile("ds_write_b64 %0, %1" :: "v"(key_base + ks_next), "v"(0ull) : "memory");
    }
    lds_barrier();   // the waves talk through LDS only; a bucket's distances are read and written by its own wave alone
    if (trace) { const unsigned long long t = __builtin_readcyclecounter(); t_barrier += t - t_mark; t_mark = t; }
    unsigned long long gkey;   // two dependent LDS reads (rounds 2 - 5: the sixteen candidates reduced again by every wave, 690 clocks)
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(gkey) : "v"(key_base + ks) : "memory");
    const uint32_t glow = static_cast<uint32_t>(gkey);
    const uint4 c = *reinterpret_cast<const uint4*>(&s_wave[buf][glow & 15u][0]);
    x1 = __uint_as_float(c.x);
    y1 = __uint_as_float(c.y);
    z1 = __uint_as_float(c.z);
    if (tid == 0) idxs[j] = index_of(kPrioTop - (glow >> 4));
    ks = ks_next;
    ks_next = ks_next == 16 ? 0 : ks_next + 8;
    if (trace) t_pick += __builtin_readcyclecounter() - t_mark;
  }
  if (trace && lane == 0 && b == 0) {
    unsigned long long* t = trace + w * 8;
    t[0] = t_test; t[1] = t_fetch; t[2] = t_first; t[3] = t_buckets; t[4] = t_best; t[5] = t_barrier; t[6] = t_pick; t[7] = n_touched;
  }
}

// ------------------------------------------------------------------ three_nn / interpolate --------
// one query per thread, known points staged through LDS; strict '<' keeps the lowest index on ties
// (interpolate_gpu.cu:37-55; the reference's double best* hold float values: float compares are identical).
struct Best3 {
  float d1, d2, d3;
  int i1, i2, i3;
};
__device__ __forceinline__ void best3_init(Best3& b) {
  b.d1 = b.d2 = b.d3 = INFINITY;  // 1e40 in the reference (:37) == +inf once cast to float (:56)
  b.i1 = b.i2 = b.i3 = 0;
}
// branch-free form of the reference's if / else-if chain (interpolate_gpu.cu:43-54): strict '<' everywhere
__device__ __forceinline__ void best3_push(Best3& b, float d, int k) {
  const bool c1 = d < b.d1, c2 = d < b.d2, c3 = d < b.d3;
  b.d3 = c2 ? b.d2 : (c3 ? d : b.d3);
  b.i3 = c2 ? b.i2 : (c3 ? k : b.i3);
  b.d2 = c1 ? b.d1 : (c2 ? d : b.d2);
  b.i2 = c1 ? b.i1 : (c2 ? k : b.i2);
  b.d1 = c1 ? d : b.d1;
  b.i1 = c1 ? k : b.i1;
}

// A workgroup = 64 queries x 4 waves; wave w scans the w-th quarter of every LDS tile of known points and the four
// partial top-3 lists are merged on (distance, index), which is exactly "strict < in ascending index order".
// 4x the wave-level parallelism of one-thread-per-query at these sizes (16384 queries = 256 workgroups).
__device__ __forceinline__ void three_nn_scan(const float* __restrict__ known, int64_t start, int m, float ux, float uy, float uz,
                                              bool live, float* tile, Best3& bst) {
  const int w = threadIdx.x >> 6;
  for (int base = 0; base < m; base += kTile) {
    const int len = min(kTile, m - base);
    __syncthreads();
    for (int e = threadIdx.x; e < len * 3; e += 256) tile[e] = known[(start + base) * 3 + e];
    __syncthreads();
    const int q = (len + 3) >> 2;
    const int k0 = w * q, k1 = min(k0 + q, len);
    if (live)
      for (int k = k0; k < k1; ++k) best3_push(bst, sqdist(ux, uy, uz, tile[k * 3], tile[k * 3 + 1], tile[k * 3 + 2]), base + k);
  }
}

__device__ __forceinline__ void best3_merge_one(Best3& b, float d, int k) {
  const bool c1 = d < b.d1 || (d == b.d1 && k < b.i1);
  const bool c2 = d < b.d2 || (d == b.d2 && k < b.i2);
  const bool c3 = d < b.d3 || (d == b.d3 && k < b.i3);
  b.d3 = c2 ? b.d2 : (c3 ? d : b.d3);
  b.i3 = c2 ? b.i2 : (c3 ? k : b.i3);
  b.d2 = c1 ? b.d1 : (c2 ? d : b.d2);
  b.i2 = c1 ? b.i1 : (c2 ? k : b.i2);
  b.d1 = c1 ? d : b.d1;
  b.i1 = c1 ? k : b.i1;
}

__device__ __forceinline__ bool three_nn_merge(Best3& bst, float* sd /*[4][64][3]*/, int* si) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  sd[(w * 64 + lane) * 3] = bst.d1; sd[(w * 64 + lane) * 3 + 1] = bst.d2; sd[(w * 64 + lane) * 3 + 2] = bst.d3;
  si[(w * 64 + lane) * 3] = bst.i1; si[(w * 64 + lane) * 3 + 1] = bst.i2; si[(w * 64 + lane) * 3 + 2] = bst.i3;
  __syncthreads();
  if (w == 0) {
    for (int ww = 1; ww < 4; ++ww)
      for (int j = 0; j < 3; ++j) {
        const float d = sd[(ww * 64 + lane) * 3 + j];
        if (d < INFINITY) best3_merge_one(bst, d, si[(ww * 64 + lane) * 3 + j]);
      }
  }
  return w == 0;
}

__global__ __launch_bounds__(256) void three_nn_batch_k(int n, int m, const float* __restrict__ unknown, const float* __restrict__ known,
                                                        float* __restrict__ dist2, int* __restrict__ idx) {
  __shared__ float tile[kTile * 3];
  __shared__ float sd[4 * 64 * 3];
  __shared__ int si[4 * 64 * 3];
  const int b = blockIdx.y, q = blockIdx.x * 64 + (threadIdx.x & 63);
  const bool live = q < n;
  float ux = 0, uy = 0, uz = 0;
  if (live) {
    const float* p = unknown + (static_cast<int64_t>(b) * n + q) * 3;
    ux = p[0]; uy = p[1]; uz = p[2];
  }
  Best3 bst;
  best3_init(bst);
  three_nn_scan(known, static_cast<int64_t>(b) * m, m, ux, uy, uz, live, tile, bst);
  if (three_nn_merge(bst, sd, si) && live) {
    const int64_t o = (static_cast<int64_t>(b) * n + q) * 3;
    dist2[o] = bst.d1; dist2[o + 1] = bst.d2; dist2[o + 2] = bst.d3;
    idx[o] = bst.i1; idx[o + 1] = bst.i2; idx[o + 2] = bst.i3;
  }
}

__global__ __launch_bounds__(256) void three_nn_stack_k(int B, int N, const float* __restrict__ unknown, const int* __restrict__ unk_cnt,
                                                        const float* __restrict__ known, const int* __restrict__ known_cnt,
                                                        float* __restrict__ dist2, int* __restrict__ idx) {
  __shared__ float tile[kTile * 3];
  __shared__ float sd[4 * 64 * 3];
  __shared__ int si[4 * 64 * 3];
  const int q = blockIdx.x * 64 + (threadIdx.x & 63);
  const bool live = q < N;
  int my_bs = -1, tmp;
  float ux = 0, uy = 0, uz = 0;
  if (live) {
    stack_locate(q, B, unk_cnt, &my_bs, &tmp);
    ux = unknown[q * 3]; uy = unknown[q * 3 + 1]; uz = unknown[q * 3 + 2];
  }
  int first_bs, last_bs;
  stack_locate(blockIdx.x * 64, B, unk_cnt, &first_bs, &tmp);
  stack_locate(min(blockIdx.x * 64 + 63, N - 1), B, unk_cnt, &last_bs, &tmp);
  Best3 bst;
  best3_init(bst);
  int my_start = 0;
  for (int bs = first_bs; bs <= last_bs; ++bs) {
    const int start = stack_start(bs, known_cnt), m = known_cnt[bs];
    const bool mine = live && my_bs == bs;
    if (mine) my_start = start;
    three_nn_scan(known, start, m, ux, uy, uz, mine, tile, bst);
  }
  if (three_nn_merge(bst, sd, si) && live) {
    dist2[q * 3] = bst.d1; dist2[q * 3 + 1] = bst.d2; dist2[q * 3 + 2] = bst.d3;
    idx[q * 3] = bst.i1 + my_start; idx[q * 3 + 1] = bst.i2 + my_start; idx[q * 3 + 2] = bst.i3 + my_start;  // global rows (:72-74)
  }
}

// ---- 3-NN through a hashed uniform grid over the known points ---------------------------------------------------------------------
// The brute-force scan above costs 8 N_u N_k flops whatever the geometry (6.6 % of the vector peak at best, 0.9 ms per level at Waymo
// size).  The decoder's known points are voxel centres: a query's three nearest sit within a cell or two of a grid whose spacing is a
// couple of lattice steps.  Exactness (same idx and dist^2 as the scan, bit for bit): every candidate's distance is sqdist() — the
// scan's float operations in the scan's order — the running best three are kept under the (distance, index) order, which is what
// "strict < in ascending index order" produces, and the search stops only when every unseen point is provably farther than the third
// best: after the block of cells within Chebyshev radius R around the query's cell, an unseen point is at least (R - 1e-3) cells away
// (the 1e-3 covers the float rounding of the cell assignment).  Queries the first block (R = 1) does not settle get a wave each
// (nn_rest_k: shells up to R = 6 dealt to the lanes, then the scan of the sample).  The cells live in an open-addressing hash table keyed by (sample, cell) — 2 slots per known point — so the spacing stays what the
// caller asked for whatever the extent of the cloud (round 3's first form was a dense grid of 2^19 cells per sample: at KITTI extent
// the spacing grew 3.5 x, a cell held ~60 points and the walk was 4 x slower than the scan it replaced).
constexpr unsigned long long kNNEmpty = ~0ull;
struct NNGeo { float lo[3]; float h, inv_h; int dim[3]; };

__device__ __forceinline__ unsigned int nn_hash(unsigned long long k) {   // 64-bit finaliser (splitmix), low bits taken by the caller
  k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27; k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return static_cast<unsigned int>(k);
}
// per-sample bounding box: grid (B, chunks), ordered-int atomics on a float box initialised by nn_box_init_k
__device__ __forceinline__ int nn_ord(float v) { const int i = __float_as_int(v); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float nn_unord(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
__global__ void nn_box_init_k(int B, int* __restrict__ bbox) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < B * 6) bbox[t] = (t % 6) < 3 ? 0x7fffffff : static_cast<int>(0x80000000u);
}
__global__ __launch_bounds__(256) void nn_bbox_k(int B, const float* __restrict__ known, const int* __restrict__ known_cnt, int* __restrict__ bbox) {
  const int bs = blockIdx.x;
  const int start = stack_start(bs, known_cnt), m = known_cnt[bs];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = blockIdx.y * 256 + threadIdx.x; k < m; k += gridDim.y * 256)
    for (int a = 0; a < 3; ++a) { const float v = known[(static_cast<int64_t>(start) + k) * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
  __shared__ float red[6][256];
  for (int a = 0; a < 3; ++a) { red[a][threadIdx.x] = lo[a]; red[3 + a][threadIdx.x] = hi[a]; }
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st)
      for (int a = 0; a < 3; ++a) {
        red[a][threadIdx.x] = fminf(red[a][threadIdx.x], red[a][threadIdx.x + st]);
        red[3 + a][threadIdx.x] = fmaxf(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + st]);
      }
    __syncthreads();
  }
  if (threadIdx.x < 3 && red[threadIdx.x][0] < INFINITY) atomicMin(&bbox[bs * 6 + threadIdx.x], nn_ord(red[threadIdx.x][0]));
  else if (threadIdx.x >= 3 && threadIdx.x < 6 && red[threadIdx.x][0] > -INFINITY) atomicMax(&bbox[bs * 6 + threadIdx.x], nn_ord(red[threadIdx.x][0]));
}
__global__ void nn_setup_k(int B, const int* __restrict__ bbox, const int* __restrict__ known_cnt, float cell, NNGeo* __restrict__ geo) {
  const int bs = blockIdx.x * blockDim.x + threadIdx.x;
  if (bs >= B) return;
  NNGeo g;
  float ext[3];
  const bool any = known_cnt[bs] > 0;
  for (int a = 0; a < 3; ++a) {
    g.lo[a] = any ? nn_unord(bbox[bs * 6 + a]) : 0.f;
    ext[a] = any ? fmaxf(nn_unord(bbox[bs * 6 + 3 + a]) - g.lo[a], 0.f) : 0.f;
    if (!(ext[a] < INFINITY) || !(g.lo[a] > -INFINITY)) { ext[a] = 0.f; g.lo[a] = 0.f; }   // non-finite input: one cell, the walk degenerates to the scan
  }
  const int m = max(known_cnt[bs], 1);
  float h = cell > 0.f ? cell : 1.5f * sqrtf(fmaxf(ext[0] * ext[1], 1e-6f) / m);   // no hint: ~2 points per cell of a surface-like cloud
  h = fmaxf(h, 1e-4f);
  // At most 4096 cells along an axis: the cell coordinate floor((v - lo) * inv_h) carries a rounding error of ~2e-7 x its value, and the
  // searches stop on "(R - 1e-3) cells away" - the margin covers coordinates up to ~4000 (a tiny caller hint on a large extent would
  // otherwise let a query stop with a nearer point just outside its block).  Larger cells only cost time, never exactness.
  for (int it = 0; it < 64; ++it) {
    if (fmaxf(fmaxf(ext[0], ext[1]), ext[2]) / h < 4095.f) break;
    h *= 2.f;
  }
  g.h = h; g.inv_h = 1.f / h;
  for (int a = 0; a < 3; ++a) g.dim[a] = static_cast<int>(floorf(ext[a] / h)) + 1;
  geo[bs] = g;
}
__device__ __forceinline__ int nn_cell_coord(float v, float lo, float inv_h) { return static_cast<int>(floorf((v - lo) * inv_h)); }
__device__ __forceinline__ unsigned long long nn_key(int bs, const NNGeo& g, int cx, int cy, int cz) {
  return (static_cast<unsigned long long>(bs) << 54) | ((static_cast<unsigned long long>(cz) * g.dim[1] + cy) * g.dim[0] + cx);
}
__device__ __forceinline__ unsigned long long nn_key_of(int bs, const NNGeo& g, float x, float y, float z) {
  const int cx = min(max(nn_cell_coord(x, g.lo[0], g.inv_h), 0), g.dim[0] - 1);
  const int cy = min(max(nn_cell_coord(y, g.lo[1], g.inv_h), 0), g.dim[1] - 1);
  const int cz = min(max(nn_cell_coord(z, g.lo[2], g.inv_h), 0), g.dim[2] - 1);
  return nn_key(bs, g, cx, cy, cz);
}
// claims (or finds) the slot of every known point's cell and counts the cell's points; slot_of[k] saves the scatter pass the probing
__global__ __launch_bounds__(256) void nn_insert_k(int B, int M, const float* __restrict__ known, const int* __restrict__ known_cnt,
                                                   const NNGeo* __restrict__ geo, unsigned int mask, unsigned long long* __restrict__ keys,
                                                   int* __restrict__ count, int* __restrict__ slot_of) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= M) return;
  int bs, st;
  stack_locate(k, B, known_cnt, &bs, &st);
  const NNGeo g = geo[bs];
  const unsigned long long key = nn_key_of(bs, g, known[k * 3], known[k * 3 + 1], known[k * 3 + 2]);
  unsigned int s = nn_hash(key) & mask;
  for (;;) {
    const unsigned long long was = atomicCAS(&keys[s], kNNEmpty, key);
    if (was == kNNEmpty || was == key) break;
    s = (s + 1) & mask;
  }
  atomicAdd(&count[s], 1);
  slot_of[k] = static_cast<int>(s);
}
// sorted[pos] = (x, y, z, index within the sample); the order inside a cell is whatever the atomics give — the search orders by (d, index)
__global__ __launch_bounds__(256) void nn_scatter_k(int B, int M, const float* __restrict__ known, const int* __restrict__ known_cnt,
                                                    const int* __restrict__ slot_of, const int* __restrict__ start, int* __restrict__ count,
                                                    float4* __restrict__ sorted) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= M) return;
  int bs, st;
  stack_locate(k, B, known_cnt, &bs, &st);
  const int s = slot_of[k];
  const int pos = start[s] + atomicSub(&count[s], 1) - 1;
  sorted[pos] = make_float4(known[k * 3], known[k * 3 + 1], known[k * 3 + 2], __int_as_float(k - st));
}
__device__ __forceinline__ int nn_find(const unsigned long long* __restrict__ keys, unsigned int mask, unsigned long long key) {
  unsigned int s = nn_hash(key) & mask;
  for (;;) {
    const unsigned long long k = keys[s];
    if (k == key) return static_cast<int>(s);
    if (k == kNNEmpty) return -1;
    s = (s + 1) & mask;
  }
}
// First pass: one query per thread, the 27 cells around its own — all 27 first probes issued together, then the segment bounds of the
// cells that exist, then their points: three rounds of loads that do not wait for each other instead of 27 dependent chains.  A query
// whose third best is provably nearer than anything outside that block is finished; the others leave their three candidates in
// dist2 / idx (local indices) and go on the `todo` list — expanding ring by ring costs ~100 dependent probes for the second ring alone,
// and a thread that does it holds its whole wave back (measured: 2.5 - 4.5 ms at the KITTI levels when stragglers walked six rings).
__global__ __launch_bounds__(256) void nn_query_k(int B, int N, const float* __restrict__ unknown, const int* __restrict__ unk_cnt,
                                                  const int* __restrict__ known_cnt, const NNGeo* __restrict__ geo, unsigned int mask,
                                                  const unsigned long long* __restrict__ keys, const int* __restrict__ start,
                                                  const float4* __restrict__ sorted, float* __restrict__ dist2, int* __restrict__ idx,
                                                  int* __restrict__ todo, int* __restrict__ todo_count) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= N) return;
  int bs, tmp;
  stack_locate(q, B, unk_cnt, &bs, &tmp);
  const NNGeo g = geo[bs];
  const float ux = unknown[q * 3], uy = unknown[q * 3 + 1], uz = unknown[q * 3 + 2];
  const int known_start = stack_start(bs, known_cnt), m = known_cnt[bs];
  // the query's own cell, clamped into int range first (a query far outside the box must not overflow the arithmetic below)
  const int qx = static_cast<int>(fminf(fmaxf(floorf((ux - g.lo[0]) * g.inv_h), -1048576.f), 1048576.f));
  const int qy = static_cast<int>(fminf(fmaxf(floorf((uy - g.lo[1]) * g.inv_h), -1048576.f), 1048576.f));
  const int qz = static_cast<int>(fminf(fmaxf(floorf((uz - g.lo[2]) * g.inv_h), -1048576.f), 1048576.f));
  const int z0 = max(qz - 1, 0), z1 = min(qz + 1, g.dim[2] - 1), y0 = max(qy - 1, 0), y1 = min(qy + 1, g.dim[1] - 1);
  const int x0 = max(qx - 1, 0), x1 = min(qx + 1, g.dim[0] - 1);
  int sl[27];
  unsigned long long want[27], got[27];
#pragma unroll
  for (int c = 0; c < 27; ++c) {
    const int cx = x0 + c % 3, cy = y0 + (c / 3) % 3, cz = z0 + c / 9;
    const bool in = cx <= x1 && cy <= y1 && cz <= z1;
    want[c] = nn_key(bs, g, in ? cx : 0, in ? cy : 0, in ? cz : 0);
    sl[c] = static_cast<int>(nn_hash(want[c]) & mask);
    got[c] = in ? keys[sl[c]] : kNNEmpty;
  }
  int pb[27], pe[27];
#pragma unroll
  for (int c = 0; c < 27; ++c) {
    while (got[c] != want[c] && got[c] != kNNEmpty) { sl[c] = static_cast<int>((static_cast<unsigned int>(sl[c]) + 1u) & mask); got[c] = keys[sl[c]]; }
    const bool hit = got[c] == want[c];
    pb[c] = hit ? start[sl[c]] : 0;
    pe[c] = hit ? start[sl[c] + 1] : 0;
  }
  Best3 bst;
  best3_init(bst);
#pragma unroll
  for (int c = 0; c < 27; ++c)
    for (int p = pb[c]; p < pe[c]; ++p) {
      const float4 v = sorted[p];
      best3_merge_one(bst, sqdist(ux, uy, uz, v.x, v.y, v.z), __float_as_int(v.w));
    }
  const float reach = (1.f - 1e-3f) * g.h;
  const bool done = m == 0 || bst.d3 < reach * reach ||
                    (x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.dim[0] - 1 && y1 == g.dim[1] - 1 && z1 == g.dim[2] - 1);   // the whole grid has been seen
  const int off = done ? known_start : 0;   // unfinished: candidates stay local, the second pass starts from them
  dist2[q * 3] = bst.d1; dist2[q * 3 + 1] = bst.d2; dist2[q * 3 + 2] = bst.d3;
  idx[q * 3] = bst.i1 + off; idx[q * 3 + 1] = bst.i2 + off; idx[q * 3 + 2] = bst.i3 + off;
  if (!done) todo[atomicAdd(todo_count, 1)] = q;
}
// the 64 lanes' lists of three merged under the (distance, index) order: three rounds of "smallest head" (float minimum, then the
// lowest index among its holders); heads are consumed.  An all-inf round yields (inf, 0), the scan's untouched slot.
__device__ __forceinline__ void nn_wave_top3(Best3 b, float (&od)[3], int (&oi)[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float mn = wave_min_f32(b.d1);
    const uint32_t mi = wave_min_u32(b.d1 == mn ? static_cast<uint32_t>(b.i1) : 0xffffffffu);
    od[r] = mn; oi[r] = static_cast<int>(mi);
    if (b.d1 == mn && static_cast<uint32_t>(b.i1) == mi && mn < INFINITY) { b.d1 = b.d2; b.i1 = b.i2; b.d2 = b.d3; b.i2 = b.i3; b.d3 = INFINITY; b.i3 = 0; }
  }
}
constexpr int kNNMaxRing = 6;
// Second pass: one WAVE per unfinished query.  The shells R = 2 .. kNNMaxRing around the query's cell are walked with the cells of a
// shell dealt to the lanes (98, 218, 386 ... cells: two to fourteen probes per lane instead of hundreds per thread); after every shell
// the lanes' candidates are merged and the bound of the header comment is tested.  What six shells do not settle scans the sample:
// lane l takes the known points l, l + 64, ... in ascending index order (strict '<' keeps the lowest index within the lane).
__global__ __launch_bounds__(256) void nn_rest_k(int B, const float* __restrict__ unknown, const int* __restrict__ unk_cnt,
                                                 const float* __restrict__ known, const int* __restrict__ known_cnt, const NNGeo* __restrict__ geo,
                                                 unsigned int mask, const unsigned long long* __restrict__ keys, const int* __restrict__ start,
                                                 const float4* __restrict__ sorted, const int* __restrict__ todo, const int* __restrict__ todo_count,
                                                 float* __restrict__ dist2, int* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int total = *todo_count, waves = gridDim.x * 4;
  for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += waves) {
    const int q = todo[t];
    int bs, tmp;
    stack_locate(q, B, unk_cnt, &bs, &tmp);
    const NNGeo g = geo[bs];
    const float ux = unknown[q * 3], uy = unknown[q * 3 + 1], uz = unknown[q * 3 + 2];
    const int known_start = stack_start(bs, known_cnt), m = known_cnt[bs];
    const int qx = static_cast<int>(fminf(fmaxf(floorf((ux - g.lo[0]) * g.inv_h), -1048576.f), 1048576.f));
    const int qy = static_cast<int>(fminf(fmaxf(floorf((uy - g.lo[1]) * g.inv_h), -1048576.f), 1048576.f));
    const int qz = static_cast<int>(fminf(fmaxf(floorf((uz - g.lo[2]) * g.inv_h), -1048576.f), 1048576.f));
    Best3 bst;
    best3_init(bst);
    if (lane == 0) {   // what the first pass found in the 27 cells (local indices)
      bst.d1 = dist2[q * 3]; bst.d2 = dist2[q * 3 + 1]; bst.d3 = dist2[q * 3 + 2];
      bst.i1 = idx[q * 3]; bst.i2 = idx[q * 3 + 1]; bst.i3 = idx[q * 3 + 2];
    }
    float od[3];
    int oi[3];
    bool done = false;
    for (int R = 2; !done && R <= kNNMaxRing; ++R) {
      // shell R = the (2R+1)^3 block minus the (2R-1)^3 block inside it: the two z faces ((2R+1)^2 cells each), then per inner z layer the
      // two y faces ((2R+1) cells each) and the two x faces ((2R-1) cells each)
      const int side = 2 * R + 1, face = side * side, ring = 4 * side - 4, cells = 2 * face + (side - 2) * ring;
      for (int c = lane; c < cells; c += 64) {
        int dx, dy, dz;
        if (c < 2 * face) {
          dz = c < face ? -R : R;
          const int e = c < face ? c : c - face;
          dy = e / side - R; dx = e % side - R;
        } else {
          const int e = c - 2 * face;
          dz = e / ring - R + 1;
          const int f = e % ring;
          if (f < 2 * side) { dy = f < side ? -R : R; dx = (f < side ? f : f - side) - R; }
          else { const int h2 = f - 2 * side, k = side - 2; dx = h2 < k ? -R : R; dy = (h2 < k ? h2 : h2 - k) - R + 1; }
        }
        const int cx = qx + dx, cy = qy + dy, cz = qz + dz;
        if (cx < 0 || cy < 0 || cz < 0 || cx >= g.dim[0] || cy >= g.dim[1] || cz >= g.dim[2]) continue;
        const int sl = nn_find(keys, mask, nn_key(bs, g, cx, cy, cz));
        if (sl < 0) continue;
        for (int p = start[sl], pe = start[sl + 1]; p < pe; ++p) {
          const float4 v = sorted[p];
          best3_merge_one(bst, sqdist(ux, uy, uz, v.x, v.y, v.z), __float_as_int(v.w));
        }
      }
      nn_wave_top3(bst, od, oi);
      const float reach = (static_cast<float>(R) - 1e-3f) * g.h;
      done = od[2] < reach * reach || (qx - R <= 0 && qy - R <= 0 && qz - R <= 0 && qx + R >= g.dim[0] - 1 && qy + R >= g.dim[1] - 1 && qz + R >= g.dim[2] - 1);
    }
    if (!done) {   // far from everything: the plain scan of this query's sample
      best3_init(bst);
      for (int k = lane; k < m; k += 64) {
        const float* p = known + (static_cast<int64_t>(known_start) + k) * 3;
        best3_push(bst, sqdist(ux, uy, uz, p[0], p[1], p[2]), k);
      }
      nn_wave_top3(bst, od, oi);
    }
    if (lane == 0) {
      dist2[q * 3] = od[0]; dist2[q * 3 + 1] = od[1]; dist2[q * 3 + 2] = od[2];
      idx[q * 3] = oi[0] + known_start; idx[q * 3 + 1] = oi[1] + known_start; idx[q * 3 + 2] = oi[2] + known_start;
    }
  }
}

// batch interpolate: points (B,C,M) channel-major, idx/weight (B,N,3) -> out (B,C,N)
__global__ void three_interp_batch_k(int64_t total, int c, int m, int n, const float* __restrict__ points, const int* __restrict__ idx,
                                     const float* __restrict__ weight, float* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t bc = t / n;
  const int pt = static_cast<int>(t % n), b = static_cast<int>(bc / c);
  const int64_t o = (static_cast<int64_t>(b) * n + pt) * 3;
  const float* p = points + bc * m;
  out[t] = weight[o] * p[idx[o]] + weight[o + 1] * p[idx[o + 1]] + weight[o + 2] * p[idx[o + 2]];
}
__global__ void three_interp_batch_grad_k(int64_t total, int c, int m, int n, const float* __restrict__ grad_out, const int* __restrict__ idx,
                                          const float* __restrict__ weight, float* __restrict__ grad_points) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t bc = t / n;
  const int pt = static_cast<int>(t % n), b = static_cast<int>(bc / c);
  const int64_t o = (static_cast<int64_t>(b) * n + pt) * 3;
  float* g = grad_points + bc * m;
  const float go = grad_out[t];
  atomicAdd(&g[idx[o]], go * weight[o]);
  atomicAdd(&g[idx[o + 1]], go * weight[o + 1]);
  atomicAdd(&g[idx[o + 2]], go * weight[o + 2]);
}
// stack interpolate: features (M,C) row-major, idx/weight (N,3) -> out (N,C)
__global__ void three_interp_stack_k(int64_t total, int c, const float* __restrict__ features, const int* __restrict__ idx,
                                     const float* __restrict__ weight, float* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t pt = t / c;
  const int ch = static_cast<int>(t % c);
  out[t] = weight[pt * 3] * features[static_cast<int64_t>(idx[pt * 3]) * c + ch] +
           weight[pt * 3 + 1] * features[static_cast<int64_t>(idx[pt * 3 + 1]) * c + ch] +
           weight[pt * 3 + 2] * features[static_cast<int64_t>(idx[pt * 3 + 2]) * c + ch];
}
__global__ void three_interp_stack_grad_k(int64_t total, int c, const float* __restrict__ grad_out, const int* __restrict__ idx,
                                          const float* __restrict__ weight, float* __restrict__ grad_features) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t pt = t / c;
  const int ch = static_cast<int>(t % c);
  const float go = grad_out[t];
  atomicAdd(&grad_features[static_cast<int64_t>(idx[pt * 3]) * c + ch], go * weight[pt * 3]);
  atomicAdd(&grad_features[static_cast<int64_t>(idx[pt * 3 + 1]) * c + ch], go * weight[pt * 3 + 1]);
  atomicAdd(&grad_features[static_cast<int64_t>(idx[pt * 3 + 2]) * c + ch], go * weight[pt * 3 + 2]);
}

// Gradient of the stack interpolation WITHOUT float atomics, in a FIXED order (round 4).  The (query, slot) entries e = 3 * query + slot
// are keyed (known row << 32 | e) and radix-sorted (stable, by the row bits): every row's entries are then one contiguous run in ascending
// e, whatever the hardware did.  The sorted sequence is cut into SEGMENTS of kInterpSeg entries, one lane group each (one lane = four
// channels): a run that lies inside one segment is summed and stored directly; a run that crosses segment borders leaves one partial sum
// per segment, and the group of the segment the run STARTS in adds the partials of the following segments in segment order.  The work per
// group is bounded whatever the list lengths (a row read by thousands of queries is hundreds of segments summed in parallel, then one
// short walk over their partials) and the association of every sum is a function of the sorted positions only: bit-identical from run to
// run.  (The first form of this round, one group per ROW walking its run, took 1.3 - 1.8 ms at the decoder's shapes when a few rows
// held thousands of entries; round 3's per-(row, channel) THREAD lists, filled with atomics, summed in an arbitrary order.)
constexpr int kInterpSeg = 32;
__global__ void interp_keys_k(int64_t entries, int m, const int* __restrict__ idx, uint64_t* __restrict__ keys) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= entries) return;
  const int r = idx[e];
  keys[e] = (static_cast<uint64_t>(r >= 0 && r < m ? r : m) << 32) | static_cast<uint64_t>(e);   // out-of-range rows sort behind every row
}
// Segment flags: bit 0 the segment's first run continues a run of the segment before (its sum is part[s][0]); bit 1 that run also
// goes on into the next segment; bit 2 the segment's last run starts here and goes on (its sum is part[s][1]).
struct Vec4 { float v[4]; };
__device__ __forceinline__ void interp_store4(float* out, const Vec4& a, bool vec, int c0, int c) {
  if (vec) *reinterpret_cast<float4*>(out) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
  else {
#pragma unroll
    for (int u = 0; u < 4; ++u) if (c0 + u < c) out[u] = a.v[u];
  }
}
__device__ __forceinline__ Vec4 interp_load4(const float* p, bool vec, int c0, int c) {
  Vec4 r;
  if (vec) { const float4 q = *reinterpret_cast<const float4*>(p); r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w; }
  else {
#pragma unroll
    for (int u = 0; u < 4; ++u) r.v[u] = c0 + u < c ? p[u] : 0.f;
  }
  return r;
}
// TPR lanes per segment (power of two, <= 64), each lane four channels per pass
template <int TPR>
__global__ __launch_bounds__(256) void interp_seg_k(int m, int c, int64_t entries, const uint64_t* __restrict__ keys,
                                                    const float* __restrict__ grad_out, const float* __restrict__ weight,
                                                    float* __restrict__ grad_features, float* __restrict__ part, int* __restrict__ flags,
                                                    int aligned) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t s = t / TPR;
  const int sub = static_cast<int>(t % TPR);
  const int64_t i0 = s * kInterpSeg;
  if (i0 >= entries) return;
  const int64_t i1 = i0 + kInterpSeg < entries ? i0 + kInterpSeg : entries;
  const uint32_t prev = i0 > 0 ? static_cast<uint32_t>(keys[i0 - 1] >> 32) : 0xffffffffu;      // rows are < 2^31: never equal to these two
  const uint32_t next = i1 < entries ? static_cast<uint32_t>(keys[i1] >> 32) : 0xfffffffeu;
  const uint32_t um = static_cast<uint32_t>(m);
  int fl = 0;
  for (int c0 = 4 * sub; c0 < c; c0 += 4 * TPR) {
    const bool vec = aligned && c0 + 3 < c && (c & 3) == 0;
    Vec4 acc = {{0.f, 0.f, 0.f, 0.f}};
    uint32_t cur = static_cast<uint32_t>(keys[i0] >> 32);
    bool at_start = true;
    auto flush = [&](bool at_end) {
      if (cur >= um) return;   // entries whose idx is out of range sort behind every row and count for nothing
      const bool from_prev = at_start && cur == prev, to_next = at_end && cur == next;
      if (!from_prev && !to_next) interp_store4(grad_features + static_cast<int64_t>(cur) * c + c0, acc, vec, c0, c);
      else if (from_prev) { interp_store4(part + (s * 2 + 0) * c + c0, acc, vec, c0, c); fl |= to_next ? 3 : 1; }
      else { interp_store4(part + (s * 2 + 1) * c + c0, acc, vec, c0, c); fl |= 4; }
    };
    for (int64_t i = i0; i < i1; i += 4) {
      // four entries in flight
      uint64_t k[4];
      bool in[4], ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        in[u] = i + u < i1;
        k[u] = in[u] ? keys[i + u] : ~0ull;
        ok[u] = in[u] && static_cast<uint32_t>(k[u] >> 32) < um;
      }
      Vec4 v[4];
      float w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t e = static_cast<uint32_t>(k[u]);
        w[u] = ok[u] ? weight[e] : 0.f;
        v[u] = interp_load4(grad_out + static_cast<int64_t>(ok[u] ? e / 3 : 0) * c + c0, vec, c0, c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (in[u]) {   // ascending e inside a run: the order is part of the result
          const uint32_t row = static_cast<uint32_t>(k[u] >> 32);
          if (row != cur) {
            flush(false);
            cur = row; at_start = false;
            acc = Vec4{{0.f, 0.f, 0.f, 0.f}};
          }
          if (ok[u]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc.v[q] = __builtin_fmaf(w[u], v[u].v[q], acc.v[q]);
          }
        }
    }
    flush(true);
  }
  if (sub == 0) flags[s] = fl;
}
// the runs that cross segment borders: the group of the segment a run starts in adds the following segments' partial sums, in order
template <int TPR>
__global__ __launch_bounds__(256) void interp_fix_k(int c, int64_t entries, int64_t segments, const uint64_t* __restrict__ keys,
                                                    const float* __restrict__ part, const int* __restrict__ flags,
                                                    float* __restrict__ grad_features, int aligned) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t s = t / TPR;
  const int sub = static_cast<int>(t % TPR);
  if (s >= segments || !(flags[s] & 4)) return;
  const int64_t i1 = (s + 1) * kInterpSeg < entries ? (s + 1) * kInterpSeg : entries;
  const int64_t row = static_cast<int64_t>(keys[i1 - 1] >> 32);
  // how far the run goes: segments s + 1 .. s + len (flags only, sixteen in flight; the last segment of the sequence never goes on)
  int64_t len = 0;
  for (bool open = true; open;) {
    int f[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int64_t q = s + 1 + len + u; f[u] = q < segments ? flags[q] : 0; }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (open) { ++len; open = (f[u] & 2) != 0; }
  }
  if (s + len >= segments) len = segments - 1 - s;   // (cannot happen on a consistent flag array)
  for (int c0 = 4 * sub; c0 < c; c0 += 4 * TPR) {
    const bool vec = aligned && c0 + 3 < c && (c & 3) == 0;
    Vec4 acc = interp_load4(part + (s * 2 + 1) * c + c0, vec, c0, c);
    // the partial sums in segment order, the next eight on their way while eight are added (their count is known: no load waits for a flag)
    auto load8 = [&](int64_t j0, Vec4 (&p)[8]) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t q = s + 1 + (j0 + u < len ? j0 + u : len - 1);
        p[u] = interp_load4(part + (q * 2 + 0) * c + c0, vec, c0, c);
      }
    };
    Vec4 cur[8], nxt[8];
    load8(0, cur);
    for (int64_t j0 = 0; j0 < len; j0 += 8) {
      if (j0 + 8 < len) load8(j0 + 8, nxt);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j0 + u < len) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc.v[j] += cur[u].v[j];
        }
#pragma unroll
      for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
    }
    interp_store4(grad_features + row * c + c0, acc, vec, c0, c);
  }
}

static int fps_ref_block(int n) {  // opt_n_threads (cuda_utils.h:10-14): 2^floor(log2 n) clamped to [1, 1024]
  int p = 1;
  while (p * 2 <= n && p < 1024) p *= 2;
  return p;
}

}  // namespace fv2p
using namespace fv2p;

#define G1D(total) dim3(static_cast<unsigned>(ceil_div((total), 256))), dim3(256)
#define STREAM(s) static_cast<hipStream_t>(s)

extern "C" int fv2p_ball_query_batch(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int* idx,
                                     fv2p_stream_t s) {
  FV2P_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 1, FV2P_EINVAL, "ball_query: bad sizes");
  if (b == 0 || m == 0) return 0;
  FV2P_REQUIRE(new_xyz && idx && (xyz || n == 0), FV2P_EINVAL, "ball_query: null pointer");
  hipLaunchKernelGGL(ball_query_batch_k, dim3((unsigned)ceil_div(m, kBqQueries), b), dim3(256), 0, STREAM(s), n, m, radius, nsample, new_xyz, xyz, idx);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_ball_query_stack(int b, int m, float radius, int nsample, const float* new_xyz, const int* new_xyz_batch_cnt,
                                     const float* xyz, const int* xyz_batch_cnt, int* idx, fv2p_stream_t s) {
  FV2P_REQUIRE(b >= 1 && m >= 0 && nsample >= 1, FV2P_EINVAL, "ball_query_stack: bad sizes");
  if (m == 0) return 0;
  FV2P_REQUIRE(new_xyz && new_xyz_batch_cnt && xyz_batch_cnt && idx, FV2P_EINVAL, "ball_query_stack: null pointer");
  hipLaunchKernelGGL(ball_query_stack_k, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, STREAM(s), b, m, radius, nsample, new_xyz,
                     new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_voxel_query_stack(int m, int r1, int r2, int r3, int nsample, float radius, int z_range, int y_range, int x_range,
                                      const float* new_xyz, const float* xyz, const int* new_coords, const int* point_indices, int* idx,
                                      fv2p_stream_t s) {
  FV2P_REQUIRE(m >= 0 && nsample >= 1 && r1 >= 1 && r2 >= 1 && r3 >= 1, FV2P_EINVAL, "voxel_query: bad sizes");
  if (m == 0) return 0;
  FV2P_REQUIRE(new_xyz && xyz && new_coords && point_indices && idx, FV2P_EINVAL, "voxel_query: null pointer");
  hipLaunchKernelGGL(voxel_query_stack_k, G1D(m), 0, STREAM(s), m, r1, r2, r3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz,
                     new_coords, point_indices, idx);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_group_points_batch(int b, int c, int n, int npoints, int nsample, const float* points, const int* idx, float* out,
                                       fv2p_stream_t s) {
  const int64_t total = static_cast<int64_t>(b) * c * npoints * nsample;
  FV2P_REQUIRE(total >= 0, FV2P_EINVAL, "group_points: bad sizes");
  if (total == 0) return 0;
  FV2P_REQUIRE(points && idx && out, FV2P_EINVAL, "group_points: null pointer");
  hipLaunchKernelGGL(group_points_batch_k, G1D(total), 0, STREAM(s), total, c, n, npoints, nsample, points, idx, out);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_group_points_batch_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out, const int* idx,
                                            float* grad_points, fv2p_stream_t s) {
  const int64_t total = static_cast<int64_t>(b) * c * npoints * nsample;
  if (total <= 0) return 0;
  FV2P_REQUIRE(grad_out && idx && grad_points, FV2P_EINVAL, "group_points_grad: null pointer");
  const int64_t rows = static_cast<int64_t>(b) * c;
  const int64_t per = static_cast<int64_t>(npoints) * nsample;
  if (n <= 16384 && per >= 64 && per < (1ll << 30) && rows < (1ll << 30)) {   // one row's accumulator fits a wave's share of LDS
    const int iper = static_cast<int>(per);
#define FV2P_GG(W)                                                                                                          \
  hipLaunchKernelGGL((group_points_batch_grad_lds_k<W>), dim3(static_cast<unsigned>((rows + W - 1) / W)), dim3(W * 64),    \
                     static_cast<size_t>(W) * n * sizeof(float), STREAM(s), rows, c, n, iper, grad_out, idx, grad_points)
    if (n <= 2048) FV2P_GG(8);
    else if (n <= 4096) FV2P_GG(4);
    else if (n <= 8192) FV2P_GG(2);
    else FV2P_GG(1);
#undef FV2P_GG
  } else {
    hipLaunchKernelGGL(group_points_batch_grad_k, G1D(total), 0, STREAM(s), total, c, n, npoints, nsample, grad_out, idx, grad_points);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_gather_points(int b, int c, int n, int npoints, const float* points, const int* idx, float* out, fv2p_stream_t s) {
  const int64_t total = static_cast<int64_t>(b) * c * npoints;
  if (total <= 0) return 0;
  FV2P_REQUIRE(points && idx && out, FV2P_EINVAL, "gather_points: null pointer");
  hipLaunchKernelGGL(gather_points_k, G1D(total), 0, STREAM(s), total, c, n, npoints, points, idx, out);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out, const int* idx, float* grad_points,
                                       fv2p_stream_t s) {
  const int64_t total = static_cast<int64_t>(b) * c * npoints;
  if (total <= 0) return 0;
  FV2P_REQUIRE(grad_out && idx && grad_points, FV2P_EINVAL, "gather_points_grad: null pointer");
  hipLaunchKernelGGL(gather_points_grad_k, G1D(total), 0, STREAM(s), total, c, n, npoints, grad_out, idx, grad_points);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_group_points_stack(int b, int m, int c, int nsample, const float* features, const int* features_batch_cnt,
                                       const int* idx, const int* idx_batch_cnt, float* out, fv2p_stream_t s) {
  const int64_t total = static_cast<int64_t>(m) * c * nsample;
  if (total <= 0) return 0;
  FV2P_REQUIRE(b >= 1 && features && features_batch_cnt && idx && idx_batch_cnt && out, FV2P_EINVAL, "group_points_stack: bad arguments");
  hipLaunchKernelGGL(group_points_stack_k, G1D(total), 0, STREAM(s), b, m, c, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_group_points_stack_grad(int b, int m, int c, int n, int nsample, const float* grad_out, const int* idx,
                                            const int* idx_batch_cnt, const int* features_batch_cnt, float* grad_features, fv2p_stream_t s) {
  const int64_t total = static_cast<int64_t>(m) * c * nsample;
  if (total <= 0) return 0;
  FV2P_REQUIRE(b >= 1 && grad_out && idx && idx_batch_cnt && features_batch_cnt && grad_features, FV2P_EINVAL, "group_points_stack_grad: bad arguments");
  hipLaunchKernelGGL(group_points_stack_grad_k, G1D(total), 0, STREAM(s), b, m, c, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt,
                     grad_features);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// measured on MI355X: 1.43-1.52 us/round against 1.62 for the plain kernel at n = 16384; the Morton pre-pass (bbox, keys,
// 3-4 radix passes) costs ~0.15 ms, so short sampling runs stay on the plain kernel
static unsigned long long* g_fps_trace = nullptr;
// test hook: the streaming sampler writes, for sample 0, trace[8 * wave + {0..7}] = clocks in {box test, issuing a pass's loads, first
// bucket of a pass (wait + distance pass + reduction), the pass's other buckets, wave arg-max, candidate exchange + barrier, winner
// selection}, touched buckets summed over the rounds.  NULL switches it off.
extern "C" int fv2p_fps_set_trace(unsigned long long* trace) { g_fps_trace = trace; return 0; }
static bool fps_bucketed_applies(int n, int m) { return n >= 2048 && n <= 1024 * kStreamBucket && m >= 256; }

extern "C" size_t fv2p_furthest_point_sampling_ws_bytes(int b, int n) {
  const int64_t total = static_cast<int64_t>(b > 0 ? b : 1) * (n > 0 ? n : 1);
  Sizer s;
  s.take<float>(static_cast<size_t>(b > 0 ? b : 1) * 6);
  s.take<uint64_t>(static_cast<size_t>(total));
  s.take<uint64_t>(static_cast<size_t>(total));
  s.take<char>(radix_sort_ws_bytes(total));
  if (n > 48 * kFpsWaves * 64)   // streaming kernel: sorted x, y, z, distance, priority, every sample padded to whole buckets
    s.take<float>(static_cast<size_t>(b > 0 ? b : 1) * static_cast<size_t>(ceil_div(n, kStreamBucket)) * kStreamBucket * 5);
  return s.bytes();
}

extern "C" int fv2p_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp, int* idxs, void* ws, size_t ws_bytes,
                                            fv2p_stream_t s) {
  FV2P_REQUIRE(b >= 0 && n >= 1 && m >= 0, FV2P_EINVAL, "furthest_point_sampling: bad sizes");
  if (b == 0 || m == 0) return 0;
  FV2P_REQUIRE(dataset && temp && idxs, FV2P_EINVAL, "furthest_point_sampling: null pointer");
  const int bs = fps_ref_block(n);
  hipStream_t st = STREAM(s);
  static int lazy = -1;   // FV2P_FPS_LAZY=0 keeps the plain kernel (the parity tests run both)
  if (lazy < 0) { const char* e = FV2P_DEV_ENV("FV2P_FPS_LAZY"); lazy = e ? atoi(e) : 1; }
  if (lazy && ws && ws_bytes >= fv2p_furthest_point_sampling_ws_bytes(b, n) && fps_bucketed_applies(n, m) && b < 65536) {
    const int64_t total = static_cast<int64_t>(b) * n;
    Carver c(ws, ws_bytes);
    float* bbox = c.take<float>(static_cast<size_t>(b) * 6);
    uint64_t* keys = c.take<uint64_t>(static_cast<size_t>(total));
    uint64_t* tmp = c.take<uint64_t>(static_cast<size_t>(total));
    const size_t rb = radix_sort_ws_bytes(total);
    char* rws = c.take<char>(rb);
    hipLaunchKernelGGL(fps_bbox_k, dim3(b), dim3(256), 0, st, n, dataset, bbox);
    hipLaunchKernelGGL(fps_keys_k, G1D(total), 0, st, b, n, dataset, bbox, keys);
    int sbits = 0;
    while ((1 << sbits) < b) ++sbits;
    if (int rc = radix_sort_u64(keys, tmp, total, 24, 48 + sbits, rws, rb, st)) return rc;
    if (n > 48 * kFpsWaves * 64) {   // does not fit one CU's registers: streaming kernel on the sorted copy
      const int ns = static_cast<int>(ceil_div(n, kStreamBucket)) * kStreamBucket;
      const int64_t padded = static_cast<int64_t>(b) * ns;
      float* sx = c.take<float>(static_cast<size_t>(padded) * 5);
      float *sy = sx + padded, *sz = sy + padded, *sd = sz + padded;
      uint32_t* sp = reinterpret_cast<uint32_t*>(sd + padded);
      hipLaunchKernelGGL(fps_stream_prep_k, G1D(padded), 0, st, padded, n, ns, dataset, temp, keys, bs, sx, sy, sz, sd, sp);
      hipLaunchKernelGGL(fps_stream_k, dim3(b), dim3(kStreamWaves * 64), 0, st, n, m, bs, dataset, sx, sy, sz, sd, sp, idxs, g_fps_trace);
      hipLaunchKernelGGL(fps_stream_post_k, G1D(total), 0, st, total, n, ns, keys, sd, temp);
      FV2P_LAUNCH_CHECK();
      return 0;
    }
    static int form = -1;   // FV2P_FPS_FORM=thread keeps the per-thread buckets of round 1 (the parity tests run every form)
    if (form < 0) { const char* e = FV2P_DEV_ENV("FV2P_FPS_FORM"); form = (e && e[0] == 't') ? 0 : 1; }
    const int slots = static_cast<int>(ceil_div(n, kFpsWaves * 64));
    const int ppt = static_cast<int>(ceil_div(n, 1024));
    static int tree = -1;   // development: FV2P_FPS_IDX=0/1 selects how the touched buckets are walked (see fps_wave_k)
    if (tree < 0) { const char* e = FV2P_DEV_ENV("FV2P_FPS_IDX"); tree = e ? atoi(e) : kFpsIdxDefault; }
    // measured (us per round, run-time index against compile-time scan): 16 384 points (32 slots) 0.601 / 0.714, 20 000 (40 slots: the
    // second vector's branch, 237 registers) 0.830 / 0.757, 24 576 (48 slots) 0.956 / 0.801 -> run-time indices up to 32 slots (2 = always)
    const bool idx_form = tree == 2 || (tree == 1 && slots <= 32);
    if (form == 1 || ppt > 16) {
#define FV2P_FPS(SS)                                                                                                       \
  do {                                                                                                                     \
    const size_t lds = static_cast<size_t>(SS) * kFpsWaves * 64 * sizeof(uint32_t);                                         \
    static bool big = false;                                                                                               \
    if (lds > 48 * 1024 && !big) {                                                                                         \
      FV2P_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_wave_k<SS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   static_cast<int>(lds)));                                                                \
      FV2P_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_wave_k<SS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   static_cast<int>(lds)));                                                                \
      FV2P_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_wave_k<SS, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   static_cast<int>(lds)));                                                                \
      FV2P_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_wave_k<SS, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   static_cast<int>(lds)));                                                                \
      big = true;                                                                                                          \
    }                                                                                                                      \
    if (g_fps_trace && idx_form) hipLaunchKernelGGL((fps_wave_k<SS, true, true>), dim3(b), dim3(kFpsWaves * 64), lds, st, n, m, bs, dataset, keys, temp, idxs, g_fps_trace); \
    else if (g_fps_trace) hipLaunchKernelGGL((fps_wave_k<SS, true>), dim3(b), dim3(kFpsWaves * 64), lds, st, n, m, bs, dataset, keys, temp, idxs, g_fps_trace); \
    else if (idx_form) hipLaunchKernelGGL((fps_wave_k<SS, false, true>), dim3(b), dim3(kFpsWaves * 64), lds, st, n, m, bs, dataset, keys, temp, idxs, nullptr); \
    else hipLaunchKernelGGL((fps_wave_k<SS, false>), dim3(b), dim3(kFpsWaves * 64), lds, st, n, m, bs, dataset, keys, temp, idxs, nullptr); \
  } while (0)
      if (slots <= 8) FV2P_FPS(8);
      else if (slots <= 16) FV2P_FPS(16);
      else if (slots <= 24) FV2P_FPS(24);
      else if (slots <= 32) FV2P_FPS(32);
      else if (slots <= 40) FV2P_FPS(40);
      else FV2P_FPS(48);
#undef FV2P_FPS
    } else if (ppt <= 4) hipLaunchKernelGGL((fps_bucket_k<4>), dim3(b), dim3(1024), 0, st, n, m, bs, dataset, keys, temp, idxs);
    else if (ppt <= 8) hipLaunchKernelGGL((fps_bucket_k<8>), dim3(b), dim3(1024), 0, st, n, m, bs, dataset, keys, temp, idxs);
    else hipLaunchKernelGGL((fps_bucket_k<16>), dim3(b), dim3(1024), 0, st, n, m, bs, dataset, keys, temp, idxs);
    FV2P_LAUNCH_CHECK();
    return 0;
  }
  if (bs == 1024) {
    const int ppt = static_cast<int>(ceil_div(n, 1024));
    if (ppt <= 4) hipLaunchKernelGGL((fps_k<1024, 4, true>), dim3(b), dim3(1024), 0, st, n, m, bs, dataset, temp, idxs);
    else if (ppt <= 8) hipLaunchKernelGGL((fps_k<1024, 8, true>), dim3(b), dim3(1024), 0, st, n, m, bs, dataset, temp, idxs);
    else if (ppt <= 16) hipLaunchKernelGGL((fps_k<1024, 16, true>), dim3(b), dim3(1024), 0, st, n, m, bs, dataset, temp, idxs);
    else hipLaunchKernelGGL((fps_k<1024, 1, false>), dim3(b), dim3(1024), 0, st, n, m, bs, dataset, temp, idxs);
  } else if (bs >= 256) {
    hipLaunchKernelGGL((fps_k<512, 2, true>), dim3(b), dim3(512), 0, st, n, m, bs, dataset, temp, idxs);  // n < 1024: <= 2 points per owner
  } else {
    hipLaunchKernelGGL((fps_k<128, 2, true>), dim3(b), dim3(128), 0, st, n, m, bs, dataset, temp, idxs);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_three_nn_batch(int b, int n, int m, const float* unknown, const float* known, float* dist2, int* idx, fv2p_stream_t s) {
  FV2P_REQUIRE(b >= 0 && n >= 0 && m >= 0, FV2P_EINVAL, "three_nn: bad sizes");
  if (b == 0 || n == 0) return 0;
  FV2P_REQUIRE(unknown && dist2 && idx && (known || m == 0), FV2P_EINVAL, "three_nn: null pointer");
  hipLaunchKernelGGL(three_nn_batch_k, dim3((unsigned)ceil_div(n, 64), b), dim3(256), 0, STREAM(s), n, m, unknown, known, dist2, idx);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_three_nn_stack(int b, int n, int m, const float* unknown, const int* unknown_batch_cnt, const float* known,
                                   const int* known_batch_cnt, float* dist2, int* idx, fv2p_stream_t s) {
  FV2P_REQUIRE(b >= 1 && n >= 0, FV2P_EINVAL, "three_nn_stack: bad sizes");
  if (n == 0) return 0;
  FV2P_REQUIRE(unknown && unknown_batch_cnt && known_batch_cnt && dist2 && idx, FV2P_EINVAL, "three_nn_stack: null pointer");
  hipLaunchKernelGGL(three_nn_stack_k, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, STREAM(s), b, n, unknown, unknown_batch_cnt, known,
                     known_batch_cnt, dist2, idx);
  FV2P_LAUNCH_CHECK();
  return 0;
}
static int64_t nn_slots(int64_t m) {   // hash-table slots: a power of two >= 2 m
  int64_t t = 1024;
  while (t < 2 * m) t <<= 1;
  return t;
}
extern "C" size_t fv2p_three_nn_grid_ws_bytes(int b, int64_t n, int64_t m) {
  const size_t bb = static_cast<size_t>(b > 0 ? b : 1), mm = static_cast<size_t>(m > 0 ? m : 1);
  const size_t slots = static_cast<size_t>(nn_slots(static_cast<int64_t>(mm)));
  return align_up(bb * 6 * sizeof(int)) + align_up(bb * sizeof(NNGeo)) + align_up(slots * sizeof(unsigned long long)) + align_up(slots * sizeof(int)) +
         align_up((slots + 1) * sizeof(int)) + align_up(mm * sizeof(int)) + align_up(mm * sizeof(float4)) +
         align_up(scan_ws_bytes(static_cast<int64_t>(slots))) + align_up(sizeof(int)) +
         align_up(static_cast<size_t>(n > 0 ? n : 1) * sizeof(int));   // ... and the list of queries the first pass leaves over
}
extern "C" int fv2p_three_nn_stack_grid(int b, int n, int m, const float* unknown, const int* unknown_batch_cnt, const float* known,
                                        const int* known_batch_cnt, float cell, float* dist2, int* idx, void* ws, size_t ws_bytes, fv2p_stream_t s) {
  FV2P_REQUIRE(b >= 1 && n >= 0 && m >= 0, FV2P_EINVAL, "three_nn_stack_grid: bad sizes");
  if (n == 0) return 0;
  FV2P_REQUIRE(unknown && unknown_batch_cnt && known_batch_cnt && dist2 && idx && (known || m == 0), FV2P_EINVAL, "three_nn_stack_grid: null pointer");
  FV2P_REQUIRE(b <= 512, FV2P_ELIMIT, "three_nn_stack_grid: at most 512 samples");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_three_nn_grid_ws_bytes(b, n, m), FV2P_EWORKSPACE, "three_nn_stack_grid: workspace too small");
  hipStream_t st = STREAM(s);
  const int64_t slots = nn_slots(m > 0 ? m : 1);
  Carver c(ws, ws_bytes);
  int* bbox = c.take<int>(static_cast<size_t>(b) * 6);
  NNGeo* geo = c.take<NNGeo>(static_cast<size_t>(b));
  unsigned long long* keys = c.take<unsigned long long>(static_cast<size_t>(slots));
  int* count = c.take<int>(static_cast<size_t>(slots));
  int* start = c.take<int>(static_cast<size_t>(slots) + 1);
  int* slot_of = c.take<int>(static_cast<size_t>(m > 0 ? m : 1));
  float4* sorted = c.take<float4>(static_cast<size_t>(m > 0 ? m : 1));
  const size_t sb = scan_ws_bytes(slots);
  void* sws = c.take<char>(sb);
  int* todo_count = c.take<int>(1);
  int* todo = c.take<int>(static_cast<size_t>(n));
  const unsigned int mask = static_cast<unsigned int>(slots - 1);
  const unsigned chunks = static_cast<unsigned>(std::min<int64_t>(64, std::max<int64_t>(1, ceil_div(m, 4096))));
  hipLaunchKernelGGL(nn_box_init_k, dim3(static_cast<unsigned>(ceil_div(b * 6, 64))), dim3(64), 0, st, b, bbox);
  hipLaunchKernelGGL(nn_bbox_k, dim3(b, chunks), dim3(256), 0, st, b, known, known_batch_cnt, bbox);
  hipLaunchKernelGGL(nn_setup_k, dim3(static_cast<unsigned>(ceil_div(b, 64))), dim3(64), 0, st, b, bbox, known_batch_cnt, cell, geo);
  FV2P_HIP(hipMemsetAsync(keys, 0xff, static_cast<size_t>(slots) * sizeof(unsigned long long), st));
  FV2P_HIP(hipMemsetAsync(count, 0, static_cast<size_t>(slots) * sizeof(int), st));
  FV2P_HIP(hipMemsetAsync(todo_count, 0, sizeof(int), st));
  if (m > 0) hipLaunchKernelGGL(nn_insert_k, G1D(m), 0, st, b, m, known, known_batch_cnt, geo, mask, keys, count, slot_of);
  if (int rc = exclusive_scan_i32(count, start, slots, start + slots, sws, sb, st)) return rc;
  if (m > 0) hipLaunchKernelGGL(nn_scatter_k, G1D(m), 0, st, b, m, known, known_batch_cnt, slot_of, start, count, sorted);
  hipLaunchKernelGGL(nn_query_k, G1D(n), 0, st, b, n, unknown, unknown_batch_cnt, known_batch_cnt, geo, mask, keys, start, sorted, dist2, idx, todo, todo_count);
  hipLaunchKernelGGL(nn_rest_k, dim3(static_cast<unsigned>(std::min<int64_t>(2048, ceil_div(n, 4)))), dim3(256), 0, st, b, unknown, unknown_batch_cnt, known, known_batch_cnt,
                     geo, mask, keys, start, sorted, todo, todo_count, dist2, idx);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_three_interpolate_batch(int b, int c, int m, int n, const float* points, const int* idx, const float* weight, float* out,
                                            fv2p_stream_t s) {
  const int64_t total = static_cast<int64_t>(b) * c * n;
  if (total <= 0) return 0;
  FV2P_REQUIRE(points && idx && weight && out, FV2P_EINVAL, "three_interpolate: null pointer");
  hipLaunchKernelGGL(three_interp_batch_k, G1D(total), 0, STREAM(s), total, c, m, n, points, idx, weight, out);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_three_interpolate_batch_grad(int b, int c, int n, int m, const float* grad_out, const int* idx, const float* weight,
                                                 float* grad_points, fv2p_stream_t s) {
  const int64_t total = static_cast<int64_t>(b) * c * n;
  if (total <= 0) return 0;
  FV2P_REQUIRE(grad_out && idx && weight && grad_points, FV2P_EINVAL, "three_interpolate_grad: null pointer");
  hipLaunchKernelGGL(three_interp_batch_grad_k, G1D(total), 0, STREAM(s), total, c, m, n, grad_out, idx, weight, grad_points);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_three_interpolate_stack(int n, int c, const float* features, const int* idx, const float* weight, float* out,
                                            fv2p_stream_t s) {
  const int64_t total = static_cast<int64_t>(n) * c;
  if (total <= 0) return 0;
  FV2P_REQUIRE(features && idx && weight && out, FV2P_EINVAL, "three_interpolate_stack: null pointer");
  hipLaunchKernelGGL(three_interp_stack_k, G1D(total), 0, STREAM(s), total, c, features, idx, weight, out);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" int fv2p_three_interpolate_stack_grad(int n, int c, const float* grad_out, const int* idx, const float* weight,
                                                 float* grad_features, fv2p_stream_t s) {
  const int64_t total = static_cast<int64_t>(n) * c;
  if (total <= 0) return 0;
  FV2P_REQUIRE(grad_out && idx && weight && grad_features, FV2P_EINVAL, "three_interpolate_stack_grad: null pointer");
  hipLaunchKernelGGL(three_interp_stack_grad_k, G1D(total), 0, STREAM(s), total, c, grad_out, idx, weight, grad_features);
  FV2P_LAUNCH_CHECK();
  return 0;
}
extern "C" size_t fv2p_three_interpolate_stack_grad_ws_bytes(int n, int c, int m) {
  const size_t entries = static_cast<size_t>(n > 0 ? n : 1) * 3;
  const size_t segments = (entries + kInterpSeg - 1) / kInterpSeg;
  (void)m;
  return 2 * align_up(entries * sizeof(uint64_t)) + align_up(radix_sort_ws_bytes(static_cast<int64_t>(entries))) +
         align_up(segments * 2 * static_cast<size_t>(c > 0 ? c : 1) * sizeof(float)) + align_up(segments * sizeof(int));
}
extern "C" int fv2p_three_interpolate_stack_grad_gather(int n, int c, int m, const float* grad_out, const int* idx, const float* weight,
                                                        float* grad_features, void* ws, size_t ws_bytes, fv2p_stream_t s) {
  FV2P_REQUIRE(n >= 0 && c >= 0 && m >= 0, FV2P_EINVAL, "three_interpolate_stack_grad_gather: bad sizes");
  const int64_t total = static_cast<int64_t>(m) * c;
  if (total <= 0) return 0;
  FV2P_REQUIRE(grad_features && (n == 0 || (grad_out && idx && weight)), FV2P_EINVAL, "three_interpolate_stack_grad_gather: null pointer");
  FV2P_REQUIRE(static_cast<int64_t>(n) * 3 < (1ll << 31), FV2P_ELIMIT, "three_interpolate_stack_grad_gather: too many queries");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_three_interpolate_stack_grad_ws_bytes(n, c, m), FV2P_EWORKSPACE, "three_interpolate_stack_grad_gather: workspace too small");
  hipStream_t st = STREAM(s);
  const int64_t entries = static_cast<int64_t>(n) * 3;
  const int64_t segments = (entries + kInterpSeg - 1) / kInterpSeg;
  Carver cv(ws, ws_bytes);
  uint64_t* keys = cv.take<uint64_t>(static_cast<size_t>(entries > 0 ? entries : 1));
  uint64_t* tmp = cv.take<uint64_t>(static_cast<size_t>(entries > 0 ? entries : 1));
  const size_t rb = radix_sort_ws_bytes(entries > 0 ? entries : 1);
  void* rws = cv.take<char>(rb);
  float* part = cv.take<float>(static_cast<size_t>(segments > 0 ? segments : 1) * 2 * c);
  int* flags = cv.take<int>(static_cast<size_t>(segments > 0 ? segments : 1));
  FV2P_HIP(hipMemsetAsync(grad_features, 0, static_cast<size_t>(total) * sizeof(float), st));   // rows nobody reads
  if (entries == 0) return 0;
  hipLaunchKernelGGL(interp_keys_k, G1D(entries), 0, st, entries, m, idx, keys);
  if (int rc = radix_sort_u64(keys, tmp, entries, 32, 32 + bits_for(static_cast<uint64_t>(m)), rws, rb, st)) return rc;
  int tpr = 1;
  while (tpr < 64 && tpr * 4 < c) tpr *= 2;
  const int64_t threads = segments * tpr;
  // 16-byte accesses only when every row of both caller buffers starts on a 16-byte boundary (a tensor view with a storage offset
  // or a C caller's odd pointer falls back to scalar accesses); `part` is carved 256-byte aligned
  const int aligned = ((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_features) | reinterpret_cast<uintptr_t>(part)) & 15) == 0;
#define FV2P_IG(T)                                                                                                                \
  {                                                                                                                               \
    hipLaunchKernelGGL((interp_seg_k<T>), G1D(threads), 0, st, m, c, entries, keys, grad_out, weight, grad_features, part, flags, aligned); \
    hipLaunchKernelGGL((interp_fix_k<T>), G1D(threads), 0, st, c, entries, segments, keys, part, flags, grad_features, aligned);            \
  }
  switch (tpr) { case 1: FV2P_IG(1) break; case 2: FV2P_IG(2) break; case 4: FV2P_IG(4) break; case 8: FV2P_IG(8) break;
                 case 16: FV2P_IG(16) break; case 32: FV2P_IG(32) break; default: FV2P_IG(64) }
#undef FV2P_IG
  FV2P_LAUNCH_CHECK();
  return 0;
}
