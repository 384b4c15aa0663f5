// Prefix-scan and LSD radix-sort primitives used by the voxeliser and the rulebook builder.
//
// Both are launch-structured (no inter-workgroup hand-offs inside a launch), so they carry no
// dispatch-order assumption.  Radix passes stage their digit histograms in LDS and rank keys
// with wavefront ballots (wave64 match-any over the 8 digit bits).
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>

namespace fv2p {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ------------------------------------------------------------------ scan ---------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;  // 2048

__global__ __launch_bounds__(kScanThreads) void scan_block_sums(const int* __restrict__ in, int64_t n,
                                                                int* __restrict__ sums) {
  __shared__ int lds_wave[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile + threadIdx.x * kScanItems;
  int s = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    int64_t i = base + j;
    if (i < n) s += in[i];
  }
  int tot;
  block_excl_scan_256(s, lds_wave, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// Single workgroup: exclusive scan of `m` block sums in place, total -> *total.
__global__ __launch_bounds__(kScanThreads) void scan_sums_inplace(int* __restrict__ sums, int64_t m,
                                                                  int* __restrict__ total) {
  __shared__ int lds_wave[4];
  int carry = 0;
  for (int64_t c = 0; c < m; c += kScanThreads) {
    int64_t i = c + threadIdx.x;
    int v = (i < m) ? sums[i] : 0;
    int tot;
    int ex = block_excl_scan_256(v, lds_wave, &tot);
    if (i < m) sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

// Whole array in one workgroup (n <= kSmallScan): one launch instead of three.  1024 threads x 32 items: every load is
// issued before the first add (the 16-tile loop this replaces was a chain of 16 dependent round trips, 19.5 us at
// n = 16384-32768), then one wave-shuffle scan per wave and a 16-entry LDS scan across the waves.
constexpr int kSmallThreads = 1024;
constexpr int kSmallScan = kSmallThreads * 32;
// kSmallItems = 32 for up to 32768 entries, 4 for up to 4096 (tile sums, digit histograms, per-offset counts: most calls)
template <int kSmallItems>
__global__ __launch_bounds__(kSmallThreads) void scan_small(const int* __restrict__ in, int* __restrict__ out, int64_t n,
                                                            int* __restrict__ total) {
  __shared__ int lds_wave[kSmallThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t base = static_cast<int64_t>(tid) * kSmallItems;
  int v[kSmallItems];
  const bool vec = (reinterpret_cast<uintptr_t>(in) & 15) == 0 && base + kSmallItems <= n;
  if (vec) {
#pragma unroll
    for (int j = 0; j < kSmallItems / 4; ++j) {
      const int4 q = reinterpret_cast<const int4*>(in + base)[j];
      v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < kSmallItems; ++j) v[j] = (base + j < n) ? in[base + j] : 0;
  }
  int s = 0;
#pragma unroll
  for (int j = 0; j < kSmallItems; ++j) s += v[j];
  int incl = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) lds_wave[w] = incl;
  __syncthreads();
  int wave_base = 0, tot = 0;
#pragma unroll
  for (int q = 0; q < kSmallThreads / 64; ++q) {
    const int t = lds_wave[q];
    if (q < w) wave_base += t;
    tot += t;
  }
  int ex = wave_base + incl - s;
  const bool vec_out = (reinterpret_cast<uintptr_t>(out) & 15) == 0 && base + kSmallItems <= n;
  if (vec_out) {
#pragma unroll
    for (int j = 0; j < kSmallItems / 4; ++j) {
      int4 q;
      q.x = ex; ex += v[4 * j];
      q.y = ex; ex += v[4 * j + 1];
      q.z = ex; ex += v[4 * j + 2];
      q.w = ex; ex += v[4 * j + 3];
      reinterpret_cast<int4*>(out + base)[j] = q;
    }
  } else {
#pragma unroll
    for (int j = 0; j < kSmallItems; ++j) {
      if (base + j < n) out[base + j] = ex;
      ex += v[j];
    }
  }
  if (tid == 0 && total) *total = tot;
}

__global__ __launch_bounds__(kScanThreads) void scan_apply(const int* __restrict__ in, int* __restrict__ out,
                                                           int64_t n, const int* __restrict__ block_off) {
  __shared__ int lds_wave[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile + threadIdx.x * kScanItems;
  int v[kScanItems];
  int s = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    int64_t i = base + j;
    v[j] = (i < n) ? in[i] : 0;
    s += v[j];
  }
  int tot;
  int ex = block_excl_scan_256(s, lds_wave, &tot) + block_off[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    int64_t i = base + j;
    if (i < n) out[i] = ex;
    ex += v[j];
  }
}

size_t scan_ws_bytes(int64_t n) {
  Sizer s;
  s.take<int>(static_cast<size_t>(ceil_div(n > 0 ? n : 1, kScanTile)) + 1);
  return s.bytes();
}

int exclusive_scan_i32(const int* in, int* out, int64_t n, int* total, void* ws, size_t ws_bytes,
                       hipStream_t stream) {
  if (n <= 0) {
    if (total) FV2P_HIP(hipMemsetAsync(total, 0, sizeof(int), stream));
    return 0;
  }
  if (n <= kSmallScan) {
    if (n <= kSmallThreads * 4) hipLaunchKernelGGL(scan_small<4>, dim3(1), dim3(kSmallThreads), 0, stream, in, out, n, total);
    else hipLaunchKernelGGL(scan_small<32>, dim3(1), dim3(kSmallThreads), 0, stream, in, out, n, total);
    FV2P_LAUNCH_CHECK();
    return 0;
  }
  FV2P_REQUIRE(ws_bytes >= scan_ws_bytes(n), FV2P_EWORKSPACE, "scan workspace too small");
  Carver c(ws, ws_bytes);
  const int64_t nb = ceil_div(n, kScanTile);
  int* sums = c.take<int>(static_cast<size_t>(nb) + 1);
  hipLaunchKernelGGL(scan_block_sums, dim3(nb), dim3(kScanThreads), 0, stream, in, n, sums);
  hipLaunchKernelGGL(scan_sums_inplace, dim3(1), dim3(kScanThreads), 0, stream, sums, nb, total);
  hipLaunchKernelGGL(scan_apply, dim3(nb), dim3(kScanThreads), 0, stream, in, out, n, sums);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ multi-region fill ---
__global__ __launch_bounds__(256) void multi_fill_k(FillJobs jobs) {
  const int j = blockIdx.y;
  unsigned* p = static_cast<unsigned*>(jobs.ptr[j]);
  const unsigned long long words = jobs.bytes[j] / 4;
  const unsigned v = jobs.value[j];
  const unsigned long long tid = static_cast<unsigned long long>(blockIdx.x) * 256 + threadIdx.x;
  const unsigned long long nthr = static_cast<unsigned long long>(gridDim.x) * 256;
  // head up to a 16-byte boundary, uint4 body, tail
  unsigned long long head = ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4;
  if (head > words) head = words;
  if (tid < head) p[tid] = v;
  uint4* body = reinterpret_cast<uint4*>(p + head);
  const unsigned long long quads = (words - head) / 4;
  for (unsigned long long q = tid; q < quads; q += nthr) body[q] = make_uint4(v, v, v, v);
  const unsigned long long done = head + quads * 4;
  if (tid < words - done) p[done + tid] = v;
}

int multi_fill(const FillJobs& jobs, hipStream_t stream) {
  if (jobs.n == 0) return 0;
  unsigned long long big = 0;
  for (int j = 0; j < jobs.n; ++j) big = jobs.bytes[j] > big ? jobs.bytes[j] : big;
  unsigned blocks = static_cast<unsigned>(ceil_div(static_cast<int64_t>(big / 16 + 1), 256 * 4));
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(multi_fill_k, dim3(blocks, jobs.n), dim3(256), 0, stream, jobs);
  FV2P_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ radix sort ---------
constexpr int kSortThreads = 256;
constexpr int kSortRounds = 4;                           // keys per thread per tile
constexpr int kSortTile = kSortThreads * kSortRounds;    // 1024 keys per workgroup
constexpr int kRadix = 256;

// hist layout [nblk][256]: hist[b][d] = number of keys of tile b whose digit is d
__global__ __launch_bounds__(kSortThreads) void radix_hist(const uint64_t* __restrict__ keys, int64_t n,
                                                           int shift, uint32_t digit_mask, int nblk,
                                                           int* __restrict__ hist) {
  __shared__ int lh[kRadix];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kSortTile;
#pragma unroll 4
  for (int r = 0; r < kSortRounds; ++r) {
    int64_t i = base + r * kSortThreads + threadIdx.x;
    if (i < n) {
      uint32_t d = static_cast<uint32_t>(keys[i] >> shift) & digit_mask;
      atomicAdd(&lh[d], 1);
    }
  }
  __syncthreads();
  hist[static_cast<int64_t>(blockIdx.x) * kRadix + threadIdx.x] = lh[threadIdx.x];
}

// The global offset of (digit d, tile b) = sum_{d' < d} total[d'] + sum_{b' < b} hist[b'][d] is computed by every
// workgroup for itself (thread d walks column d of hist: nblk coalesced loads), which removes the separate scan launch.
__global__ __launch_bounds__(kSortThreads) void radix_scatter(const uint64_t* __restrict__ keys,
                                                              uint64_t* __restrict__ out, int64_t n, int shift,
                                                              uint32_t digit_mask, int nblk,
                                                              const int* __restrict__ hist /*[nblk][256]*/) {
  __shared__ int base[kRadix];           // running output position per digit for this tile
  __shared__ int wave_cnt[4][kRadix];    // per-round, per-wave digit counts
  __shared__ int lds_wave[4];
  const int tid = threadIdx.x, w = tid >> 6;
  {
    int before = 0, total = 0;
    for (int b = 0; b < nblk; ++b) {
      const int v = hist[static_cast<int64_t>(b) * kRadix + tid];
      total += v;
      if (b < static_cast<int>(blockIdx.x)) before += v;
    }
    int tot;
    const int ex = block_excl_scan_256(total, lds_wave, &tot);  // exclusive over digits
    base[tid] = ex + before;
  }
  const int64_t tile0 = static_cast<int64_t>(blockIdx.x) * kSortTile;
  for (int r = 0; r < kSortRounds; ++r) {
    const int64_t i = tile0 + r * kSortThreads + tid;
    if (tile0 + r * kSortThreads >= n) break;  // uniform
    // zero the per-wave counters of this round
    wave_cnt[0][tid] = 0; wave_cnt[1][tid] = 0; wave_cnt[2][tid] = 0; wave_cnt[3][tid] = 0;
    __syncthreads();
    const bool valid = i < n;
    uint64_t key = valid ? keys[i] : 0;
    uint32_t d = static_cast<uint32_t>(key >> shift) & digit_mask;
    // wave64 match-any on the 8 digit bits
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      uint64_t vote = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? vote : ~vote;
    }
    const int rank_in_wave = __popcll(peers & lanemask_lt());
    if (valid && rank_in_wave == 0) wave_cnt[w][d] = __popcll(peers);
    __syncthreads();
    if (valid) {
      int pos = base[d] + rank_in_wave;
      for (int ww = 0; ww < w; ++ww) pos += wave_cnt[ww][d];
      out[pos] = key;
    }
    __syncthreads();
    base[tid] += wave_cnt[0][tid] + wave_cnt[1][tid] + wave_cnt[2][tid] + wave_cnt[3][tid];
    __syncthreads();
  }
}

size_t radix_sort_ws_bytes(int64_t n) {
  const int64_t nblk = ceil_div(n > 0 ? n : 1, kSortTile);
  Sizer s;
  s.take<int>(static_cast<size_t>(kRadix * nblk));
  return s.bytes();
}

int radix_sort_u64(uint64_t* keys, uint64_t* tmp, int64_t n, int bit_lo, int bit_hi, void* ws,
                   size_t ws_bytes, hipStream_t stream) {
  if (n <= 1 || bit_hi <= bit_lo) return 0;
  FV2P_REQUIRE(ws_bytes >= radix_sort_ws_bytes(n), FV2P_EWORKSPACE, "radix sort workspace too small");
  const int nblk = static_cast<int>(ceil_div(n, kSortTile));
  Carver c(ws, ws_bytes);
  int* hist = c.take<int>(static_cast<size_t>(kRadix) * nblk);
  uint64_t* src = keys;
  uint64_t* dst = tmp;
  for (int lo = bit_lo; lo < bit_hi; lo += 8) {
    const int nb = (bit_hi - lo) < 8 ? (bit_hi - lo) : 8;
    const uint32_t dm = (1u << nb) - 1u;
    hipLaunchKernelGGL(radix_hist, dim3(nblk), dim3(kSortThreads), 0, stream, src, n, lo, dm, nblk, hist);
    hipLaunchKernelGGL(radix_scatter, dim3(nblk), dim3(kSortThreads), 0, stream, src, dst, n, lo, dm, nblk, hist);
    uint64_t* t = src; src = dst; dst = t;
  }
  FV2P_LAUNCH_CHECK();
  if (src != keys) FV2P_HIP(hipMemcpyAsync(keys, src, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
  return 0;
}

}  // namespace fv2p

extern "C" const char* fv2p_last_error(void) { return fv2p::g_err; }
extern "C" int fv2p_abi_version(void) { return FV2P_ABI_VERSION; }

// Test hooks for the primitives (exercised by tests/test_primitives.py through the C ABI).
extern "C" size_t fv2p_scan_ws_bytes(int64_t n) { return fv2p::scan_ws_bytes(n); }
extern "C" int fv2p_exclusive_scan_i32(const int* in, int* out, int64_t n, int* total, void* ws,
                                       size_t ws_bytes, void* stream) {
  return fv2p::exclusive_scan_i32(in, out, n, total, ws, ws_bytes, static_cast<hipStream_t>(stream));
}
extern "C" size_t fv2p_radix_sort_ws_bytes(int64_t n) { return fv2p::radix_sort_ws_bytes(n); }
extern "C" int fv2p_radix_sort_u64(uint64_t* keys, uint64_t* tmp, int64_t n, int bit_lo, int bit_hi,
                                   void* ws, size_t ws_bytes, void* stream) {
  return fv2p::radix_sort_u64(keys, tmp, n, bit_lo, bit_hi, ws, ws_bytes, static_cast<hipStream_t>(stream));
}
