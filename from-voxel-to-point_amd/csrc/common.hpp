// Shared host/device helpers for libfv2p_ops (gfx950 only).
//
// Conventions used by every translation unit in csrc/:
//  * every extern "C" entry point returns 0 on success or a negative FV2P_E* code and
//    records a message retrievable through fv2p_last_error();
//  * no hipMalloc/hipFree inside an op: scratch comes from the caller-provided workspace;
//  * kernels are launched on the caller's stream, never on the legacy default stream;
//  * wave64 everywhere (ballot masks are 64 bit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include "../../include/fv2p_ops.h"

namespace fv2p {

constexpr int kWave = 64;

int set_error(int code, const char* fmt, ...);

#define FV2P_REQUIRE(cond, code, ...)                      \
  do {                                                     \
    if (!(cond)) return ::fv2p::set_error((code), __VA_ARGS__); \
  } while (0)

#define FV2P_HIP(expr)                                                              \
  do {                                                                              \
    hipError_t e__ = (expr);                                                        \
    if (e__ != hipSuccess)                                                          \
      return ::fv2p::set_error(FV2P_EHIP, "%s failed: %s (%s:%d)", #expr,           \
                               hipGetErrorString(e__), __FILE__, __LINE__);         \
  } while (0)

#define FV2P_LAUNCH_CHECK() FV2P_HIP(hipGetLastError())

// Development switches (tuning overrides read from the environment, timing-only ablation kernels) exist only in a library built
// with -DFV2P_DEV=1 (`make DEV=1` -> lib/dev/, used by tools/ and the profile scripts through FV2P_LIB_DIR).  The release library
// reads no environment variable at all: its kernel choice is a function of the call's arguments and of the run-time setters declared
// in include/fv2p_ops.h (fv2p_sparse_conv_set_impl / _set_paths, fv2p_dcn_set_colg_cap, ...), and tests/test_abi.py checks that the
// names below do not occur in its string table.
#ifdef FV2P_DEV
#define FV2P_DEV_ENV(name) getenv(name)
#else
#define FV2P_DEV_ENV(name) (static_cast<const char*>(nullptr))
#endif

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Carves typed, 256-byte aligned regions out of the caller's workspace.
struct Carver {
  char* base;
  size_t off;
  size_t cap;
  Carver(void* p, size_t bytes) : base(static_cast<char*>(p)), off(0), cap(bytes) {}
  template <typename T>
  T* take(size_t n) {
    size_t o = align_up(off);
    off = o + n * sizeof(T);
    return reinterpret_cast<T*>(base + o);
  }
  bool ok() const { return off <= cap; }
};
// Same arithmetic without a buffer, for *_ws_bytes queries.
struct Sizer {
  size_t off = 0;
  template <typename T>
  void take(size_t n) { off = align_up(off) + n * sizeof(T); }
  size_t bytes() const { return align_up(off); }
};

// One launch that fills up to 8 regions with a 32-bit pattern each (replaces a run of hipMemsetAsync calls: every
// memset is its own dispatch, ~3 us of queue time each).  Regions are 4-byte aligned, sizes multiples of 4 bytes.
struct FillJobs {
  void* ptr[8];
  unsigned long long bytes[8];
  unsigned value[8];
  int n = 0;
  void add(void* p, size_t nbytes, unsigned v) {
    if (p && nbytes) { ptr[n] = p; bytes[n] = nbytes; value[n] = v; ++n; }
  }
};
int multi_fill(const FillJobs& jobs, hipStream_t stream);

// ---- device-side helpers -------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ uint64_t lanemask_lt() {
  return (1ull << (threadIdx.x & 63)) - 1ull;
}

// 64-bit mix (splitmix64 finaliser) used by every open-addressing table in the library.
__device__ __forceinline__ uint32_t hash_u64(uint64_t k, uint32_t mask) {
  k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27; k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return static_cast<uint32_t>(k) & mask;
}

// Packed hash slot: high 40 bits = key (flat voxel index), low 24 bits = payload (row / point id).
constexpr int kValBits = 24;
constexpr uint64_t kValMask = (1ull << kValBits) - 1ull;
constexpr uint64_t kEmptySlot = ~0ull;
constexpr int64_t kMaxRows = (1ll << kValBits) - 2;      // payload 0xFFFFFF is reserved
constexpr int64_t kMaxKey = (1ll << 40) - 2;

__device__ __forceinline__ uint64_t slot_pack(uint64_t key, uint32_t val) {
  return (key << kValBits) | static_cast<uint64_t>(val);
}
__device__ __forceinline__ uint64_t slot_key(uint64_t s) { return s >> kValBits; }
__device__ __forceinline__ uint32_t slot_val(uint64_t s) { return static_cast<uint32_t>(s & kValMask); }

// Lookup in a packed table; returns payload or -1.
__device__ __forceinline__ int table_find(const uint64_t* __restrict__ table, uint32_t mask, uint64_t key) {
  uint32_t h = hash_u64(key, mask);
  while (true) {
    uint64_t s = table[h];
    if (s == kEmptySlot) return -1;
    if (slot_key(s) == key) return static_cast<int>(slot_val(s));
    h = (h + 1) & mask;
  }
}

// Block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix,
// *block_total receives the sum (valid in all threads).
__device__ __forceinline__ int block_excl_scan_256(int v, int* lds_wave /*[4]*/, int* block_total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) lds_wave[w] = incl;
  __syncthreads();
  int w0 = lds_wave[0], w1 = lds_wave[1], w2 = lds_wave[2], w3 = lds_wave[3];
  int base = (w > 0 ? w0 : 0) + (w > 1 ? w1 : 0) + (w > 2 ? w2 : 0);
  *block_total = w0 + w1 + w2 + w3;
  __syncthreads();
  return base + incl - v;
}

// accumulator rows of the conv-epilogue BatchNorm statistics = partials that are folded afterwards.  64 until round 5; 16 since the conv
// launch's last workgroup folds and clears them in its own tail (conv_stats_done): every slot it reads and zeroes lengthens the launch
// (64 slots: +7 us per conv at 64 / 128 columns), while 4 x more fp64 atomics per address (~60 of ~10 ns) cost nothing measurable.
constexpr int kStatSlots = 16;

int bn_column_sums(const float* x, int64_t n, int c, double* stats, hipStream_t stream);   // batchnorm.hip
int bn_backward_sums(const float* x, const float* dy, int64_t n, int c, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, int relu, double* stats, hipStream_t stream);

// ---- internal primitives (sort_scan.hip) ---------------------------------------------
size_t scan_ws_bytes(int64_t n);
// Exclusive prefix sum of int32; in == out allowed. total (device int*, may be null) gets the sum.
int exclusive_scan_i32(const int* in, int* out, int64_t n, int* total, void* ws, size_t ws_bytes,
                       hipStream_t stream);
size_t radix_sort_ws_bytes(int64_t n);
// Stable LSD radix sort of 64-bit words on bits [bit_lo, bit_hi). Result ends in `keys`;
// `tmp` is an n-word ping-pong buffer.
int radix_sort_u64(uint64_t* keys, uint64_t* tmp, int64_t n, int bit_lo, int bit_hi, void* ws,
                   size_t ws_bytes, hipStream_t stream);

static inline uint32_t next_pow2(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return static_cast<uint32_t>(p);
}
static inline int bits_for(uint64_t max_value) {  // number of bits needed to hold max_value
  int b = 0;
  while (max_value) { ++b; max_value >>= 1; }
  return b < 1 ? 1 : b;
}

}  // namespace fv2p
