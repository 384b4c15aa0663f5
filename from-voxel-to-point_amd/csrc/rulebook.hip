// A3/A4 — hashed rulebook builder for submanifold / strided / transposed sparse conv.
//
// Replaces the reference's dense-grid builders (pcdet/ops/spconv/include/spconv/spconv_ops.h:27-140,
// src/indice_cuda.cu:29-135, include/spconv/indice.cu.h:22-203): instead of a B*Z*Y*X int grid
// (1.48 GB per call at KITTI level 1) the active set lives in an open-addressing table of packed
// {flat index : 40, row : 24} words, and the rulebook is produced as two dense neighbour tables
//     tab_in [K][n_in ] : output row reached from input row i through kernel offset k (or -1)
//     tab_out[K][n_out] : input row that feeds output row o through kernel offset k   (or -1)
// which is what the fused output-stationary conv kernels consume (every (k, o) has at most one
// input, every (k, i) at most one output).  The reference's pair lists [K,2,n_in] are derived
// from tab_in by an ordered stream compaction (canonical order: ascending input row).
//
// Output rows of a strided/transposed conv are the sorted distinct flat output indices
// (b*V + (z*Y + y)*X + x), exactly the order torch::_unique gives the GPU reference
// (spconv_ops.h:130-131): candidates are de-duplicated in a hash set, radix sorted, ranked.
// Neighbour enumeration and the kernel-offset formula follow geometry.h:24-142 literally
// (C integer division, the `m * (...) / dilation` precedence included).
#include "common.hpp"

namespace fv2p {

template <int ND>
struct RbGeomT {
  int in_shape[ND], out_shape[ND], k[ND], s[ND], p[ND], d[ND], E[ND];
  int kvol, emax, transpose;
  long long out_vol;
};
using RbGeom = RbGeomT<3>;   // the 2-D / 3-D rulebooks (2-D: a leading dimension of size 1); RbGeomT<4>: SparseConv4d / SubMConv4d

// Enumerated candidate e of input position `in` -> output position + kernel offset (geometry.h:24-142), any number of dimensions
template <int ND>
__device__ __forceinline__ bool enum_out(const RbGeomT<ND>& g, const int* in, int e, int* out, int* offset) {
  int lower[ND], upper[ND], cs[ND];
  int total = 1;
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    if (g.transpose) {
      lower[j] = in[j] * g.s[j] - g.p[j];
      upper[j] = lower[j] + (g.k[j] - 1) * g.d[j];
    } else {
      lower[j] = (in[j] - (g.k[j] - 1) * g.d[j] - 1 + g.s[j] + g.p[j]) / g.s[j];
      upper[j] = (in[j] + g.p[j]) / g.s[j];
    }
    cs[j] = (upper[j] - lower[j]) / g.d[j] + 1;
    if (cs[j] <= 0) return false;
    total *= cs[j];
  }
  if (e >= total) return false;
  int c[ND];
#pragma unroll
  for (int j = ND - 1; j > 0; --j) { c[j] = e % cs[j]; e /= cs[j]; }
  c[0] = e;
  bool valid = true;
  int m = 1, off = 0;
#pragma unroll
  for (int j = ND - 1; j >= 0; --j) {
    const int val = upper[j] - c[j] * g.d[j];
    out[j] = val;
    if (val < 0 || val > g.out_shape[j] - 1) valid = false;
    if (g.transpose) off += m * (val - lower[j]) / g.d[j];
    else off += m * (in[j] - val * g.s[j] + g.p[j]) / g.d[j];
    m *= g.k[j];
  }
  *offset = off;
  return valid;
}

template <int ND>
__device__ __forceinline__ uint64_t flat_key_nd(int b, const int* pos, const int* shape, long long vol) {
  uint64_t cell = static_cast<uint64_t>(pos[0]);
#pragma unroll
  for (int j = 1; j < ND; ++j) cell = cell * shape[j] + pos[j];
  return static_cast<uint64_t>(b) * vol + cell;
}
__device__ __forceinline__ uint64_t flat_key(int b, const int pos[3], const int shape[3], long long vol) { return flat_key_nd<3>(b, pos, shape, vol); }
// row i of the index tensor [n][1 + ND]: batch, position (3-D rows are one 16-byte load)
template <int ND>
__device__ __forceinline__ int load_row(const int* __restrict__ ind, int i, int* pos) {
  if constexpr (ND == 3) {
    const int4 c = reinterpret_cast<const int4*>(ind)[i];
    pos[0] = c.y; pos[1] = c.z; pos[2] = c.w;
    return c.x;
  } else {
    const int* r = ind + static_cast<int64_t>(i) * (ND + 1);
#pragma unroll
    for (int j = 0; j < ND; ++j) pos[j] = r[1 + j];
    return r[0];
  }
}

// subM: table key(position of row i) -> i ; duplicates: highest row wins (geometry.h:275-280).
template <int ND>
__global__ void rb_insert_rows(const int* __restrict__ ind, int n, RbGeomT<ND> g, uint64_t* __restrict__ table, uint32_t mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int pos[ND];
  const int b = load_row<ND>(ind, i, pos);
  const uint64_t key = flat_key_nd<ND>(b, pos, g.out_shape, g.out_vol);
  const uint64_t word = slot_pack(key, static_cast<uint32_t>(i));
  uint32_t h = hash_u64(key, mask);
  while (true) {
    unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[h]), (unsigned long long)kEmptySlot,
                                       (unsigned long long)word);
    if (old == kEmptySlot) break;
    if (slot_key(old) == key) { atomicMax(reinterpret_cast<unsigned long long*>(&table[h]), (unsigned long long)word); break; }
    h = (h + 1) & mask;
  }
}

// strided / transposed: hash-set of candidate output keys; the inserting thread appends the key to uniq[].
template <int ND>
__global__ void rb_insert_outputs(const int* __restrict__ ind, int n, RbGeomT<ND> g, uint64_t* __restrict__ table, uint32_t mask,
                                  uint64_t* __restrict__ uniq, int* __restrict__ n_uniq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = blockIdx.y;
  bool fresh = false;
  uint64_t key = 0;
  if (i < n) {
    int in[ND];
    const int b = load_row<ND>(ind, i, in);
    int out[ND], off;
    if (enum_out<ND>(g, in, e, out, &off)) {
      key = flat_key_nd<ND>(b, out, g.out_shape, g.out_vol);
      const uint64_t word = slot_pack(key, static_cast<uint32_t>(kValMask));
      uint32_t h = hash_u64(key, mask);
      while (true) {
        // plain (L2-coherent) read first: most candidates are duplicates of a key that is already present
        unsigned long long old = __hip_atomic_load(reinterpret_cast<unsigned long long*>(&table[h]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == kEmptySlot)
          old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[h]), (unsigned long long)kEmptySlot, (unsigned long long)word);
        if (old == kEmptySlot) { fresh = true; break; }
        if (slot_key(old) == key) break;
        h = (h + 1) & mask;
      }
    }
  }
  // one atomic per wave: the lanes that inserted a new key take consecutive slots of uniq[]
  const uint64_t vote = __ballot(fresh);
  if (vote) {
    int base = 0;
    const int leader = __ffsll(static_cast<long long>(vote)) - 1;
    if ((threadIdx.x & 63) == leader) base = atomicAdd(n_uniq, __popcll(vote));
    base = __shfl(base, leader, 64);
    if (fresh) uniq[base + __popcll(vote & lanemask_lt())] = key;
  }
}

// rank r -> table payload, decoded output coordinates
template <int ND>
__global__ void rb_assign_outputs(const uint64_t* __restrict__ uniq, int n_out, RbGeomT<ND> g, uint64_t* __restrict__ table,
                                  uint32_t mask, int* __restrict__ out_ind) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_out) return;
  const uint64_t key = uniq[r];
  uint32_t h = hash_u64(key, mask);
  while (slot_key(table[h]) != key) h = (h + 1) & mask;
  table[h] = slot_pack(key, static_cast<uint32_t>(r));
  uint64_t rem = key % static_cast<uint64_t>(g.out_vol);
  int o[ND + 1];
  o[0] = static_cast<int>(key / static_cast<uint64_t>(g.out_vol));
#pragma unroll
  for (int j = ND - 1; j > 0; --j) { o[1 + j] = static_cast<int>(rem % g.out_shape[j]); rem /= g.out_shape[j]; }
  o[1] = static_cast<int>(rem);
  if constexpr (ND == 3) reinterpret_cast<int4*>(out_ind)[r] = make_int4(o[0], o[1], o[2], o[3]);
  else {
#pragma unroll
    for (int j = 0; j <= ND; ++j) out_ind[static_cast<int64_t>(r) * (ND + 1) + j] = o[j];
  }
}

// one thread per (input row, candidate): probe and fill the neighbour tables
template <int ND>
__global__ void rb_fill_tables(const int* __restrict__ ind, int n_in, int n_out, RbGeomT<ND> g, const uint64_t* __restrict__ table,
                               uint32_t mask, int* __restrict__ tab_in, int* __restrict__ tab_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = blockIdx.y;
  if (i >= n_in) return;
  int in[ND];
  const int b = load_row<ND>(ind, i, in);
  int out[ND], off;
  if (!enum_out<ND>(g, in, e, out, &off)) return;
  const int o = table_find(table, mask, flat_key_nd<ND>(b, out, g.out_shape, g.out_vol));
  if (o < 0) return;
  tab_in[static_cast<int64_t>(off) * n_in + i] = o;
  if (tab_out) tab_out[static_cast<int64_t>(off) * n_out + o] = i;
}

// ---- strided / transposed rulebook through a bitmap of the output grid -----------------------------------------------------
// When the output grid is small enough to afford one bit per cell (batch * volume <= 2^28: 32 MB; every KITTI / Waymo level
// is below 2^26) the distinct output cells need neither a hash set nor a sort: candidates set their bit, an exclusive
// popcount scan over the words ranks the cells in ascending flat index — the order torch::_unique gives the GPU reference
// (spconv_ops.h:130-131) — and a neighbour's output row is prefix[word] + popc(bits below): two loads instead of a probe.
// Replaces {hash-set insert, 4 radix passes, rank write-back} = 11 launches by {mark, tile sums, scan of sums, emit} = 4.
constexpr int kBmTile = 256;             // words per workgroup in the popcount passes: one per thread (a dense word is 32 serial steps)
constexpr long long kBmMaxCells = 1ll << 28;

__global__ void rb_mark_outputs(const int* __restrict__ ind, int n, int batch, RbGeom g, uint32_t* __restrict__ bitmap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = blockIdx.y;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(ind)[i];
  if (static_cast<unsigned>(c.x) >= static_cast<unsigned>(batch)) return;   // a batch index outside the bitmap: the row takes no part
  const int in[3] = {c.y, c.z, c.w};
  int out[3], off;
  if (!enum_out<3>(g, in, e, out, &off)) return;
  const uint64_t key = flat_key(c.x, out, g.out_shape, g.out_vol);
  const uint32_t bit = 1u << (key & 31);
  uint32_t* wp = bitmap + (key >> 5);
  if (!(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(wp, bit);   // most candidates are repeats
}
// sums[b] = number of set bits in tile b
__global__ __launch_bounds__(256) void rb_tile_popc(const uint32_t* __restrict__ bitmap, int64_t words, int* __restrict__ sums) {
  __shared__ int ws[4];
  const int64_t wi = static_cast<int64_t>(blockIdx.x) * kBmTile + threadIdx.x;
  int c = wi < words ? __popc(bitmap[wi]) : 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// prefix[w] = set bits before word w (tile offset from the scanned sums + scan inside the tile); out_ind[row] = decoded
// cell.  One word per lane: the serial part is the <= 32 bits of a word, neighbouring lanes hold neighbouring words of
// similar density, a cell is decoded with three 32-bit divisions per non-empty word plus carries per bit.  (Per-thread
// runs of 8 words with two 64-bit divisions per bit took 100 us on any grid; 2048-word tiles in 8 rounds still 25 us.)
__global__ __launch_bounds__(256) void rb_emit_outputs(const uint32_t* __restrict__ bitmap, int64_t words, const int* __restrict__ tile_off,
                                                       RbGeom g, int* __restrict__ prefix, int* __restrict__ out_ind) {
  __shared__ int lds_wave[4];
  const uint32_t W = g.out_shape[2], H = g.out_shape[1], D = g.out_shape[0];
  const uint32_t HW = H * W, vol = static_cast<uint32_t>(g.out_vol);   // cells <= 2^28: everything fits 32 bits
  const int64_t wi = static_cast<int64_t>(blockIdx.x) * kBmTile + threadIdx.x;
  uint32_t bits = wi < words ? bitmap[wi] : 0u;
  int tot;
  int run = tile_off[blockIdx.x] + block_excl_scan_256(__popc(bits), lds_wave, &tot);
  if (wi < words) prefix[wi] = run;
  if (bits) {
    const uint32_t key = static_cast<uint32_t>(wi) << 5;
    const uint32_t b0 = key / vol, r1 = key - b0 * vol;
    const uint32_t z0 = r1 / HW, r2 = r1 - z0 * HW;
    const uint32_t y0 = r2 / W, x0 = r2 - y0 * W;
    do {
      const int bit = __ffs(static_cast<int>(bits)) - 1;
      bits &= bits - 1;
      uint32_t x = x0 + bit, y = y0, z = z0, b = b0;
      while (x >= W) {
        x -= W;
        if (++y == H) { y = 0; if (++z == D) { z = 0; ++b; } }
      }
      reinterpret_cast<int4*>(out_ind)[run++] = make_int4((int)b, (int)z, (int)y, (int)x);
    } while (bits);
  }
}
__global__ void rb_fill_tables_bm(const int* __restrict__ ind, int n_in, int n_out, int batch, RbGeom g, const uint32_t* __restrict__ bitmap,
                                  const int* __restrict__ prefix, int* __restrict__ tab_in, int* __restrict__ tab_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = blockIdx.y;
  if (i >= n_in) return;
  const int4 c = reinterpret_cast<const int4*>(ind)[i];
  if (static_cast<unsigned>(c.x) >= static_cast<unsigned>(batch)) return;
  const int in[3] = {c.y, c.z, c.w};
  int out[3], off;
  if (!enum_out<3>(g, in, e, out, &off)) return;
  const uint64_t key = flat_key(c.x, out, g.out_shape, g.out_vol);
  const uint32_t word = bitmap[key >> 5], bit = 1u << (key & 31);
  if (!(word & bit)) return;
  const int o = prefix[key >> 5] + __popc(word & (bit - 1u));
  tab_in[static_cast<int64_t>(off) * n_in + i] = o;
  if (tab_out) tab_out[static_cast<int64_t>(off) * n_out + o] = i;
}

// indice_num[k] = #valid entries of tab[k][:]
__global__ void rb_count_rows(const int* __restrict__ tab, int n, int* __restrict__ num) {
  const int k = blockIdx.y;
  int cnt = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    cnt += tab[static_cast<int64_t>(k) * n + i] >= 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&num[k], cnt);
}

// ---- reference-format pair lists: ordered compaction of tab_in[k][:] -----------------------------
constexpr int kPairTile = 1024;  // rows per workgroup (256 threads x 4)
__global__ __launch_bounds__(256) void rb_pair_counts(const int* __restrict__ tab, int n, int nblk, int* __restrict__ cnt) {
  const int k = blockIdx.y;
  const int base = blockIdx.x * kPairTile;
  int c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = base + j * 256 + threadIdx.x;
    c += (i < n) && tab[static_cast<int64_t>(k) * n + i] >= 0;
  }
  __shared__ int ws[4];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) cnt[k * nblk + blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// offs = exclusive scan of cnt over the flattened [K][nblk] array; row start of k = offs[k*nblk]
__global__ __launch_bounds__(256) void rb_pair_write(const int* __restrict__ tab, int n, int nblk, const int* __restrict__ offs,
                                                     int* __restrict__ pairs, int64_t ld /* row length of pairs */,
                                                     int* __restrict__ pair_num /* [K] or null */) {
  const int k = blockIdx.y;
  const int base = blockIdx.x * kPairTile;
  __shared__ int wave_cnt[4];
  __shared__ int run;
  if (threadIdx.x == 0) {
    run = offs[k * nblk + blockIdx.x] - offs[k * nblk];
    if (pair_num && blockIdx.x == 0) pair_num[k] = offs[(k + 1) * nblk] - offs[k * nblk];  // offs has K*nblk + 1 entries
  }
  __syncthreads();
  const int w = threadIdx.x >> 6;
  for (int j = 0; j < 4; ++j) {
    const int i = base + j * 256 + threadIdx.x;
    const int o = (i < n) ? tab[static_cast<int64_t>(k) * n + i] : -1;
    const uint64_t vote = __ballot(o >= 0);
    if ((threadIdx.x & 63) == 0) wave_cnt[w] = __popcll(vote);
    __syncthreads();
    int pos = run + __popcll(vote & lanemask_lt());
    for (int ww = 0; ww < w; ++ww) pos += wave_cnt[ww];
    if (o >= 0) {
      pairs[(static_cast<int64_t>(k) * 2 + 0) * ld + pos] = i;
      pairs[(static_cast<int64_t>(k) * 2 + 1) * ld + pos] = o;
    }
    __syncthreads();
    if (threadIdx.x == 0) run += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}

// pair lists -> tables (for rulebooks supplied in the reference format)
__global__ void rb_pairs_to_tables(const int* __restrict__ pairs, const int* __restrict__ num, int64_t ld, int n_in, int n_out,
                                   int* __restrict__ tab_in, int* __restrict__ tab_out) {
  const int k = blockIdx.y;
  const int cnt = num[k];
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < cnt; s += gridDim.x * blockDim.x) {
    const int i = pairs[(static_cast<int64_t>(k) * 2 + 0) * ld + s];
    const int o = pairs[(static_cast<int64_t>(k) * 2 + 1) * ld + s];
    if (i < 0 || o < 0 || i >= n_in || o >= n_out) continue;
    if (tab_in) tab_in[static_cast<int64_t>(k) * n_in + i] = o;
    if (tab_out) tab_out[static_cast<int64_t>(k) * n_out + o] = i;
  }
}

struct RbWs {
  uint64_t* table; uint32_t cap;
  uint64_t* uniq; uint64_t* tmp; int* n_uniq;
  char* aux; size_t aux_bytes;
};

template <int ND>
static int rb_geom(RbGeomT<ND>* g, const int* in_shape, const int* out_shape, const int* ksize, const int* stride,
                   const int* padding, const int* dilation, int subm, int transpose) {
  g->kvol = 1; g->emax = 1; g->out_vol = 1; g->transpose = transpose && !subm;
  for (int j = 0; j < ND; ++j) {
    FV2P_REQUIRE(ksize[j] >= 1 && stride[j] >= 1 && dilation[j] >= 1 && in_shape[j] >= 1 && out_shape[j] >= 1, FV2P_EINVAL,
                 "rulebook: bad geometry in dim %d", j);
    g->in_shape[j] = in_shape[j]; g->out_shape[j] = out_shape[j]; g->k[j] = ksize[j]; g->d[j] = dilation[j];
    if (subm) { g->s[j] = 1; g->p[j] = ksize[j] / 2; }   // spconv_ops.h:76-80
    else { g->s[j] = stride[j]; g->p[j] = padding[j]; }
    g->E[j] = g->transpose ? ksize[j] : ((ksize[j] - 1) * dilation[j] / g->s[j]) / dilation[j] + 1;
    g->kvol *= ksize[j]; g->emax *= g->E[j]; g->out_vol *= out_shape[j];
  }
  FV2P_REQUIRE(g->kvol <= 4096, FV2P_ELIMIT, "rulebook: kernel volume > 4096 (spconv_ops.h:52)");
  return 0;
}

template <typename C>
static void rb_carve(C& c, int64_t n_in, int emax, int subm, RbWs* w) {
  const uint64_t items = subm ? static_cast<uint64_t>(n_in) : static_cast<uint64_t>(n_in) * emax;
  const uint32_t cap = next_pow2((items > 512 ? items : 512) * 2);
  const int64_t nu = subm ? 1 : static_cast<int64_t>(items);
  const size_t aux = radix_sort_ws_bytes(nu);
  uint64_t* table = c.template take<uint64_t>(cap);
  int* n_uniq = c.template take<int>(4);
  uint64_t* uniq = c.template take<uint64_t>(nu);
  uint64_t* tmp = c.template take<uint64_t>(nu);
  char* auxp = c.template take<char>(aux);
  if (w) { w->table = table; w->cap = cap; w->n_uniq = n_uniq; w->uniq = uniq; w->tmp = tmp; w->aux = auxp; w->aux_bytes = aux; }
}
struct SizerC : Sizer { template <typename T> T* take(size_t n) { Sizer::take<T>(n); return nullptr; } };

struct BmWs {
  uint32_t* bitmap; int* prefix; int* sums; int64_t words; int tiles;
  char* aux; size_t aux_bytes;
};
static bool bm_applies(long long cells, int subm) { return !subm && cells > 0 && cells <= kBmMaxCells; }
template <typename C>
static void bm_carve(C& c, long long cells, BmWs* w) {
  const int64_t words = (cells + 31) / 32;
  const int tiles = static_cast<int>(ceil_div(words, kBmTile));
  const size_t aux = scan_ws_bytes(tiles + 1);
  uint32_t* bitmap = c.template take<uint32_t>(static_cast<size_t>(words));
  int* prefix = c.template take<int>(static_cast<size_t>(words));
  int* sums = c.template take<int>(static_cast<size_t>(tiles) + 1);
  char* auxp = c.template take<char>(aux);
  if (w) { w->bitmap = bitmap; w->prefix = prefix; w->sums = sums; w->words = words; w->tiles = tiles; w->aux = auxp; w->aux_bytes = aux; }
}

// The hash-set + sort rulebook of any dimension (the bitmap path above is 3-D only): begin = candidate outputs into the set (submanifold:
// the rows themselves) and the host-side count, finish = sort, rank, tables.
template <int ND>
static int rulebook_begin_hashed(const int* indices, int64_t n_in, const RbGeomT<ND>& g, int subm, int64_t* n_out_host, void* ws, size_t ws_bytes,
                                 hipStream_t stream) {
  Carver c(ws, ws_bytes);
  RbWs w;
  rb_carve(c, n_in, g.emax, subm, &w);
  FillJobs fill;
  fill.add(w.table, sizeof(uint64_t) * w.cap, 0xFFFFFFFFu);
  if (!subm) fill.add(w.n_uniq, sizeof(int) * 4, 0u);
  if (int rc = multi_fill(fill, stream)) return rc;
  const int T = 256;
  const unsigned nb = static_cast<unsigned>(ceil_div(n_in, T));
  if (subm) {
    hipLaunchKernelGGL(rb_insert_rows<ND>, dim3(nb), dim3(T), 0, stream, indices, (int)n_in, g, w.table, w.cap - 1);
    FV2P_LAUNCH_CHECK();
    *n_out_host = n_in;
    return 0;
  }
  FV2P_REQUIRE(static_cast<int64_t>(n_in) * g.emax <= kMaxRows, FV2P_ELIMIT, "rulebook: too many candidate outputs");
  hipLaunchKernelGGL(rb_insert_outputs<ND>, dim3(nb, g.emax), dim3(T), 0, stream, indices, (int)n_in, g, w.table, w.cap - 1, w.uniq,
                     w.n_uniq);
  FV2P_LAUNCH_CHECK();
  int n_out = 0;
  FV2P_HIP(hipMemcpyAsync(&n_out, w.n_uniq, sizeof(int), hipMemcpyDeviceToHost, stream));
  // Output row count = a host-side shape (the reference synchronises at the same place, spconv_ops.h:131-139).  Blocking
  // wait on purpose: polling hipStreamQuery returns ~85 us sooner when nothing else runs, but its spinning contends with
  // the training thread's launches inside the runtime and the whole step got 15 % slower (measured).
  FV2P_HIP(hipStreamSynchronize(stream));
  *n_out_host = n_out;
  return 0;
}

template <int ND>
static int rulebook_finish_hashed(const int* indices, int64_t n_in, int batch, const RbGeomT<ND>& g, int subm, int64_t n_out, int* out_indices,
                                  int* tab_in, int* tab_out, int* indice_num, FillJobs& fill, void* ws, size_t ws_bytes, hipStream_t stream) {
  Carver c(ws, ws_bytes);
  RbWs w;
  rb_carve(c, n_in, g.emax, subm, &w);
  const int T = 256;
  const unsigned nb = static_cast<unsigned>(ceil_div(n_in, T));
  fill.add(tab_in, sizeof(int) * (size_t)g.kvol * n_in, 0xFFFFFFFFu);
  if (tab_out && n_out > 0) fill.add(tab_out, sizeof(int) * (size_t)g.kvol * n_out, 0xFFFFFFFFu);
  if (int rc = multi_fill(fill, stream)) return rc;
  if (!subm) {
    FV2P_REQUIRE(out_indices || n_out == 0, FV2P_EINVAL, "rulebook_finish: out_indices is null");
    if (n_out > 0) {
      const int bits = bits_for(static_cast<uint64_t>(batch) * g.out_vol);
      if (int rc = radix_sort_u64(w.uniq, w.tmp, n_out, 0, bits, w.aux, w.aux_bytes, stream)) return rc;
      hipLaunchKernelGGL(rb_assign_outputs<ND>, dim3((unsigned)ceil_div(n_out, T)), dim3(T), 0, stream, w.uniq, (int)n_out, g, w.table,
                         w.cap - 1, out_indices);
    }
  } else {
    FV2P_REQUIRE(n_out == n_in, FV2P_EINVAL, "rulebook_finish: subm needs n_out == n_in");
  }
  hipLaunchKernelGGL(rb_fill_tables<ND>, dim3(nb, g.emax), dim3(T), 0, stream, indices, (int)n_in, (int)n_out, g, w.table, w.cap - 1,
                     tab_in, tab_out);
  if (indice_num) {
    const unsigned cb = static_cast<unsigned>(ceil_div(n_in, 1024) < 64 ? ceil_div(n_in, 1024) : 64);
    hipLaunchKernelGGL(rb_count_rows, dim3(cb, g.kvol), dim3(T), 0, stream, tab_in, (int)n_in, indice_num);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}

}  // namespace fv2p

using namespace fv2p;

extern "C" size_t fv2p_rulebook_ws_bytes(int64_t n_in, const int ksize[3], const int stride[3], const int dilation[3], int subm,
                                         int transpose) {
  RbGeom g;
  const int one[3] = {1, 1, 1}, zero[3] = {0, 0, 0};
  if (rb_geom<3>(&g, one, one, ksize, stride, zero, dilation, subm, transpose)) return 0;
  SizerC s;
  rb_carve(s, n_in > 0 ? n_in : 1, g.emax, subm, static_cast<RbWs*>(nullptr));
  return s.bytes();
}

extern "C" size_t fv2p_rulebook_ws_bytes_grid(int64_t n_in, int batch, const int out_shape[3], const int ksize[3], const int stride[3],
                                              const int dilation[3], int subm, int transpose) {
  const size_t hashed = fv2p_rulebook_ws_bytes(n_in, ksize, stride, dilation, subm, transpose);
  const long long cells = static_cast<long long>(batch > 0 ? batch : 1) * out_shape[0] * out_shape[1] * out_shape[2];
  if (!bm_applies(cells, subm)) return hashed;
  SizerC s;
  s.take<int>(4);
  bm_carve(s, cells, static_cast<BmWs*>(nullptr));
  return s.bytes() > hashed ? s.bytes() : hashed;
}

// the bitmap path is taken when it applies and the caller's workspace was sized for it (fv2p_rulebook_ws_bytes_grid)
static int g_rb_path = 0;   // 0 = bitmap when it applies, 1 = always the hash set + sort (tests compare the two)
extern "C" int fv2p_rulebook_set_path(int path) { g_rb_path = path; return 0; }
static bool use_bitmap(const RbGeom& g, int batch, int subm, size_t ws_bytes) {
  if (g_rb_path == 1) return false;
  const long long cells = static_cast<long long>(batch) * g.out_vol;
  if (!bm_applies(cells, subm)) return false;
  SizerC s;
  s.take<int>(4);
  bm_carve(s, cells, static_cast<BmWs*>(nullptr));
  return ws_bytes >= s.bytes();
}

extern "C" int fv2p_rulebook_begin(const int* indices, int64_t n_in, int batch, const int in_shape[3], const int out_shape[3],
                                   const int ksize[3], const int stride[3], const int padding[3], const int dilation[3], int subm,
                                   int transpose, int64_t* n_out_host, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  RbGeom g;
  if (int rc = rb_geom<3>(&g, in_shape, out_shape, ksize, stride, padding, dilation, subm, transpose)) return rc;
  FV2P_REQUIRE(n_out_host, FV2P_EINVAL, "rulebook_begin: n_out_host is null");
  FV2P_REQUIRE(n_in >= 0 && n_in <= kMaxRows, FV2P_ELIMIT, "rulebook: n_in=%lld outside [0, 2^24)", (long long)n_in);
  FV2P_REQUIRE(batch >= 1 && static_cast<double>(batch) * g.out_vol <= static_cast<double>(kMaxKey), FV2P_ELIMIT,
               "rulebook: batch*volume exceeds 2^40");
  if (subm)
    for (int j = 0; j < 3; ++j)
      FV2P_REQUIRE(in_shape[j] == out_shape[j], FV2P_EINVAL, "rulebook: subm needs out_shape == in_shape");
  if (n_in == 0) { *n_out_host = 0; return 0; }
  FV2P_REQUIRE(indices && ws && ws_bytes >= fv2p_rulebook_ws_bytes(n_in, ksize, stride, dilation, subm, transpose), FV2P_EWORKSPACE,
               "rulebook: workspace too small");
  if (use_bitmap(g, batch, subm, ws_bytes)) {
    FV2P_REQUIRE(static_cast<int64_t>(n_in) * g.emax <= kMaxRows, FV2P_ELIMIT, "rulebook: too many candidate outputs");
    Carver cb(ws, ws_bytes);
    int* total = cb.take<int>(4);
    BmWs b;
    bm_carve(cb, static_cast<long long>(batch) * g.out_vol, &b);
    FillJobs fill;
    fill.add(b.bitmap, sizeof(uint32_t) * static_cast<size_t>(b.words), 0u);
    fill.add(total, sizeof(int) * 4, 0u);
    if (int rc = multi_fill(fill, stream)) return rc;
    hipLaunchKernelGGL(rb_mark_outputs, dim3(static_cast<unsigned>(ceil_div(n_in, 256)), g.emax), dim3(256), 0, stream, indices, (int)n_in, batch, g,
                       b.bitmap);
    hipLaunchKernelGGL(rb_tile_popc, dim3(b.tiles), dim3(256), 0, stream, b.bitmap, b.words, b.sums);
    if (int rc = exclusive_scan_i32(b.sums, b.sums, b.tiles, total, b.aux, b.aux_bytes, stream)) return rc;
    int n_out = 0;
    FV2P_HIP(hipMemcpyAsync(&n_out, total, sizeof(int), hipMemcpyDeviceToHost, stream));
    FV2P_HIP(hipStreamSynchronize(stream));  // output row count = host-side shape (spconv_ops.h:131-139), see below
    *n_out_host = n_out;
    return 0;
  }
  return rulebook_begin_hashed<3>(indices, n_in, g, subm, n_out_host, ws, ws_bytes, stream);
}

extern "C" int fv2p_rulebook_finish(const int* indices, int64_t n_in, int batch, const int in_shape[3], const int out_shape[3],
                                    const int ksize[3], const int stride[3], const int padding[3], const int dilation[3], int subm,
                                    int transpose, int64_t n_out, int* out_indices, int* tab_in, int* tab_out, int* indice_num,
                                    void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  RbGeom g;
  if (int rc = rb_geom<3>(&g, in_shape, out_shape, ksize, stride, padding, dilation, subm, transpose)) return rc;
  FillJobs fill;
  if (indice_num) fill.add(indice_num, sizeof(int) * g.kvol, 0u);
  if (n_in == 0) return multi_fill(fill, stream);
  FV2P_REQUIRE(tab_in, FV2P_EINVAL, "rulebook_finish: null tab_in");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_rulebook_ws_bytes(n_in, ksize, stride, dilation, subm, transpose), FV2P_EWORKSPACE,
               "rulebook: workspace too small");
  if (use_bitmap(g, batch, subm, ws_bytes)) {
    Carver cb(ws, ws_bytes);
    cb.take<int>(4);
    BmWs b;
    bm_carve(cb, static_cast<long long>(batch) * g.out_vol, &b);
    fill.add(tab_in, sizeof(int) * (size_t)g.kvol * n_in, 0xFFFFFFFFu);
    if (tab_out && n_out > 0) fill.add(tab_out, sizeof(int) * (size_t)g.kvol * n_out, 0xFFFFFFFFu);
    if (int rc = multi_fill(fill, stream)) return rc;
    FV2P_REQUIRE(out_indices || n_out == 0, FV2P_EINVAL, "rulebook_finish: out_indices is null");
    if (n_out > 0) {
      hipLaunchKernelGGL(rb_emit_outputs, dim3(b.tiles), dim3(256), 0, stream, b.bitmap, b.words, b.sums, g, b.prefix, out_indices);
      hipLaunchKernelGGL(rb_fill_tables_bm, dim3(static_cast<unsigned>(ceil_div(n_in, 256)), g.emax), dim3(256), 0, stream, indices, (int)n_in,
                         (int)n_out, batch, g, b.bitmap, b.prefix, tab_in, tab_out);
    }
    if (indice_num) {
      const unsigned cbk = static_cast<unsigned>(ceil_div(n_in, 1024) < 64 ? ceil_div(n_in, 1024) : 64);
      hipLaunchKernelGGL(rb_count_rows, dim3(cbk, g.kvol), dim3(256), 0, stream, tab_in, (int)n_in, indice_num);
    }
    FV2P_LAUNCH_CHECK();
    return 0;
  }
  return rulebook_finish_hashed<3>(indices, n_in, batch, g, subm, n_out, out_indices, tab_in, tab_out, indice_num, fill, ws, ws_bytes, stream);
}

// ---- 4-D rulebooks (SparseConv4d / SubMConv4d; spconv_ops.h:143-258 getIndicePair 4-D instantiations, all.cc:22-33) -------------------
// The same hash set + sort on rows of five ints (batch, four coordinates); no bitmap path and no transposed form (the reference has no
// SparseConvTranspose4d, conv.py:233-480).
extern "C" size_t fv2p_rulebook4d_ws_bytes(int64_t n_in, const int ksize[4], const int stride[4], const int dilation[4], int subm) {
  RbGeomT<4> g;
  const int one[4] = {1, 1, 1, 1}, zero[4] = {0, 0, 0, 0};
  if (rb_geom<4>(&g, one, one, ksize, stride, zero, dilation, subm, 0)) return 0;
  SizerC s;
  rb_carve(s, n_in > 0 ? n_in : 1, g.emax, subm, static_cast<RbWs*>(nullptr));
  return s.bytes();
}

extern "C" int fv2p_rulebook4d_begin(const int* indices, int64_t n_in, int batch, const int in_shape[4], const int out_shape[4],
                                     const int ksize[4], const int stride[4], const int padding[4], const int dilation[4], int subm,
                                     int64_t* n_out_host, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  RbGeomT<4> g;
  if (int rc = rb_geom<4>(&g, in_shape, out_shape, ksize, stride, padding, dilation, subm, 0)) return rc;
  FV2P_REQUIRE(n_out_host, FV2P_EINVAL, "rulebook4d_begin: n_out_host is null");
  FV2P_REQUIRE(n_in >= 0 && n_in <= kMaxRows, FV2P_ELIMIT, "rulebook4d: n_in=%lld outside [0, 2^24)", (long long)n_in);
  double cells = static_cast<double>(batch);
  for (int j = 0; j < 4; ++j) cells *= out_shape[j];
  FV2P_REQUIRE(batch >= 1 && cells <= static_cast<double>(kMaxKey), FV2P_ELIMIT, "rulebook4d: batch*volume exceeds 2^40");
  if (subm)
    for (int j = 0; j < 4; ++j)
      FV2P_REQUIRE(in_shape[j] == out_shape[j], FV2P_EINVAL, "rulebook4d: subm needs out_shape == in_shape");
  if (n_in == 0) { *n_out_host = 0; return 0; }
  FV2P_REQUIRE(indices && ws && ws_bytes >= fv2p_rulebook4d_ws_bytes(n_in, ksize, stride, dilation, subm), FV2P_EWORKSPACE,
               "rulebook4d: workspace too small");
  return rulebook_begin_hashed<4>(indices, n_in, g, subm, n_out_host, ws, ws_bytes, stream);
}

extern "C" int fv2p_rulebook4d_finish(const int* indices, int64_t n_in, int batch, const int in_shape[4], const int out_shape[4],
                                      const int ksize[4], const int stride[4], const int padding[4], const int dilation[4], int subm,
                                      int64_t n_out, int* out_indices, int* tab_in, int* tab_out, int* indice_num, void* ws, size_t ws_bytes,
                                      fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  RbGeomT<4> g;
  if (int rc = rb_geom<4>(&g, in_shape, out_shape, ksize, stride, padding, dilation, subm, 0)) return rc;
  FillJobs fill;
  if (indice_num) fill.add(indice_num, sizeof(int) * g.kvol, 0u);
  if (n_in == 0) return multi_fill(fill, stream);
  FV2P_REQUIRE(tab_in, FV2P_EINVAL, "rulebook4d_finish: null tab_in");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_rulebook4d_ws_bytes(n_in, ksize, stride, dilation, subm), FV2P_EWORKSPACE,
               "rulebook4d: workspace too small");
  return rulebook_finish_hashed<4>(indices, n_in, batch, g, subm, n_out, out_indices, tab_in, tab_out, indice_num, fill, ws, ws_bytes, stream);
}

// ---- row order for the backward-data conv of a strided layer -------------------------------------------------------------
// Input rows grouped by the residue class of (coordinate + padding) mod stride: rows of one class can only reach the same
// few kernel offsets (sparse_conv.hip, conv_rows_act).  Stable counting sort in two launches, no atomics: workgroup b
// counts the classes of its 256 rows into hist[b][8]; then every workgroup sums the (small) table into its class bases
// and ranks its rows with per-class wave ballots.  (One workgroup walking contiguous runs per thread was 70 us at 45 k
// rows: ~90 instructions per row on a single CU.)
constexpr int kPermClasses = 8;
__device__ __forceinline__ int perm_class(const int* __restrict__ ind, int i, int sy, int sx, int mz, int my, int mx, int pz, int py, int px) {
  const int4 c = reinterpret_cast<const int4*>(ind)[i];
  return ((((c.y + pz) & mz) * sy + ((c.z + py) & my)) * sx) + ((c.w + px) & mx);   // strides are 1 or 2: residues by mask
}
__global__ __launch_bounds__(256) void rb_class_hist(const int* __restrict__ ind, int n, int sy, int sx, int mz, int my, int mx, int pz, int py, int px,
                                                     int* __restrict__ hist) {
  __shared__ int h[4][kPermClasses];
  const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = i < n ? perm_class(ind, i, sy, sx, mz, my, mx, pz, py, px) : -1;
#pragma unroll
  for (int q = 0; q < kPermClasses; ++q) {
    const unsigned long long b = __ballot(c == q);
    if (lane == 0) h[w][q] = __popcll(b);
  }
  __syncthreads();
  if (threadIdx.x < kPermClasses) hist[blockIdx.x * kPermClasses + threadIdx.x] = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void rb_class_scatter(const int* __restrict__ ind, int n, int sy, int sx, int mz, int my, int mx, int pz, int py,
                                                        int px, const int* __restrict__ hist, int nblk, int* __restrict__ perm) {
  __shared__ int part[256][2];          // per thread: rows of class q in all / in earlier workgroups
  __shared__ int base[kPermClasses], wcount[4][kPermClasses], tot[2][kPermClasses];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  // thread (q = t % 8, s = t / 8) sums hist[b][q] over b = s, s + 32, ...
  const int q = t & 7, s = t >> 3;
  int all = 0, before = 0;
  for (int b = s; b < nblk; b += 32) {
    const int v = hist[b * kPermClasses + q];
    all += v;
    if (b < static_cast<int>(blockIdx.x)) before += v;
  }
  part[t][0] = all; part[t][1] = before;
  __syncthreads();
  if (t < 2 * kPermClasses) {   // 16 threads fold the 32 slices of (class, all | before), then 8 take the prefix over the classes
    const int q2 = t & 7, which = t >> 3;
    int v = 0;
    for (int s2 = 0; s2 < 32; ++s2) v += part[s2 * 8 + q2][which];
    tot[which][q2] = v;
  }
  __syncthreads();
  if (t < kPermClasses) {
    int lower = 0;
    for (int q2 = 0; q2 < t; ++q2) lower += tot[0][q2];
    base[t] = lower + tot[1][t];
  }
  const int i = blockIdx.x * 256 + t;
  const int c = i < n ? perm_class(ind, i, sy, sx, mz, my, mx, pz, py, px) : -1;
  int rank = 0;
#pragma unroll
  for (int q2 = 0; q2 < kPermClasses; ++q2) {
    const unsigned long long b = __ballot(c == q2);
    if (c == q2) rank = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wcount[w][q2] = __popcll(b);
  }
  __syncthreads();
  if (c >= 0) {
    int pos = base[c] + rank;
    for (int w2 = 0; w2 < w; ++w2) pos += wcount[w2][c];
    perm[pos] = i;
  }
}

extern "C" size_t fv2p_rulebook_class_perm_ws_bytes(int64_t n) { return static_cast<size_t>(ceil_div(n > 0 ? n : 1, 256)) * kPermClasses * sizeof(int); }

extern "C" int fv2p_rulebook_class_perm(const int* indices, int64_t n, const int stride[3], const int padding[3], int* perm, void* ws,
                                        size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(n >= 0 && n <= kMaxRows && (n == 0 || (indices && perm && ws)), FV2P_EINVAL, "rulebook_class_perm: bad arguments");
  for (int j = 0; j < 3; ++j)
    FV2P_REQUIRE((stride[j] == 1 || stride[j] == 2) && padding[j] >= 0, FV2P_ELIMIT, "rulebook_class_perm: strides 1 or 2, padding >= 0");
  FV2P_REQUIRE(ws_bytes >= fv2p_rulebook_class_perm_ws_bytes(n), FV2P_EWORKSPACE, "rulebook_class_perm: workspace too small");
  if (n == 0) return 0;
  const int nblk = static_cast<int>(ceil_div(n, 256));
  int* hist = static_cast<int*>(ws);
  hipLaunchKernelGGL(rb_class_hist, dim3(nblk), dim3(256), 0, stream, indices, static_cast<int>(n), stride[1], stride[2], stride[0] - 1, stride[1] - 1,
                     stride[2] - 1, padding[0], padding[1], padding[2], hist);
  hipLaunchKernelGGL(rb_class_scatter, dim3(nblk), dim3(256), 0, stream, indices, static_cast<int>(n), stride[1], stride[2], stride[0] - 1,
                     stride[1] - 1, stride[2] - 1, padding[0], padding[1], padding[2], hist, nblk, perm);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_rulebook_count(const int* tab_in, int64_t n_in, int kvol, int* indice_num, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(indice_num && kvol >= 1 && n_in >= 0, FV2P_EINVAL, "rulebook_count: bad arguments");
  FV2P_HIP(hipMemsetAsync(indice_num, 0, sizeof(int) * kvol, stream));
  if (n_in == 0) return 0;
  FV2P_REQUIRE(tab_in, FV2P_EINVAL, "rulebook_count: null table");
  const unsigned cb = static_cast<unsigned>(ceil_div(n_in, 1024) < 64 ? ceil_div(n_in, 1024) : 64);
  hipLaunchKernelGGL(rb_count_rows, dim3(cb, kvol), dim3(256), 0, stream, tab_in, (int)n_in, indice_num);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t fv2p_rulebook_pairs_ws_bytes(int64_t n_in, int kvol) {
  const int64_t nblk = ceil_div(n_in > 0 ? n_in : 1, kPairTile);
  Sizer s;
  s.take<int>(static_cast<size_t>(kvol) * nblk + 1);
  s.take<char>(scan_ws_bytes(static_cast<int64_t>(kvol) * nblk + 1));
  return s.bytes();
}

extern "C" int fv2p_rulebook_pairs(const int* tab_in, int64_t n_in, int kvol, int pad, int* pairs, int* pair_num, void* ws,
                                   size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(kvol >= 1 && n_in >= 0, FV2P_EINVAL, "rulebook_pairs: bad sizes");
  if (n_in == 0) {
    if (pair_num) FV2P_HIP(hipMemsetAsync(pair_num, 0, sizeof(int) * kvol, stream));
    return 0;
  }
  FV2P_REQUIRE(tab_in && pairs, FV2P_EINVAL, "rulebook_pairs: null pointer");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_rulebook_pairs_ws_bytes(n_in, kvol), FV2P_EWORKSPACE, "rulebook_pairs: workspace too small");
  const int nblk = static_cast<int>(ceil_div(n_in, kPairTile));
  Carver c(ws, ws_bytes);
  int* cnt = c.take<int>(static_cast<size_t>(kvol) * nblk + 1);
  const size_t sb = scan_ws_bytes(static_cast<int64_t>(kvol) * nblk + 1);
  char* sws = c.take<char>(sb);
  FillJobs fill;
  if (pad) fill.add(pairs, sizeof(int) * 2 * (size_t)kvol * n_in, 0xFFFFFFFFu);  // -1 padding, spconv_ops.h:55-57
  fill.add(cnt + static_cast<size_t>(kvol) * nblk, sizeof(int), 0u);            // the scan's extra entry = total
  if (int rc = multi_fill(fill, stream)) return rc;
  hipLaunchKernelGGL(rb_pair_counts, dim3(nblk, kvol), dim3(256), 0, stream, tab_in, (int)n_in, nblk, cnt);
  if (int rc = exclusive_scan_i32(cnt, cnt, static_cast<int64_t>(kvol) * nblk + 1, nullptr, sws, sb, stream)) return rc;
  hipLaunchKernelGGL(rb_pair_write, dim3(nblk, kvol), dim3(256), 0, stream, tab_in, (int)n_in, nblk, cnt, pairs, n_in, pair_num);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_pairs_to_tables(const int* pairs, const int* indice_num, int kvol, int64_t pair_len, int64_t n_in, int64_t n_out,
                                    int* tab_in, int* tab_out, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(pairs && indice_num && kvol >= 1 && pair_len >= 0, FV2P_EINVAL, "pairs_to_tables: bad arguments");
  if (tab_in && n_in > 0) FV2P_HIP(hipMemsetAsync(tab_in, 0xFF, sizeof(int) * (size_t)kvol * n_in, stream));
  if (tab_out && n_out > 0) FV2P_HIP(hipMemsetAsync(tab_out, 0xFF, sizeof(int) * (size_t)kvol * n_out, stream));
  if (pair_len == 0) return 0;
  const unsigned nb = static_cast<unsigned>(ceil_div(pair_len, 256) < 256 ? ceil_div(pair_len, 256) : 256);
  hipLaunchKernelGGL(rb_pairs_to_tables, dim3(nb, kvol), dim3(256), 0, stream, pairs, indice_num, pair_len, (int)n_in, (int)n_out,
                     tab_in, tab_out);
  FV2P_LAUNCH_CHECK();
  return 0;
}
