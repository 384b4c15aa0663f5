// A5/A6 — fused sparse convolution (gather + per-offset GEMM + accumulate) on fp32 MFMA.
//
// Replaces the reference's per-offset gather -> torch::mm -> scatterAdd loop
// (pcdet/ops/spconv/include/spconv/spconv_ops.h:260-457, reordering.cu.h:21-157: ~78 launches,
// two staging buffers and a D2H sync per layer) with ONE output-stationary kernel per direction:
//
//   dst[r, :] = sum_k  src[tab[k][r], :] * W_k        (rows with tab == -1 contribute nothing)
//
// Forward uses tab_out, backward-data uses tab_in with W_k transposed, inverse conv uses tab_in
// (see rulebook.hip).  Every destination row is owned by exactly one wavefront, so there are no
// atomics and the k = 0..K-1 summation order is fixed (deterministic).  The per-offset GEMM runs on
// v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain): a wave owns 16 destination rows x all
// output columns; the A fragment is gathered straight from HBM/L2 as one 16-byte load per lane
// (lane (r, g) reads channels 16j+4g..+3 of row r, a K-permutation that both operands share), the
// B fragment (W_k) is staged once per workgroup in LDS in fragment order so every lane does a
// conflict-free ds_read_b128.
//
// Weight gradient: dW_k = sum_r src[tab[k][r], :]^T * grad[r, :], one wave per (row chunk, k,
// cin-block group); valid rows are compacted with a wave64 ballot so only real pairs reach the MFMA.
#include "common.hpp"
#include "bn_fold.hpp"
#include <stdlib.h>
#include <type_traits>
#include <algorithm>
#include <mutex>
#include <vector>

namespace fv2p {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, A>{}), ..., f(std::integral_constant<int, B - 1>{})
template <int A, int B, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (A < B) {
    f(std::integral_constant<int, A>{});
    static_for<A + 1, B>(f);
  }
}

struct ConvArgs {
  const float* src; int ld_src; int c_src;          // c_src: valid source channels in this launch
  const float* w; long long w_kstride; int w_ld;     // W_k = w + k*w_kstride, row stride w_ld
  const int* tab; int n_dst; int kvol; int flip;
  const float* bias;
  float* dst; int ld_dst; int c_dst;                 // c_dst: valid destination channels in this launch
  int accumulate;                                    // dst += result (used when the host splits c_src)
  unsigned long long* trace;                         // optional per-workgroup placement/timing trace (test hook)
  double* stats;                                     // optional [kStatSlots][2][stats_ld]: per-column sum / sum of squares of dst
  int stats_ld;
  // backward-data conv whose result is the gradient of a BatchNorm(+ReLU) output: stats then receive that layer's
  // backward sums  (sum dz, sum dz * xhat), dz = dst * [y > 0], from the BatchNorm input bn_x (same shape as dst)
  const float* bn_x; const float* bn_mean; const float* bn_invstd; const float* bn_gamma; const float* bn_beta; int bn_relu;
  // statistics finalised by the launch's LAST workgroup (conv_stats_done; bn_fold.hpp): on when fin_counter != nullptr
  unsigned* fin_counter;                             // zero before the launch, zero again after it
  double* fin_rows; double* fin_gslots; int fin_c;   // [tiles][2][stats_ld] rows, [groups][2][stats_ld] group slots (this launch's columns, unshifted by the kernels' column split), their number
  int fin_groups, fin_col0;                          // groups the tiles fall into (tile % groups, <= kFinSubs); first column of the block a column-split kernel computes (set by the kernel)
  BnFwdFin fin_fwd;                                  // forward statistics (bn_x == nullptr): mean / invstd / running statistics of these columns
  BnBwdFin fin_bwd;                                  // backward sums (bn_x != nullptr): dgamma / dbeta / coef of these columns; coef rows are stats_ld apart
  int fin_bump;                                      // forward: this launch also advances num_batches_tracked (the last column block of a layer)
  // BatchNorm (+ReLU) of the SOURCE rows applied on the gather (the reference's bn1 -> relu -> conv2 of a residual block in one kernel):
  // gathered value v of source channel c becomes relu?((v - pre_mean[c]) * pre_invstd[c] * pre_gamma[c] + pre_beta[c]); "no neighbour" stays 0
  const float* pre_mean; const float* pre_invstd; const float* pre_gamma; const float* pre_beta; int pre_relu;
  int dry;                                           // host only: choose the kernel, launch nothing (fv2p_sparse_conv_prenorm_supported)
  const int* perm;                                   // optional row order: tile t owns destination rows perm[64t .. 64t+63]
  const int* plan;                                   // optional cost-balanced tiling (conv_rows_ksplit): tile t owns rows [plan[t], plan[t+1])
  int col_blocks;                                    // conv_rows_ksplit: 64-column blocks of a tile, decoded from blockIdx.x (0 / 1: one)
};

// ---- BatchNorm statistics in the epilogue --------------------------------------------------------
// Every conv of the reference's backbones feeds BatchNorm1d (spconv_backbone.py:8-27); its per-channel sum and sum of
// squares are taken here from the accumulators instead of by a separate pass over dst (fp64 from the first add on:
// E[x^2] - E[x]^2 cancels).  vals[i][reg]: value stored to row rows[reg] (-1: past the end), column col[i].  Two forms:
//   * fv2p_sparse_conv_rows_stats (rounds 2 - 5): a wave reduces its 16 rows and adds to one of kStatSlots accumulator rows with fp64
//     atomics - fire and forget, the kernel boundary orders them before the consumer (bn_apply_fwd_k folds the slots);
//   * fv2p_sparse_conv_rows_bnfin (round 6, acc != nullptr): the lane only ACCUMULATES here; conv_stats_done combines the workgroup's
//     sums, publishes them as the tile's row of a [tiles][2][C] buffer and the launch's last workgroups fold the rows in a fixed order:
//     no float atomics, no slot clearing, statistics bit-identical from run to run.
template <int NV>
struct StatAcc {
  double s1[NV], s2[NV];
  __device__ __forceinline__ StatAcc() {
#pragma unroll
    for (int i = 0; i < NV; ++i) s1[i] = s2[i] = 0.0;
  }
};
template <int NV>
__device__ __forceinline__ void tile_stats_rows(const ConvArgs& a, const float (&vals)[NV][4], const int (&col)[NV], const int (&rows)[4], StatAcc<NV>* acc) {
  const int lane = threadIdx.x & 63, q = lane >> 4;   // rows[reg]: destination row of value reg, -1 past the end
  const int slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) % kStatSlots;
  double* base = a.stats + static_cast<long long>(slot) * 2 * a.stats_ld;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s1 = 0.0, s2 = 0.0;
    const bool live = col[i] < a.c_dst;
    if (a.bn_x == nullptr) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        if (rows[reg] >= 0) { const double v = vals[i][reg]; s1 += v; s2 += v * v; }
      }
    } else if (live) {   // same arithmetic as bn_reduce_k<BWD> (batchnorm.hip)
      const float m = a.bn_mean[col[i]], is = a.bn_invstd[col[i]];
      const float ga = a.bn_gamma ? a.bn_gamma[col[i]] : 1.f, be = a.bn_beta ? a.bn_beta[col[i]] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = rows[reg];
        if (row >= 0) {
          const float xhat = (a.bn_x[static_cast<long long>(row) * a.ld_dst + col[i]] - m) * is;
          const float y = xhat * ga + be;
          const float dz = (a.bn_relu && !(y > 0.f)) ? 0.f : vals[i][reg];
          s1 += dz; s2 += static_cast<double>(dz) * xhat;
        }
      }
    }
    if (a.fin_counter) {   // uniform: summed up by conv_stats_done
      acc->s1[i] += s1; acc->s2[i] += s2;
      continue;
    }
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    if (q == 0 && live) {
      __hip_atomic_fetch_add(base + col[i], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(base + a.stats_ld + col[i], s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
template <int NV>
__device__ __forceinline__ void tile_stats(const ConvArgs& a, const float (&vals)[NV][4], const int (&col)[NV], int row0, StatAcc<NV>* acc) {
  const int q = (threadIdx.x & 63) >> 4;
  int rows[4];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) rows[reg] = (row0 + q * 4 + reg < a.n_dst) ? row0 + q * 4 + reg : -1;
  tile_stats_rows<NV>(a, vals, col, rows, acc);
}

// ---- finalisation by the last workgroups of the launch ------------------------------------------------------------------------------
// Every workgroup calls this once, at its very end (fv2p_sparse_conv_rows_bnfin / _bnbwd_fin only: a.fin_counter).
//   1. the lanes' sums (StatAcc) are reduced over the wave's four row quads (shuffles) and over the workgroup's waves (LDS, wave
//      order), and published as row `tile` of fin_rows [tiles][2][stats_ld] (stat_publish: exchange atomics), one row per tile
//      whichever column block the workgroup computed;
//   2. fin_rows_done (bn_fold.hpp): the group / top-word protocol that folds the rows in a fixed order and writes mean / invstd /
//      running statistics (forward) or dgamma / dbeta / c1 / c2 (backward sums).
// No fence (a release fence writes the whole L2 back on gfx950: 20 us), no float atomic adds, nothing to clear: rows and slots are
// overwritten by the next launch.  Measured against the first form of this round (fp64 atomic adds into 16 slots, one last workgroup
// folding and clearing them: + 8 - 13 us per launch, the waves waiting for ~250 k memory-side atomics): profiles/README.md.
// scratch: kStatsDoneLds bytes of the kernel's DYNAMIC LDS that nothing else uses any more (no static LDS: the K-split tile fills a
// CU's 160 KB with two workgroups to within 768 bytes).
constexpr size_t kStatsDoneLds = 9 * 1024;   // the largest [waves][2][columns] in use + the fold's buffer
template <int NV>
__device__ __forceinline__ void conv_stats_done(const ConvArgs& a, void* scratch, const StatAcc<NV>& acc, const int (&cols)[NV], int tile, int n_tiles, int ncb) {
  if (!a.fin_counter) return;   // uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, n = lane & 15, q = lane >> 4;
  constexpr int LC = NV * 16;   // columns this workgroup computed (local index i * 16 + n)
  double* part = reinterpret_cast<double*>(scratch);                    // [nw][2][LC]
  int* colmap = reinterpret_cast<int*>(part + nw * 2 * LC);             // [LC] launch column of a local column (-1: past the end)
  unsigned* flag = reinterpret_cast<unsigned*>(static_cast<char*>(scratch) + kStatsDoneLds - 16);   // behind everything the fold reuses
  double s1[NV], s2[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    s1[i] = acc.s1[i]; s2[i] = acc.s2[i];
    s1[i] += __shfl_xor(s1[i], 16, 64); s2[i] += __shfl_xor(s2[i], 16, 64);
    s1[i] += __shfl_xor(s1[i], 32, 64); s2[i] += __shfl_xor(s2[i], 32, 64);
  }
  __syncthreads();   // every wave is done with whatever the scratch region held
  if (q == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      part[(wave * 2 + 0) * LC + i * 16 + n] = s1[i];
      part[(wave * 2 + 1) * LC + i * 16 + n] = s2[i];
      if (wave == 0) colmap[i * 16 + n] = cols[i] < a.c_dst ? a.fin_col0 + cols[i] : -1;
    }
  }
  __syncthreads();
  const int ld = a.stats_ld;
  if (tid < 2 * LC) {
    const int which = tid / LC, lc = tid % LC, gc = colmap[lc];
    if (gc >= 0) {
      double t = 0.0;
      for (int w = 0; w < nw; ++w) t += part[(w * 2 + which) * LC + lc];
      stat_publish(a.fin_rows + (static_cast<long long>(tile) * 2 + which) * ld + gc, t);
    }
  }
  FinOut fo;
  fo.bwd = a.bn_x != nullptr; fo.n = a.n_dst; fo.ff = a.fin_fwd; fo.bf = a.fin_bwd; fo.bump = a.fin_bump; fo.coef_ld = ld;
  fin_rows_done(a.fin_rows, a.fin_gslots, a.fin_counter, tile, n_tiles, ncb, a.fin_groups, a.fin_c, ld, flag, reinterpret_cast<double (*)[256]>(part), fo);
}

// source-row BatchNorm (+ReLU) on the gather: v holds source channels c0 .. c0 + 3 of a row that exists (`valid`; "no neighbour" stays 0).
// Same operations in the same order as bn_apply_fwd_k (batchnorm.hip): a consumer that normalises on the fly and one that reads the
// materialised rows multiply bit-identical operands.
struct PreNorm4 { f32x4 mean, is, ga, be; };
__device__ __forceinline__ void prenorm_load(const ConvArgs& a, int c0, PreNorm4& p) {
  p.mean = *reinterpret_cast<const f32x4*>(a.pre_mean + c0);
  p.is = *reinterpret_cast<const f32x4*>(a.pre_invstd + c0);
  p.ga = a.pre_gamma ? *reinterpret_cast<const f32x4*>(a.pre_gamma + c0) : f32x4{1.f, 1.f, 1.f, 1.f};
  p.be = a.pre_beta ? *reinterpret_cast<const f32x4*>(a.pre_beta + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f32x4 prenorm_apply(const PreNorm4& p, f32x4 v, bool valid, int relu) {
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float xhat = (v[i] - p.mean[i]) * p.is[i];
    const float t = xhat * p.ga[i] + p.be[i];
    o[i] = valid ? ((relu && t <= 0.f) ? 0.f : t) : 0.f;
  }
  return o;
}

// ---- B staging --------------------------------------------------------------------------------
// VEC layout: element B[c][col] at (((c>>4)*NB + (col>>4))*64 + ((c>>2)&3)*16 + (col&15))*4 + (c&3)
template <int CINP, int NB, bool WT>
__device__ __forceinline__ void stage_w_vec(const ConvArgs& a, const float* __restrict__ wk, float* __restrict__ lds) {
  constexpr int COLS = NB * 16;
  if (!WT) {
    // W_k[c][col], row-major: float4 along col
    for (int e = threadIdx.x; e < CINP * (COLS / 4); e += 256) {
      const int c = e / (COLS / 4), col = (e % (COLS / 4)) * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (c < a.c_src) {
        const float* p = wk + static_cast<long long>(c) * a.w_ld + col;
        if (col + 3 < a.c_dst && (a.w_ld & 3) == 0) {
          const float4 q = *reinterpret_cast<const float4*>(p);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) if (col + u < a.c_dst) v[u] = p[u];
        }
      }
      const int j = c >> 4, g = (c >> 2) & 3, t = c & 3;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = col + u;
        lds[(((j * NB + (cc >> 4)) * 64) + g * 16 + (cc & 15)) * 4 + t] = v[u];
      }
    }
  } else {
    // B[c][col] = W_k[col][c]: float4 along c -> one 16-byte LDS store
    for (int e = threadIdx.x; e < COLS * (CINP / 4); e += 256) {
      const int col = e / (CINP / 4), c = (e % (CINP / 4)) * 4;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col < a.c_dst) {
        const float* p = wk + static_cast<long long>(col) * a.w_ld + c;
        if (c + 3 < a.c_src && (a.w_ld & 3) == 0) q = *reinterpret_cast<const float4*>(p);
        else {
          if (c + 0 < a.c_src) q.x = p[0];
          if (c + 1 < a.c_src) q.y = p[1];
          if (c + 2 < a.c_src) q.z = p[2];
          if (c + 3 < a.c_src) q.w = p[3];
        }
      }
      const int j = c >> 4, g = (c >> 2) & 3;
      *reinterpret_cast<float4*>(&lds[(((j * NB + (col >> 4)) * 64) + g * 16 + (col & 15)) * 4]) = q;
    }
  }
}

// SCALAR layout: element B[c][col] at ((c>>2)*NB + (col>>4))*64 + (c&3)*16 + (col&15)
template <int STEPS, int NB, bool WT>
__device__ __forceinline__ void stage_w_scalar(const ConvArgs& a, const float* __restrict__ wk, float* __restrict__ lds) {
  constexpr int COLS = NB * 16;
  for (int e = threadIdx.x; e < STEPS * 4 * COLS; e += 256) {
    const int c = e / COLS, col = e % COLS;
    float v = 0.f;
    if (c < a.c_src && col < a.c_dst)
      v = WT ? wk[static_cast<long long>(col) * a.w_ld + c] : wk[static_cast<long long>(c) * a.w_ld + col];
    lds[((c >> 2) * NB + (col >> 4)) * 64 + (c & 3) * 16 + (col & 15)] = v;
  }
}

template <int NB>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const f32x4 (&acc)[NB], int row0, StatAcc<NB>* sacc) {
  const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int col = nb * 16 + n;
    if (col >= a.c_dst) continue;
    const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = row0 + q * 4 + reg;
      if (row < a.n_dst) {
        float* p = a.dst + static_cast<long long>(row) * a.ld_dst + col;
        const float v = acc[nb][reg] + b;
        *p = a.accumulate ? (*p + v) : v;
      }
    }
  }
  if (a.stats) {   // uniform; never together with accumulate (host)
    float vals[NB][4];
    int cols[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      cols[nb] = nb * 16 + n;
      const float b = (a.bias && cols[nb] < a.c_dst) ? a.bias[cols[nb]] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) vals[nb][reg] = acc[nb][reg] + b;
    }
    tile_stats<NB>(a, vals, cols, row0, sacc);
  }
}
// the columns conv_epilogue's lane holds (accumulator nb of lane n: column 16 nb + n), for conv_stats_done
template <int NB>
__device__ __forceinline__ void epilogue_cols(int (&cols)[NB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) cols[nb] = nb * 16 + (threadIdx.x & 15);
}

// CINP: source channels padded to a multiple of 16 (c_src % 4 == 0 required, 16-byte row loads)
template <int CINP, int NB, bool WT>
__global__ __launch_bounds__(256) void conv_rows_vec(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int J = CINP / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 64 + wave * 16;
  const int my_row = row0 + r;
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < a.kvol; ++k) {
    const int trow = a.flip ? (a.kvol - 1 - k) : k;
    const int idx = (my_row < a.n_dst) ? a.tab[static_cast<long long>(trow) * a.n_dst + my_row] : -1;
    float4 av[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx >= 0 && (16 * j + 4 * g) < a.c_src)
        av[j] = *reinterpret_cast<const float4*>(a.src + static_cast<long long>(idx) * a.ld_src + 16 * j + 4 * g);
    }
    const bool any = __ballot(idx >= 0) != 0ull;
    __syncthreads();
    stage_w_vec<CINP, NB, WT>(a, a.w + static_cast<long long>(k) * a.w_kstride, lds);
    __syncthreads();
    if (any) {
#pragma unroll
      for (int j = 0; j < J; ++j) {
        float4 bv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = *reinterpret_cast<const float4*>(&lds[((j * NB + nb) * 64 + lane) * 4]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[nb].x, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[nb].y, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[nb].z, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[nb].w, acc[nb], 0, 0, 0);
      }
    }
  }
  StatAcc<NB> sacc;
  conv_epilogue<NB>(a, acc, row0, &sacc);
  int scols[NB];
  epilogue_cols<NB>(scols);
  conv_stats_done<NB>(a, lds, sacc, scols, blockIdx.x, gridDim.x, 1);
}

// ---- software-pipelined variant (CINP*NB <= 256) ---------------------------------------------------
// Same tile as conv_rows_vec, but W_k is double-buffered in LDS and staged through registers (global loads issued
// before the MFMA block, LDS stores after it), the A fragments of the next offsets and the table rows of the ones
// after are fetched while the current offsets are on the matrix pipe, and a single barrier per iteration remains.
template <int CINP, int NB, bool WT>
struct WStage {
  static constexpr int COLS = NB * 16;
  static constexpr int UNITS = CINP * COLS / 4;             // float4 units of one W_k
  static constexpr int R = (UNITS + 255) / 256;             // units per thread
  float4 v[R];
  __device__ __forceinline__ void load(const ConvArgs& a, const float* __restrict__ wk) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int e = threadIdx.x + u * 256;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < UNITS) {
        if (!WT) {
          const int c = e / (COLS / 4), col = (e % (COLS / 4)) * 4;
          if (c < a.c_src) {
            const float* p = wk + static_cast<long long>(c) * a.w_ld + col;
            if (col + 3 < a.c_dst && (a.w_ld & 3) == 0) q = *reinterpret_cast<const float4*>(p);
            else {
              if (col + 0 < a.c_dst) q.x = p[0];
              if (col + 1 < a.c_dst) q.y = p[1];
              if (col + 2 < a.c_dst) q.z = p[2];
              if (col + 3 < a.c_dst) q.w = p[3];
            }
          }
        } else {
          // unit e in LDS fragment order: n fastest, then g, nb, j  ->  the LDS store below is perfectly linear
          const int n = e & 15, gq = (e >> 4) & 3, rest = e >> 6;
          const int col = (rest % NB) * 16 + n, c = (rest / NB) * 16 + gq * 4;
          if (col < a.c_dst) {
            const float* p = wk + static_cast<long long>(col) * a.w_ld + c;
            if (c + 3 < a.c_src && (a.w_ld & 3) == 0) q = *reinterpret_cast<const float4*>(p);
            else {
              if (c + 0 < a.c_src) q.x = p[0];
              if (c + 1 < a.c_src) q.y = p[1];
              if (c + 2 < a.c_src) q.z = p[2];
              if (c + 3 < a.c_src) q.w = p[3];
            }
          }
        }
      }
      v[u] = q;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int e = threadIdx.x + u * 256;
      if (e >= UNITS) continue;
      if (!WT) {
        const int c = e / (COLS / 4), col = (e % (COLS / 4)) * 4;
        const int j = c >> 4, g = (c >> 2) & 3, t = c & 3;
        const float q[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int cc = col + x;
          lds[(((j * NB + (cc >> 4)) * 64) + g * 16 + (cc & 15)) * 4 + t] = q[x];
        }
      } else {
        *reinterpret_cast<float4*>(&lds[e * 4]) = v[u];
      }
    }
  }
};

// KU kernel offsets are processed per loop iteration (KU = 2 for the usual 27-offset kernels): the gathers are
// latency bound, so issuing the loads of KU offsets at once halves the number of exposed round trips.
template <int CINP, int NB, bool WT, int KU>
__global__ __launch_bounds__(256) void conv_rows_pipe(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 (double buffer) x KU x W_k
  constexpr int J = CINP / 16;
  constexpr int WSZ = CINP * NB * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 64 + wave * 16;
  const int my_row = row0 + r;
  const bool row_ok = my_row < a.n_dst;
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto tab_at = [&](int k) -> int {
    if (k >= a.kvol || !row_ok) return -1;
    const int trow = a.flip ? (a.kvol - 1 - k) : k;
    return a.tab[static_cast<long long>(trow) * a.n_dst + my_row];
  };
  auto gather = [&](int idx, float4 (&av)[J]) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx >= 0 && (16 * j + 4 * g) < a.c_src)
        av[j] = *reinterpret_cast<const float4*>(a.src + static_cast<long long>(idx) * a.ld_src + 16 * j + 4 * g);
    }
  };
  WStage<CINP, NB, WT> ws[KU];
  float4 a_cur[KU][J], a_nxt[KU][J];
  int idx_cur[KU], idx_nxt[KU], idx_nn[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) { idx_cur[u] = tab_at(u); idx_nxt[u] = tab_at(KU + u); }
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    if (u < a.kvol) ws[u].load(a, a.w + static_cast<long long>(u) * a.w_kstride);
    gather(idx_cur[u], a_cur[u]);
  }
#pragma unroll
  for (int u = 0; u < KU; ++u)
    if (u < a.kvol) ws[u].store(lds + u * WSZ);
  __syncthreads();
  int it = 0;
  for (int k0 = 0; k0 < a.kvol; k0 += KU, ++it) {
    float* wcur = lds + (it & 1) * KU * WSZ;
    float* wnxt = lds + ((it + 1) & 1) * KU * WSZ;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int kn = k0 + KU + u;
      idx_nn[u] = tab_at(kn + KU);
      if (kn < a.kvol) {
        ws[u].load(a, a.w + static_cast<long long>(kn) * a.w_kstride);
        gather(idx_nxt[u], a_nxt[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (k0 + u < a.kvol && __ballot(idx_cur[u] >= 0) != 0ull) {
        const float* wk = wcur + u * WSZ;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          float4 bv[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bv[nb] = *reinterpret_cast<const float4*>(&wk[((j * NB + nb) * 64 + lane) * 4]);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][j].x, bv[nb].x, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][j].y, bv[nb].y, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][j].z, bv[nb].z, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][j].w, bv[nb].w, acc[nb], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < KU; ++u)
      if (k0 + KU + u < a.kvol) ws[u].store(wnxt + u * WSZ);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < KU; ++u) {
#pragma unroll
      for (int j = 0; j < J; ++j) a_cur[u][j] = a_nxt[u][j];
      idx_cur[u] = idx_nxt[u];
      idx_nxt[u] = idx_nn[u];
    }
  }
  StatAcc<NB> sacc;
  conv_epilogue<NB>(a, acc, row0, &sacc);
  int scols[NB];
  epilogue_cols<NB>(scols);
  conv_stats_done<NB>(a, lds, sacc, scols, blockIdx.x, gridDim.x, 1);
}

// ---- LDS-DMA variant (the default for full 16-channel multiples) --------------------------------------
// W_k goes global -> LDS with `global_load_lds_dwordx4` (no staging VGPRs, no ds_write pass): the LDS image of one
// wave-instruction is lane-linear, so the fragment order is produced on the SOURCE side — every lane fetches the 16
// contiguous bytes of W that its own later `ds_read_b128` expects:
//   forward  (W_k[c][col], contiguous in col): lane (g, n), K-step (j, t), column block cb reads
//            W_k[16j+4g+t][64cb+4n .. +3]; the four components feed four MFMAs whose accumulators are the output
//            columns 64cb+4n+{0,1,2,3}  (needs c_dst % 64 == 0);
//   backward (B[c][col] = W_k[col][c], contiguous in c): lane (g, n), block (j, nb) reads W_k[NB*n+nb][16j+4g .. +3];
//            the components are the four K-steps of one MFMA column, accumulator nb is output column NB*n+nb.
// Either way a lane ends up owning runs of 4 adjacent output columns, so the epilogue writes float4s (256 B per row
// per 16 lanes).  Tiles are handed to workgroups XCD-major (blockIdx % 8 = XCD): each XCD's L2 sees one contiguous
// slab of destination rows and therefore one spatial slab of gathered source rows.
#define FV2P_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define FV2P_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// One LDS-DMA piece: 64 lanes x 16 B, lane l lands at lds_dst + 16*l.  Issued as asm so that hipcc does not drain
// vmcnt before the MFMA phase's ds_reads of the OTHER buffer (with the builtin it waits vmcnt(0) at the next LDS
// read); the kernel waits vmcnt(0) itself right before the barrier that publishes the buffer.  hipcc's own vmcnt
// bookkeeping for ordinary loads stays safe: unknown extra loads only make its counted waits conservative.
__device__ __forceinline__ void glds16(const float* gsrc, float* lds_dst) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds_dst)));
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

__device__ __forceinline__ int xcd_major_tile(int b, int nblk) {
  const int x = b & 7, slot = b >> 3;
  return x * (nblk >> 3) + min(x, nblk & 7) + slot;
}

// Loads whose completion the kernel tracks itself (hipcc keeps no vmcnt score for asm): they are issued before the
// MFMA phase and awaited by wait_loads() right before the barrier, so nothing in the MFMA phase stalls on them.
template <int OFF>
__device__ __forceinline__ void load16_at(f32x4& d, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void load4_async(int& d, const int* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
// s_waitcnt vmcnt(0) that every asynchronously loaded register passes through ("+v"): uses are ordered after it
template <int J>
__device__ __forceinline__ void wait_loads(f32x4 (&v)[J], int& idx) {
  if constexpr (J == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(idx) : : "memory");
  else if constexpr (J == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(idx) : : "memory");
  else if constexpr (J == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(idx) : : "memory");
  else {
    static_assert(J == 8, "J in {1,2,4,8}");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(idx) : : "memory");
  }
}
template <int J, int JJ = 0>
__device__ __forceinline__ void gather_async(f32x4 (&v)[J], const float* p) {
  if constexpr (JJ < J) {
    load16_at<64 * JJ>(v[JJ], p);
    gather_async<J, JJ + 1>(v, p);
  }
}

__device__ __attribute__((aligned(256))) float g_zero_row[256];  // stands in for "no neighbour": A fragment = 0

template <int CINP, int NB, bool WT>
__global__ __launch_bounds__(256) void conv_rows_dma(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 x W_k
  constexpr int J = CINP / 16;
  constexpr int WSZ = CINP * NB * 16;
  constexpr int UNITS = WSZ / 4;
  constexpr int R = (UNITS + 255) / 256;
  constexpr int NCB = NB / 4;  // forward only
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, g = lane >> 4;
  const int row0 = xcd_major_tile(blockIdx.x, gridDim.x) * 64 + wave * 16;
  if (gridDim.y > 1) {
    // column split (128 x 128 layers: two 64 KB weight buffers do not fit next to each other): workgroup y computes the
    // destination columns [16 NB y, 16 NB (y + 1)) from its own slice of W_k; every per-column pointer moves with it
    const int off = blockIdx.y * NB * 16;
    a.w += WT ? static_cast<long long>(off) * a.w_ld : off;
    a.dst += off;
    a.fin_col0 = off;
    a.c_dst = NB * 16;
    if (a.bias) a.bias += off;
    if (a.stats) a.stats += off;
    if (a.bn_x) {
      a.bn_x += off; a.bn_mean += off; a.bn_invstd += off;
      if (a.bn_gamma) a.bn_gamma += off;
      if (a.bn_beta) a.bn_beta += off;
    }
  }
  unsigned long long t_begin = 0;
  if (a.trace) t_begin = __builtin_readcyclecounter();
  StatAcc<NB> sacc;
  // rows past the end compute on the last row's neighbours and are never stored (no masks in the loop)
  const int my_row = min(row0 + r, a.n_dst - 1);
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane source offset (floats, relative to W_k) of every unit this lane fetches
  long long woff[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int rest = u * 4 + wave;
    if constexpr (!WT) {
      const int cb = rest % NCB, jt = rest / NCB;
      woff[u] = static_cast<long long>(16 * (jt >> 2) + 4 * g + (jt & 3)) * a.w_ld + 64 * cb + 4 * r;
    } else {
      const int nb = rest % NB, j = rest / NB;
      woff[u] = static_cast<long long>(NB * r + nb) * a.w_ld + 16 * j + 4 * g;
    }
  }
  auto issue_w = [&](int k, float* buf) {
    const float* wk = a.w + static_cast<long long>(k) * a.w_kstride;
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int e0 = (u * 4 + wave) * 64;
      if (UNITS % 256 == 0 || e0 < UNITS) glds16(wk + woff[u], buf + e0 * 4);
    }
  };
  const int* tab_me = a.tab + my_row;
  auto tab_ptr = [&](int k) -> const int* {
    const int kk = min(k, a.kvol - 1);  // past-the-end offsets re-read the last one; the value is never used
    return tab_me + static_cast<long long>(a.flip ? (a.kvol - 1 - kk) : kk) * a.n_dst;
  };
  auto row_ptr = [&](int idx) -> const float* {
    return idx >= 0 ? a.src + static_cast<long long>(idx) * a.ld_src + 4 * g : g_zero_row + 4 * g;
  };
  f32x4 a_cur[J], a_nxt[J];
  int idx_cur = *tab_ptr(0), idx_nxt = *tab_ptr(1), idx_nn;
  issue_w(0, lds);
  gather_async<J>(a_cur, row_ptr(idx_cur));
  wait_loads<J>(a_cur, idx_nxt);
  __syncthreads();
  unsigned long long t_pro = 0, t_wait = 0, t_mark = 0;
  if (a.trace) t_pro = __builtin_readcyclecounter();
  for (int k = 0; k < a.kvol; ++k) {
    const float* wcur = lds + (k & 1) * WSZ;
    if (k + 1 < a.kvol) issue_w(k + 1, lds + ((k + 1) & 1) * WSZ);
    gather_async<J>(a_nxt, row_ptr(idx_nxt));
    load4_async(idx_nn, tab_ptr(k + 2));
    if (__ballot(idx_cur >= 0) != 0ull) {
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if constexpr (!WT) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float4 bv[NCB > 0 ? NCB : 1];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) bv[cb] = *reinterpret_cast<const float4*>(&wcur[(((j * 4 + t) * NCB + cb) * 64 + lane) * 4]);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
              acc[cb * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][t], bv[cb].x, acc[cb * 4 + 0], 0, 0, 0);
              acc[cb * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][t], bv[cb].y, acc[cb * 4 + 1], 0, 0, 0);
              acc[cb * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][t], bv[cb].z, acc[cb * 4 + 2], 0, 0, 0);
              acc[cb * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][t], bv[cb].w, acc[cb * 4 + 3], 0, 0, 0);
            }
          }
        } else {
          float4 bv[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bv[nb] = *reinterpret_cast<const float4*>(&wcur[((j * NB + nb) * 64 + lane) * 4]);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][0], bv[nb].x, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][1], bv[nb].y, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][2], bv[nb].z, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][3], bv[nb].w, acc[nb], 0, 0, 0);
        }
      }
    }
    if (a.trace) t_mark = __builtin_readcyclecounter();
    wait_loads<J>(a_nxt, idx_nn);
    __syncthreads();
    if (a.trace) t_wait += __builtin_readcyclecounter() - t_mark;
#pragma unroll
    for (int j = 0; j < J; ++j) a_cur[j] = a_nxt[j];
    idx_cur = idx_nxt;
    idx_nxt = idx_nn;
  }
  if (a.trace && threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.trace[blockIdx.x * 8 + 0] = hw;
    a.trace[blockIdx.x * 8 + 1] = xcc;
    a.trace[blockIdx.x * 8 + 2] = t_begin;
    a.trace[blockIdx.x * 8 + 3] = __builtin_readcyclecounter();
    a.trace[blockIdx.x * 8 + 4] = t_pro;
    a.trace[blockIdx.x * 8 + 5] = t_wait;   // clocks wave 0 spent between the end of its MFMA block and the barrier's release
  }
  // epilogue: accumulator i of lane (q, n) is output column  WT ? NB*n + i : 64*(i/4) + 4*n + i%4
  const int q = lane >> 4, n = lane & 15;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = row0 + q * 4 + reg;
    if (row >= a.n_dst) continue;
    float* prow = a.dst + static_cast<long long>(row) * a.ld_dst;
    if constexpr (NB % 4 == 0) {
#pragma unroll
      for (int v = 0; v < NB / 4; ++v) {
        const int col = WT ? (NB * n + 4 * v) : (64 * v + 4 * n);
        float4 o = make_float4(acc[4 * v][reg], acc[4 * v + 1][reg], acc[4 * v + 2][reg], acc[4 * v + 3][reg]);
        if (a.bias) { const float4 b = *reinterpret_cast<const float4*>(a.bias + col); o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w; }
        float4* p = reinterpret_cast<float4*>(prow + col);
        if (a.accumulate) { const float4 old = *p; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *p = o;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int col = NB * n + i;  // WT only (forward requires NB % 4 == 0)
        const float v = acc[i][reg] + (a.bias ? a.bias[col] : 0.f);
        prow[col] = a.accumulate ? (prow[col] + v) : v;
      }
    }
  }
  if (a.stats) {
    float vals[NB][4];
    int cols[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      cols[i] = (NB % 4 == 0) ? (WT ? (NB * n + i) : (64 * (i / 4) + 4 * n + (i % 4))) : (NB * n + i);
      const float b = a.bias ? a.bias[cols[i]] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) vals[i][reg] = acc[i][reg] + b;
    }
    tile_stats<NB>(a, vals, cols, row0, &sacc);
  }
  {
    int scols[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) scols[i] = (NB % 4 == 0) ? (WT ? (NB * n + i) : (64 * (i / 4) + 4 * n + (i % 4))) : (NB * n + i);
    conv_stats_done<NB>(a, lds, sacc, scols, blockIdx.x, gridDim.x, gridDim.y);
  }
}

// ---- LDS-DMA variant for permuted rows: visits only the offsets its tile uses ----------------------------------------
// Same data flow as conv_rows_dma, for launches that come with a row permutation (a.perm: tile t owns destination rows
// perm[64t .. 64t+63]).  The backward-data conv of a strided layer is the case: an input voxel of a k=3, s=2 conv can
// only reach the offsets the parity of its coordinates allows (1, 2, 4 or 8 of 27), so with the rows grouped by parity
// class a tile uses a few offsets and the rest cost neither the weight DMA nor the barrier nor MFMAs on zero rows (in
// flat-index order neighbouring rows alternate parity and every tile visits all 27 at 10 % row density).  The prologue
// loads the tile's slice of the table into LDS ([kvol <= 32][64] ints) and ORs the mask of used offsets.  Every row's
// sum still runs over ascending k: results are bit-identical to the unpermuted launch.
template <int CINP, int NB, bool WT>
__global__ __launch_bounds__(256) void conv_rows_act(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 x W_k | table slice [32][64] | mask
  constexpr int J = CINP / 16;
  constexpr int WSZ = CINP * NB * 16;
  constexpr int UNITS = WSZ / 4;
  constexpr int R = (UNITS + 255) / 256;
  constexpr int NCB = NB / 4;  // forward only
  int* idx_lds = reinterpret_cast<int*>(lds + 2 * WSZ);
  unsigned* mask_lds = reinterpret_cast<unsigned*>(idx_lds + 32 * 64);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, g = lane >> 4;
  const int slot0 = xcd_major_tile(blockIdx.x, gridDim.x) * 64 + wave * 16;
  const int my_slot = min(slot0 + r, a.n_dst - 1);
  const int my_row = a.perm ? a.perm[my_slot] : my_slot;
  if (threadIdx.x == 0) *mask_lds = 0u;
  unsigned m = 0u;
  {
    const int* tab_me = a.tab + my_row;
    for (int k = g; k < a.kvol; k += 4) {
      const int v = tab_me[static_cast<long long>(a.flip ? (a.kvol - 1 - k) : k) * a.n_dst];
      idx_lds[k * 64 + wave * 16 + r] = v;
      if (v >= 0 && slot0 + r < a.n_dst) m |= 1u << k;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) m |= __shfl_xor(m, d, 64);
  }
  __syncthreads();
  if (lane == 0 && m) atomicOr(mask_lds, m);
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  long long woff[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int rest = u * 4 + wave;
    if constexpr (!WT) {
      const int cb = rest % (NCB > 0 ? NCB : 1), jt = rest / (NCB > 0 ? NCB : 1);
      woff[u] = static_cast<long long>(16 * (jt >> 2) + 4 * g + (jt & 3)) * a.w_ld + 64 * cb + 4 * r;
    } else {
      const int nb = rest % NB, j = rest / NB;
      woff[u] = static_cast<long long>(NB * r + nb) * a.w_ld + 16 * j + 4 * g;
    }
  }
  auto issue_w = [&](int k, float* buf) {
    const float* wk = a.w + static_cast<long long>(k) * a.w_kstride;
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int e0 = (u * 4 + wave) * 64;
      if (UNITS % 256 == 0 || e0 < UNITS) glds16(wk + woff[u], buf + e0 * 4);
    }
  };
  auto row_ptr = [&](int idx) -> const float* {
    return idx >= 0 ? a.src + static_cast<long long>(idx) * a.ld_src + 4 * g : g_zero_row + 4 * g;
  };
  __syncthreads();
  unsigned todo = __builtin_amdgcn_readfirstlane(*mask_lds);
  if (todo) {
    f32x4 a_cur[J], a_nxt[J];
    int k = __ffs(static_cast<int>(todo)) - 1;
    todo &= todo - 1;
    int idx_cur = idx_lds[k * 64 + wave * 16 + r], idx_nxt = -1;
    issue_w(k, lds);
    gather_async<J>(a_cur, row_ptr(idx_cur));
    wait_loads<J>(a_cur, idx_nxt);
    __syncthreads();
    for (int it = 0;; ++it) {
      const float* wcur = lds + (it & 1) * WSZ;
      const bool more = todo != 0u;   // uniform
      if (more) {
        k = __ffs(static_cast<int>(todo)) - 1;
        todo &= todo - 1;
        issue_w(k, lds + ((it + 1) & 1) * WSZ);
        idx_nxt = idx_lds[k * 64 + wave * 16 + r];
        if constexpr (J < 8) gather_async<J>(a_nxt, row_ptr(idx_nxt));
        else {
          // 128 source channels: the compiler placed the a_cur = a_nxt copies ABOVE the hand-written wait (a copy of registers the loads
          // had not reached; tools/check_async_asm.py) — plain tracked loads here (this shape runs only when the K-split tile is switched off)
          const float* p = row_ptr(idx_nxt);
#pragma unroll
          for (int j = 0; j < J; ++j) a_nxt[j] = *reinterpret_cast<const f32x4*>(p + 16 * j);
        }
      }
      if (__ballot(idx_cur >= 0) != 0ull) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          if constexpr (!WT) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              float4 bv[NCB > 0 ? NCB : 1];
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb) bv[cb] = *reinterpret_cast<const float4*>(&wcur[(((j * 4 + t) * NCB + cb) * 64 + lane) * 4]);
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb) {
                acc[cb * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][t], bv[cb].x, acc[cb * 4 + 0], 0, 0, 0);
                acc[cb * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][t], bv[cb].y, acc[cb * 4 + 1], 0, 0, 0);
                acc[cb * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][t], bv[cb].z, acc[cb * 4 + 2], 0, 0, 0);
                acc[cb * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][t], bv[cb].w, acc[cb * 4 + 3], 0, 0, 0);
              }
            }
          } else {
            float4 bv[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = *reinterpret_cast<const float4*>(&wcur[((j * NB + nb) * 64 + lane) * 4]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][0], bv[nb].x, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][1], bv[nb].y, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][2], bv[nb].z, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j][3], bv[nb].w, acc[nb], 0, 0, 0);
          }
        }
      }
      if (!more) break;
      if constexpr (J < 8) wait_loads<J>(a_nxt, idx_nxt);
      else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");   // the weights' LDS-DMA
      __syncthreads();
#pragma unroll
      for (int j = 0; j < J; ++j) a_cur[j] = a_nxt[j];
      idx_cur = idx_nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");   // nothing in flight (the last round issued no loads): stated for tools/check_async_asm.py
  }
  // epilogue: accumulator i of lane (q, n) is output column  WT ? NB*n + i : 64*(i/4) + 4*n + i%4; rows through perm
  const int q = lane >> 4, n = lane & 15;
  StatAcc<NB> sacc;
  int rows[4];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int slot = slot0 + q * 4 + reg;
    rows[reg] = slot < a.n_dst ? (a.perm ? a.perm[slot] : slot) : -1;
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    if (rows[reg] < 0) continue;
    float* prow = a.dst + static_cast<long long>(rows[reg]) * a.ld_dst;
    if constexpr (NB % 4 == 0) {
#pragma unroll
      for (int v = 0; v < NB / 4; ++v) {
        const int col = WT ? (NB * n + 4 * v) : (64 * v + 4 * n);
        float4 o = make_float4(acc[4 * v][reg], acc[4 * v + 1][reg], acc[4 * v + 2][reg], acc[4 * v + 3][reg]);
        if (a.bias) { const float4 b = *reinterpret_cast<const float4*>(a.bias + col); o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w; }
        float4* p = reinterpret_cast<float4*>(prow + col);
        if (a.accumulate) { const float4 old = *p; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *p = o;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int col = NB * n + i;  // WT only (forward requires NB % 4 == 0)
        const float v = acc[i][reg] + (a.bias ? a.bias[col] : 0.f);
        prow[col] = a.accumulate ? (prow[col] + v) : v;
      }
    }
  }
  if (a.stats) {
    float vals[NB][4];
    int cols[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      cols[i] = (NB % 4 == 0) ? (WT ? (NB * n + i) : (64 * (i / 4) + 4 * n + (i % 4))) : (NB * n + i);
      const float b = a.bias ? a.bias[cols[i]] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) vals[i][reg] = acc[i][reg] + b;
    }
    tile_stats_rows<NB>(a, vals, cols, rows, &sacc);
  }
  {
    int scols[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) scols[i] = (NB % 4 == 0) ? (WT ? (NB * n + i) : (64 * (i / 4) + 4 * n + (i % 4))) : (NB * n + i);
    conv_stats_done<NB>(a, lds, sacc, scols, blockIdx.x, gridDim.x, 1);
  }
}

// ---- thin layers (16 / 32 channels, 27 offsets): a wave, 16 destination rows, nothing staged, nothing shared -------------------------
// At 16 / 32 channels an offset is 4 / 16 MFMAs per 16 rows: conv_rows_pipe / conv_rows_dma (W_k through LDS, a barrier per offset,
// gathers one offset ahead) spend their time on the barriers and on exposed gather latency - subm 32 -> 32 at 39 k rows took 35 us with
// the matrix pipe 0.28 busy.  Here a wave works alone: its 27 table entries are read up front, the rows AND the weight fragment of
// offset k + 2 are requested (straight from L2 / L1 into registers, 1 - 4 KB per offset) while offset k multiplies, and offsets at which
// none of the 16 rows has a neighbour skip the MFMAs (wave-uniform; their loads are issued all the same: every iteration has the same
// loads, so the compiler's counted waits stay exact).  No LDS, no barrier, occupancy limited by registers only.
// Non-transposed product (rows as A, weights as B): the accumulators have the layout conv_epilogue expects.  B of lane (n, g), k-step
// (j, t), column tile nb = W_k[16 j + 4 g + t][16 nb + n]: 16 bytes along k in the transposed layout (WT), four dwords otherwise.
// PRE: BatchNorm (+ReLU) of the source rows applied on the gather (ConvArgs::pre_*: the lane's 4 J channels' parameters in registers).
// DEPTH: offsets whose rows and weight fragment are in flight (8 registers per stage at 16 channels).  Two until round 6; at 35 k rows the
// launch has two waves per SIMD and every wave walks 27 dependent-latency offsets - see the dispatcher for what deeper prefetch measured.
template <int CINP, int NB, bool WT, int KVOL, bool PRE = false, int DEPTH = 2>
__global__ __launch_bounds__(256) void conv_rows_thin(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // only conv_stats_done's scratch
  constexpr int J = CINP / 16, D = DEPTH, NST = D + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int row0 = (xcd_major_tile(blockIdx.x, gridDim.x) * 4 + wave) * 16;
  StatAcc<NB> sacc;
  if (row0 < a.n_dst) {   // (no barrier in the conv itself; waves past the end only take part in conv_stats_done)
  const int my_row = row0 + r;
  PreNorm4 pn[PRE ? J : 1];
  if constexpr (PRE) {
#pragma unroll
    for (int j = 0; j < J; ++j) prenorm_load(a, 16 * j + 4 * g, pn[j]);
  }
  const bool row_ok = my_row < a.n_dst;
  int tv[KVOL];
#pragma unroll
  for (int k = 0; k < KVOL; ++k) tv[k] = row_ok ? a.tab[static_cast<long long>(a.flip ? (KVOL - 1 - k) : k) * a.n_dst + my_row] : -1;
  unsigned live = 0;   // bit k: some row of the wave has a neighbour at offset k
#pragma unroll
  for (int k = 0; k < KVOL; ++k) live |= (__ballot(tv[k] >= 0) != 0ull ? 1u : 0u) << k;
  struct Stage { float4 x[J]; float w[J][NB][4]; };
  Stage st[NST];
  auto request = [&](auto k_, Stage& s) {
    constexpr int k = decltype(k_)::value;
    const float* p = tv[k] >= 0 ? a.src + static_cast<long long>(tv[k]) * a.ld_src + 4 * g : g_zero_row + 4 * g;
    const float* wk = a.w + static_cast<long long>(k) * a.w_kstride;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      s.x[j] = *reinterpret_cast<const float4*>(p + 16 * j);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if constexpr (WT) {
          const float4 q = *reinterpret_cast<const float4*>(wk + static_cast<long long>(16 * nb + r) * a.w_ld + 16 * j + 4 * g);
          s.w[j][nb][0] = q.x; s.w[j][nb][1] = q.y; s.w[j][nb][2] = q.z; s.w[j][nb][3] = q.w;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) s.w[j][nb][t] = wk[static_cast<long long>(16 * j + 4 * g + t) * a.w_ld + 16 * nb + r];
        }
      }
    }
  };
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  static_for<0, (D < KVOL ? D : KVOL)>([&](auto k_) { request(k_, st[decltype(k_)::value % NST]); });
  static_for<0, KVOL>([&](auto k_) {
    constexpr int k = decltype(k_)::value;
    if constexpr (k + D < KVOL) request(std::integral_constant<int, k + D>{}, st[(k + D) % NST]);
    __builtin_amdgcn_sched_barrier(0);   // the requests of k + 2 stay in front of the products of k
    if ((live >> k) & 1u) {
      const Stage& s = st[k % NST];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        float xv[4] = {s.x[j].x, s.x[j].y, s.x[j].z, s.x[j].w};
        if constexpr (PRE) {
          const f32x4 v = prenorm_apply(pn[j], f32x4{xv[0], xv[1], xv[2], xv[3]}, tv[k] >= 0, a.pre_relu);
          xv[0] = v[0]; xv[1] = v[1]; xv[2] = v[2]; xv[3] = v[3];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t], s.w[j][nb][t], acc[nb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  });
  conv_epilogue<NB>(a, acc, row0, &sacc);
  }
  int scols[NB];
  epilogue_cols<NB>(scols);
  conv_stats_done<NB>(a, lds, sacc, scols, blockIdx.x, gridDim.x, 1);
}

// ---- thin layers with ALL 27 weight matrices resident in LDS (32 channels: 108 KB, one workgroup of 16 waves per CU) -------------------
// conv_rows_pipe / conv_rows_dma bring W_k (4 KB at 32 -> 32) into LDS once per offset and 64-row workgroup, with a barrier per offset: at
// 8.5 pairs per row that is 108 KB of weights through the vector-memory path for 70 KB of gathered rows, and the path's issue rate is what
// bounds these kernels (see conv_rows_thin).  Here a workgroup stays on its CU: the 27 matrices are laid down in LDS once, in MFMA
// fragment order, and the 16 waves then walk 16-row groups on their own (conv_rows_thin's loop with the weight fragment read from LDS:
// table entries up front, rows two offsets ahead, MFMAs skipped where no row has a neighbour, no barrier after the prologue).  Workgroup
// b takes the b-th of gridDim.x contiguous ranges of groups, XCD-major.
// PRE (16 source channels only): BatchNorm (+ReLU) of the source rows on the gather, parameters of the lane's four channels in registers
// (at 32 channels the same pushed this 128-register kernel into scratch: 66.8 against 47.1 us)
template <int CINP, int NB, bool WT, int KVOL, bool PRE = false>
__global__ __launch_bounds__(1024) void conv_rows_res(ConvArgs a, int groups_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [k][j][nb][lane][4]: B fragment of lane (n, g), k-step (j, t), tile nb at t
  constexpr int J = CINP / 16, D = 2, NST = D + 1, WSZ = CINP * NB * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  for (int e = threadIdx.x; e < KVOL * WSZ; e += 1024) {
    const int k = e / WSZ, q = e % WSZ;
    int c, col;
    if (WT) { col = q / CINP; c = q % CINP; } else { c = q / (NB * 16); col = q % (NB * 16); }   // coalesced in either layout
    const float v = a.w[static_cast<long long>(k) * a.w_kstride + (WT ? static_cast<long long>(col) * a.w_ld + c : static_cast<long long>(c) * a.w_ld + col)];
    lds[k * WSZ + ((((c >> 4) * NB + (col >> 4)) * 64) + ((c >> 2) & 3) * 16 + (col & 15)) * 4 + (c & 3)] = v;
  }
  __syncthreads();
  StatAcc<NB> sacc;
  PreNorm4 pn[PRE ? J : 1];
  if constexpr (PRE) {
#pragma unroll
    for (int j = 0; j < J; ++j) prenorm_load(a, 16 * j + 4 * g, pn[j]);
  }
  const int groups = (a.n_dst + 15) / 16;
  const int first = xcd_major_tile(blockIdx.x, gridDim.x) * groups_per_wg;
  const int last = min(first + groups_per_wg, groups);
  for (int grp = first + wave; grp < last; grp += 16) {
    const int row0 = grp * 16, my_row = row0 + r;
    const bool row_ok = my_row < a.n_dst;
    int tv[KVOL];
#pragma unroll
    for (int k = 0; k < KVOL; ++k) tv[k] = row_ok ? a.tab[static_cast<long long>(a.flip ? (KVOL - 1 - k) : k) * a.n_dst + my_row] : -1;
    unsigned live = 0, mine = 0;   // bit k: some row of the group / this lane's row has a neighbour at offset k
#pragma unroll
    for (int k = 0; k < KVOL; ++k) { live |= (__ballot(tv[k] >= 0) != 0ull ? 1u : 0u) << k; if constexpr (PRE) mine |= (tv[k] >= 0 ? 1u : 0u) << k; }
    float4 x[NST][J];
    auto request = [&](auto k_, float4 (&dst)[J]) {
      constexpr int k = decltype(k_)::value;
      const float* p = tv[k] >= 0 ? a.src + static_cast<long long>(tv[k]) * a.ld_src + 4 * g : g_zero_row + 4 * g;
#pragma unroll
      for (int j = 0; j < J; ++j) dst[j] = *reinterpret_cast<const float4*>(p + 16 * j);
    };
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_for<0, D>([&](auto k_) { request(k_, x[decltype(k_)::value % NST]); });
    // the weight fragment of offset k + 1 is read from LDS while offset k multiplies (every offset: an LDS read costs less than a branch)
    float4 bv[2][J][NB];
    auto fragment = [&](int k, float4 (&dst)[J][NB]) {
      const float* wk = lds + k * WSZ;
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dst[j][nb] = *reinterpret_cast<const float4*>(&wk[((j * NB + nb) * 64 + lane) * 4]);
    };
    fragment(0, bv[0]);
    static_for<0, KVOL>([&](auto k_) {
      constexpr int k = decltype(k_)::value;
      if constexpr (k + D < KVOL) request(std::integral_constant<int, k + D>{}, x[(k + D) % NST]);
      if constexpr (k + 1 < KVOL) fragment(k + 1, bv[(k + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // the requests of k + 2 and the fragment of k + 1 stay in front of the products of k
      if ((live >> k) & 1u) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          float4 xv = x[k % NST][j];
          if constexpr (PRE) {
            const f32x4 v = prenorm_apply(pn[j], f32x4{xv.x, xv.y, xv.z, xv.w}, (mine >> k) & 1u, a.pre_relu);
            xv = make_float4(v[0], v[1], v[2], v[3]);
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.x, bv[k & 1][j][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.y, bv[k & 1][j][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.z, bv[k & 1][j][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.w, bv[k & 1][j][nb].w, acc[nb], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    conv_epilogue<NB>(a, acc, row0, &sacc);
  }
  int scols[NB];
  epilogue_cols<NB>(scols);
  conv_stats_done<NB>(a, lds, sacc, scols, blockIdx.x, gridDim.x, 1);
}

// ---- pair-compacted tile with the reduction dimension split over the waves ---------------------------------------------------
// At the 5-10 pairs per row of the backbones' deep levels an output-stationary 16-row MFMA group is mostly zero rows: of
// the 27 offsets a tile visits, a row has a neighbour at ~38 % of them, so ~60 % of the MFMA issue slots of conv_rows_dma
// multiply zeros (and its 64 KB weight double buffer limits a CU to two workgroups, which the 155 tiles of a 10 k-row level
// cannot even fill).  Here the MFMA work follows the pair count:
//   * prologue: the tile's slice of the table goes through one ballot compaction per offset into LDS
//     ([offset][slot] -> source row / tile row, padded with "no row"), offsets without any pair are dropped from the loop;
//   * wave w owns the source channels [w CINP/4, (w+1) CINP/4) for ALL rows of the tile and all 64 columns of the launch:
//     its slice of W_k (CINP/4 x 64 floats = 16 or 32 VGPRs) comes straight from L2 into registers, no LDS staging, no
//     barrier in the offset loop, no imbalance between the waves whatever the compaction leaves;
//   * per offset only ceil(pairs / 16) row groups are gathered (one or two 16-byte loads per lane) and multiplied.  Three
//     offsets are in flight per wave: the slot lists of offset j+2 are being read from LDS, the weights and rows of offset
//     j+1 are on their way from L2, offset j multiplies;
//   * the products are taken transposed (weights as the MFMA's A operand, gathered rows as B): a lane then holds 16
//     columns of ONE compacted row, as four 4-register accumulators.  The wave's own LDS buffer keeps a tile row's partial
//     sums in exactly that order, so a group's accumulators are four 16-byte reads straight into the MFMA C operand (issued
//     while the group before multiplies) and four 16-byte writes after — no adds, no register shuffles (LDS float atomics
//     measured 10x slower).  A row occurs once per offset and a wave's LDS operations execute in order, so every sum runs
//     over ascending k: deterministic; the epilogue adds the four waves' partial sums in wave order.
// LDS at TM = 64 rows per tile: 4 x 65 x 68 accumulators + the compacted table = 79 KB -> two workgroups per CU; TM = 32: 40 KB.
// ABL (development, FV2P_KSPLIT_ABL, results INVALID - timing only; profiles/r05_ksplit_ablation.txt): bit 0 drops the steady-state global
// loads, bit 1 the accumulator round trip through LDS, bit 2 replaces every MFMA by four lane-wise FMAs, bit 3 drops the step bookkeeping
// (every step re-uses step 0's slot lists).
// PRE: BatchNorm (+ReLU) of the source rows applied to the gathered pieces before they enter the MFMAs (ConvArgs::pre_*; the parameters
// of the lane's 4 JS source channels stay in registers for the whole launch).
template <int CINP, bool WT, int TM, int GPS = 1, int ABL = 0, bool PRE = false>
__global__ __launch_bounds__(256) void conv_rows_ksplit(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KS = CINP / 4;      // source channels per wave
  constexpr int JS = KS / 16;       // 16-byte row pieces per lane and group
  constexpr int NBV = KS / 4;       // float4 weight registers per wave and offset
  constexpr int MAXK = 32;
  constexpr int MAXG = TM / 16;     // row groups an offset can need
  constexpr int ROWS = TM + 1;      // tile rows + one row that takes the padding slots' sums of zeros
  constexpr int LDR = 68;           // floats per accumulator row: 64 + 4, consecutive rows start 4 banks apart
  float* accl = lds;                                              // [wave][row ROWS][LDR]; column c of a row at (c >> 4) * 16 + (c & 3) * 4 + ((c >> 2) & 3)
  int* s_idx = reinterpret_cast<int*>(lds + 4 * ROWS * LDR);      // [MAXK][TM] source row of slot p (-1: padding)
  unsigned char* s_row = reinterpret_cast<unsigned char*>(s_idx + MAXK * TM);   // [MAXK][TM] tile row of slot p
  int* s_cnt = reinterpret_cast<int*>(s_row + MAXK * TM);         // [MAXK]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;
  // rows of this workgroup: a fixed TM-row tile, or — with a plan (fv2p_conv_plan_build) — the t-th of gridDim.x row ranges of
  // equal cost (pairs per row, floored), taken in sub-tiles of at most TM rows of equal size
  // One grid dimension, (tile, column block) decoded XCD-major: the two 64-column halves of a 128-column layer are neighbours in that
  // order, i.e. run on the SAME XCD at about the same time, and the rows the second one gathers are L2 hits.  As grid.y they were
  // gridDim.x workgroups apart - on another XCD, whose L2 fetched every row again from the fabric.
  const int ncb = a.col_blocks > 1 ? a.col_blocks : 1;
  const int lp = xcd_major_tile(blockIdx.x, gridDim.x);
  const int tile = lp / ncb, cblk = lp % ncb;
  int r_begin = tile * TM, r_end = min(r_begin + TM, a.n_dst);
  if (a.plan) { r_begin = a.plan[tile]; r_end = a.plan[tile + 1]; }
  const int n_sub = (r_end - r_begin + TM - 1) / TM;
  const int sub_rows = n_sub > 0 ? (r_end - r_begin + n_sub - 1) / n_sub : 0;
  StatAcc<4> sacc;
  if (ncb > 1) {   // column split as in conv_rows_dma
    const int off = cblk * 64;
    a.w += WT ? static_cast<long long>(off) * a.w_ld : off;
    a.dst += off;
    a.fin_col0 = off;
    if (a.bias) a.bias += off;
    if (a.stats) a.stats += off;
    if (a.bn_x) {
      a.bn_x += off; a.bn_mean += off; a.bn_invstd += off;
      if (a.bn_gamma) a.bn_gamma += off;
      if (a.bn_beta) a.bn_beta += off;
    }
  }
  a.c_dst = 64;
  PreNorm4 pn[PRE ? JS : 1];
  if constexpr (PRE) {
#pragma unroll
    for (int jj = 0; jj < JS; ++jj) prenorm_load(a, KS * wave + 16 * jj + 4 * (lane >> 4), pn[jj]);
  }
  unsigned long long t_begin = 0, t_pro = 0, t_wait = 0, t_mark = 0, t_issue = 0, t_comp = 0;   // t_issue: the 100 MHz SoC clock at the start (one time base for all XCDs)
  if (a.trace) { t_begin = __builtin_readcyclecounter(); t_issue = wall_clock64(); }
  for (int row0 = r_begin; row0 < r_end; row0 += sub_rows) {
  const int row_end = min(row0 + sub_rows, r_end);
  {
    float4* z = reinterpret_cast<float4*>(accl);
    for (int e = tid; e < 4 * ROWS * LDR / 4; e += 256) z[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  {
    // the wave's offsets k = wave, wave + 4, ...: all table loads first, then one ballot compaction each
    const int row = row0 + lane;
    const bool mine = lane < TM && row < row_end;
    int tv[MAXK / 4];
#pragma unroll
    for (int u = 0; u < MAXK / 4; ++u) {
      const int k = wave + 4 * u;
      tv[u] = (mine && k < a.kvol) ? a.tab[static_cast<long long>(a.flip ? (a.kvol - 1 - k) : k) * a.n_dst + row] : -1;
    }
#pragma unroll
    for (int u = 0; u < MAXK / 4; ++u) {
      const int k = wave + 4 * u;
      if (k < a.kvol) {   // uniform
        const int t = tv[u];
        const unsigned long long vote = __ballot(t >= 0);
        const int cnt = __popcll(vote), below = __popcll(vote & lanemask_lt());
        const int slot = t >= 0 ? below : cnt + (lane - below);   // lanes < TM: a permutation of 0..TM-1, every slot is written
        if (slot < TM) {
          s_idx[k * TM + slot] = t;
          s_row[k * TM + slot] = static_cast<unsigned char>(t >= 0 ? lane : TM);   // padding: the spare row
        }
        if (lane == 0) s_cnt[k] = cnt;
      }
    }
  }
  __syncthreads();
  const int cntv = lane < a.kvol ? s_cnt[lane] : 0;   // lane k: pairs of offset k in this tile
  // per-lane byte offsets of the weight registers relative to W_k
  int boff[NBV];
#pragma unroll
  for (int u = 0; u < NBV; ++u) {
    const int jj = u >> 2, x = u & 3;   // !WT: x = t (k-step), WT: x = i (column block)
    boff[u] = WT ? ((4 * r + x) * a.w_ld + KS * wave + 16 * jj + 4 * g) * 4 : ((KS * wave + 16 * jj + 4 * g + x) * a.w_ld + 4 * r) * 4;
  }
  const float* zero_row = g_zero_row + KS * wave + 4 * g;
  const long long src_col = KS * wave + 4 * g;
  float* my_acc = accl + wave * ROWS * LDR + g * 16;   // + row * LDR: this lane's 16 columns of a row (four 16-byte words)

  // ---- the tile's work as a flat list of STEPS, one 16-slot row group of one offset each (offsets ascending, groups ascending) --------
  // Round 2 walked the offsets and issued all loads of the next offset in one burst before the MFMA block.  Measured (round 3, trace of
  // wave 0): that burst costs 285 clocks per group with one workgroup per CU and 900 - 1070 with two — the CU's single vector-memory
  // address unit takes 16 clocks per dwordx4 wave instruction (64 B/clk), eight waves queue on it, and a wave that is stuck issuing
  // loads issues no MFMA: the matrix pipes idled 43 % of the time at either occupancy.  Now every step has the same shape (the weight
  // slice: NBV loads, one row group: JS loads), so its loads are spread over the MFMA block of the step before, one load behind each of
  // the first MFMAs (32 clocks apart, twice the address unit's time per load): the queue never builds up and the pipes keep running.
  // A second group of an offset reloads the offset's weights (L1 hits, +8 loads for ~1 step in 5).  Same shape for every step also
  // means no conditional load: nothing for the register allocator to merge (tools/check_async_asm.py).
  unsigned char* my_steps = reinterpret_cast<unsigned char*>(s_cnt + MAXK) + wave * 112;   // [wave][<= 27 * MAXG + 1] : k | first group << 5 | past-the-end << 7
  // GPS row groups per step (template): 1 where an offset rarely has more than 16 pairs in a tile (the KITTI levels under a tiling
  // plan), 2 for full 64-row tiles of dense levels (three groups per offset are the rule there: the weights are loaded once per two
  // groups; an offset with an odd number of groups multiplies one group of zero rows)
  int n_steps;
  {
    const int ns = (((cntv + 15) >> 4) + GPS - 1) / GPS;   // lane k: steps of offset k
    int incl = ns;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    n_steps = __builtin_amdgcn_readlane(incl, 31);
    const int first = incl - ns;
    for (int gq = 0; gq < ns; ++gq) my_steps[first + gq] = static_cast<unsigned char>(lane | ((gq * GPS) << 5));
    if (lane == 0) my_steps[n_steps] = 0x80;   // one step past the end (offset 0's weights, the zero row): loads that nobody uses
  }
  struct Step { int k, ng; int idx[GPS], row[GPS]; };   // ng: row groups of the step that hold pairs (uniform)
  auto read_step = [&](int j, Step& m) {   // j <= n_steps
    if constexpr (ABL & 8) { if (j > 1) return; }
    const int st = __builtin_amdgcn_readfirstlane(static_cast<int>(my_steps[j]));
    m.k = st & 0x1f;
    const int grp = (st >> 5) & 3;
    const bool dummy = (st & 0x80) != 0;   // uniform
    m.ng = dummy ? 0 : min(GPS, ((__builtin_amdgcn_readlane(cntv, m.k) + 15) >> 4) - grp);
#pragma unroll
    for (int u = 0; u < GPS; ++u) {
      const bool none = dummy || grp + u >= MAXG;   // uniform; slots past the offset's pairs hold (-1, spare row) already
      m.idx[u] = none ? -1 : s_idx[m.k * TM + (grp + u) * 16 + r];
      m.row[u] = none ? TM : s_row[m.k * TM + (grp + u) * 16 + r];
    }
  };
  constexpr int NLOAD = NBV + GPS * JS;
  // load number q of a step: the weight registers first, then the row pieces
  auto load_one = [&](auto q_, const Step& m, f32x4 (&A)[GPS][JS], f32x4 (&B)[NBV]) {
    constexpr int q = decltype(q_)::value;
    if constexpr (q < NBV) {
      const float* wk = a.w + static_cast<long long>(m.k) * a.w_kstride;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(B[q]) : "v"(boff[q]), "s"(wk) : "memory");
    } else if constexpr (q < NLOAD) {
      constexpr int u = (q - NBV) / JS, jj = (q - NBV) % JS;
      const float* p = m.idx[u] >= 0 ? a.src + static_cast<long long>(m.idx[u]) * a.ld_src + src_col : zero_row;
      load16_at<64 * jj>(A[u][jj], p);
    }
  };
  auto issue_all = [&](const Step& m, f32x4 (&A)[GPS][JS], f32x4 (&B)[NBV]) {
    static_for<0, NLOAD>([&](auto q) { load_one(q, m, A, B); });
  };
  auto wait = [&](f32x4 (&A)[GPS][JS], f32x4 (&B)[NBV]) {
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
    for (int u = 0; u < GPS; ++u)
#pragma unroll
      for (int jj = 0; jj < JS; ++jj) asm volatile("" : "+v"(A[u][jj]));
#pragma unroll
    for (int u = 0; u < NBV; ++u) asm volatile("" : "+v"(B[u]));
  };
  // one step: per group the accumulators of its 16 slots from LDS, JS * 16 MFMAs, accumulators back; the loads of step `nxt` sit
  // behind the first MFMAs.  acc[i] of lane (g, r) is column 16 g + 4 reg + i of slot r
  auto compute = [&](const Step& m, const f32x4 (&A)[GPS][JS], const f32x4 (&B)[NBV], const Step& nxt, f32x4 (&An)[GPS][JS], f32x4 (&Bn)[NBV]) {
    constexpr int QUADS = JS * 4;                                  // blocks of four MFMAs (128 clocks of matrix pipe) of one group
    constexpr int LPQ = (NLOAD + QUADS - 2) / (QUADS - 1);         // loads behind each block of the FIRST group (always multiplied): every
    static_for<0, GPS>([&](auto u_) {                              // load of the next step is issued whatever this step's group count
      constexpr int u = decltype(u_)::value;
      if (u == 0 || u < m.ng) {   // uniform: a step's later groups exist for offsets with more than 16 u pairs only
        f32x4* p = reinterpret_cast<f32x4*>(my_acc + m.row[u] * LDR);
        // Two-level sum, as the reference's per-offset GEMM + scatter-add (spconv_ops.h:300-362): the offset's product starts from zero
        // and is ADDED to the row's running sum afterwards.  One fused multiply-add chain over all (offset, channel) terms - the round 1-3
        // form, accumulators read from LDS before the MFMAs - is 27 x longer and sat 1.5 - 2.5 x further from a float64 run than
        // torch's blocked GEMMs (tools/f64_gap.py); the 16 extra adds per group are free beside 16 JS MFMAs.
        f32x4 c[4], sum[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { c[i] = f32x4{0.f, 0.f, 0.f, 0.f}; if constexpr (ABL & 2) sum[i] = A[u][0]; else sum[i] = p[i]; }   // the running sums travel under the MFMAs
        f32x4 Av[JS];
#pragma unroll
        for (int jj = 0; jj < JS; ++jj) {
          if constexpr (PRE) Av[jj] = prenorm_apply(pn[jj], A[u][jj], m.idx[u] >= 0, a.pre_relu);
          else Av[jj] = A[u][jj];
        }
        static_for<0, JS * 4>([&](auto q_) {
          constexpr int q = decltype(q_)::value, jj = q / 4, t = q % 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (ABL & 4) c[i][t] += (WT ? B[jj * 4 + i][t] : B[jj * 4 + t][i]) * Av[jj][t];
            else c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(WT ? B[jj * 4 + i][t] : B[jj * 4 + t][i], Av[jj][t], c[i], 0, 0, 0);
          }
          if constexpr (u == 0 && !(ABL & 1))
            static_for<0, LPQ>([&](auto v_) { load_one(std::integral_constant<int, q * LPQ + decltype(v_)::value>{}, nxt, An, Bn); });
        });
        if constexpr (ABL & 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { f32x4 v = sum[i] + c[i]; asm volatile("" : : "v"(v)); }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) p[i] = sum[i] + c[i];
        }
      }
    });
    static_assert(LPQ * QUADS >= NLOAD, "every load of the next step must be issued inside the first group's block");
  };
  if (a.trace) t_pro = __builtin_readcyclecounter();
  if (n_steps > 0) {
    f32x4 A0[GPS][JS], B0[NBV], A1[GPS][JS], B1[NBV];
    Step m0, m1;
    read_step(0, m0);
    issue_all(m0, A0, B0);
    read_step(1, m1);
    for (int j = 0; j < n_steps; j += 2) {
      if (a.trace) t_mark = __builtin_readcyclecounter();
      wait(A0, B0);
      if (a.trace) { const unsigned long long t = __builtin_readcyclecounter(); t_wait += t - t_mark; t_mark = t; }
      compute(m0, A0, B0, m1, A1, B1);
      if (a.trace) t_comp += __builtin_readcyclecounter() - t_mark;
      if (j + 1 >= n_steps) break;
      read_step(j + 2 <= n_steps ? j + 2 : n_steps, m0);
      if (a.trace) t_mark = __builtin_readcyclecounter();
      wait(A1, B1);
      if (a.trace) { const unsigned long long t = __builtin_readcyclecounter(); t_wait += t - t_mark; t_mark = t; }
      compute(m1, A1, B1, m0, A0, B0);
      if (a.trace) t_comp += __builtin_readcyclecounter() - t_mark;
      read_step(j + 3 <= n_steps ? j + 3 : n_steps, m1);
    }
  }
  // the loads of the step past the end (issued during the last step, used by nobody) are drained before their registers are reused
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
  if (a.trace && tid == 0) {   // the trace buffer has 4 records per 64 rows: every workgroup of a <= 2 x 2 split launch
    const int rec = blockIdx.x;
    if (rec < 4 * ((a.n_dst + 63) / 64)) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      a.trace[rec * 8 + 0] = hw;
      a.trace[rec * 8 + 1] = (xcc & 0xffu) | ((wall_clock64() - t_issue) << 32);   // workgroup lifetime in 10 ns ticks above the XCC id
      a.trace[rec * 8 + 2] = t_begin;
      a.trace[rec * 8 + 3] = __builtin_readcyclecounter();
      a.trace[rec * 8 + 4] = t_pro;
      a.trace[rec * 8 + 5] = t_wait;   // clocks wave 0 spent waiting for the prefetched rows and weights
      a.trace[rec * 8 + 6] = t_issue;  // SoC clock (10 ns ticks) at the workgroup's start
      a.trace[rec * 8 + 7] = t_comp;   // ... in the accumulator read / MFMA / accumulator write blocks
    }
  }
  __syncthreads();
  // epilogue: wave w stores tile rows 16 w .. 16 w + 15; lane (q, n) holds rows 4 q + reg, columns 4 n + i like an MFMA tile
  if (wave < MAXG) {
  const int q = lane >> 4, n = lane & 15;
  f32x4 acc[4];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* at = accl + (16 * wave + 4 * q + reg) * LDR + (n >> 2) * 16 + i * 4 + (n & 3);   // column 4 n + i
      acc[i][reg] = ((at[0] + at[ROWS * LDR]) + at[2 * ROWS * LDR]) + at[3 * ROWS * LDR];
    }
  const int wrow0 = row0 + 16 * wave;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = wrow0 + q * 4 + reg;
    if (row >= row_end) continue;
    float4 o = make_float4(acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]);
    if (a.bias) { const float4 b = *reinterpret_cast<const float4*>(a.bias + 4 * n); o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w; }
    float4* p = reinterpret_cast<float4*>(a.dst + static_cast<long long>(row) * a.ld_dst + 4 * n);
    if (a.accumulate) { const float4 old = *p; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
    *p = o;
  }
  if (a.stats) {
    float vals[4][4];
    int cols[4], rows[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      cols[i] = 4 * n + i;
      const float b = a.bias ? a.bias[cols[i]] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) vals[i][reg] = acc[i][reg] + b;
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) rows[reg] = (wrow0 + q * 4 + reg < row_end) ? wrow0 + q * 4 + reg : -1;
    tile_stats_rows<4>(a, vals, cols, rows, &sacc);
  }
  }
  if (row0 + sub_rows < r_end) __syncthreads();   // the next sub-tile clears the accumulators the epilogue above reads
  }
  {
    int scols[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) scols[i] = 4 * (lane & 15) + i;
    conv_stats_done<4>(a, lds, sacc, scols, tile, static_cast<int>(gridDim.x) / ncb, ncb);
  }
}

// Compacted variant: the workgroup owns TM destination rows whose accumulators live in LDS.  For every kernel
// offset the rows that actually have a neighbour are compacted (wave64 ballot + prefix) into a list, and only
// ceil(n_k / 16) MFMA row groups are issued (round-robin over the 4 waves) instead of TM/16 — the MFMA work follows
// the real pair count P instead of K*N.  A group loads its 16 accumulator rows from LDS as the MFMA C operand,
// gathers its A fragment from global memory, and writes the rows back; a row occurs at most once per offset, so
// groups of one offset never collide, and the barrier between offsets orders the LDS traffic.
template <int CINP, int NB, bool WT, int TM>
__global__ __launch_bounds__(256) void conv_rows_cmp(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int J = CINP / 16;
  constexpr int COLS = NB * 16;
  constexpr int LDA = COLS + 16;               // accumulator row stride (floats)
  float* wl = lds;                             // W_k fragments: CINP*COLS floats
  float* accl = lds + CINP * COLS;             // [TM][LDA]
  int* lsrc = reinterpret_cast<int*>(accl + TM * LDA);  // [TM]
  int* lrow = lsrc + TM;                       // [TM]
  __shared__ int wave_cnt[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * TM;
  for (int e = tid; e < TM * LDA; e += 256) accl[e] = 0.f;
  for (int k = 0; k < a.kvol; ++k) {
    const int trow = a.flip ? (a.kvol - 1 - k) : k;
    int idx[TM / 256 > 0 ? TM / 256 : 1];
    int total = 0;
    // ---- compaction of this offset's valid rows (TM may exceed 256: process in slabs of 256 rows)
#pragma unroll
    for (int slab = 0; slab < (TM + 255) / 256; ++slab) {
      const int t = slab * 256 + tid;
      const int row = row0 + t;
      const int v = (t < TM && row < a.n_dst) ? a.tab[static_cast<long long>(trow) * a.n_dst + row] : -1;
      idx[slab] = v;
      const uint64_t vote = __ballot(v >= 0);
      if (lane == 0) wave_cnt[w] = __popcll(vote);
      __syncthreads();  // (first slab: also closes the previous offset's group phase)
      int pos = total + __popcll(vote & lanemask_lt());
      for (int ww = 0; ww < w; ++ww) pos += wave_cnt[ww];
      if (v >= 0) { lsrc[pos] = v; lrow[pos] = t; }
      total += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      if (slab + 1 < (TM + 255) / 256) __syncthreads();
    }
    if (total == 0) { __syncthreads(); continue; }
    stage_w_vec<CINP, NB, WT>(a, a.w + static_cast<long long>(k) * a.w_kstride, wl);
    __syncthreads();
    const int ngroups = (total + 15) >> 4;
    for (int gi = w; gi < ngroups; gi += 4) {
      const int e = gi * 16 + r;
      const int src = e < total ? lsrc[e] : -1;
      float4 av[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src >= 0 && (16 * j + 4 * g) < a.c_src)
          av[j] = *reinterpret_cast<const float4*>(a.src + static_cast<long long>(src) * a.ld_src + 16 * j + 4 * g);
      }
      int rr[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ee = gi * 16 + g * 4 + reg;
        rr[reg] = ee < total ? lrow[ee] : -1;
      }
      f32x4 acc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[nb][reg] = rr[reg] >= 0 ? accl[rr[reg] * LDA + nb * 16 + r] : 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        float4 bv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = *reinterpret_cast<const float4*>(&wl[((j * NB + nb) * 64 + lane) * 4]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[nb].x, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[nb].y, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[nb].z, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[nb].w, acc[nb], 0, 0, 0);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          if (rr[reg] >= 0) accl[rr[reg] * LDA + nb * 16 + r] = acc[nb][reg];
    }
  }
  __syncthreads();
  // epilogue: whole rows, coalesced
  for (int e = tid; e < TM * COLS; e += 256) {
    const int t = e / COLS, col = e % COLS;
    const int row = row0 + t;
    if (row < a.n_dst && col < a.c_dst) {
      float v = accl[t * LDA + col] + (a.bias ? a.bias[col] : 0.f);
      float* p = a.dst + static_cast<long long>(row) * a.ld_dst + col;
      *p = a.accumulate ? (*p + v) : v;
    }
  }
}

// any c_src <= 4*STEPS (scalar row loads): first layer (4 or 5 point features) and odd channel counts
template <int STEPS, int NB, bool WT>
__global__ __launch_bounds__(256) void conv_rows_scalar(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 64 + wave * 16;
  const int my_row = row0 + r;
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < a.kvol; ++k) {
    const int trow = a.flip ? (a.kvol - 1 - k) : k;
    const int idx = (my_row < a.n_dst) ? a.tab[static_cast<long long>(trow) * a.n_dst + my_row] : -1;
    float av[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      av[s] = 0.f;
      if (idx >= 0 && (4 * s + g) < a.c_src) av[s] = a.src[static_cast<long long>(idx) * a.ld_src + 4 * s + g];
    }
    const bool any = __ballot(idx >= 0) != 0ull;
    __syncthreads();
    stage_w_scalar<STEPS, NB, WT>(a, a.w + static_cast<long long>(k) * a.w_kstride, lds);
    __syncthreads();
    if (any) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], lds[(s * NB + nb) * 64 + lane], acc[nb], 0, 0, 0);
    }
  }
  StatAcc<NB> sacc;
  conv_epilogue<NB>(a, acc, row0, &sacc);
  int scols[NB];
  epilogue_cols<NB>(scols);
  conv_stats_done<NB>(a, lds, sacc, scols, blockIdx.x, gridDim.x, 1);
}

// ---- the first layer of a backbone: 4 (or 5 .. 8) point features -> 16 / 32 channels, full 3 x 3 x 3 kernel ---------------------------
// conv_rows_scalar stages every W_k (64 - 256 floats) through LDS with two barriers per offset: 27 x 2 barriers for 27 MFMAs per wave -
// 26.6 - 35.6 us for the 35 k rows of `conv_input`, twice the 16 -> 16 layer that follows it with four times the work.  Here a wave works
// alone, as in conv_rows_thin: its 27 table entries, then its 27 x STEPS feature dwords (lane (r, g): channel 4 s + g of row r's neighbour
// at offset k: the four lanes of a row read one 16-byte row piece) and the 27 x STEPS x NB weight dwords of ITS fragment position are all
// requested up front and stay in registers; then 27 x STEPS x NB MFMAs in conv_rows_scalar's order (same products, same sums: bit-identical
// results), skipped where no row of the wave has a neighbour.  No LDS, no barrier.
template <int STEPS, int NB, int KVOL>
__global__ __launch_bounds__(256) void conv_rows_first(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // only conv_stats_done's scratch
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int row0 = (blockIdx.x * 4 + wave) * 16;
  StatAcc<NB> sacc;
  if (row0 < a.n_dst) {
    const int my_row = row0 + r;
    const bool row_ok = my_row < a.n_dst;
    int tv[KVOL];
#pragma unroll
    for (int k = 0; k < KVOL; ++k) tv[k] = row_ok ? a.tab[static_cast<long long>(a.flip ? (KVOL - 1 - k) : k) * a.n_dst + my_row] : -1;
    float w[KVOL][STEPS][NB], f[KVOL][STEPS];
#pragma unroll
    for (int k = 0; k < KVOL; ++k)
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int ch = 4 * st + g;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int col = 16 * nb + r;
          w[k][st][nb] = (ch < a.c_src && col < a.c_dst) ? a.w[static_cast<long long>(k) * a.w_kstride + static_cast<long long>(ch) * a.w_ld + col] : 0.f;
        }
        f[k][st] = (tv[k] >= 0 && ch < a.c_src) ? a.src[static_cast<long long>(tv[k]) * a.ld_src + ch] : 0.f;
      }
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KVOL; ++k) {
      if (__ballot(tv[k] >= 0) != 0ull) {
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[k][st], w[k][st][nb], acc[nb], 0, 0, 0);
      }
    }
    conv_epilogue<NB>(a, acc, row0, &sacc);
  }
  int scols[NB];
  epilogue_cols<NB>(scols);
  conv_stats_done<NB>(a, lds, sacc, scols, blockIdx.x, gridDim.x, 1);
}

// ---- weight gradient ----------------------------------------------------------------------------
struct WgradArgs {
  const float* src; int ld_src; int c_src;     // gathered operand (features), rows = tab values
  const float* grad; int ld_grad; int c_grad;  // row-aligned operand (gradient wrt dst rows)
  const int* tab; int n_dst; int kvol; int flip;
  float* dw; long long dw_kstride; int dw_ld;   // dW_k[c_src][c_grad]
  int rows_per_chunk;
  int skip_k;                                    // offset handled by another launch, or -1
  int k_base, k_count;                           // this launch covers offsets [k_base, k_base + k_count)
  // BatchNorm (+ReLU) of the gathered operand applied where it enters the MFMA (the conv's forward pass normalised its source rows on the
  // gather: ConvArgs::pre_*); null = the rows are used as they are
  const float* pre_mean; const float* pre_invstd; const float* pre_gamma; const float* pre_beta; int pre_relu;
};
// the lane's channel `ch` of the gathered operand: value v -> relu?((v - mean) * invstd * gamma + beta), as bn_apply_fwd_k computes it.
// Padding rows (zeros) become relu?(beta - mean * ...) but meet a zero gradient row: their product stays zero.
struct PreNorm1 { float mean, is, ga, be; };
template <class Args>
__device__ __forceinline__ PreNorm1 prenorm1_load(const Args& a, int ch, int c_src) {
  PreNorm1 p{0.f, 1.f, 1.f, 0.f};
  if (a.pre_mean && ch < c_src) { p.mean = a.pre_mean[ch]; p.is = a.pre_invstd[ch]; p.ga = a.pre_gamma ? a.pre_gamma[ch] : 1.f; p.be = a.pre_beta ? a.pre_beta[ch] : 0.f; }
  return p;
}
__device__ __forceinline__ float prenorm1_apply(const PreNorm1& p, float v, int relu) {
  const float xhat = (v - p.mean) * p.is;
  const float t = xhat * p.ga + p.be;
  return (relu && t <= 0.f) ? 0.f : t;
}

// One WORKGROUP per (row chunk, kernel offset k).  The gathers are latency bound (a dependent table read, then rows
// scattered over L2/HBM), so the kernel is organised around memory-level parallelism:
//   1. the 256 threads compact the chunk's valid rows into an LDS pair list (ordered: ballot + wave prefix);
//   2. per stage of 64 pairs they fetch both operands COOPERATIVELY — every thread issues (CIN+COUT)/16 independent
//      16-byte loads (whole 256-byte rows, coalesced) for the next stage into registers while the current stage is on
//      the matrix pipe, then parks them in LDS (F[64][CIN], G[64][COUT], padded rows);
//   3. wave w owns cin block(s) w*MBW.. and all cout blocks: A_mfma[m = cin][kk = pair] and B_mfma[kk = pair][n = cout]
//      are 4-byte LDS reads, 16 k-slices per stage.
// The [Cin, Cout] tile goes to partial[chunk][k] straight from the accumulators (no atomics); a second kernel adds
// the chunks in a fixed order, so the gradient is deterministic.
constexpr int kWgradMaxChunk = 1024;
template <int MB, int NB>     // CINP = 16*MB, COUTP = 16*NB ; MBW = max(MB/4, 1) cin blocks per wave
__global__ __launch_bounds__(256) void conv_wgrad(WgradArgs a, float* __restrict__ partial) {
  constexpr int CINP = MB * 16, COUTP = NB * 16;
  constexpr int kWgStage = (MB * NB >= 64) ? 32 : 64;  // pairs per stage (keeps the LDS footprint under 64 KB)
  constexpr int MBW = MB >= 4 ? MB / 4 : 1;
  constexpr int LDF = CINP + 4, LDG = COUTP + 4;
  constexpr int RF = (kWgStage * CINP / 4 + 255) / 256, RG = (kWgStage * COUTP / 4 + 255) / 256;  // float4 units per thread per stage
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* F = lds;                                   // [64][LDF]
  float* G = F + kWgStage * LDF;                    // [64][LDG]
  int* s_src = reinterpret_cast<int*>(G + kWgStage * LDG);  // [rows_per_chunk]
  int* s_row = s_src + a.rows_per_chunk;
  __shared__ int wave_cnt[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 15, g = lane >> 4;
  const int kl = blockIdx.y, k = a.k_base + kl;
  if (k == a.skip_k) return;  // whole workgroup
  const int trow = a.flip ? (a.kvol - 1 - k) : k;
  const int r_begin = blockIdx.x * a.rows_per_chunk;
  const int r_end = min(r_begin + a.rows_per_chunk, a.n_dst);
  // ---- 1. ordered compaction
  int cnt = 0;
  for (int base = r_begin; base < r_end; base += 256) {
    const int row = base + tid;
    const int idx = (row < r_end) ? a.tab[static_cast<long long>(trow) * a.n_dst + row] : -1;
    const uint64_t vote = __ballot(idx >= 0);
    if (lane == 0) wave_cnt[w] = __popcll(vote);
    __syncthreads();
    int pos = cnt + __popcll(vote & lanemask_lt());
    for (int ww = 0; ww < w; ++ww) pos += wave_cnt[ww];
    if (idx >= 0) { s_src[pos] = idx; s_row[pos] = row; }
    cnt += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  f32x4 acc[MBW][NB];
#pragma unroll
  for (int i = 0; i < MBW; ++i)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[i][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- 2./3. staged gather + MFMA
  float4 rf[RF], rg[RG];
  auto fetch = [&](int p0) {
#pragma unroll
    for (int u = 0; u < RF; ++u) {
      const int e = tid + u * 256, pr = e / (CINP / 4), c = (e % (CINP / 4)) * 4;
      const int sr = (pr < kWgStage && p0 + pr < cnt) ? s_src[p0 + pr] : -1;
      rf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sr >= 0 && c < a.c_src) {
        const float* q = a.src + static_cast<long long>(sr) * a.ld_src + c;
        if (c + 3 < a.c_src && (a.ld_src & 3) == 0) rf[u] = *reinterpret_cast<const float4*>(q);
        else { rf[u].x = q[0]; if (c + 1 < a.c_src) rf[u].y = q[1]; if (c + 2 < a.c_src) rf[u].z = q[2]; if (c + 3 < a.c_src) rf[u].w = q[3]; }
      }
    }
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      const int e = tid + u * 256, pr = e / (COUTP / 4), c = (e % (COUTP / 4)) * 4;
      const int rr = (pr < kWgStage && p0 + pr < cnt) ? s_row[p0 + pr] : -1;
      rg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rr >= 0 && c < a.c_grad) {
        const float* q = a.grad + static_cast<long long>(rr) * a.ld_grad + c;
        if (c + 3 < a.c_grad && (a.ld_grad & 3) == 0) rg[u] = *reinterpret_cast<const float4*>(q);
        else { rg[u].x = q[0]; if (c + 1 < a.c_grad) rg[u].y = q[1]; if (c + 2 < a.c_grad) rg[u].z = q[2]; if (c + 3 < a.c_grad) rg[u].w = q[3]; }
      }
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int u = 0; u < RF; ++u) {
      const int e = tid + u * 256, pr = e / (CINP / 4), c = (e % (CINP / 4)) * 4;
      if (pr < kWgStage) *reinterpret_cast<float4*>(&F[pr * LDF + c]) = rf[u];
    }
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      const int e = tid + u * 256, pr = e / (COUTP / 4), c = (e % (COUTP / 4)) * 4;
      if (pr < kWgStage) *reinterpret_cast<float4*>(&G[pr * LDG + c]) = rg[u];
    }
  };
  PreNorm1 pn[MBW];
  const bool pre = a.pre_mean != nullptr;   // uniform
#pragma unroll
  for (int i = 0; i < MBW; ++i) pn[i] = prenorm1_load(a, (w * MBW + i) * 16 + m, a.c_src);
  if (cnt > 0) fetch(0);
  for (int p0 = 0; p0 < cnt; p0 += kWgStage) {
    __syncthreads();  // previous stage's LDS reads are done
    park();
    __syncthreads();
    if (p0 + kWgStage < cnt) fetch(p0 + kWgStage);
    const int npair = min(kWgStage, cnt - p0);
    if (w * MBW < MB) {
      for (int sl = 0; sl * 4 < npair; ++sl) {
        const int pr = sl * 4 + g;  // rows beyond npair were parked as zeros
        float av[MBW], bv[NB];
#pragma unroll
        for (int i = 0; i < MBW; ++i) { av[i] = F[pr * LDF + (w * MBW + i) * 16 + m]; if (pre) av[i] = prenorm1_apply(pn[i], av[i], a.pre_relu); }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = G[pr * LDG + nb * 16 + m];
#pragma unroll
        for (int i = 0; i < MBW; ++i)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[nb], acc[i][nb], 0, 0, 0);
      }
    }
  }
  // partial[chunk][k][c_src][c_grad]: C layout row = 4g + reg (cin), col = m (cout)
  if (w * MBW < MB) {
    float* out = partial + (static_cast<long long>(blockIdx.x) * a.k_count + kl) * a.c_src * a.c_grad;
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int cr = (w * MBW + i) * 16 + g * 4 + reg, cc = nb * 16 + m;
          if (cr < a.c_src && cc < a.c_grad) out[static_cast<long long>(cr) * a.c_grad + cc] = acc[i][nb][reg];
        }
  }
}

// ---- weight gradient from pair lists ---------------------------------------------------------------------------
// Same staged gather + MFMA as conv_wgrad, but the valid (source row, gradient row) pairs come from the rulebook's
// compacted pair lists (the reference's indice_pairs [K][2][n] + indice_pair_num [K], spconv_ops.h:403-455 walks the
// same lists): no per-launch compaction prologue, every workgroup gets pairs_per_chunk() real pairs whatever the offset's
// density (the dense centre offset of a submanifold conv needs no launch of its own), and workgroups past an offset's
// pair count exit at once.  The pair lists are built once per rulebook, off the training stream when prefetched.
constexpr int kPairsMax = 512;     // LDS room for a workgroup's pair indices
// pairs per workgroup: 256 up to ~1e5 rows (512 leaves a 2.3-round tail at KITTI size), 512 above (fewer partial tiles to fold)
static int pairs_per_chunk(long long pair_len) { return pair_len > 100000 ? 512 : 256; }

struct WgradPairArgs {
  const float* src; int ld_src; int c_src;     // operand indexed by pairs[k][side_src]
  const float* grad; int ld_grad; int c_grad;  // operand indexed by pairs[k][1 - side_src]
  const int* pairs; long long pair_len;         // [kvol][2][pair_len]
  const int* pair_num;                          // [kvol]
  int side_src;                                 // 0: src rows are pairs[k][0] (forward conv), 1: pairs[k][1] (inverse conv)
  int kvol;
  int chunk;                                    // pairs per workgroup (<= kPairsMax)
  const float* pre_mean; const float* pre_invstd; const float* pre_gamma; const float* pre_beta; int pre_relu;   // as WgradArgs
};

template <int MB, int NB>
__global__ __launch_bounds__(256) void conv_wgrad_pairs(WgradPairArgs a, float* __restrict__ partial) {
  constexpr int CINP = MB * 16, COUTP = NB * 16;
  constexpr int kWgStage = (MB * NB >= 64) ? 32 : 64;
  constexpr int MBW = MB >= 4 ? MB / 4 : 1;
  constexpr int LDF = CINP + 4, LDG = COUTP + 4;
  constexpr int RF = (kWgStage * CINP / 4 + 255) / 256, RG = (kWgStage * COUTP / 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* F = lds;
  float* G = F + kWgStage * LDF;
  int* s_src = reinterpret_cast<int*>(G + kWgStage * LDG);  // [kPairsMax]
  int* s_row = s_src + kPairsMax;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 15, g = lane >> 4;
  const int k = blockIdx.y;
  const int total = a.pair_num[k];
  const int p_begin = blockIdx.x * a.chunk;
  if (p_begin >= total) return;  // whole workgroup
  const int cnt = min(a.chunk, total - p_begin);
  {
    const int* pa = a.pairs + (static_cast<long long>(k) * 2 + a.side_src) * a.pair_len + p_begin;
    const int* pb = a.pairs + (static_cast<long long>(k) * 2 + 1 - a.side_src) * a.pair_len + p_begin;
    for (int e = tid; e < cnt; e += 256) { s_src[e] = pa[e]; s_row[e] = pb[e]; }
  }
  __syncthreads();
  f32x4 acc[MBW][NB];
#pragma unroll
  for (int i = 0; i < MBW; ++i)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[i][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 rf[RF], rg[RG];
  auto fetch = [&](int p0) {
#pragma unroll
    for (int u = 0; u < RF; ++u) {
      const int e = tid + u * 256, pr = e / (CINP / 4), c = (e % (CINP / 4)) * 4;
      const int sr = (pr < kWgStage && p0 + pr < cnt) ? s_src[p0 + pr] : -1;
      rf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sr >= 0 && c < a.c_src) {
        const float* q = a.src + static_cast<long long>(sr) * a.ld_src + c;
        if (c + 3 < a.c_src && (a.ld_src & 3) == 0) rf[u] = *reinterpret_cast<const float4*>(q);
        else { rf[u].x = q[0]; if (c + 1 < a.c_src) rf[u].y = q[1]; if (c + 2 < a.c_src) rf[u].z = q[2]; if (c + 3 < a.c_src) rf[u].w = q[3]; }
      }
    }
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      const int e = tid + u * 256, pr = e / (COUTP / 4), c = (e % (COUTP / 4)) * 4;
      const int rr = (pr < kWgStage && p0 + pr < cnt) ? s_row[p0 + pr] : -1;
      rg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rr >= 0 && c < a.c_grad) {
        const float* q = a.grad + static_cast<long long>(rr) * a.ld_grad + c;
        if (c + 3 < a.c_grad && (a.ld_grad & 3) == 0) rg[u] = *reinterpret_cast<const float4*>(q);
        else { rg[u].x = q[0]; if (c + 1 < a.c_grad) rg[u].y = q[1]; if (c + 2 < a.c_grad) rg[u].z = q[2]; if (c + 3 < a.c_grad) rg[u].w = q[3]; }
      }
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int u = 0; u < RF; ++u) {
      const int e = tid + u * 256, pr = e / (CINP / 4), c = (e % (CINP / 4)) * 4;
      if (pr < kWgStage) *reinterpret_cast<float4*>(&F[pr * LDF + c]) = rf[u];
    }
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      const int e = tid + u * 256, pr = e / (COUTP / 4), c = (e % (COUTP / 4)) * 4;
      if (pr < kWgStage) *reinterpret_cast<float4*>(&G[pr * LDG + c]) = rg[u];
    }
  };
  PreNorm1 pn[MBW];
  const bool pre = a.pre_mean != nullptr;   // uniform
#pragma unroll
  for (int i = 0; i < MBW; ++i) pn[i] = prenorm1_load(a, (w * MBW + i) * 16 + m, a.c_src);
  fetch(0);
  for (int p0 = 0; p0 < cnt; p0 += kWgStage) {
    __syncthreads();
    park();
    __syncthreads();
    if (p0 + kWgStage < cnt) fetch(p0 + kWgStage);
    const int npair = min(kWgStage, cnt - p0);
    if (w * MBW < MB) {
      for (int sl = 0; sl * 4 < npair; ++sl) {
        const int pr = sl * 4 + g;
        float av[MBW], bv[NB];
#pragma unroll
        for (int i = 0; i < MBW; ++i) { av[i] = F[pr * LDF + (w * MBW + i) * 16 + m]; if (pre) av[i] = prenorm1_apply(pn[i], av[i], a.pre_relu); }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = G[pr * LDG + nb * 16 + m];
#pragma unroll
        for (int i = 0; i < MBW; ++i)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[nb], acc[i][nb], 0, 0, 0);
      }
    }
  }
  if (w * MBW < MB) {
    float* out = partial + (static_cast<long long>(blockIdx.x) * a.kvol + k) * a.c_src * a.c_grad;
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int cr = (w * MBW + i) * 16 + g * 4 + reg, cc = nb * 16 + m;
          if (cr < a.c_src && cc < a.c_grad) out[static_cast<long long>(cr) * a.c_grad + cc] = acc[i][nb][reg];
        }
  }
}

// LDS-DMA variant for 64 / 128 channels on both sides (AS = c_src / 64, BS = c_grad / 64).  The pair rows of a stage are
// gathered straight into LDS by `global_load_lds_dwordx4` (lane l of a piece covers 16 bytes of row l / 16: whole 256-byte
// rows, no staging registers, no ds_write pass), double buffered, one barrier per stage.  MFMA operands come from LDS as
// the 16 bytes a lane owns: lane (m, g) reads unit m of pair row 4*sl + g — for the gradient side all four components
// (four MFMAs, output columns 4m + {0..3}), for the feature side component w of wave w (output rows 4m + w): one
// ds_read_b128 + one ds_read_b32 per four MFMAs instead of five ds_read_b32.  A lane therefore owns runs of four
// adjacent dW columns and the partial tile is written with float4 stores.
template <int AS, int BS, int ST>
__global__ __launch_bounds__(256) void conv_wgrad_pairs_dma(WgradPairArgs a, float* __restrict__ partial) {
  constexpr int CS = AS * 64, CG = BS * 64;             // ST = pairs per stage
  constexpr int ROWF = CS, ROWG = CG;                   // floats per staged row
  constexpr int STAGE = ST * (ROWF + ROWG);             // floats per stage buffer
  constexpr int PF = ST * (CS / 4) / 64, PG = ST * (CG / 4) / 64;  // DMA pieces (64 lanes x 16 B) per stage
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int* s_src = reinterpret_cast<int*>(lds + 2 * STAGE);
  int* s_row = s_src + kPairsMax;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, g = lane >> 4;
  const int k = blockIdx.y;
  const int total = a.pair_num[k];
  const int p_begin = blockIdx.x * a.chunk;
  if (p_begin >= total) return;  // whole workgroup
  const int cnt = min(a.chunk, total - p_begin);
  {
    const int* pa = a.pairs + (static_cast<long long>(k) * 2 + a.side_src) * a.pair_len + p_begin;
    const int* pb = a.pairs + (static_cast<long long>(k) * 2 + 1 - a.side_src) * a.pair_len + p_begin;
    for (int e = tid; e < a.chunk; e += 256) { s_src[e] = e < cnt ? pa[e] : -1; s_row[e] = e < cnt ? pb[e] : -1; }
  }
  __syncthreads();
  // piece q of a stage: rows [q * 64 / U, ...) of F (q < PF) or G; lane l -> row q*64/U + l/U, unit l%U  (U = units per row)
  auto issue = [&](int p0, float* buf) {
#pragma unroll
    for (int q = 0; q < (PF + PG + 3) / 4; ++q) {
      const int piece = q * 4 + w;
      if (piece < PF) {
        constexpr int U = CS / 4;
        const int row = piece * (64 / U) + lane / U, unit = lane % U;
        const int sr = s_src[p0 + row];
        const float* src = sr >= 0 ? a.src + static_cast<long long>(sr) * a.ld_src + unit * 4 : g_zero_row + unit * 4;
        glds16(src, buf + piece * 256);
      } else if (piece < PF + PG) {
        constexpr int U = CG / 4;
        const int pg = piece - PF;
        const int row = pg * (64 / U) + lane / U, unit = lane % U;
        const int rr = s_row[p0 + row];
        const float* src = rr >= 0 ? a.grad + static_cast<long long>(rr) * a.ld_grad + unit * 4 : g_zero_row + unit * 4;
        glds16(src, buf + ST * ROWF + pg * 256);
      }
    }
  };
  f32x4 acc[AS][BS * 4];
#pragma unroll
  for (int i = 0; i < AS; ++i)
#pragma unroll
    for (int j = 0; j < BS * 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  PreNorm1 pn[AS];
  const bool pre = a.pre_mean != nullptr;   // uniform
#pragma unroll
  for (int i = 0; i < AS; ++i) pn[i] = prenorm1_load(a, 4 * (m + 16 * i) + w, CS);
  issue(0, lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int it = 0;
  for (int p0 = 0; p0 < cnt; p0 += ST, ++it) {
    const float* F = lds + (it & 1) * STAGE;
    const float* G = F + ST * ROWF;
    if (p0 + ST < cnt) issue(p0 + ST, lds + ((it + 1) & 1) * STAGE);
    const int npair = min(ST, cnt - p0);
    for (int sl = 0; sl * 4 < npair; ++sl) {
      const int pr = sl * 4 + g;  // rows beyond npair were staged from the zero row
      float av[AS];
      float4 bv[BS];
#pragma unroll
      for (int i = 0; i < AS; ++i) { av[i] = F[pr * ROWF + (m + 16 * i) * 4 + w]; if (pre) av[i] = prenorm1_apply(pn[i], av[i], a.pre_relu); }
#pragma unroll
      for (int j = 0; j < BS; ++j) bv[j] = *reinterpret_cast<const float4*>(&G[pr * ROWG + (m + 16 * j) * 4]);
#pragma unroll
      for (int i = 0; i < AS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          acc[i][j * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j].x, acc[i][j * 4 + 0], 0, 0, 0);
          acc[i][j * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j].y, acc[i][j * 4 + 1], 0, 0, 0);
          acc[i][j * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j].z, acc[i][j * 4 + 2], 0, 0, 0);
          acc[i][j * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j].w, acc[i][j * 4 + 3], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // C layout: row = 4 * (lane >> 4) + reg is the A lane index m' -> cin = 4 * (m' + 16 i) + w; col n = lane & 15 -> cout = 4 * (n + 16 j) + t
  float* out = partial + (static_cast<long long>(blockIdx.x) * a.kvol + k) * CS * CG;
#pragma unroll
  for (int i = 0; i < AS; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int cin = 4 * (4 * g + reg + 16 * i) + w;
#pragma unroll
      for (int j = 0; j < BS; ++j)
        *reinterpret_cast<float4*>(&out[static_cast<long long>(cin) * CG + 4 * (m + 16 * j)]) =
            make_float4(acc[i][j * 4 + 0][reg], acc[i][j * 4 + 1][reg], acc[i][j * 4 + 2][reg], acc[i][j * 4 + 3][reg]);
    }
}

// Fixed-order sum of the tiles of the chunks that exist for each offset: ceil(pair_num[k] / chunk).
__global__ __launch_bounds__(256) void wgrad_reduce_pairs(const float* __restrict__ partial, const int* __restrict__ pair_num, int kvol, int chunk,
                                                          int c_src, int c_grad, float* __restrict__ dw, long long dw_kstride, int dw_ld) {
  __shared__ float part[16][17];
  const int e = threadIdx.x & 15, cg = threadIdx.x >> 4;
  const long long per_chunk = static_cast<long long>(kvol) * c_src * c_grad;
  const long long t = static_cast<long long>(blockIdx.x) * 16 + e;
  float s = 0.f;
  int k = 0;
  if (t < per_chunk) {
    k = static_cast<int>(t / (static_cast<long long>(c_grad) * c_src));
    const int chunks = (pair_num[k] + chunk - 1) / chunk;
    for (int c = cg; c < chunks; c += 16) s += partial[c * per_chunk + t];
  }
  part[cg][e] = s;
  __syncthreads();
  if (cg == 0 && t < per_chunk) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += part[q][e];
    const int cc = static_cast<int>(t % c_grad), cr = static_cast<int>((t / c_grad) % c_src);
    dw[k * dw_kstride + static_cast<long long>(cr) * dw_ld + cc] = tot;
  }
}

// Sums the per-chunk tiles in a fixed order.  A workgroup covers 16 consecutive elements x 16 chunk groups (64-byte
// coalesced reads, chunks/16 independent loads per thread), then folds the 16 partial sums through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ partial, int chunks, int k_base, int k_count, int c_src,
                                                    int c_grad, float* __restrict__ dw, long long dw_kstride, int dw_ld, int skip_k) {
  __shared__ float part[16][17];
  const int e = threadIdx.x & 15, cg = threadIdx.x >> 4;
  const long long per_chunk = static_cast<long long>(k_count) * c_src * c_grad;
  const long long t = static_cast<long long>(blockIdx.x) * 16 + e;
  float s = 0.f;
  if (t < per_chunk)
    for (int c = cg; c < chunks; c += 16) s += partial[c * per_chunk + t];
  part[cg][e] = s;
  __syncthreads();
  if (cg == 0 && t < per_chunk) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += part[q][e];
    const int cc = static_cast<int>(t % c_grad), cr = static_cast<int>((t / c_grad) % c_src);
    const int k = k_base + static_cast<int>(t / (static_cast<long long>(c_grad) * c_src));
    if (k != skip_k) dw[k * dw_kstride + static_cast<long long>(cr) * dw_ld + cc] = tot;
  }
}

// ---- dispatch -----------------------------------------------------------------------------------
// kernel variant: 0 = heuristic (LDS-DMA tile when the shapes allow, else the register-staged pipeline), 1 = plain
// dense tile, 2 = compacted tile, 3 = register-staged pipeline.  FV2P_CONV_IMPL=dense|cmp|pipe presets it;
// fv2p_sparse_conv_set_impl() lets the parity tests run every variant in one process.
static int g_conv_impl = -1;
static int g_ksplit_auto = 1;   // FV2P_CONV_KSPLIT=0 keeps the round-1 kernels in auto mode (comparison runs)
static int g_ksplit_tm = 0;     // FV2P_KSPLIT_TM: rows per tile (32 or 64); 0 = by launch shape
static int g_ksplit_gps = 0;    // FV2P_KSPLIT_GPS=1: one row group per step at every size (comparison runs)
static size_t g_ksplit_pad = 0; // FV2P_KSPLIT_PAD: extra dynamic LDS bytes per workgroup (limits the workgroups resident on a CU)
static int conv_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    else cus = 256;
  }
  return cus;
}
// thin-layer kernels of the auto mode: conv_rows_thin (16 -> 16) and conv_rows_res (32 -> 32 / 32 -> 16, weights resident in LDS).
// -1 = not read yet (FV2P_CONV_THIN / FV2P_CONV_RES preset them, default on); fv2p_sparse_conv_set_paths() switches them at run time
// so that the parity tests hold each kernel against the staged kernel it replaces in one process.
static int g_thin_on = -1, g_res_on = -1, g_first_on = -1;
constexpr int kThinDepth = 2;      // offsets in flight per wave of conv_rows_thin (measured: see launch_vec)
constexpr int kRes16Default = 1;   // conv_rows_res at 16 source channels: the plain forward 16 -> 16 (see launch_vec)
static bool path_on(int& flag, const char* e) {   // e: the development preset (FV2P_DEV_ENV), null in the release library
  if (flag < 0) flag = (!e || atoi(e) != 0) ? 1 : 0;
  return flag != 0;
}
static int g_wgrad_dma = 1;   // pair-split weight gradient: 1 = LDS-DMA kernel where the shapes allow, 0 = register-staged kernel
static unsigned long long* g_conv_trace = nullptr;
static int conv_impl() {
  if (g_conv_impl < 0) {
    const char* e = FV2P_DEV_ENV("FV2P_CONV_IMPL");
    g_conv_impl = (e && e[0] == 'd') ? 1 : (e && e[0] == 'c') ? 2 : (e && e[0] == 'p') ? 3 : (e && e[0] == 'k') ? 4 : 0;
    const char* ks = FV2P_DEV_ENV("FV2P_CONV_KSPLIT");
    g_ksplit_auto = ks ? atoi(ks) : 1;
    const char* tm = FV2P_DEV_ENV("FV2P_KSPLIT_TM");
    g_ksplit_tm = tm ? atoi(tm) : 0;
    const char* gps = FV2P_DEV_ENV("FV2P_KSPLIT_GPS");
    g_ksplit_gps = gps ? atoi(gps) : 0;
    const char* pad = FV2P_DEV_ENV("FV2P_KSPLIT_PAD");
    g_ksplit_pad = pad ? static_cast<size_t>(atoi(pad)) : 0;
  }
  return g_conv_impl;
}

// ---- cost-balanced tiling plan (fv2p_conv_plan_build) ---------------------------------------------------------------------------
// At the 5-20 pairs per row of the backbones' deep levels equal-ROW tiles carry unequal work: the 620 workgroups of the res4
// layer lasted 90 k clocks at the median and 180 k at the maximum, all resident at once, so the launch ended with the heaviest
// tile while the MFMA pipes were 0.37 busy chip-wide (profiles/r02_pmc_mfma.json).  A plan cuts the rows into T contiguous
// ranges of equal COST instead, cost(row) = max(pairs(row), kPlanFloor): the floor prices the per-offset overhead of a sparse
// tile (a tile visits every non-empty offset whatever its fill) and bounds the rows of a range.  The plan lives in the ints that
// follow the table ([kvol][n_dst]) in the caller's allocation: [magic, levels, then per level l the T_l + 1 bounds], levels
// T_l = 256, 384, 512, 768, ... up to about n_dst / 16, so that one plan serves every launch shape (column blocks, direction).
constexpr int kPlanMagic = 0x706c616e;   // "plan"
constexpr int kPlanFloor = 8;
static inline int plan_tiles(int level) { return ((level & 1) ? 384 : 256) << (level >> 1); }
static inline int plan_levels(int64_t n_dst) {
  int l = 1;
  while (l < 24 && plan_tiles(l) <= n_dst / 16) ++l;
  return l;
}
static inline int64_t plan_level_offset(int level) {   // ints from the plan header to the bounds of `level`
  int64_t off = 2;
  for (int l = 0; l < level; ++l) off += plan_tiles(l) + 1;
  return off;
}
static int g_plan_rows = 0;   // FV2P_PLAN_ROWS: rows per tile the level choice aims at (0 = default)
static int g_plan_on = -1;    // FV2P_CONV_PLAN=0 ignores plans (comparison runs)
// Level (tile count) of a launch: the chip holds 512 of these workgroups at once (2 per CU: LDS), so a launch runs in
// ceil(tiles x column blocks / 512) rounds of about n_dst / tiles rows each — the level with the smallest product wins, the fewer
// (larger: less weight traffic per pair) tiles on a tie; a level is eligible while its average tile leaves the balancer room below
// the 64-row limit (<= 58 rows).  Round 3 first took the level nearest to 44 rows per tile: 29 446 rows x 64 channels got 768 tiles =
// one and a half rounds, 63 us where 512 tiles take one round.  FV2P_PLAN_ROWS forces the old rule with that row target.
static int plan_pick_level(int64_t n_dst, int col_blocks) {
  if (g_plan_on < 0) {
    const char* e = FV2P_DEV_ENV("FV2P_CONV_PLAN"); g_plan_on = e ? atoi(e) : 1;
    const char* r = FV2P_DEV_ENV("FV2P_PLAN_ROWS"); g_plan_rows = r ? atoi(r) : 0;
  }
  if (!g_plan_on || n_dst > 65536) return -1;   // beyond ~1000 tiles the dispatcher balances by itself; full 64-row tiles reload the fewest weights
  const int levels = plan_levels(n_dst);
  if (g_plan_rows > 0) {
    double want = static_cast<double>(n_dst) / g_plan_rows;
    if (want * col_blocks < 512.0) want = 512.0 / col_blocks;
    int best = 0; double best_r = 1e30;
    for (int l = 0; l < levels; ++l) {
      const double t = plan_tiles(l), ratio = t > want ? t / want : want / t;
      if (ratio < best_r) { best_r = ratio; best = l; }
    }
    return best;
  }
  int best = levels - 1; double best_cost = 1e30;
  for (int l = 0; l < levels; ++l) {
    const double t = plan_tiles(l), rows = static_cast<double>(n_dst) / t;
    if (rows > 58.0) continue;
    const double cost = static_cast<double>(ceil_div(static_cast<int64_t>(t) * col_blocks, 512)) * rows;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = l; }   // ascending tile counts: a tie keeps the fewer tiles
  }
  return best;
}

static constexpr size_t ksplit_lds(int tm) { return 4 * static_cast<size_t>(tm + 1) * 68 * sizeof(float) + 32 * tm * sizeof(int) + 32 * tm + 32 * sizeof(int) + 4 * 112; }
// impl 4 forces the pair-compacted K-split tile wherever its shapes allow.  In auto mode it takes every launch with 64 or 128
// source channels and whole 64-column blocks (measured against conv_rows_dma, profiles/r02_microbench.txt: 128 -> 128 at
// 9 919 rows 127 -> 77 us, 64 -> 64 at 22 331 rows 66 -> 54 us, at 141 294 rows 264 -> 232 us), except permuted launches with
// 64 source channels: there the offset-skipping tile on parity-ordered rows stays ahead (17.9 vs 61 us at 32 -> 64).
template <int CINP>
static bool ksplit_wanted(const ConvArgs& a) {
  const int impl = conv_impl();
  if (impl == 4) return true;
  return impl == 0 && g_ksplit_auto && a.kvol > 1 && !(a.perm && CINP == 64);
}
// rows per tile: 32 fill a 10 k-row level's launch with twice the workgroups (forward), 64 halve the weight traffic per pair
template <bool WT>
static int ksplit_rows(const ConvArgs& a) {
  if (g_ksplit_tm == 32 || g_ksplit_tm == 64) return g_ksplit_tm;
  return (!WT && a.n_dst < 65536) ? 32 : 64;
}

// launch unless the call only asks which kernel would run (a.dry: fv2p_sparse_conv_prenorm_supported)
#define FV2P_LAUNCH(kern, grid, block, lds, stream, ...) do { if (!a.dry) hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__); } while (0)
template <int CINP, int NB, bool WT>
static int launch_vec(const ConvArgs& a, hipStream_t s) {   // 0: launched (or chosen, a.dry); 1: the chosen kernel cannot normalise its source rows (a.pre_mean)
  constexpr int TM = 128;
  constexpr size_t cmp_lds = (static_cast<size_t>(CINP) * NB * 16 + static_cast<size_t>(TM) * (NB * 16 + 16)) * sizeof(float) + 2 * TM * sizeof(int);
  const int impl = conv_impl();
  const bool use_cmp = impl == 2 && cmp_lds <= 64 * 1024;  // measured slower than the dense tile at KITTI sizes (profiles/r01_*): opt-in
  if (use_cmp) {
    if (a.pre_mean) return 1;
    const unsigned blocks = static_cast<unsigned>(ceil_div(a.n_dst, TM));
    FV2P_LAUNCH((conv_rows_cmp<CINP, NB, WT, TM>), dim3(blocks), dim3(256), cmp_lds, s, a);
    return 0;
  }
  const unsigned blocks = static_cast<unsigned>(ceil_div(a.n_dst, 64));
  if constexpr ((CINP == 64 || CINP == 128) && NB % 4 == 0) {
    // pair-compacted K-split tile: whole 64-column blocks, whole fragments
    const bool whole = a.c_src == CINP && a.c_dst == NB * 16 && (a.w_ld & 3) == 0 && (a.w_kstride & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 && (a.ld_dst & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.dst) & 15) == 0 && (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0);
    if (whole && a.kvol <= 27 && ksplit_wanted<CINP>(a)) {   // 27: a wave's step list holds 27 * MAXG + 1 entries (112 bytes, conv_rows_ksplit)
      ConvArgs b = a; b.trace = g_conv_trace;
      const int level = a.plan ? plan_pick_level(a.n_dst, NB / 4) : -1;   // a.plan: the plan header behind the table
      b.plan = level >= 0 ? a.plan + plan_level_offset(level) : nullptr;
      const unsigned tiles = level >= 0 ? static_cast<unsigned>(plan_tiles(level)) : 0u;
      // row groups per step (FV2P_KSPLIT_GPS=2: two, the second multiplied only where it holds pairs)
      const int gps = g_ksplit_gps == 2 ? 2 : 1;   // measured: two groups per step gain nothing at either size (KITTI 55 vs 53 us, Waymo 254 vs 256): opt-in
      const size_t lds64 = ksplit_lds(64) + g_ksplit_pad;
      if (level >= 0 || ksplit_rows<WT>(a) == 64) {
        b.col_blocks = NB / 4;
        const dim3 grid((level >= 0 ? tiles : blocks) * (NB / 4));
#ifdef FV2P_DEV   // timing-only ablation instances: development builds only (results INVALID by construction)
        static const int abl = [] { const char* e = FV2P_DEV_ENV("FV2P_KSPLIT_ABL"); return e ? atoi(e) : 0; }();
        if constexpr (CINP == 128 && !WT) {
          if (abl) {   // timing-only ablations of the roofline layer's forward kernel (see the template's comment)
#define FV2P_ABL(V) case V: { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_ksplit<CINP, WT, 64, 1, V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                              FV2P_LAUNCH((conv_rows_ksplit<CINP, WT, 64, 1, V>), grid, dim3(256), lds64, s, b); return 0; }
            switch (abl) { FV2P_ABL(1) FV2P_ABL(2) FV2P_ABL(3) FV2P_ABL(4) FV2P_ABL(5) FV2P_ABL(6) FV2P_ABL(7) FV2P_ABL(8) FV2P_ABL(9) FV2P_ABL(11) FV2P_ABL(15) default: break; }
#undef FV2P_ABL
          }
        }
#endif
        if (a.pre_mean) {   // source rows normalised on the gather: forward convs only
          if constexpr (!WT) {
            static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_ksplit<CINP, WT, 64, 1, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
            if (once) { FV2P_LAUNCH((conv_rows_ksplit<CINP, WT, 64, 1, 0, true>), grid, dim3(256), lds64, s, b); return 0; }
          }
          return 1;
        }
        if (gps == 2) {
          static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_ksplit<CINP, WT, 64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
          if (once) { FV2P_LAUNCH((conv_rows_ksplit<CINP, WT, 64, 2>), grid, dim3(256), lds64, s, b); return 0; }
        } else {
          static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_ksplit<CINP, WT, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
          if (once) { FV2P_LAUNCH((conv_rows_ksplit<CINP, WT, 64>), grid, dim3(256), lds64, s, b); return 0; }
        }
      }
      b.plan = nullptr;
      {
        const size_t lds = ksplit_lds(32) + g_ksplit_pad;
        static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_ksplit<CINP, WT, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
        b.col_blocks = NB / 4;
        if (a.pre_mean) {
          if constexpr (!WT) {
            static bool once_pre = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_ksplit<CINP, WT, 32, 1, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
            if (once_pre) { FV2P_LAUNCH((conv_rows_ksplit<CINP, WT, 32, 1, 0, true>), dim3(static_cast<unsigned>(ceil_div(a.n_dst, 32)) * (NB / 4)), dim3(256), lds, s, b); return 0; }
          }
          return 1;
        }
        if (once) { FV2P_LAUNCH((conv_rows_ksplit<CINP, WT, 32>), dim3(static_cast<unsigned>(ceil_div(a.n_dst, 32)) * (NB / 4)), dim3(256), lds, s, b); return 0; }
      }
    }
  }
  if constexpr ((CINP == 32 || CINP == 16) && NB <= 2) {
    // 32 source channels, full 3 x 3 x 3 kernel: every W_k resident in LDS, one persistent workgroup of 16 waves per CU
    // 16 source channels, FORWARD only: 11.7 against 14.0 us of conv_rows_thin at 35 k rows (whose non-transposed weight fragment is four
    // dword loads per offset and wave); backward data (one 16-byte load) is equal on both, 11.7 / 11.8, and stays on conv_rows_thin.
    // Development switch FV2P_CONV_RES16: bit 0 the plain 16 -> 16 forward, bit 1 its PRE form, bit 2 16 -> 32 as well.  Measured, residual
    // backbone (batch 4) / plain backbone, ms per step: 0: 3.738 / 1.690, 1: 3.758 / 1.645, 3: 3.703 / -, 7: 3.835 / 1.669 (all within the
    // +- 0.03 of repeated runs).  Inside the FV2P step (rocprofv3, launches that also finalise their BatchNorm sums): plain forward 22.7 us
    // on either kernel, the PRE form 26.1 here against 24.2 on conv_rows_thin -> 1: the plain forward only.
    static const int res16 = [] { const char* e = FV2P_DEV_ENV("FV2P_CONV_RES16"); return e ? atoi(e) : kRes16Default; }();
    const bool pre = a.pre_mean != nullptr;
    const bool res16_here = !WT && (res16 & 1) && (NB == 1 || (res16 & 4)) && (!pre || (res16 & 2));
    const bool res_on = path_on(g_res_on, FV2P_DEV_ENV("FV2P_CONV_RES")) && (CINP == 32 || res16_here);
    const bool whole = a.c_src == CINP && a.c_dst == NB * 16 && (a.ld_src & 3) == 0 && (reinterpret_cast<uintptr_t>(a.src) & 15) == 0;
    constexpr size_t res_lds = static_cast<size_t>(27) * CINP * NB * 16 * sizeof(float);
    // measured (tools/microbench.py conv): subm 32 -> 32 at 39 k rows 29.4 us forward / 29.4 us backward data against 34.7 / 32.3 us of the
    // staged kernels; at 286 k rows (Waymo) 170 / 170 against 168 / 156 us - with several groups per wave the staged kernels' shared fragment wins back
    if (impl == 0 && res_on && whole && a.kvol == 27 && !a.perm && a.n_dst <= 65536) {
      const int groups = static_cast<int>(ceil_div(a.n_dst, 16));
      const int wgs = std::min(groups, conv_cu_count());
      const int per = static_cast<int>(ceil_div(groups, wgs));
      if (pre) {
        // (no PRE form at 32 channels: normalising the gathered rows pushed this 128-register, 16-wave kernel into scratch - 66.8 against
        //  47.1 us at 39 k rows, more than the 5.6 us apply launch it saves; 32-channel consumers read materialised rows)
        if constexpr (CINP == 16 && !WT) {
          static bool once_pre = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_res<CINP, NB, WT, 27, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
          if (once_pre) { FV2P_LAUNCH((conv_rows_res<CINP, NB, WT, 27, true>), dim3(static_cast<unsigned>(ceil_div(groups, per))), dim3(1024), std::max(res_lds, kStatsDoneLds), s, a, per); return 0; }
        }
        return 1;
      }
      static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_res<CINP, NB, WT, 27>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
      if (once) {
        FV2P_LAUNCH((conv_rows_res<CINP, NB, WT, 27>), dim3(static_cast<unsigned>(ceil_div(groups, per))), dim3(1024), std::max(res_lds, kStatsDoneLds), s, a, per);
        return 0;
      }
    }
  }
  if constexpr (CINP == 16 && NB == 1) {
    // 16 -> 16 with the full 3 x 3 x 3 kernel: one wave per 16 rows, operands straight into registers two offsets ahead.  Measured
    // (FV2P_RES=1 tools/microbench.py conv, 35 k rows): 13.4 us forward / 11.5 us backward data against 14.9 / 13.5 us of the staged
    // kernels.  NOT at 32 channels: 49 against 35 us (subm 32 -> 32, 39 k rows), 22.4 against 20.1 us (16 -> 32) - a 4 KB weight
    // fragment per wave and offset through the vector-memory path costs more than the barrier it removes (LDS staging shares it
    // between the four waves of a workgroup).
    const bool thin_on = path_on(g_thin_on, FV2P_DEV_ENV("FV2P_CONV_THIN"));
    const bool whole = a.c_src == CINP && a.c_dst == NB * 16 && (a.ld_src & 3) == 0 && (reinterpret_cast<uintptr_t>(a.src) & 15) == 0 &&
                       (!WT || ((a.w_ld & 3) == 0 && (a.w_kstride & 3) == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0));
    if (impl == 0 && thin_on && whole && a.kvol == 27 && !a.perm) {
      const unsigned groups = static_cast<unsigned>(ceil_div(a.n_dst, 16));
      static const int depth = [] { const char* e = FV2P_DEV_ENV("FV2P_THIN_DEPTH"); return e ? atoi(e) : kThinDepth; }();   // development: 2 / 4 / 6
      if (a.pre_mean) {
        if constexpr (!WT) {
          if (depth == 4) FV2P_LAUNCH((conv_rows_thin<CINP, NB, WT, 27, true, 4>), dim3(ceil_div(groups, 4u)), dim3(256), kStatsDoneLds, s, a);
          else if (depth == 6) FV2P_LAUNCH((conv_rows_thin<CINP, NB, WT, 27, true, 6>), dim3(ceil_div(groups, 4u)), dim3(256), kStatsDoneLds, s, a);
          else FV2P_LAUNCH((conv_rows_thin<CINP, NB, WT, 27, true>), dim3(ceil_div(groups, 4u)), dim3(256), kStatsDoneLds, s, a);
          return 0;
        }
        return 1;
      }
      if (depth == 4) FV2P_LAUNCH((conv_rows_thin<CINP, NB, WT, 27, false, 4>), dim3(ceil_div(groups, 4u)), dim3(256), kStatsDoneLds, s, a);
      else if (depth == 6) FV2P_LAUNCH((conv_rows_thin<CINP, NB, WT, 27, false, 6>), dim3(ceil_div(groups, 4u)), dim3(256), kStatsDoneLds, s, a);
      else FV2P_LAUNCH((conv_rows_thin<CINP, NB, WT, 27>), dim3(ceil_div(groups, 4u)), dim3(256), kStatsDoneLds, s, a);
      return 0;
    }
  }
  if (a.pre_mean) return 1;   // the staged kernels below read their source rows as they are
  if constexpr (CINP * NB <= 512 && (WT || NB % 4 == 0)) {
    // LDS-DMA kernel: whole fragments only (every lane's 16-byte source must exist and be aligned)
    const bool whole = a.c_src == CINP && a.c_dst == NB * 16 && (a.w_ld & 3) == 0 && (a.w_kstride & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 && (a.ld_dst & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.dst) & 15) == 0 && (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0);
    constexpr size_t act_lds = 2 * CINP * NB * 16 * sizeof(float) + 32 * 64 * sizeof(int) + 16;
    if (a.perm && impl == 0 && whole && a.kvol > 1 && a.kvol <= 32 && act_lds <= 64 * 1024) {
      FV2P_LAUNCH((conv_rows_act<CINP, NB, WT>), dim3(blocks), dim3(256), act_lds, s, a);
      return 0;
    }
    if (impl == 0 && whole && a.kvol > 1) {
      ConvArgs b = a; b.trace = g_conv_trace;
      FV2P_LAUNCH((conv_rows_dma<CINP, NB, WT>), dim3(blocks), dim3(256), std::max<size_t>(2 * CINP * NB * 16 * sizeof(float), kStatsDoneLds), s, b);
      return 0;
    }
  }
  if constexpr (CINP * NB == 1024 && NB == 8) {
    // 128 -> 128: the LDS-DMA kernel on two column halves (grid.y), 2 x 32 KB of weights per workgroup.  The gathered rows are
    // read by both halves (L2 hits); twice the workgroups also fill the chip better at the 10 k rows these layers have.
    const bool whole = a.c_src == CINP && a.c_dst == NB * 16 && (a.w_ld & 3) == 0 && (a.w_kstride & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 && (a.ld_dst & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.dst) & 15) == 0 && (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0);
    if (impl == 0 && whole && a.kvol > 1) {
      ConvArgs b = a; b.trace = nullptr;
      FV2P_LAUNCH((conv_rows_dma<CINP, 4, WT>), dim3(blocks, 2), dim3(256), 2 * CINP * 4 * 16 * sizeof(float), s, b);
      return 0;
    }
  }
  if constexpr (CINP * NB <= 256) {
    if (impl != 1 && a.kvol > 1) {  // FV2P_CONV_IMPL=dense keeps the unpipelined tile, =pipe the register-staged pipeline (parity tests run all)
      // KU = 2 (two offsets per barrier) measured slower than KU = 1 on MI355X (93.9 vs 89.6 us, 64->64 N=29k)
      FV2P_LAUNCH((conv_rows_pipe<CINP, NB, WT, 1>), dim3(blocks), dim3(256), std::max<size_t>(2 * CINP * NB * 16 * sizeof(float), kStatsDoneLds), s, a);
      return 0;
    }
  }
  FV2P_LAUNCH((conv_rows_vec<CINP, NB, WT>), dim3(blocks), dim3(256), std::max<size_t>(CINP * NB * 16 * sizeof(float), kStatsDoneLds), s, a);
  return 0;
}
template <int STEPS, int NB, bool WT>
static int launch_scalar(const ConvArgs& a, hipStream_t s) {
  if (a.pre_mean) return 1;
  const unsigned blocks = static_cast<unsigned>(ceil_div(a.n_dst, 64));
  if constexpr (!WT && STEPS * NB <= 2) {
    // the backbones' first layer (4 or 5 point features -> 16 / 32 channels, 27 offsets): everything in registers, no barrier.  Measured
    // (FV2P_RES=1 tools/microbench.py conv, subm 4 -> 16 at 35 146 rows): see profiles/README.md, round 6
    if (conv_impl() == 0 && path_on(g_first_on, FV2P_DEV_ENV("FV2P_CONV_FIRST")) && a.kvol == 27 && !a.perm) {
      FV2P_LAUNCH((conv_rows_first<STEPS, NB, 27>), dim3(blocks), dim3(256), kStatsDoneLds, s, a);
      return 0;
    }
  }
  FV2P_LAUNCH((conv_rows_scalar<STEPS, NB, WT>), dim3(blocks), dim3(256), std::max<size_t>(STEPS * 4 * NB * 16 * sizeof(float), kStatsDoneLds), s, a);
  return 0;
}

template <bool WT>
static int dispatch_conv(const ConvArgs& a, hipStream_t s) {
  const int nb = static_cast<int>(ceil_div(a.c_dst, 16));
  const int nbp = nb <= 1 ? 1 : nb <= 2 ? 2 : nb <= 4 ? 4 : 8;
  const bool vec = (a.c_src % 4 == 0) && (a.ld_src % 4 == 0) && a.c_src >= 16 &&
                   (reinterpret_cast<uintptr_t>(a.src) % 16 == 0);
  int unsupported = 0;   // 1: a.pre_mean with a kernel that cannot normalise its source rows (nothing was launched)
#define FV2P_VEC_CASE(CINP)                                            \
  switch (nbp) {                                                       \
    case 1: unsupported = launch_vec<CINP, 1, WT>(a, s); break;        \
    case 2: unsupported = launch_vec<CINP, 2, WT>(a, s); break;        \
    case 4: unsupported = launch_vec<CINP, 4, WT>(a, s); break;        \
    default: unsupported = launch_vec<CINP, 8, WT>(a, s); break;       \
  }
#define FV2P_SCALAR_CASE(STEPS)                                        \
  switch (nbp) {                                                       \
    case 1: unsupported = launch_scalar<STEPS, 1, WT>(a, s); break;    \
    case 2: unsupported = launch_scalar<STEPS, 2, WT>(a, s); break;    \
    case 4: unsupported = launch_scalar<STEPS, 4, WT>(a, s); break;    \
    default: unsupported = launch_scalar<STEPS, 8, WT>(a, s); break;   \
  }
  if (vec) {
    if (a.c_src <= 16) { FV2P_VEC_CASE(16) }
    else if (a.c_src <= 32) { FV2P_VEC_CASE(32) }
    else if (a.c_src <= 64) { FV2P_VEC_CASE(64) }
    else { FV2P_VEC_CASE(128) }
  } else {
    if (a.c_src <= 4) { FV2P_SCALAR_CASE(1) }
    else if (a.c_src <= 8) { FV2P_SCALAR_CASE(2) }
    else if (a.c_src <= 16) { FV2P_SCALAR_CASE(4) }
    else if (a.c_src <= 32) { FV2P_SCALAR_CASE(8) }
    else return set_error(FV2P_ELIMIT, "sparse conv: %d source channels (not a multiple of 4) unsupported above 32", a.c_src);
  }
#undef FV2P_VEC_CASE
#undef FV2P_SCALAR_CASE
  return unsupported;   // (> 0: not an error code; conv_rows_impl turns it into one or into the answer of the support query)
}

// ---- exact plan for tables of <= kPlanExactRows rows ------------------------------------------------------------------------------
// A tile's time is its number of MFMA row groups, sum_k ceil(pairs_k / 16) (measured: correlation 0.92, 3.5 k clocks per group at 128
// channels), and equal PAIRS do not mean equal groups: a planar neighbourhood has 9 busy offsets, a full 3-D one 27, so equal-cost
// tiles of the res4 level span 23 .. 45 groups.  Here the bounds come from the groups themselves: for a budget G, reach_G(s) = the
// longest run of rows from s (at most 64) whose groups stay <= G; greedy tiles s -> s + reach_G(s) are the fewest tiles of cost <= G
// (the cost of a run only grows with its length).  All starts in parallel for 16 budgets at once, one workgroup per budget walks
// its chain in LDS (pointer doubling, then 8 steps per anchor), and every level takes the smallest budget whose tile count fits:
// 28 .. 36 groups per tile instead of 23 .. 45.
constexpr int kPlanExactRows = 32767;
constexpr int kPlanBudgets = 16;
__constant__ int c_plan_budget[kPlanBudgets] = {28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 52, 56, 60, 64, 72};
constexpr unsigned long long kPlanOnes = 0x0102040810204081ull;   // 9 fields of 7 bits

// three 64-bit words per row: offset k counts in field k % 9 of word k / 9
__global__ __launch_bounds__(256) void plan_rowwords_k(const int* __restrict__ tab, int kvol, int n_dst, unsigned long long* __restrict__ words) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= n_dst) return;
  unsigned long long w[3] = {0ull, 0ull, 0ull};
  for (int k = 0; k < kvol; ++k)
    if (tab[static_cast<long long>(k) * n_dst + row] >= 0) w[k / 9] |= 1ull << (7 * (k % 9));
  words[row] = w[0]; words[n_dst + row] = w[1]; words[2 * static_cast<long long>(n_dst) + row] = w[2];
}
__device__ __forceinline__ int plan_groups(unsigned long long c) {   // sum over the 9 fields of ceil(count / 16), counts <= 64
  const unsigned long long g = ((c + 15ull * kPlanOnes) >> 4) & (7ull * kPlanOnes);
  return static_cast<int>((g * kPlanOnes) >> 56) & 0x7f;
}
__global__ __launch_bounds__(256) void plan_reach_k(const unsigned long long* __restrict__ words, int n_dst, unsigned char* __restrict__ reach) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n_dst) return;
  unsigned long long c0 = 0ull, c1 = 0ull, c2 = 0ull;
  const int end = min(n_dst, s + 64);
  int j = 0, r = s;
  for (; r < end && j < kPlanBudgets; ++r) {
    c0 += words[r]; c1 += words[n_dst + r]; c2 += words[2 * static_cast<long long>(n_dst) + r];
    const int g = plan_groups(c0) + plan_groups(c1) + plan_groups(c2);
    while (j < kPlanBudgets && g > c_plan_budget[j]) {   // rows s .. r - 1 fit budget j, row r does not
      reach[static_cast<long long>(j) * n_dst + s] = static_cast<unsigned char>(max(r - s, 1));
      ++j;
    }
  }
  for (; j < kPlanBudgets; ++j) reach[static_cast<long long>(j) * n_dst + s] = static_cast<unsigned char>(end - s);
}
// one workgroup per budget: tile starts 0, 0 + reach(0), ... -> walk[j][t], count[j]
__global__ __launch_bounds__(1024) void plan_walk_k(const unsigned char* __restrict__ reach, int n_dst, int* __restrict__ walk, int* __restrict__ count) {
  extern __shared__ unsigned short plan_lds[];
  unsigned short* A = plan_lds;                  // [n + 1]
  unsigned short* B = plan_lds + (n_dst + 1);    // [n + 1]
  const int j = blockIdx.x, tid = threadIdx.x;
  const unsigned char* rj = reach + static_cast<long long>(j) * n_dst;
  int* wj = walk + static_cast<long long>(j) * n_dst;
  for (int s = tid; s <= n_dst; s += 1024) A[s] = static_cast<unsigned short>(s < n_dst ? min(s + rj[s], n_dst) : n_dst);
  __syncthreads();
  for (int s = tid; s <= n_dst; s += 1024) B[s] = A[A[s]];   // 2 steps
  __syncthreads();
  for (int s = tid; s <= n_dst; s += 1024) A[s] = B[B[s]];   // 4 steps
  __syncthreads();
  for (int s = tid; s <= n_dst; s += 1024) B[s] = A[A[s]];   // 8 steps
  __syncthreads();
  __shared__ int n_anchor;
  if (tid == 0) {   // anchors: every 8th tile start, kept in A (no longer needed)
    int p = 0, a = 0;
    while (p < n_dst) { A[a++] = static_cast<unsigned short>(p); p = B[p]; }
    n_anchor = a;
  }
  __syncthreads();
  const int na = n_anchor;
  for (int a = tid; a < na; a += 1024) {
    int p = A[a], i = 0;
    for (; i < 8 && p < n_dst; ++i) { wj[8 * a + i] = p; p = min(p + rj[p], n_dst); }
    if (a == na - 1) count[j] = 8 * a + i;
  }
}
// per level: the smallest budget whose tiles fit the level's tile count replaces the equal-cost bounds
__global__ __launch_bounds__(256) void plan_select_k(const int* __restrict__ walk, const int* __restrict__ count, int n_dst, int levels, int* __restrict__ plan) {
  const long long e = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  long long base = 0;
  for (int l = 0; l < levels; ++l) {
    const int tiles = ((l & 1) ? 384 : 256) << (l >> 1);
    if (e < base + tiles + 1) {
      const int t = static_cast<int>(e - base);
      if (e == 0) { plan[0] = kPlanMagic; plan[1] = levels; }
      int pick = -1;
      for (int j = 0; j < kPlanBudgets; ++j)
        if (count[j] <= tiles) { pick = j; break; }
      if (pick >= 0) plan[2 + e] = (t < count[pick]) ? walk[static_cast<long long>(pick) * n_dst + t] : n_dst;
      else plan[2 + e] = static_cast<int>(min(static_cast<long long>(n_dst), static_cast<long long>(t) * ((n_dst + tiles - 1) / tiles)));   // no budget fits: equal rows
      return;
    }
    base += tiles + 1;
  }
}

// cost(row) = max(pairs of the row, kPlanFloor): one thread per destination row, coalesced over the rows of every offset
__global__ __launch_bounds__(256) void plan_cost_k(const int* __restrict__ tab, int kvol, int n_dst, int* __restrict__ cost) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= n_dst) return;
  int c = 0;
  for (int k = 0; k < kvol; ++k) c += tab[static_cast<long long>(k) * n_dst + row] >= 0;
  cost[row] = max(c, kPlanFloor);
}
// bounds of every level: entry t of a level with T tiles = first row whose exclusive cost prefix reaches total * t / T
__global__ __launch_bounds__(256) void plan_bounds_k(const int* __restrict__ prefix, const int* __restrict__ total_p, int n_dst, int levels,
                                                     int* __restrict__ plan) {
  const long long e = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (e == 0) { plan[0] = kPlanMagic; plan[1] = levels; }
  long long base = 0;
  for (int l = 0; l < levels; ++l) {
    const int tiles = ((l & 1) ? 384 : 256) << (l >> 1);
    if (e < base + tiles + 1) {
      const int t = static_cast<int>(e - base);
      int lo = 0, hi = n_dst;   // first row in [0, n_dst] with prefix[row] >= target (prefix[n_dst] = total)
      if (t >= tiles) lo = n_dst;
      else if (t > 0) {
        const long long target = (static_cast<long long>(*total_p) * t + tiles - 1) / tiles;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (prefix[mid] >= target) hi = mid; else lo = mid + 1;
        }
      } else lo = 0;
      plan[2 + e] = lo;
      return;
    }
    base += tiles + 1;
  }
}

}  // namespace fv2p

using namespace fv2p;

extern "C" int fv2p_sparse_conv_set_trace(unsigned long long* trace) {
  fv2p::g_conv_trace = trace;
  return 0;
}

// ---- in-step probe: event pairs around the launches of one layer shape (bench.py roofline.in_step_us) ---------------------------
namespace {
struct ConvProbe {
  bool armed = false;
  int c_src = 0, c_dst = 0, kvol = 0, flip = 0;
  int64_t n_dst = 0;
  int used = 0;
  std::vector<hipEvent_t> ev;   // 2 per launch
  std::mutex mu;
};
ConvProbe g_probe;
constexpr int kProbePairs = 512;
}  // namespace

extern "C" int fv2p_sparse_conv_probe_arm(int c_src, int c_dst, int kvol, int64_t n_dst, int flip) {
  std::lock_guard<std::mutex> lock(g_probe.mu);
  if (g_probe.ev.empty()) {
    g_probe.ev.resize(2 * kProbePairs);
    for (auto& e : g_probe.ev) FV2P_HIP(hipEventCreate(&e));
  }
  g_probe.c_src = c_src; g_probe.c_dst = c_dst; g_probe.kvol = kvol; g_probe.n_dst = n_dst; g_probe.flip = flip & 1;
  g_probe.used = 0;
  g_probe.armed = true;
  return 0;
}

extern "C" int fv2p_sparse_conv_probe_read(double* sum_us, int* launches) {
  FV2P_REQUIRE(sum_us && launches, FV2P_EINVAL, "sparse_conv_probe_read: null pointer");
  std::lock_guard<std::mutex> lock(g_probe.mu);
  double total = 0.0;
  for (int i = 0; i < g_probe.used; ++i) {
    FV2P_HIP(hipEventSynchronize(g_probe.ev[2 * i + 1]));
    float ms = 0.f;
    FV2P_HIP(hipEventElapsedTime(&ms, g_probe.ev[2 * i], g_probe.ev[2 * i + 1]));
    total += static_cast<double>(ms) * 1e3;
  }
  *sum_us = total;
  *launches = g_probe.used;
  g_probe.used = 0;
  g_probe.armed = false;
  return 0;
}

extern "C" int fv2p_sparse_conv_set_impl(int impl) {
  FV2P_REQUIRE(impl >= 0 && impl <= 4, FV2P_EINVAL, "impl must be 0 (auto), 1 (dense), 2 (compacted), 3 (pipelined) or 4 (pair-compacted K-split)");
  fv2p::g_conv_impl = impl;
  fv2p::g_wgrad_dma = impl == 0;   // forcing any variant also selects the register-staged pair-split weight gradient
  return 0;
}

extern "C" int fv2p_sparse_conv_set_paths(int thin_on, int res_on) {
  FV2P_REQUIRE(thin_on >= -1 && thin_on <= 1 && res_on >= -1 && res_on <= 1, FV2P_EINVAL, "set_paths: 1 = on, 0 = off, -1 = back to the environment's preset");
  fv2p::g_thin_on = thin_on;
  fv2p::g_res_on = res_on;
  fv2p::g_first_on = thin_on;   // the first-layer kernel (conv_rows_first) is a thin-layer kernel too: one switch
  return 0;
}

// tiles any kernel of this file cuts n_dst rows into, at most: 16-row groups without a plan; with one, the level the launch picks - the
// levels go 256, 384, 512, ... and the largest a table has is the first ABOVE n_dst / 16 (a 2 000-row table still has its 256-tile level:
// the first form of this bound, n_dst / 16 + 64, let such a launch write its rows over the group slots - garbage statistics in the reduced
// detector of tests/test_ddp_gpu.py while every full-size layer passed)
static int64_t fin_tile_cap(int64_t n_dst) {
  const int64_t groups = ceil_div(n_dst > 0 ? n_dst : 1, 16);
  int64_t cap = groups + 64;
  if (n_dst > 0) cap = std::max<int64_t>(cap, plan_tiles(plan_levels(n_dst) - 1));
  return cap + 64;
}
extern "C" size_t fv2p_sparse_conv_fin_ws_bytes(int64_t n_dst, int c_dst) {
  return align_up(static_cast<size_t>(fin_tile_cap(n_dst) + kFinSubs) * 2 * static_cast<size_t>(c_dst > 0 ? c_dst : 1) * sizeof(double));
}
// what a call adds to the plain conv: statistics finalised by the last workgroup, source rows normalised on the gather, a support query
struct ConvExtra {
  unsigned* fin_counter = nullptr;
  BnFwdFin fwd = {nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};   // forward statistics (all columns; moved per column block here)
  BnBwdFin bwd = {nullptr, nullptr, nullptr, 1};                              // backward sums
  const float* pre_mean = nullptr; const float* pre_invstd = nullptr; const float* pre_gamma = nullptr; const float* pre_beta = nullptr; int pre_relu = 0;
  int dry = 0;
};
static int conv_rows_impl(const float* src, int64_t n_src, int c_src, const float* weight, int kvol, const int* tab, int64_t n_dst, int c_dst,
                          int flip_k, int transpose_w, const float* bias, float* dst, double* stats, hipStream_t stream,
                          const ConvArgs* bn = nullptr, const int* perm = nullptr, const ConvExtra* ex = nullptr) {
  FV2P_REQUIRE(c_src >= 1 && c_dst >= 1 && kvol >= 1 && n_dst >= 0 && n_src >= 0, FV2P_EINVAL, "sparse_conv_rows: bad sizes");
  if (n_dst == 0) return 0;
  FV2P_REQUIRE(weight && tab && dst && (src || n_src == 0), FV2P_EINVAL, "sparse_conv_rows: null pointer");
  FV2P_REQUIRE(n_dst < (1ll << 31) - 64, FV2P_ELIMIT, "sparse_conv_rows: too many rows");
  // weight is [K][Cin][Cout] in the reference layout (spconv/conv.py:98-99); with transpose_w the roles of the
  // two channel axes are swapped: W_k^T is used, i.e. weight is [K][c_dst][c_src].
  const int w_rows = transpose_w ? c_dst : c_src, w_cols = transpose_w ? c_src : c_dst;
  int probe_slot = -1;
  if (g_probe.armed && !(ex && ex->dry)) {   // (unlocked read of a flag: the probe is armed and read by the thread that measures)
    std::lock_guard<std::mutex> lock(g_probe.mu);
    if (g_probe.armed && g_probe.used < kProbePairs && c_src == g_probe.c_src && c_dst == g_probe.c_dst && kvol == g_probe.kvol &&
        n_dst == g_probe.n_dst && (flip_k & 1) == g_probe.flip && !transpose_w && !bn && !(ex && ex->pre_mean)) {   // the plain-gather launches: the kernel instance the isolated probe times
      probe_slot = g_probe.used++;
      FV2P_HIP(hipEventRecord(g_probe.ev[2 * probe_slot], stream));
    }
  }
  for (int d0 = 0; d0 < c_dst; d0 += 128) {
    const int cd = (c_dst - d0) < 128 ? (c_dst - d0) : 128;
    for (int s0 = 0; s0 < c_src; s0 += 128) {
      const int cs = (c_src - s0) < 128 ? (c_src - s0) : 128;
      ConvArgs a;
      a.trace = nullptr;
      a.col_blocks = 0;
      a.stats = stats ? stats + d0 : nullptr; a.stats_ld = c_dst;
      a.bn_x = nullptr; a.bn_mean = a.bn_invstd = a.bn_gamma = a.bn_beta = nullptr; a.bn_relu = 0;
      a.perm = perm;   // honoured by the LDS-DMA tile for permuted rows, ignored (plain row order, same result) by the others
      a.fin_counter = nullptr; a.fin_rows = a.fin_gslots = nullptr; a.fin_c = 0; a.fin_groups = 1; a.fin_col0 = 0; a.fin_bump = 0; a.dry = ex ? ex->dry : 0;
      a.fin_fwd = BnFwdFin{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};
      a.fin_bwd = BnBwdFin{nullptr, nullptr, nullptr, 1};
      a.pre_mean = a.pre_invstd = a.pre_gamma = a.pre_beta = nullptr; a.pre_relu = 0;
      if (ex && ex->pre_mean) {   // parameters of the SOURCE channels of this launch
        a.pre_mean = ex->pre_mean + s0; a.pre_invstd = ex->pre_invstd + s0;
        a.pre_gamma = ex->pre_gamma ? ex->pre_gamma + s0 : nullptr; a.pre_beta = ex->pre_beta ? ex->pre_beta + s0 : nullptr; a.pre_relu = ex->pre_relu;
      }
      if (ex && ex->fin_counter && stats && c_src <= 128) {   // one launch per column block holds the whole sum: its last workgroup finalises
        // stats: the workspace of fv2p_sparse_conv_fin_ws_bytes - [tile cap][2][c_dst] rows, then kFinSubs group slots
        a.fin_counter = ex->fin_counter; a.fin_rows = stats + d0; a.fin_c = cd;
        a.fin_gslots = stats + static_cast<long long>(fin_tile_cap(n_dst)) * 2 * c_dst + d0;
        a.fin_groups = n_dst > 32768 ? static_cast<int>(kFinSubs) : 16;   // ~15 - 35 rows per group fold either way
        if (bn) {
          a.fin_bwd = BnBwdFin{ex->bwd.dgamma + d0, ex->bwd.dbeta + d0, ex->bwd.coef + d0, ex->bwd.batch_stats};
        } else {
          a.fin_fwd = ex->fwd;
          a.fin_fwd.mean += d0; a.fin_fwd.invstd += d0;
          if (a.fin_fwd.running_mean) { a.fin_fwd.running_mean += d0; a.fin_fwd.running_var += d0; }
          a.fin_bump = d0 + 128 >= c_dst;
        }
      }
      if (bn && stats) {   // per-column pointers move with the column block, bn_x is addressed like dst (row * ld_dst + col)
        a.bn_x = bn->bn_x + d0; a.bn_mean = bn->bn_mean + d0; a.bn_invstd = bn->bn_invstd + d0;
        a.bn_gamma = bn->bn_gamma ? bn->bn_gamma + d0 : nullptr; a.bn_beta = bn->bn_beta ? bn->bn_beta + d0 : nullptr; a.bn_relu = bn->bn_relu;
      }
      a.src = src + s0; a.ld_src = c_src; a.c_src = cs;
      a.w = transpose_w ? weight + static_cast<long long>(d0) * w_cols + s0 : weight + static_cast<long long>(s0) * w_cols + d0;
      a.w_kstride = static_cast<long long>(w_rows) * w_cols; a.w_ld = w_cols;
      a.tab = tab; a.n_dst = static_cast<int>(n_dst); a.kvol = kvol; a.flip = flip_k & 1;
      // FV2P_TAB_PLANNED: the caller's table allocation continues with the plan fv2p_conv_plan_build wrote
      a.plan = (flip_k & 2) ? tab + static_cast<long long>(kvol) * n_dst : nullptr;
      a.bias = (bias && s0 == 0) ? bias + d0 : nullptr;
      a.dst = dst + d0; a.ld_dst = c_dst; a.c_dst = cd; a.accumulate = s0 > 0;
      int rc = transpose_w ? dispatch_conv<true>(a, stream) : dispatch_conv<false>(a, stream);
      if (rc > 0) return a.dry ? 1 : set_error(FV2P_EINVAL, "sparse conv: the kernel for %d -> %d channels, %d offsets cannot normalise its source rows "
                                                            "(ask fv2p_sparse_conv_prenorm_supported first)", c_src, c_dst, kvol);
      if (rc) return rc;
    }
  }
  if (probe_slot >= 0) FV2P_HIP(hipEventRecord(g_probe.ev[2 * probe_slot + 1], stream));
  if (ex && ex->dry) return 0;   // nothing was launched
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_sparse_conv_rows(const float* src, int64_t n_src, int c_src, const float* weight, int kvol, const int* tab,
                                     int64_t n_dst, int c_dst, int flip_k, int transpose_w, const float* bias, float* dst,
                                     fv2p_stream_t stream_) {
  return conv_rows_impl(src, n_src, c_src, weight, kvol, tab, n_dst, c_dst, flip_k, transpose_w, bias, dst, nullptr,
                        static_cast<hipStream_t>(stream_));
}

extern "C" int fv2p_sparse_conv_rows_perm(const float* src, int64_t n_src, int c_src, const float* weight, int kvol, const int* tab,
                                          int64_t n_dst, int c_dst, int flip_k, int transpose_w, const float* bias, float* dst, const int* perm,
                                          fv2p_stream_t stream_) {
  return conv_rows_impl(src, n_src, c_src, weight, kvol, tab, n_dst, c_dst, flip_k, transpose_w, bias, dst, nullptr,
                        static_cast<hipStream_t>(stream_), nullptr, perm);
}

extern "C" int fv2p_sparse_conv_stat_slots(void) { return kStatSlots; }
extern "C" int fv2p_sparse_conv_fin_counter_words(void) { return static_cast<int>((1 + kFinSubs) * kFinStride); }
extern "C" int64_t fv2p_conv_plan_ints(int64_t n_dst) {
  if (n_dst <= 0) return 0;
  return plan_level_offset(plan_levels(n_dst));
}

extern "C" size_t fv2p_conv_plan_ws_bytes(int64_t n_dst) {
  const int64_t n = n_dst > 0 ? n_dst : 1;
  size_t b = align_up(static_cast<size_t>(n) * sizeof(int)) + align_up(static_cast<size_t>(n) * sizeof(int)) + align_up(sizeof(int)) + align_up(scan_ws_bytes(n));
  if (n <= kPlanExactRows)   // row words, reach bytes, walks and counts of the exact plan
    b += align_up(3 * static_cast<size_t>(n) * sizeof(unsigned long long)) + align_up(static_cast<size_t>(kPlanBudgets) * n) +
         align_up(static_cast<size_t>(kPlanBudgets) * n * sizeof(int)) + align_up(kPlanBudgets * sizeof(int));
  return b;
}

extern "C" int fv2p_conv_plan_build(int* tab, int kvol, int64_t n_dst, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(kvol >= 1 && n_dst >= 0, FV2P_EINVAL, "conv_plan_build: bad sizes");
  if (n_dst == 0) return 0;
  FV2P_REQUIRE(tab && ws, FV2P_EINVAL, "conv_plan_build: null pointer");
  FV2P_REQUIRE(n_dst < (1ll << 31) - 64 && static_cast<long long>(kvol) * n_dst < (1ll << 31), FV2P_ELIMIT, "conv_plan_build: too many rows");
  FV2P_REQUIRE(ws_bytes >= fv2p_conv_plan_ws_bytes(n_dst), FV2P_EWORKSPACE, "conv_plan_build: workspace too small");
  Carver c(ws, ws_bytes);
  int* cost = c.take<int>(static_cast<size_t>(n_dst));
  int* prefix = c.take<int>(static_cast<size_t>(n_dst));
  int* total = c.take<int>(1);
  const size_t sb = scan_ws_bytes(n_dst);
  void* sws = c.take<char>(sb);
  const int n = static_cast<int>(n_dst);
  const int levels = plan_levels(n_dst);
  const int64_t entries = plan_level_offset(levels) - 2;
  int* plan = tab + static_cast<long long>(kvol) * n_dst;
  static int exact = -1;   // FV2P_PLAN_EXACT=0 keeps the equal-cost bounds (comparison runs)
  if (exact < 0) { const char* e = FV2P_DEV_ENV("FV2P_PLAN_EXACT"); exact = e ? atoi(e) : 1; }
  const size_t lds = 2 * static_cast<size_t>(n_dst + 1) * sizeof(unsigned short);
  static bool big = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&plan_walk_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) == hipSuccess; }();
  if (exact && kvol >= 8 && kvol <= 27 && n_dst >= 1024 && n_dst <= kPlanExactRows && (big || lds <= 48 * 1024)) {
    // group-balanced bounds: four launches (row words, reach of 16 budgets, one chain walk per budget, selection per level)
    unsigned long long* words = c.take<unsigned long long>(3 * static_cast<size_t>(n_dst));
    unsigned char* reach = c.take<unsigned char>(static_cast<size_t>(kPlanBudgets) * n_dst);
    int* walk = c.take<int>(static_cast<size_t>(kPlanBudgets) * n_dst);
    int* count = c.take<int>(kPlanBudgets);
    const unsigned blocks = static_cast<unsigned>(ceil_div(n_dst, 256));
    hipLaunchKernelGGL(plan_rowwords_k, dim3(blocks), dim3(256), 0, stream, tab, kvol, n, words);
    hipLaunchKernelGGL(plan_reach_k, dim3(blocks), dim3(256), 0, stream, words, n, reach);
    hipLaunchKernelGGL(plan_walk_k, dim3(kPlanBudgets), dim3(1024), lds, stream, reach, n, walk, count);
    hipLaunchKernelGGL(plan_select_k, dim3(static_cast<unsigned>(ceil_div(entries, 256))), dim3(256), 0, stream, walk, count, n, levels, plan);
  } else {
    // equal-cost bounds: cost per row, prefix sums, binary search per bound
    hipLaunchKernelGGL(plan_cost_k, dim3(static_cast<unsigned>(ceil_div(n_dst, 256))), dim3(256), 0, stream, tab, kvol, n, cost);
    if (int rc = exclusive_scan_i32(cost, prefix, n_dst, total, sws, sb, stream)) return rc;
    hipLaunchKernelGGL(plan_bounds_k, dim3(static_cast<unsigned>(ceil_div(entries, 256))), dim3(256), 0, stream, prefix, total, n, levels, plan);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}


extern "C" int fv2p_sparse_conv_rows_stats(const float* src, int64_t n_src, int c_src, const float* weight, int kvol, const int* tab,
                                           int64_t n_dst, int c_dst, int flip_k, int transpose_w, const float* bias, float* dst,
                                           double* stats, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(stats, FV2P_EINVAL, "sparse_conv_rows_stats: null stats");
  // epilogue statistics need the whole sum in one launch per column block and a kernel with the shared epilogue;
  // otherwise the columns are summed by the BatchNorm reduce pass right after the conv (same slots, same meaning)
  const bool fused = c_src <= 128 && conv_impl() != 2 && n_dst > 0;
  if (int rc = conv_rows_impl(src, n_src, c_src, weight, kvol, tab, n_dst, c_dst, flip_k, transpose_w, bias, dst, fused ? stats : nullptr, stream))
    return rc;
  if (!fused && n_dst > 0) return bn_column_sums(dst, n_dst, c_dst, stats, stream);
  return 0;
}

// conv whose statistics are finalised by its own last workgroup (mean / invstd / running statistics in memory when the launch ends, the
// slots zero again), and / or whose source rows pass through a BatchNorm (+ReLU) on the gather.  stats == NULL: no statistics.
extern "C" int fv2p_sparse_conv_rows_bnfin(const float* src, int64_t n_src, int c_src, const float* weight, int kvol, const int* tab,
                                           int64_t n_dst, int c_dst, int flip_k, int transpose_w, const float* bias, float* dst,
                                           double* stats, unsigned* counter, float eps, float momentum, float* running_mean, float* running_var,
                                           int64_t* num_batches_tracked, float* mean, float* invstd, const float* pre_mean, const float* pre_invstd,
                                           const float* pre_gamma, const float* pre_beta, int pre_relu, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(!stats || (counter && mean && invstd), FV2P_EINVAL, "sparse_conv_rows_bnfin: statistics need counter, mean and invstd");
  FV2P_REQUIRE((running_mean == nullptr) == (running_var == nullptr), FV2P_EINVAL, "sparse_conv_rows_bnfin: running_mean and running_var come together");
  FV2P_REQUIRE((pre_mean == nullptr) == (pre_invstd == nullptr), FV2P_EINVAL, "sparse_conv_rows_bnfin: pre_mean and pre_invstd come together");
  FV2P_REQUIRE(!pre_mean || ((c_src & 3) == 0 && (reinterpret_cast<uintptr_t>(pre_mean) & 15) == 0 && (reinterpret_cast<uintptr_t>(pre_invstd) & 15) == 0 &&
                            (reinterpret_cast<uintptr_t>(pre_gamma) & 15) == 0 && (reinterpret_cast<uintptr_t>(pre_beta) & 15) == 0),
               FV2P_EINVAL, "sparse_conv_rows_bnfin: source BatchNorm parameters must be 16-byte aligned, channels a multiple of 4");
  ConvExtra ex;
  ex.pre_mean = pre_mean; ex.pre_invstd = pre_invstd; ex.pre_gamma = pre_gamma; ex.pre_beta = pre_beta; ex.pre_relu = pre_relu;
  const bool fused = stats && c_src <= 128 && conv_impl() != 2 && n_dst > 0;
  const BnFwdFin ff{mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), momentum, eps};
  if (fused) { ex.fin_counter = counter; ex.fwd = ff; }
  if (int rc = conv_rows_impl(src, n_src, c_src, weight, kvol, tab, n_dst, c_dst, flip_k, transpose_w, bias, dst, fused ? stats : nullptr, stream,
                              nullptr, nullptr, &ex))
    return rc;
  if (stats && !fused && n_dst > 0) {   // sums by a pass over dst, finalised by one workgroup (the workspace's head serves as its slots)
    FV2P_HIP(hipMemsetAsync(stats, 0, sizeof(double) * kStatSlots * 2 * static_cast<size_t>(c_dst), stream));
    if (int rc = bn_column_sums(dst, n_dst, c_dst, stats, stream)) return rc;
    return bn_finalize_forward(stats, n_dst, c_dst, ff, stream);
  }
  return 0;
}

// 1 when fv2p_sparse_conv_rows_bnfin can normalise the source rows of a conv of this shape on the gather (the kernel the dispatcher would
// choose has the PRE form: conv_rows_thin 16 -> 16, conv_rows_ksplit 64 / 128 source channels), else 0.
extern "C" int fv2p_sparse_conv_prenorm_supported(int c_src, int c_dst, int kvol, int64_t n_dst, int flip_k, int transpose_w) {
  if (c_src < 1 || c_dst < 1 || kvol < 1 || n_dst < 1 || transpose_w || c_src > 128 || c_dst > 128 || (c_src & 3)) return 0;
  ConvExtra ex;
  ex.dry = 1;
  float* fake = reinterpret_cast<float*>(static_cast<uintptr_t>(4096));   // never dereferenced: a dry run launches nothing
  ex.pre_mean = ex.pre_invstd = fake;
  const int rc = conv_rows_impl(fake, n_dst, c_src, fake, kvol, reinterpret_cast<const int*>(fake), n_dst, c_dst, flip_k, 0, nullptr, fake, nullptr, nullptr,
                                nullptr, nullptr, &ex);
  return rc == 0 ? 1 : 0;
}

extern "C" int fv2p_sparse_conv_rows_bnbwd_fin(const float* src, int64_t n_src, int c_src, const float* weight, int kvol, const int* tab,
                                               int64_t n_dst, int c_dst, int flip_k, int transpose_w, float* dst, const float* bn_x,
                                               const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int relu,
                                               double* stats, unsigned* counter, int batch_stats, float* dgamma, float* dbeta, float* coef,
                                               const int* perm, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(stats && counter && bn_x && bn_mean && bn_invstd && dgamma && dbeta && coef, FV2P_EINVAL, "sparse_conv_rows_bnbwd_fin: null pointer");
  const bool fused = c_src <= 128 && conv_impl() != 2 && n_dst > 0;
  ConvArgs bn;
  bn.bn_x = bn_x; bn.bn_mean = bn_mean; bn.bn_invstd = bn_invstd; bn.bn_gamma = bn_gamma; bn.bn_beta = bn_beta; bn.bn_relu = relu;
  ConvExtra ex;
  const BnBwdFin bf{dgamma, dbeta, coef, batch_stats};
  if (fused) { ex.fin_counter = counter; ex.bwd = bf; }
  if (int rc = conv_rows_impl(src, n_src, c_src, weight, kvol, tab, n_dst, c_dst, flip_k, transpose_w, nullptr, dst, fused ? stats : nullptr, stream,
                              fused ? &bn : nullptr, perm, &ex))
    return rc;
  if (!fused && n_dst > 0) {
    FV2P_HIP(hipMemsetAsync(stats, 0, sizeof(double) * kStatSlots * 2 * static_cast<size_t>(c_dst), stream));
    if (int rc = bn_backward_sums(bn_x, dst, n_dst, c_dst, bn_mean, bn_invstd, bn_gamma, bn_beta, relu, stats, stream)) return rc;
    return bn_finalize_backward(stats, n_dst, c_dst, bf, stream);
  }
  return 0;
}

extern "C" int fv2p_sparse_conv_rows_bnbwd(const float* src, int64_t n_src, int c_src, const float* weight, int kvol, const int* tab,
                                           int64_t n_dst, int c_dst, int flip_k, int transpose_w, float* dst, const float* bn_x,
                                           const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int relu,
                                           double* stats, const int* perm, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(stats && bn_x && bn_mean && bn_invstd, FV2P_EINVAL, "sparse_conv_rows_bnbwd: null pointer");
  const bool fused = c_src <= 128 && conv_impl() != 2 && n_dst > 0;   // as in fv2p_sparse_conv_rows_stats
  ConvArgs bn;
  bn.bn_x = bn_x; bn.bn_mean = bn_mean; bn.bn_invstd = bn_invstd; bn.bn_gamma = bn_gamma; bn.bn_beta = bn_beta; bn.bn_relu = relu;
  if (int rc = conv_rows_impl(src, n_src, c_src, weight, kvol, tab, n_dst, c_dst, flip_k, transpose_w, nullptr, dst, fused ? stats : nullptr, stream,
                              fused ? &bn : nullptr, perm))
    return rc;
  if (!fused && n_dst > 0) return bn_backward_sums(bn_x, dst, n_dst, c_dst, bn_mean, bn_invstd, bn_gamma, bn_beta, relu, stats, stream);
  return 0;
}

static int wgrad_rows_per_chunk(int64_t n_dst) {
  static int forced = -1;  // FV2P_WGRAD_RPC: tuning override (multiple of 64)
  if (forced < 0) { const char* e = FV2P_DEV_ENV("FV2P_WGRAD_RPC"); forced = e ? atoi(e) : 0; }
  if (forced >= 64 && forced <= kWgradMaxChunk) return forced;
  return n_dst > 200000 ? 1024 : 512;
}
static const int kDenseRpc = 256;  // rows per wave for an identity (every row valid) offset

extern "C" size_t fv2p_sparse_conv_wgrad_ws_bytes(int64_t n_dst, int c_src, int c_dst, int kvol) {
  const int64_t n = n_dst > 0 ? n_dst : 1;
  const int cd = c_dst < 128 ? c_dst : 128, cs = c_src < 128 ? c_src : 128;
  Sizer s;
  s.take<float>(static_cast<size_t>(ceil_div(n, wgrad_rows_per_chunk(n_dst))) * kvol * cs * cd);
  s.take<float>(static_cast<size_t>(ceil_div(n, kDenseRpc)) * cs * cd);
  return s.bytes();
}

template <int MB, int NB>
static void wgrad_launch(const WgradArgs& a, float* partial, unsigned chunks, hipStream_t stream) {
  const dim3 grid(chunks, static_cast<unsigned>(a.k_count), 1), block(256);
  constexpr int stage = (MB * NB >= 64) ? 32 : 64;
  const size_t lds = static_cast<size_t>(stage) * (MB * 16 + 4 + NB * 16 + 4) * sizeof(float) + 2 * a.rows_per_chunk * sizeof(int);
  hipLaunchKernelGGL((conv_wgrad<MB, NB>), grid, block, lds, stream, a, partial);
}

extern "C" int fv2p_sparse_conv_wgrad(const float* src, int64_t n_src, int c_src, const float* grad, const int* tab, int64_t n_dst,
                                      int c_dst, int kvol, int flip_k, int dense_k, float* dweight, void* ws, size_t ws_bytes,
                                      fv2p_stream_t stream_) {
  return fv2p_sparse_conv_wgrad_pre(src, n_src, c_src, grad, tab, n_dst, c_dst, kvol, flip_k, dense_k, dweight, nullptr, nullptr, nullptr, nullptr, 0,
                                    ws, ws_bytes, stream_);
}

// ... with the gathered operand passed through a BatchNorm (+ReLU) on its way into the MFMAs (the forward conv did the same on its gather)
extern "C" int fv2p_sparse_conv_wgrad_pre(const float* src, int64_t n_src, int c_src, const float* grad, const int* tab, int64_t n_dst,
                                          int c_dst, int kvol, int flip_k, int dense_k, float* dweight, const float* pre_mean,
                                          const float* pre_invstd, const float* pre_gamma, const float* pre_beta, int pre_relu, void* ws,
                                          size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE((pre_mean == nullptr) == (pre_invstd == nullptr), FV2P_EINVAL, "sparse_conv_wgrad: pre_mean and pre_invstd come together");
  FV2P_REQUIRE(c_src >= 1 && c_dst >= 1 && kvol >= 1 && n_dst >= 0 && dense_k < kvol, FV2P_EINVAL, "sparse_conv_wgrad: bad sizes");
  FV2P_REQUIRE(dweight, FV2P_EINVAL, "sparse_conv_wgrad: null dweight");
  if (n_dst == 0 || n_src == 0) {
    FV2P_HIP(hipMemsetAsync(dweight, 0, sizeof(float) * (size_t)kvol * c_src * c_dst, stream));
    return 0;
  }
  FV2P_REQUIRE(src && grad && tab, FV2P_EINVAL, "sparse_conv_wgrad: null pointer");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_sparse_conv_wgrad_ws_bytes(n_dst, c_src, c_dst, kvol), FV2P_EWORKSPACE,
               "sparse_conv_wgrad: workspace too small");
  const int rpc = wgrad_rows_per_chunk(n_dst);
  const unsigned chunks = static_cast<unsigned>(ceil_div(n_dst, rpc));
  const unsigned dchunks = static_cast<unsigned>(ceil_div(n_dst, kDenseRpc));
  for (int d0 = 0; d0 < c_dst; d0 += 128) {
    const int cd = (c_dst - d0) < 128 ? (c_dst - d0) : 128;
   for (int s0 = 0; s0 < c_src; s0 += 128) {
    const int cs = (c_src - s0) < 128 ? (c_src - s0) : 128;
    Carver c(ws, ws_bytes);
    float* partial = c.take<float>(static_cast<size_t>(chunks) * kvol * cs * cd);
    float* dpartial = c.take<float>(static_cast<size_t>(dchunks) * cs * cd);
    WgradArgs a;
    a.src = src + s0; a.ld_src = c_src; a.c_src = cs;
    a.grad = grad + d0; a.ld_grad = c_dst; a.c_grad = cd;
    a.tab = tab; a.n_dst = static_cast<int>(n_dst); a.kvol = kvol; a.flip = flip_k & 1;
    a.dw = dweight + static_cast<long long>(s0) * c_dst + d0; a.dw_kstride = static_cast<long long>(c_src) * c_dst; a.dw_ld = c_dst;
    a.pre_mean = pre_mean ? pre_mean + s0 : nullptr; a.pre_invstd = pre_invstd ? pre_invstd + s0 : nullptr;
    a.pre_gamma = pre_gamma ? pre_gamma + s0 : nullptr; a.pre_beta = pre_beta ? pre_beta + s0 : nullptr; a.pre_relu = pre_relu;
    const int mb = static_cast<int>(ceil_div(cs, 16));
    const int nb = static_cast<int>(ceil_div(cd, 16));
    const int nbp = nb <= 1 ? 1 : nb <= 2 ? 2 : nb <= 4 ? 4 : 8;
    const int mbp = mb <= 1 ? 1 : mb <= 2 ? 2 : mb <= 4 ? 4 : 8;
    FV2P_REQUIRE(mb <= 8, FV2P_ELIMIT, "sparse_conv_wgrad: more than 128 source channels per launch (split on the host)");
    // pass 0: every offset except the dense one, rpc rows per workgroup; pass 1: the dense (identity) offset, whose pair
    // count equals the row count, in shorter chunks so that it is not the straggler of the launch
    for (int pass = 0; pass < (dense_k >= 0 ? 2 : 1); ++pass) {
      a.k_base = pass ? dense_k : 0;
      a.k_count = pass ? 1 : kvol;
      a.skip_k = pass ? -1 : dense_k;
      a.rows_per_chunk = pass ? kDenseRpc : rpc;
      float* pp = pass ? dpartial : partial;
      const unsigned nch = pass ? dchunks : chunks;
#define FV2P_WG(MB, NB) wgrad_launch<MB, NB>(a, pp, nch, stream)
#define FV2P_WG_NB(MB)                                                                   \
      switch (nbp) { case 1: FV2P_WG(MB, 1); break; case 2: FV2P_WG(MB, 2); break;      \
                     case 4: FV2P_WG(MB, 4); break; default: FV2P_WG(MB, 8); break; }
      switch (mbp) { case 1: FV2P_WG_NB(1) break; case 2: FV2P_WG_NB(2) break; case 4: FV2P_WG_NB(4) break; default: FV2P_WG_NB(8) break; }
#undef FV2P_WG_NB
#undef FV2P_WG
      const long long per_chunk = static_cast<long long>(a.k_count) * cs * cd;
      hipLaunchKernelGGL(wgrad_reduce, dim3(static_cast<unsigned>(ceil_div(per_chunk, 16))), dim3(256), 0, stream, pp, (int)nch, a.k_base,
                         a.k_count, cs, cd, a.dw, a.dw_kstride, a.dw_ld, a.skip_k);
    }
   }
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}

template <int MB, int NB>
static void wgrad_pairs_launch(const WgradPairArgs& a, float* partial, unsigned chunks, hipStream_t stream) {
  const dim3 grid(chunks, static_cast<unsigned>(a.kvol), 1), block(256);
  constexpr int stage = (MB * NB >= 64) ? 32 : 64;
  const size_t lds = static_cast<size_t>(stage) * (MB * 16 + 4 + NB * 16 + 4) * sizeof(float) + 2 * kPairsMax * sizeof(int);
  hipLaunchKernelGGL((conv_wgrad_pairs<MB, NB>), grid, block, lds, stream, a, partial);
}

extern "C" size_t fv2p_sparse_conv_wgrad_pairs_ws_bytes(int64_t pair_len, int c_src, int c_dst, int kvol) {
  const int cs = c_src < 128 ? c_src : 128, cd = c_dst < 128 ? c_dst : 128;
  Sizer s;
  s.take<float>(static_cast<size_t>(ceil_div(pair_len > 0 ? pair_len : 1, pairs_per_chunk(pair_len))) * kvol * cs * cd);
  return s.bytes();
}

extern "C" int fv2p_sparse_conv_wgrad_pairs(const float* src, int64_t n_src, int c_src, const float* grad, int64_t n_grad, int c_dst,
                                            const int* pairs, const int* pair_num, int kvol, int64_t pair_len, int side_src,
                                            float* dweight, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  return fv2p_sparse_conv_wgrad_pairs_pre(src, n_src, c_src, grad, n_grad, c_dst, pairs, pair_num, kvol, pair_len, side_src, dweight, nullptr, nullptr,
                                          nullptr, nullptr, 0, ws, ws_bytes, stream_);
}

extern "C" int fv2p_sparse_conv_wgrad_pairs_pre(const float* src, int64_t n_src, int c_src, const float* grad, int64_t n_grad, int c_dst,
                                                const int* pairs, const int* pair_num, int kvol, int64_t pair_len, int side_src,
                                                float* dweight, const float* pre_mean, const float* pre_invstd, const float* pre_gamma,
                                                const float* pre_beta, int pre_relu, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE((pre_mean == nullptr) == (pre_invstd == nullptr), FV2P_EINVAL, "sparse_conv_wgrad_pairs: pre_mean and pre_invstd come together");
  FV2P_REQUIRE(c_src >= 1 && c_dst >= 1 && kvol >= 1 && pair_len >= 0 && (side_src == 0 || side_src == 1), FV2P_EINVAL,
               "sparse_conv_wgrad_pairs: bad sizes");
  FV2P_REQUIRE(dweight, FV2P_EINVAL, "sparse_conv_wgrad_pairs: null dweight");
  if (pair_len == 0 || n_src == 0 || n_grad == 0) {
    FV2P_HIP(hipMemsetAsync(dweight, 0, sizeof(float) * (size_t)kvol * c_src * c_dst, stream));
    return 0;
  }
  FV2P_REQUIRE(src && grad && pairs && pair_num, FV2P_EINVAL, "sparse_conv_wgrad_pairs: null pointer");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_sparse_conv_wgrad_pairs_ws_bytes(pair_len, c_src, c_dst, kvol), FV2P_EWORKSPACE,
               "sparse_conv_wgrad_pairs: workspace too small");
  const int chunk = pairs_per_chunk(pair_len);
  const unsigned chunks = static_cast<unsigned>(ceil_div(pair_len, chunk));
  for (int d0 = 0; d0 < c_dst; d0 += 128) {
    const int cd = (c_dst - d0) < 128 ? (c_dst - d0) : 128;
    for (int s0 = 0; s0 < c_src; s0 += 128) {
      const int cs = (c_src - s0) < 128 ? (c_src - s0) : 128;
      Carver c(ws, ws_bytes);
      float* partial = c.take<float>(static_cast<size_t>(chunks) * kvol * cs * cd);
      WgradPairArgs a;
      a.src = src + s0; a.ld_src = c_src; a.c_src = cs;
      a.grad = grad + d0; a.ld_grad = c_dst; a.c_grad = cd;
      a.pairs = pairs; a.pair_len = pair_len; a.pair_num = pair_num; a.side_src = side_src; a.kvol = kvol; a.chunk = chunk;
      a.pre_mean = pre_mean ? pre_mean + s0 : nullptr; a.pre_invstd = pre_invstd ? pre_invstd + s0 : nullptr;
      a.pre_gamma = pre_gamma ? pre_gamma + s0 : nullptr; a.pre_beta = pre_beta ? pre_beta + s0 : nullptr; a.pre_relu = pre_relu;
      const int mb = static_cast<int>(ceil_div(cs, 16)), nb = static_cast<int>(ceil_div(cd, 16));
      const int nbp = nb <= 1 ? 1 : nb <= 2 ? 2 : nb <= 4 ? 4 : 8;
      const int mbp = mb <= 1 ? 1 : mb <= 2 ? 2 : mb <= 4 ? 4 : 8;
      const bool dma_ok = (cs == 64 || cs == 128) && (cd == 64 || cd == 128) && (c_src & 3) == 0 && (c_dst & 3) == 0 &&
                          (reinterpret_cast<uintptr_t>(a.src) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.grad) & 15) == 0 && g_wgrad_dma;
      if (dma_ok) {
        const dim3 grid(chunks, static_cast<unsigned>(kvol), 1), block(256);
        const int as = cs / 64, bs = cd / 64;
        // 64 x 64: 64-pair stages (64 KB LDS, 2 workgroups per CU) when the launch has about one live workgroup per CU,
        // 32-pair stages (4 per CU) when it has several (measured: 35 vs 43 us at N = 13k, 54 vs 61 us at N = 29k)
        static int st_env = -1;
        if (st_env < 0) { const char* e = FV2P_DEV_ENV("FV2P_WGRAD_ST"); st_env = e ? atoi(e) : 0; }
        const int st64 = st_env ? st_env : (static_cast<long long>(chunks) * kvol > 1200 ? 32 : 64);
        const int st = (as + bs <= 2) ? st64 : 32;
        const size_t lds = static_cast<size_t>(2) * st * (cs + cd) * sizeof(float) + 2 * kPairsMax * sizeof(int);
        if (as == 1 && bs == 1 && st == 64) hipLaunchKernelGGL((conv_wgrad_pairs_dma<1, 1, 64>), grid, block, lds, stream, a, partial);
        else if (as == 1 && bs == 1) hipLaunchKernelGGL((conv_wgrad_pairs_dma<1, 1, 32>), grid, block, lds, stream, a, partial);
        else if (as == 2 && bs == 1) hipLaunchKernelGGL((conv_wgrad_pairs_dma<2, 1, 32>), grid, block, lds, stream, a, partial);
        else if (as == 1 && bs == 2) hipLaunchKernelGGL((conv_wgrad_pairs_dma<1, 2, 32>), grid, block, lds, stream, a, partial);
        else hipLaunchKernelGGL((conv_wgrad_pairs_dma<2, 2, 32>), grid, block, lds, stream, a, partial);
      } else {
#define FV2P_WGP(MB, NB) wgrad_pairs_launch<MB, NB>(a, partial, chunks, stream)
#define FV2P_WGP_NB(MB)                                                                    \
      switch (nbp) { case 1: FV2P_WGP(MB, 1); break; case 2: FV2P_WGP(MB, 2); break;      \
                     case 4: FV2P_WGP(MB, 4); break; default: FV2P_WGP(MB, 8); break; }
      switch (mbp) { case 1: FV2P_WGP_NB(1) break; case 2: FV2P_WGP_NB(2) break; case 4: FV2P_WGP_NB(4) break; default: FV2P_WGP_NB(8) break; }
#undef FV2P_WGP_NB
#undef FV2P_WGP
      }
      const long long per_chunk = static_cast<long long>(kvol) * cs * cd;
      hipLaunchKernelGGL(wgrad_reduce_pairs, dim3(static_cast<unsigned>(ceil_div(per_chunk, 16))), dim3(256), 0, stream, partial, pair_num, kvol,
                         chunk, cs, cd, dweight + static_cast<long long>(s0) * c_dst + d0, static_cast<long long>(c_src) * c_dst, c_dst);
    }
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}
