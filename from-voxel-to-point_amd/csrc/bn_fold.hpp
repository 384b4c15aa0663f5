// BatchNorm1d statistics: the fold of the per-slot partial sums and the finalisation (mean / invstd / running statistics forward;
// dgamma / dbeta / the two means of the input gradient backward), shared by batchnorm.hip and the conv epilogues of sparse_conv.hip.
//
// Who finalises (round 6).  The conv kernels leave per-column partial sums in kStatSlots accumulator rows (fp64 atomics from their
// epilogues).  Rounds 2 - 5 folded those slots in EVERY workgroup of the BatchNorm apply kernel that followed (up to 512 workgroups
// x 64 slots x 2 x C doubles: most of a 12 us launch at the backbones' sizes).  Now the LAST workgroup of the conv launch itself folds
// them once (conv_stats_done, sparse_conv.hip) and leaves mean / invstd (or dgamma, dbeta, c1, c2) in memory: apply kernels read
// 2 - 4 floats per column, and a consumer conv can normalise its gathered rows without any BatchNorm launch in between.
// The fold order is the one the apply kernels always used (fold_chunk), so the statistics are bit-identical whoever folds.
#pragma once
#include "common.hpp"

namespace fv2p {

struct BnGeom {
  long long n;
  int c, tcols, rpp, nblk;
  long long rows_per_block;
};

struct BnFwdFin {   // outputs of the forward finalisation
  float* mean; float* invstd;
  float* running_mean; float* running_var; long long* num_batches_tracked;
  float momentum;   // < 0: cumulative moving average (momentum=None)
  float eps;
};
struct BnBwdFin {
  float* dgamma; float* dbeta; float* coef;  // coef[0][c] = mean(dz), coef[1][c] = mean(dz * xhat) (0 when running stats were used)
  int batch_stats;
};

// Plain loads (the partials were written by an earlier launch) or agent-scope atomic loads (the partials were accumulated by other
// workgroups of THIS launch with agent-scope atomics: a plain load may be served from this XCD's L2, which is not coherent with the others)
template <bool COHERENT>
__device__ __forceinline__ double stat_load(const double* p) {
  if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}

// Publishing a value other workgroups of the SAME launch will read after a counter says so.  A plain or agent-scope STORE is
// acknowledged to the wave when this XCD's L2 has taken it (write-through, but the acknowledgement does not wait for the fabric): the
// counter atomic issued after `s_waitcnt vmcnt(0)` can then be performed at the memory side BEFORE the data lands there, and a reader on
// another XCD gets the old contents - seen as garbage statistics once two processes shared the GPU (tests/test_ddp_gpu.py, round 6).
// A read-modify-write atomic is executed by the memory-side atomic unit itself and acknowledged after it was performed, exactly like the
// counter: same-path ordering.  Hence an exchange instead of a store.
__device__ __forceinline__ void stat_publish(double* p, double v) {
  (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Folds the [nblk][2][c] partials of channels [e0, e0 + cfold) in a fixed order: L = 256 / cfold lanes per channel take
// interleaved slices, then an ordered LDS fold.  Returns the two sums of channel e0 + tid (valid for tid < cfold).  256 threads.
template <bool COHERENT>
__device__ __forceinline__ void fold_chunk(int nblk, int c, const double* __restrict__ partial, int e0, int cfold, double (*red)[256],
                                           double* a_out, double* b_out, int ld = 0) {
  if (ld == 0) ld = c;   // row stride of the partials (a conv launch on a column block of a wider layer: ld = all columns)
  const int tid = threadIdx.x, L = 256 / cfold;
  const int e = e0 + tid % cfold, lane_q = tid / cfold;
  double a = 0.0, b = 0.0;
  if (lane_q < L && e < c) {
#pragma unroll 8
    for (int q = lane_q; q < nblk; q += L) {
      a += stat_load<COHERENT>(partial + (static_cast<long long>(q) * 2 + 0) * ld + e);
      b += stat_load<COHERENT>(partial + (static_cast<long long>(q) * 2 + 1) * ld + e);
    }
  }
  __syncthreads();
  red[0][tid] = a; red[1][tid] = b;
  __syncthreads();
  a = 0.0; b = 0.0;
  if (tid < cfold)
    for (int q = 0; q < L; ++q) { a += red[0][q * cfold + tid]; b += red[1][q * cfold + tid]; }
  *a_out = a; *b_out = b;
}

// forward: batch mean / invstd of channel e from its folded sums (a = sum x, b = sum x^2 over n rows); `write` (one workgroup of the
// launch) also stores them and moves the running statistics.  Same arithmetic wherever it runs.
__device__ __forceinline__ void bn_fwd_channel(double a, double b, long long n_rows, const BnFwdFin& ff, int e, bool write, float* mu_f, float* is_f) {
  const double n = static_cast<double>(n_rows);
  const double mu = a / n;
  double var = b / n - mu * mu;
  if (var < 0.0) var = 0.0;
  *mu_f = static_cast<float>(mu);
  *is_f = static_cast<float>(1.0 / sqrt(var + static_cast<double>(ff.eps)));
  if (write) {
    ff.mean[e] = *mu_f;
    ff.invstd[e] = *is_f;
    if (ff.running_mean) {
      double f = ff.momentum;
      if (ff.momentum < 0.f) f = 1.0 / static_cast<double>(ff.num_batches_tracked ? (*ff.num_batches_tracked + 1) : 1);
      const double unbiased = n_rows > 1 ? var * n / (n - 1.0) : var;
      ff.running_mean[e] = static_cast<float>((1.0 - f) * ff.running_mean[e] + f * mu);
      ff.running_var[e] = static_cast<float>((1.0 - f) * ff.running_var[e] + f * unbiased);
    }
  }
}

// ---- finalisation of per-tile rows by the launch's last workgroups (conv_stats_done in sparse_conv.hip, bn_reduce_fin_k in batchnorm.hip) ------
// Protocol (every workgroup of the launch calls fin_rows_done once, after it PUBLISHED its row with stat_publish):
//   s_waitcnt vmcnt(0): the row's atomics have been performed; the workgroup counts itself in on the word of its group (tile % G);
//   whoever completes a group folds the group's rows, in row order, into the group's slot and counts the group in on the top word;
//   whoever completes the top word folds the slots, in group order, and writes the results (FinOut).
// Other workgroups' rows are read with agent-scope loads.  No fence, no float atomic add, nothing to clear; the sums are bit-identical
// from run to run.  counter: fv2p_sparse_conv_fin_counter_words() zeroed words, zero again when the launch ends.
constexpr unsigned kFinSubs = 64, kFinStride = 32;   // counter: [0] the top word, [(1 + g) * kFinStride] group g
struct FinOut {
  int bwd;          // 0: (sum x, sum x^2) -> mean / invstd / running statistics;  1: (sum dz, sum dz * xhat) -> dgamma / dbeta / coef
  long long n;      // rows of the statistic
  BnFwdFin ff;
  BnBwdFin bf;
  int bump;         // forward: advance num_batches_tracked
  int coef_ld;      // backward: coef[0][c] at coef, coef[1][c] at coef + coef_ld
};
// one thread: the sum over `count` doubles p[0], p[stride], ... in that order, with the loads batched sixteen deep (an agent-scope
// load is a ~0.4 us round trip past the L2)
__device__ __forceinline__ double fin_sum(const double* p, int count, long long stride) {
  double acc = 0.0;
  for (int r0 = 0; r0 < count; r0 += 16) {
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (r0 + i < count) ? stat_load<true>(p + (r0 + i) * stride) : 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i];   // (x + 0.0 == x)
  }
  return acc;
}
// c <= 128 columns (one (column, which) pair per thread of the first 256); ld: row stride of rows / slots in doubles;
// flag, red: LDS scratch (one word, [2][256] doubles) nothing else uses any more
__device__ __forceinline__ void fin_rows_done(double* rows, double* gslots, unsigned* counter, int tile, int n_tiles, int ncb, int G, int c, int ld,
                                              unsigned* flag, double (*red)[256], const FinOut& fo) {
  const int tid = threadIdx.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int grp = tile % G;
  if (tid == 0) {
    const unsigned expect = static_cast<unsigned>((n_tiles - grp + G - 1) / G) * static_cast<unsigned>(ncb);
    unsigned* word = counter + (1 + grp) * kFinStride;
    unsigned role = 0u;
    if (__hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expect - 1u) {
      __hip_atomic_store(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      role = 1u;
    }
    *flag = role;
  }
  __syncthreads();
  if (*flag == 0u || tid >= 256) return;   // (a 1024-thread workgroup folds with its first four waves)
  const int which = tid / c, e = tid % c;
  const bool mine = tid < 2 * c;
  double* gslot = gslots + static_cast<long long>(grp) * 2 * ld;
  if (mine) {
    const int count = (n_tiles - grp + G - 1) / G;
    stat_publish(gslot + static_cast<long long>(which) * ld + e, fin_sum(rows + (static_cast<long long>(grp) * 2 + which) * ld + e, count, static_cast<long long>(G) * 2 * ld));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int groups = n_tiles < G ? n_tiles : G;
  if (tid == 0) *flag = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == static_cast<unsigned>(groups) - 1u ? 2u : 0u;
  __syncthreads();
  if (*flag != 2u) return;
  if (mine) red[which][e] = fin_sum(gslots + static_cast<long long>(which) * ld + e, groups, 2ll * ld);
  __syncthreads();
  if (tid < c) {
    const double sa = red[0][tid], sb = red[1][tid];
    if (!fo.bwd) {
      float mu, is;
      bn_fwd_channel(sa, sb, fo.n, fo.ff, tid, true, &mu, &is);
    } else {
      const double nn = static_cast<double>(fo.n);
      fo.bf.dbeta[tid] = static_cast<float>(sa);
      fo.bf.dgamma[tid] = static_cast<float>(sb);
      fo.bf.coef[tid] = fo.bf.batch_stats ? static_cast<float>(sa / nn) : 0.f;
      fo.bf.coef[fo.coef_ld + tid] = fo.bf.batch_stats ? static_cast<float>(sb / nn) : 0.f;
    }
  }
  if (tid == 0) {
    if (!fo.bwd && fo.bump && fo.ff.running_mean && fo.ff.num_batches_tracked) *fo.ff.num_batches_tracked += 1;
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// one-workgroup finalisation of sums a separate pass took (batchnorm.hip)
int bn_finalize_forward(double* stats, int64_t n, int c, const BnFwdFin& ff, hipStream_t stream);
int bn_finalize_backward(double* stats, int64_t n, int c, const BnBwdFin& bf, hipStream_t stream);

}  // namespace fv2p
