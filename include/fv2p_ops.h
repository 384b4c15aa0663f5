/*
 * libfv2p_ops — C ABI of the MI355X-native (gfx950) voxel-to-point hot path.
 *
 * This header is the drop-in boundary underneath the reference's pybind/torch extension
 * modules.  Every entry point takes plain device pointers, explicit sizes,
 * a caller-provided workspace and a HIP stream (passed as void*), and returns 0 or a
 * negative FV2P_E* code; fv2p_last_error() gives the message.  Nothing here allocates
 * device memory, touches the legacy default stream or calls exit().
 *
 * Each declaration cites the reference interface it replaces (paths relative to the
 * reference checkout, jialeli1/From-Voxel-to-Point).
 */
#ifndef FV2P_OPS_H_
#define FV2P_OPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FV2P_ABI_VERSION 1

#define FV2P_EINVAL      (-1)  /* bad argument (shape, range, null pointer)            */
#define FV2P_EWORKSPACE  (-2)  /* workspace smaller than the matching *_ws_bytes query */
#define FV2P_EHIP        (-3)  /* a HIP runtime call failed                            */
#define FV2P_ELIMIT      (-4)  /* problem exceeds a documented packing limit           */

typedef void* fv2p_stream_t;   /* hipStream_t */

const char* fv2p_last_error(void);
int fv2p_abi_version(void);

/* ---- primitives (exported for tests; used internally by voxeliser and rulebook) ------- */
size_t fv2p_scan_ws_bytes(int64_t n);
int fv2p_exclusive_scan_i32(const int* in, int* out, int64_t n, int* total, void* ws, size_t ws_bytes,
                            fv2p_stream_t stream);
size_t fv2p_radix_sort_ws_bytes(int64_t n);
int fv2p_radix_sort_u64(uint64_t* keys, uint64_t* tmp, int64_t n, int bit_lo, int bit_hi, void* ws,
                        size_t ws_bytes, fv2p_stream_t stream);

/* ---- A1: points_to_voxel ----------------------------------------------------------------
 * Replaces pcdet/datasets/processor/voxel_generator.py:75-133 (points_to_voxel) and
 * :136-207 (_points_to_voxel_reverse_kernel): first-come voxelisation, coords stored (z,y,x),
 * fp32 floor((p-lo)/vs), the whole scan stops at the first point that would open voxel number
 * max_voxels+1 (:198-199).
 *   points      [n_points, ndim] f32 (device), xyz first
 *   voxel_size  [3] f32 host (x,y,z);  range_lo [3] f32 host;  grid [3] i32 host (x,y,z)
 *   voxels      [max_voxels, max_points, ndim] f32 (device) — rows >= *num_voxels untouched zero
 *   coors       [max_voxels, 3] i32 (z,y,x);  num_points_per_voxel [max_voxels] i32
 *   num_voxels  device int: number of voxels produced (<= max_voxels)
 * All outputs are fully written (zero padded) by the call.
 */
size_t fv2p_points_to_voxel_ws_bytes(int64_t n_points, int max_voxels);
int fv2p_points_to_voxel(const float* points, int64_t n_points, int ndim, const float voxel_size[3],
                         const float range_lo[3], const int grid[3], int max_points, int max_voxels,
                         float* voxels, int* coors, int* num_points_per_voxel, int* num_voxels,
                         void* ws, size_t ws_bytes, fv2p_stream_t stream);

/* Host entry point of A1: the reference calls VoxelGenerator.generate on numpy arrays inside forked DataLoader workers
 * (pcdet/datasets/processor/data_processor.py:43-81 -> voxel_generator.py:75-207), where HIP cannot be initialised.  All
 * pointers are HOST pointers, the scan runs on the calling thread and makes no HIP call; same outputs as above (voxels,
 * num_points_per_voxel zero-filled by the call), *num_voxels is a host int. */
int fv2p_points_to_voxel_host(const float* points, int64_t n_points, int ndim, const float voxel_size[3],
                              const float range_lo[3], const int grid[3], int max_points, int max_voxels,
                              float* voxels, int* coors, int* num_points_per_voxel, int* num_voxels);

/* ---- A3/A4: sparse-conv rulebook ----------------------------------------------------------
 * Replaces sparse_conv_ext.get_indice_pairs_{2d,3d} (pcdet/ops/spconv/src/all.cc:22-33 ->
 * spconv_ops.h:27-140 getIndicePair<NDim>, kernels indice.cu.h:22-203, CPU geometry.h:24-297).
 * Geometry arrays are host int[3] in (z,y,x) order (2-D problems pass a unit leading dim).
 * subm!=0: stride:=1, padding:=ksize/2 (spconv_ops.h:76-80), out_shape must equal in_shape.
 *
 * The build is split around the one host-visible quantity, the number of output rows:
 *   begin : hashes the active set (subm) or the distinct candidate outputs (conv / transposed);
 *           *n_out_host = number of output rows (conv: synchronises `stream`, like the
 *           reference's numActOut, spconv_ops.h:131-139).
 *   finish: fills   out_indices [n_out,4] (b,z,y,x) sorted by flat index (conv only; null for subm),
 *                   tab_in  [K, n_in ]  output row fed by input row i through offset k, or -1,
 *                   tab_out [K, n_out]  input row feeding output row o through offset k, or -1
 *                                       (may be null: for subm with odd ksize and dilation 1
 *                                        tab_out[k] == tab_in[K-1-k]),
 *                   indice_num [K]      pairs per offset (the reference's indiceNum); may be null and obtained
 *                                       later with fv2p_rulebook_count (the fused conv kernels never read it).
 * begin and finish must be given the same workspace (>= fv2p_rulebook_ws_bytes) and arguments.
 * Kernel offset index k = (kz*Ky + ky)*Kx + kx with k_j = in_j - out_j*s_j + p_j (geometry.h:62-73).
 */
size_t fv2p_rulebook_ws_bytes(int64_t n_in, const int ksize[3], const int stride[3], const int dilation[3],
                              int subm, int transpose);
/* Same, plus room for the bitmap path of strided / transposed rulebooks: when batch * output volume <= 2^28 cells and the
 * workspace has this size, begin/finish rank the output cells with one bit per cell and a popcount scan (ascending
 * flat index = the reference's sorted-unique order) instead of the hash set + radix sort; results are identical. */
size_t fv2p_rulebook_ws_bytes_grid(int64_t n_in, int batch, const int out_shape[3], const int ksize[3],
                                   const int stride[3], const int dilation[3], int subm, int transpose);
/* 0 (default): bitmap path whenever it applies; 1: always the hash set + radix sort. Process-wide; for tests / A-B runs. */
int fv2p_rulebook_set_path(int path);
int fv2p_rulebook_begin(const int* indices, int64_t n_in, int batch, const int in_shape[3],
                        const int out_shape[3], const int ksize[3], const int stride[3], const int padding[3],
                        const int dilation[3], int subm, int transpose, int64_t* n_out_host, void* ws,
                        size_t ws_bytes, fv2p_stream_t stream);
int fv2p_rulebook_finish(const int* indices, int64_t n_in, int batch, const int in_shape[3],
                         const int out_shape[3], const int ksize[3], const int stride[3], const int padding[3],
                         const int dilation[3], int subm, int transpose, int64_t n_out, int* out_indices,
                         int* tab_in, int* tab_out, int* indice_num, void* ws, size_t ws_bytes,
                         fv2p_stream_t stream);
/* 4-D rulebooks (SparseConv4d / SubMConv4d: spconv_ops.h:143-258 and the 4-D instantiations of getIndicePair, all.cc:22-33; Python
 * entry point get_indice_pairs_4d, ops.py:96-150 of the reference).  indices [n_in, 5] = (batch, four coordinates), out_indices
 * [n_out, 5]; the same two-phase hash set + radix sort as above (output rows in ascending (batch, cell) order), no bitmap path and
 * no transposed form (the reference has no SparseConvTranspose4d).  Kernel offset index: row-major over ksize[0..3]. */
size_t fv2p_rulebook4d_ws_bytes(int64_t n_in, const int ksize[4], const int stride[4], const int dilation[4], int subm);
int fv2p_rulebook4d_begin(const int* indices, int64_t n_in, int batch, const int in_shape[4],
                          const int out_shape[4], const int ksize[4], const int stride[4], const int padding[4],
                          const int dilation[4], int subm, int64_t* n_out_host, void* ws, size_t ws_bytes,
                          fv2p_stream_t stream);
int fv2p_rulebook4d_finish(const int* indices, int64_t n_in, int batch, const int in_shape[4],
                           const int out_shape[4], const int ksize[4], const int stride[4], const int padding[4],
                           const int dilation[4], int subm, int64_t n_out, int* out_indices, int* tab_in,
                           int* tab_out, int* indice_num, void* ws, size_t ws_bytes, fv2p_stream_t stream);
/* Rows [n,4] (b,z,y,x) ordered by the residue class of (coordinate + padding) mod stride, stable inside a class:
 * input rows of one class of a strided conv reach the same few kernel offsets.  Strides 1 or 2. */
size_t fv2p_rulebook_class_perm_ws_bytes(int64_t n);
int fv2p_rulebook_class_perm(const int* indices, int64_t n, const int stride[3], const int padding[3], int* perm,
                             void* ws, size_t ws_bytes, fv2p_stream_t stream);
int fv2p_rulebook_count(const int* tab_in, int64_t n_in, int kvol, int* indice_num, fv2p_stream_t stream);
/* Reference-format pair lists indicePairs [K,2,n_in] from tab_in; within one offset pairs are ordered by ascending
 * input row (the CPU reference's order, geometry.h:281-295; the GPU reference's slot order is atomic-order).
 * pad != 0 fills the unused tail with -1 (spconv_ops.h:55-57); pad == 0 leaves it unwritten (enough for
 * fv2p_sparse_conv_wgrad_pairs, which reads pair_num[k] entries).  pair_num [K] (optional) receives indiceNum. */
size_t fv2p_rulebook_pairs_ws_bytes(int64_t n_in, int kvol);
int fv2p_rulebook_pairs(const int* tab_in, int64_t n_in, int kvol, int pad, int* pairs, int* pair_num, void* ws,
                        size_t ws_bytes, fv2p_stream_t stream);
/* Tables from caller-supplied pair lists [K,2,pair_len] + indice_num [K] (device). */
int fv2p_pairs_to_tables(const int* pairs, const int* indice_num, int kvol, int64_t pair_len, int64_t n_in,
                         int64_t n_out, int* tab_in, int* tab_out, fv2p_stream_t stream);

/* ---- A5/A6: fused sparse convolution ------------------------------------------------------
 * Replaces sparse_conv_ext.indice_conv_fp32 / indice_conv_backward_fp32 / fused_indice_conv_fp32
 * (all.cc:34-51 -> spconv_ops.h:260-457, fused_spconv_ops.h:28-131; gather/scatter kernels
 * reordering.cu.h:21-157 and the torch::mm between them).
 *   dst[r,:] = (bias) + sum_k src[tab[k][r],:] . W_k           rows with tab == -1 are skipped
 * weight is the reference parameter layout [K][Cin][Cout] (spconv/conv.py:98-99).
 *   forward        : src=features[n_in,Cin],  tab=tab_out, n_dst=n_out, c_dst=Cout, transpose_w=0
 *   backward data  : src=dOut[n_out,Cout],    tab=tab_in,  n_dst=n_in,  c_dst=Cin,  transpose_w=1
 *   inverse conv   : src=features[n_out,Cin'], tab=tab_in, n_dst=n_in,  transpose_w=0
 * flip_k is a flag word: bit 0 (FV2P_TAB_FLIP) reads table row K-1-k for offset k (subm symmetry, lets subm reuse
 * tab_in as tab_out); bit 1 (FV2P_TAB_PLANNED, conv entry points only) promises that the table's allocation continues
 * with the tiling plan fv2p_conv_plan_build wrote behind it (see below).  The reference has no counterpart of the
 * plan: its gather -> mm -> scatter loop balances by construction (one GEMM per offset, spconv_ops.h:300-357).
 * fp32 in / fp32 accumulate on v_mfma_f32_16x16x4_f32; deterministic (fixed k order, no atomics).
 */
#define FV2P_TAB_FLIP 1
#define FV2P_TAB_PLANNED 2
/* Cost-balanced tiling of a table's destination rows for the fused conv kernels.  A table `tab` [kvol][n_dst] allocated
 * with fv2p_conv_plan_ints(n_dst) extra ints behind it can be given a plan once (it depends on the table only, so every
 * conv that shares the rulebook — forward, flipped submanifold backward, any channel count — reuses it): the rows are cut
 * into contiguous ranges of equal cost, cost(row) = max(pairs of the row, 8), for tile counts 256, 384, 512, 768, ...;
 * a launch picks the level that suits its shape and a workgroup processes one range.  Results are bit-identical with
 * and without a plan (every row's sum still runs over ascending k inside one workgroup); only the work per workgroup
 * changes: equal-row tiles carry up to 2x the median pairs at the 5-20 pairs per row of the backbones' deep levels. */
int64_t fv2p_conv_plan_ints(int64_t n_dst);
size_t fv2p_conv_plan_ws_bytes(int64_t n_dst);
int fv2p_conv_plan_build(int* tab, int kvol, int64_t n_dst, void* ws, size_t ws_bytes, fv2p_stream_t stream);
/* Test / tuning hook: force one kernel variant of fv2p_sparse_conv_rows for the calls that follow
 * (0 = heuristic, 1 = plain dense tile, 2 = compacted tile, 3 = register-staged pipeline).  All variants compute
 * the same sum in the same k order. */
int fv2p_sparse_conv_set_impl(int impl);
/* Test / tuning hook for the two thin-layer kernels the heuristic (impl 0) adds for the full 3 x 3 x 3 kernel: conv_rows_thin
 * (16 -> 16, any row count; with it conv_rows_first: <= 8 source channels that are not a multiple of 4 or fewer than 16, <= 32 destination
 * channels - the backbones' first layer) and conv_rows_res (32 source channels, <= 32 destination channels, <= 65 536 rows).  1 = on, 0 = off
 * (the staged kernels take those launches), -1 = back to the FV2P_CONV_THIN / FV2P_CONV_RES preset (default on).  Same sum, same k order. */
int fv2p_sparse_conv_set_paths(int thin_on, int res_on);
/* Profiling hook: when non-NULL, every workgroup of the LDS-DMA conv kernel writes {HW_ID, XCC_ID, start, end of the
 * offset loop, end of the prologue, clocks spent waiting at the per-offset barrier (wave 0), 0, 0} (shader clock) to
 * trace[8*blockIdx .. +7] (device memory, >= 8*ceil(n_dst/64) entries).  NULL switches it off. */
int fv2p_sparse_conv_set_trace(unsigned long long* trace);
/* Measurement hook (bench.py's roofline.in_step_us): while armed, every conv launch of fv2p_sparse_conv_rows* with exactly these
 * channel counts, kernel volume, destination rows and table direction (flip_k & 1) - forward convs that gather their source rows as
 * they are (not the PRE form of fv2p_sparse_conv_rows_bnfin: another kernel instance) - is bracketed by a pair of HIP events on the
 * launch stream (at most 512 pairs; further launches run unbracketed).  fv2p_sparse_conv_probe_read waits for the recorded pairs,
 * returns their number and the sum of their elapsed times in microseconds, and disarms.  No reference counterpart (its timers are
 * commented out: spconv_ops.h:305-360). */
int fv2p_sparse_conv_probe_arm(int c_src, int c_dst, int kvol, int64_t n_dst, int flip);
int fv2p_sparse_conv_probe_read(double* sum_us, int* launches);

/* fv2p_sparse_conv_rows that also leaves the per-column sum and sum of squares of dst (fp64) in `stats`
 * [fv2p_sparse_conv_stat_slots()][2][c_dst]: the caller passes it zero-filled, the conv epilogue (or, for shapes
 * whose sum takes several launches, a reduce pass after the conv) adds into it; the total over the slots is what
 * BatchNorm1d needs (fv2p_batchnorm_forward_stats).  Declared after fv2p_sparse_conv_rows below. */
int fv2p_sparse_conv_stat_slots(void);
int fv2p_sparse_conv_rows(const float* src, int64_t n_src, int c_src, const float* weight, int kvol,
                          const int* tab, int64_t n_dst, int c_dst, int flip_k, int transpose_w,
                          const float* bias, float* dst, fv2p_stream_t stream);
int fv2p_sparse_conv_rows_stats(const float* src, int64_t n_src, int c_src, const float* weight, int kvol,
                                const int* tab, int64_t n_dst, int c_dst, int flip_k, int transpose_w,
                                const float* bias, float* dst, double* stats, fv2p_stream_t stream);
/* Backward-data conv (no bias) whose result dst is the gradient of a BatchNorm1d(+ReLU) output: `stats` (zero-filled,
 * same slot layout) additionally receives that layer's backward sums, sum dz and sum dz * xhat with
 * dz = dst * [y > 0] and xhat, y recomputed from the BatchNorm input bn_x [n_dst, c_dst] — what
 * fv2p_batchnorm_backward_stats needs (c_src > 128: summed by a pass after the conv instead of in its epilogue). */
int fv2p_sparse_conv_rows_bnbwd(const float* src, int64_t n_src, int c_src, const float* weight, int kvol,
                                const int* tab, int64_t n_dst, int c_dst, int flip_k, int transpose_w, float* dst,
                                const float* bn_x, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                const float* bn_beta, int relu, double* stats, const int* perm, fv2p_stream_t stream);
/* ---- round 6: BatchNorm statistics finalised by the conv launch, BatchNorm (+ReLU) of the source rows on the gather ------------
 * The reference runs conv -> nn.BatchNorm1d -> nn.ReLU as three modules (spconv_backbone.py:8-27, modules.py:86-100) and, inside a
 * residual block, conv1 -> bn1 -> relu -> conv2 (spconv_backbone.py:47-68).  Here
 *   fv2p_sparse_conv_rows_bnfin     = the conv that finalises the BatchNorm statistics of its own output: every workgroup stores its
 *                                     tile's column sums as one row of `stats` (a workspace of fv2p_sparse_conv_fin_ws_bytes(n_dst, c_dst)
 *                                     bytes, no initialisation needed), the launch's last workgroups fold the rows in a fixed order
 *                                     (no float atomics: the statistics are bit-identical from run to run) and when the launch ends
 *                                     mean / invstd [c_dst] (and the running statistics, num_batches_tracked) are final and
 *                                     counter[] (fv2p_sparse_conv_fin_counter_words() zeroed device words the caller keeps per stream
 *                                     and direction) zero again - BatchNorm apply kernels
 *                                     (fv2p_batchnorm_apply_res) and consumer convs read 2 - 4 floats per column instead of folding
 *                                     64 slots in every workgroup.  stats == NULL: no statistics.
 *                                     pre_mean / pre_invstd / pre_gamma / pre_beta [c_src] (NULL = off): every gathered source row passes
 *                                     through relu?((v - mean) * invstd * gamma + beta) on its way into the MFMAs - the producer's
 *                                     BatchNorm(+ReLU) output is never materialised.  Same operations in the same order as
 *                                     fv2p_batchnorm_apply: results are bit-identical to the conv over materialised rows.
 *                                     Kernels with that form: ask fv2p_sparse_conv_prenorm_supported (else FV2P_EINVAL, nothing launched).
 *   fv2p_sparse_conv_rows_bnbwd_fin = fv2p_sparse_conv_rows_bnbwd whose last workgroup leaves dgamma, dbeta [c_dst] and
 *                                     coef [2][c_dst] = (mean dz, mean dz * xhat; zeros when batch_stats == 0) for fv2p_batchnorm_backward_fin.
 *   fv2p_sparse_conv_wgrad(_pairs)_pre = the weight gradient with the same normalisation of the gathered operand. */
int fv2p_sparse_conv_rows_bnfin(const float* src, int64_t n_src, int c_src, const float* weight, int kvol, const int* tab,
                                int64_t n_dst, int c_dst, int flip_k, int transpose_w, const float* bias, float* dst,
                                double* stats, unsigned* counter, float eps, float momentum, float* running_mean,
                                float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                                const float* pre_mean, const float* pre_invstd, const float* pre_gamma, const float* pre_beta,
                                int pre_relu, fv2p_stream_t stream);
int fv2p_sparse_conv_prenorm_supported(int c_src, int c_dst, int kvol, int64_t n_dst, int flip_k, int transpose_w);
int fv2p_sparse_conv_fin_counter_words(void);
size_t fv2p_sparse_conv_fin_ws_bytes(int64_t n_dst, int c_dst);
int fv2p_sparse_conv_rows_bnbwd_fin(const float* src, int64_t n_src, int c_src, const float* weight, int kvol, const int* tab,
                                    int64_t n_dst, int c_dst, int flip_k, int transpose_w, float* dst, const float* bn_x,
                                    const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                    int relu, double* stats, unsigned* counter, int batch_stats, float* dgamma, float* dbeta,
                                    float* coef, const int* perm, fv2p_stream_t stream);
int fv2p_sparse_conv_wgrad_pre(const float* src, int64_t n_src, int c_src, const float* grad, const int* tab, int64_t n_dst,
                               int c_dst, int kvol, int flip_k, int dense_k, float* dweight, const float* pre_mean,
                               const float* pre_invstd, const float* pre_gamma, const float* pre_beta, int pre_relu, void* ws,
                               size_t ws_bytes, fv2p_stream_t stream);
int fv2p_sparse_conv_wgrad_pairs_pre(const float* src, int64_t n_src, int c_src, const float* grad, int64_t n_grad, int c_dst,
                                     const int* pairs, const int* pair_num, int kvol, int64_t pair_len, int side_src,
                                     float* dweight, const float* pre_mean, const float* pre_invstd, const float* pre_gamma,
                                     const float* pre_beta, int pre_relu, void* ws, size_t ws_bytes, fv2p_stream_t stream);
/* fv2p_sparse_conv_rows with a row order: perm [n_dst] (NULL = identity) says which destination rows share a 64-row
 * tile; every row's sum is unchanged (bit-identical results), only the work per tile changes — a tile visits just the
 * kernel offsets its rows use.  Meant for the backward-data conv of strided layers with fv2p_rulebook_class_perm's
 * order (also accepted by fv2p_sparse_conv_rows_bnbwd). */
int fv2p_sparse_conv_rows_perm(const float* src, int64_t n_src, int c_src, const float* weight, int kvol,
                               const int* tab, int64_t n_dst, int c_dst, int flip_k, int transpose_w,
                               const float* bias, float* dst, const int* perm, fv2p_stream_t stream);
/* dW_k[c_src][c_dst] = sum_r src[tab[k][r],:]^T grad[r,:]   (dweight [K][c_src][c_dst], fully written here).
 *   forward conv's dW : src=features, grad=dOut [n_out,Cout], tab=tab_out, n_dst=n_out.
 * Per-chunk partial tiles go through the workspace and are summed in a fixed order (deterministic, no atomics).
 * dense_k >= 0 names an offset known to pair (almost) every row — the centre of a submanifold conv — which is then split
 * into many short row chunks of its own so that it does not become the straggler of the launch (-1: none). */
size_t fv2p_sparse_conv_wgrad_ws_bytes(int64_t n_dst, int c_src, int c_dst, int kvol);
int fv2p_sparse_conv_wgrad(const float* src, int64_t n_src, int c_src, const float* grad, const int* tab,
                           int64_t n_dst, int c_dst, int kvol, int flip_k, int dense_k, float* dweight, void* ws,
                           size_t ws_bytes, fv2p_stream_t stream);
/* The same weight gradient from the rulebook's compacted pair lists — the reference's indice_pairs [K][2][pair_len]
 * (-1 padded) and indice_pair_num [K] (spconv_ops.h:403-455 walks the same lists; fv2p_rulebook_pairs / _count build
 * them): dW_k = sum_{p < pair_num[k]} src[pairs[k][side_src][p], :]^T grad[pairs[k][1 - side_src][p], :].
 * side_src = 0 for a forward / submanifold conv (src = features indexed by the input side), 1 for an inverse conv.
 * Work is split by pairs, not rows, so dense and sparse offsets cost what their pair counts say. */
size_t fv2p_sparse_conv_wgrad_pairs_ws_bytes(int64_t pair_len, int c_src, int c_dst, int kvol);
int fv2p_sparse_conv_wgrad_pairs(const float* src, int64_t n_src, int c_src, const float* grad, int64_t n_grad,
                                 int c_dst, const int* pairs, const int* pair_num, int kvol, int64_t pair_len,
                                 int side_src, float* dweight, void* ws, size_t ws_bytes, fv2p_stream_t stream);

/* ---- A7: sparse max-pool / neighbour group over the same tables ------------------------------
 * Replace sparse_conv_ext.indice_maxpool_fp32(+backward) (all.cc:52-63 -> pool_ops.h:25-94; output starts
 * at zero so the result is max(0, .)) and indice_group_fp32(+backward) (all.cc:64-71 -> group_ops.h:29-291):
 *   maxpool fwd : out[o,c]   = max(0, max_k in[tab_out[k][o], c])
 *   maxpool bwd : din[i,c]   = sum_k [in[i,c]==out[o,c]] dout[o,c],  o = tab_in[k][i]
 *   group   fwd : out[k,o,:] = in[tab_out[k][o], :] or 0             ([K, n_out, C])
 *   group   bwd : din[i,:]   = sum_k grad[k, tab_in[k][i], :]
 */
int fv2p_sparse_maxpool_fwd(const float* in, int64_t n_in, int c, const int* tab, int kvol, int64_t n_out,
                            int flip_k, float* out, fv2p_stream_t stream);
int fv2p_sparse_maxpool_bwd(const float* in, const float* out, const float* dout, int64_t n_in, int c,
                            const int* tab_in, int kvol, float* din, fv2p_stream_t stream);
int fv2p_sparse_group_fwd(const float* in, int64_t n_in, int c, const int* tab, int kvol, int64_t n_out,
                          int flip_k, float* out, fv2p_stream_t stream);
int fv2p_sparse_group_bwd(const float* grad, int64_t n_out, int c, const int* tab, int kvol, int64_t n_in,
                          int flip_k, float* din, fv2p_stream_t stream);

/* ---- (f).2: bilinear gather of BEV features at key points --------------------------------------------------------
 * Replaces bilinear_interpolate_torch and BEVGridPooling.interpolate_from_bev_features
 * (pcdet/models/backbones_3d/pfe/bev_grid_pooling.py:11-45, 68-83).  bev: [B, C, H, W] (channels_first, the layout the
 * detector holds; transposed once into the workspace) or [B, H, W, C]; x, y [B, n]: pixel coordinates (already divided
 * by voxel size and stride); out [B, n, C].  Corners floor / floor + 1 clamped to the map, weights from the clamped
 * corners, sum order a, b, c, d — the reference's arithmetic, no contraction.  bwd: gradient of the map (zero-filled
 * here, float atomics: unordered like torch's index backward); x, y get no gradient. */
size_t fv2p_bev_interp_ws_bytes(int batch, int c, int h, int w, int channels_first);
int fv2p_bev_interp_fwd(const float* bev, int batch, int c, int h, int w, int channels_first, const float* x,
                        const float* y, int64_t n, float* out, void* ws, size_t ws_bytes, fv2p_stream_t stream);
int fv2p_bev_interp_bwd(const float* grad_out, int batch, int c, int h, int w, int channels_first, const float* x,
                        const float* y, int64_t n, float* grad_bev, void* ws, size_t ws_bytes, fv2p_stream_t stream);

/* ---- (f).2: SparseConvTensor.dense() and its gradient ---------------------------------------------------------
 * Replaces scatter_nd + permute + contiguous (pcdet/ops/spconv/structure.py:5-18, 57-66; consumer HeightCompression,
 * pcdet/models/backbones_2d/map_to_bev/height_compression.py:10-26).  indices [n, 1+ndim] (batch, z, y, x) or
 * (batch, y, x); dense is [B, C, *spatial] (channels_first) or [B, *spatial, C]; it is zero-filled here.
 * Duplicate coordinates: last writer in thread order wins (the reference's index_put is equally unordered). */
int fv2p_sparse_to_dense(const float* features, const int* indices, int64_t n, int c, int ndim, int batch,
                         const int spatial[3], int channels_first, float* dense, fv2p_stream_t stream);
int fv2p_dense_to_sparse(const float* dense, const int* indices, int64_t n, int c, int ndim, int batch,
                         const int spatial[3], int channels_first, float* rows, fv2p_stream_t stream);

/* ---- (f).1: MeanVFE + collate of one voxelised cloud -------------------------------------------------------------
 * feats[v, :] = sum of the zero-padded point slots of voxel v / max(num_points[v], 1)   (vfe/mean_vfe.py:14-31),
 * coords[v, :] = (batch_idx, z, y, x)   (dataset.collate_batch, pcdet/datasets/dataset.py:165-171), for the
 * v < min(*num_voxels, max_voxels) voxels of fv2p_points_to_voxel's output; feats / coords point at this cloud's
 * offset inside the batch tensors.  *num_voxels is read on the device. */
int fv2p_voxel_mean_collate(const float* voxels, const int* coors, const int* num_points, const int* num_voxels,
                            int max_voxels, int max_points, int ndim, int batch_idx, float* feats, int* coords,
                            fv2p_stream_t stream);

/* ---- A16: rotated BEV overlap / IoU, rotated and axis-aligned NMS ----------------------------
 * Replace iou3d_nms_cuda.* (pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17):
 *   boxes_overlap_bev_gpu (iou3d_nms.cpp:49-68,  kernel iou3d_nms_kernel.cu:236-249)
 *   boxes_iou_bev_gpu     (iou3d_nms.cpp:70-88,  kernel :251-265)
 *   nms_gpu / nms_normal_gpu (iou3d_nms.cpp:90-187, kernels :267-372, host greedy loop :121-135)
 *   boxes_iou_bev_cpu     (iou3d_cpu.cpp:232-252) — host pointers, runs on the calling thread.
 * boxes are [n,7] f32 (x, y, z, dx, dy, dz, heading).  fv2p_nms expects boxes already sorted by descending
 * score; keep [n] i64 and num_keep [1] i32 are DEVICE buffers (the greedy pass runs on the GPU); survivors
 * are written in ascending index order, exactly the reference's keep list.  normal!=0 ignores headings.
 */
int fv2p_boxes_overlap_bev(const float* boxes_a, int num_a, const float* boxes_b, int num_b,
                           float* ans_overlap, fv2p_stream_t stream);
int fv2p_boxes_iou_bev(const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* ans_iou,
                       fv2p_stream_t stream);
size_t fv2p_nms_ws_bytes(int n);
int fv2p_nms(const float* boxes, int n, float thresh, int normal, int64_t* keep, int* num_keep, void* ws,
             size_t ws_bytes, fv2p_stream_t stream);
/* Batched, truncated form for proposal layers (roi_head_template.py:46-101 loops `class_agnostic_nms` over the samples and
 * keeps `selected[:NMS_POST_MAXSIZE]`, model_nms_utils.py:16-20): boxes [batch, n, 7] sorted by descending score per sample,
 * keep [batch, keep_stride] i64, num_keep [batch] i32, both on the device.  max_keep > 0 stops each sample's greedy pass
 * at its max_keep-th survivor (identical to the head of the full list); max_keep <= 0 keeps all. */
size_t fv2p_nms_batch_ws_bytes(int batch, int n, int max_keep);
int fv2p_nms_batch(const float* boxes, int batch, int n, float thresh, int normal, int max_keep, int64_t* keep,
                   int keep_stride, int* num_keep, void* ws, size_t ws_bytes, fv2p_stream_t stream);
/* boxes_iou3d_gpu (pcdet/ops/iou3d_nms/iou3d_nms_utils.py:454-491) for a batch in one launch: boxes_a [batch, num_a, 7], boxes_b
 * [batch, num_b, b_stride >= 7] (e.g. gt boxes with their class id), ans_iou [batch, num_a, num_b] = BEV overlap x height overlap
 * / max(vol_a + vol_b - overlap, 1e-6), clamped to [0, 1] — the same float operations in the same order as the per-sample
 * composition of boxes_overlap_bev_gpu with torch ops. */
int fv2p_boxes_iou3d_batch(const float* boxes_a, int batch, int num_a, const float* boxes_b, int num_b, int b_stride, float* ans_iou,
                           fv2p_stream_t stream);
int fv2p_boxes_iou_bev_cpu(const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* ans_iou);

/* ---- (f).4: second-stage target sampling --------------------------------------------------------
 * Replaces ProposalTargetLayer.sample_rois_for_rcnn / subsample_rois
 * (pcdet/models/roi_heads/target_assigner/proposal_target_layer.py:92-217; thresholds of ROI_HEAD.TARGET_CONFIG) for a whole
 * batch in one launch, without nonzero() / .item() round trips.  iou (B,R,G): 3-D IoU of every RoI with every (zero padded)
 * ground-truth box; rois (B,R,7); gt (B,G,gt_w); uniforms (B,R+n) in [0,1): the first R order the foreground set (random
 * permutation), the last n pick with replacement.  Per sample: best box per RoI (first maximum), foreground = overlap >=
 * fg_thresh, hard background = [bg_lo, reg_fg), easy = < bg_lo; min(foreground, fg_quota) foreground RoIs in permuted order
 * (all n picked with replacement when there is no background), the rest hard (floor(rest * hard_ratio), capped) then easy
 * background, with replacement.  Outputs s_rois (B,n,7), s_gt (B,n,gt_w), s_iou (B,n), s_index (B,n) i32. */
int fv2p_roi_sample_targets(const float* iou, const float* rois, const float* gt, const float* uniforms, int batch, int r, int g, int n,
                            int gt_w, float fg_thresh, float bg_lo, float reg_fg, int fg_quota, float hard_ratio, float* s_rois,
                            float* s_gt, float* s_iou, int* s_index, fv2p_stream_t stream);

/* First-stage target assignment for a batch (AxisAlignedTargetAssigner.assign_targets_single,
 * pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:66-210, one class): anchor_bev [A,4] and gt_bev
 * [B,G,4] are the nearest-BEV footprints (x0,y0,x1,y1) of anchors [A,7] and gt [B,G,gt_w] (class id at [7], zero rows = padding).
 * labels [B,A] i32: the best box's class where its overlap >= matched_thr or the anchor attains a box's non-zero maximum, 0 below
 * unmatched_thr, -1 in between; reg [B,A,7]: ResidualCoder target of the positive anchors, zero elsewhere. */
size_t fv2p_anchor_assign_ws_bytes(int batch, int g);
int fv2p_anchor_assign(const float* anchor_bev, const float* anchors, int n_anchor, const float* gt_bev, const float* gt, int batch, int g,
                       int gt_w, float matched_thr, float unmatched_thr, int* labels, float* reg, void* ws, size_t ws_bytes,
                       fv2p_stream_t stream);

/* First-stage losses of a batch in one pass (AnchorHeadTemplate.get_cls_layer_loss / get_box_reg_layer_loss,
 * pcdet/models/dense_heads/anchor_head_template.py:98-206; SigmoidFocalClassificationLoss with gamma = 2, WeightedSmoothL1Loss on
 * the sin-difference encoded residuals, two direction bins): cls [B,A,1], box [B,A,7], dirs [B,A,2] logits, labels [B,A] i32 and
 * reg_t [B,A,7] from fv2p_anchor_assign, anchor_rot [A].  loss4 = {total, cls, loc, dir} (already weighted and divided by B);
 * dcls / dbox / ddirs = d total / d logits. */
size_t fv2p_anchor_loss_ws_bytes(int batch, int n_anchor);
int fv2p_anchor_loss(const float* cls, const float* box, const float* dirs, const int* labels, const float* reg_t, const float* anchor_rot,
                     int batch, int n_anchor, float alpha, float beta, float dir_offset, float w_cls, float w_loc, float w_dir, float* loss4,
                     float* dcls, float* dbox, float* ddirs, void* ws, size_t ws_bytes, fv2p_stream_t stream);

/* ---- A15 / A18: point-in-box, RoI-aware voxel pooling, RoI point pooling ------------------------
 * Replace roiaware_pool3d_cuda.{points_in_boxes_gpu, points_in_boxes_cpu, forward, backward}
 * (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:29-177, kernels roiaware_pool3d_kernel.cu:16-359) and
 * roipoint_pool3d_cuda.forward (pcdet/ops/roipoint_pool3d/src/roipoint_pool3d.cpp:22-56, kernels
 * roipoint_pool3d_kernel.cu:16-165).  boxes: [x,y,z,dx,dy,dz,heading]; MARGIN 1e-5 on the GPU paths, 1e-2 in the
 * host path, as in the reference.
 *   points_in_boxes      boxes (B,T,7), pts (B,M,3) -> box_idx_of_points (B,M): first containing box or -1
 *   points_in_boxes_cpu  host pointers; boxes (N,7), pts (M,3) -> pts_indices (N,M) 0/1
 *   roipoint_pool3d      xyz (B,N,3), boxes3d (B,M,7), pts_feature (B,N,C) -> pooled (B,M,S,3+C) = the first S
 *                        inside points in index order, wrapped when fewer; pooled_empty_flag (B,M)
 *   roiaware_pool3d_fwd  rois (R,7), pts (P,3), pts_feature (P,C) -> pts_idx_of_voxels (R,ox,oy,oz,max_pts)
 *                        [slot 0 = count], argmax (R,ox,oy,oz,C), pooled (R,ox,oy,oz,C); pool_method 0 max / 1 avg
 *   roiaware_pool3d_bwd  grad_in (P,C) must be zeroed by the caller (roiaware_pool3d_utils.py:104), accumulated.
 */
int fv2p_points_in_boxes(const float* boxes, const float* pts, int batch, int boxes_num, int pts_num,
                         int* box_idx_of_points, fv2p_stream_t stream);
int fv2p_points_in_boxes_cpu(const float* boxes, const float* pts, int boxes_num, int pts_num, int* pts_indices);
int fv2p_roipoint_pool3d(const float* xyz, const float* boxes3d, const float* pts_feature, int batch, int pts_num,
                         int boxes_num, int feature_len, int sampled_pts_num, float* pooled_features,
                         int* pooled_empty_flag, fv2p_stream_t stream);
int fv2p_roiaware_pool3d_fwd(const float* rois, const float* pts, const float* pts_feature, int boxes_num,
                             int pts_num, int channels, int max_pts_each_voxel, int out_x, int out_y, int out_z,
                             int pool_method, int* argmax, int* pts_idx_of_voxels, float* pooled_features,
                             fv2p_stream_t stream);
int fv2p_roiaware_pool3d_bwd(const int* pts_idx_of_voxels, const int* argmax, const float* grad_out, int boxes_num,
                             int out_x, int out_y, int out_z, int channels, int max_pts_each_voxel, int pool_method,
                             float* grad_in, fv2p_stream_t stream);

/* ---- A8-A12: pointnet2 (batch and stacked layouts) ------------------------------------------------
 * Replace pointnet2_batch_cuda.* (pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:10-24) and
 * pointnet2_stack_cuda.* (pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:11-23).  Argument order and
 * meaning follow the reference wrappers (ints first, then tensors); outputs are caller-allocated.
 *  batch layouts : xyz (B,N,3), features channel-major (B,C,N), idx (B,M,nsample) / (B,M) / (B,N,3)
 *  stack layouts : rows of all samples concatenated, per-sample row counts in *_batch_cnt (int32 [B], device)
 * ball_query      first `nsample` points with d2 <  r^2 in index order, padded with the first hit
 *                 (ball_query_gpu.cu:15-51; stack: idx[0] = -1 when the ball is empty, :16-66); idx pre-zeroed by caller
 * voxel_query     (2r+1)^3 neighbourhood of a dense (B,Z,Y,X) int volume, d2 <= r^2 (voxel_query_gpu.cu:10-89)
 * furthest_point_sampling  temp (B,N) must hold 1e10 (pointnet2_utils.py:26); first index 0; ties as the reference's
 *                 strided tree reduction (sampling_gpu.cu:93-216); temp holds the final distances on return
 * three_nn        3 smallest squared distances (strict <, lowest index wins) + indices (stack: global rows)
 * *_grad          accumulate into grad buffers the caller zeroed (atomic adds, as the reference)
 */
int fv2p_ball_query_batch(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz,
                          int* idx, fv2p_stream_t stream);
int fv2p_ball_query_stack(int b, int m, float radius, int nsample, const float* new_xyz,
                          const int* new_xyz_batch_cnt, const float* xyz, const int* xyz_batch_cnt, int* idx,
                          fv2p_stream_t stream);
int fv2p_voxel_query_stack(int m, int r1, int r2, int r3, int nsample, float radius, int z_range, int y_range,
                           int x_range, const float* new_xyz, const float* xyz, const int* new_coords,
                           const int* point_indices, int* idx, fv2p_stream_t stream);
int fv2p_group_points_batch(int b, int c, int n, int npoints, int nsample, const float* points, const int* idx,
                            float* out, fv2p_stream_t stream);
int fv2p_group_points_batch_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out,
                                 const int* idx, float* grad_points, fv2p_stream_t stream);
int fv2p_gather_points(int b, int c, int n, int npoints, const float* points, const int* idx, float* out,
                       fv2p_stream_t stream);
int fv2p_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out, const int* idx,
                            float* grad_points, fv2p_stream_t stream);
int fv2p_group_points_stack(int b, int m, int c, int nsample, const float* features,
                            const int* features_batch_cnt, const int* idx, const int* idx_batch_cnt, float* out,
                            fv2p_stream_t stream);
int fv2p_group_points_stack_grad(int b, int m, int c, int n, int nsample, const float* grad_out, const int* idx,
                                 const int* idx_batch_cnt, const int* features_batch_cnt, float* grad_features,
                                 fv2p_stream_t stream);
/* ws / ws_bytes (fv2p_furthest_point_sampling_ws_bytes, may be NULL): scratch of the bucketed kernel — Morton-sorted
 * point order, per-bucket boxes; with it 2048 <= n <= 16384, m >= 1024 run the lazy variant, whose indices and final
 * `temp` are bit-identical to the plain kernel's (and the reference's, sampling_gpu.cu:100-216). */
size_t fv2p_furthest_point_sampling_ws_bytes(int b, int n);
int fv2p_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp, int* idxs, void* ws,
                                 size_t ws_bytes, fv2p_stream_t stream);
/* Profiling hook of the streaming sampler (n > 24 576): when non-NULL (device memory, 16 x 8 entries), sample 0 writes
 * trace[8 * wave + {0..7}] = shader clocks spent in {box test, issuing the loads of a pass over the touched buckets, the first
 * bucket of a pass (wait + distance pass + reduction), the pass's other buckets, wave arg-max, candidate exchange + barrier,
 * winner selection}, the touched buckets summed over the rounds.  NULL switches it off. */
int fv2p_fps_set_trace(unsigned long long* trace);
int fv2p_three_nn_batch(int b, int n, int m, const float* unknown, const float* known, float* dist2, int* idx,
                        fv2p_stream_t stream);
int fv2p_three_nn_stack(int b, int n, int m, const float* unknown, const int* unknown_batch_cnt,
                        const float* known, const int* known_batch_cnt, float* dist2, int* idx,
                        fv2p_stream_t stream);
/* fv2p_three_nn_stack through a hashed uniform grid over the known points (2 table slots per point): the same idx / dist2, bit
 * for bit.  Every distance is the scan's float expression, the three best are kept under the (distance, index) order = "strict <
 * over ascending index", and a query is settled by the 27 cells around its own only when every point outside them is provably
 * farther than its third best; the other queries scan their sample, one wave each.  Time goes with the points near each query
 * instead of all of them.  cell: grid spacing in the points' unit (e.g. two voxel pitches of the level the known points are
 * centres of); <= 0 lets the library estimate one.  n = queries, m = known points.  The reference has no counterpart
 * (interpolate_gpu.cu:16-73 scans). */
size_t fv2p_three_nn_grid_ws_bytes(int b, int64_t n, int64_t m);
int fv2p_three_nn_stack_grid(int b, int n, int m, const float* unknown, const int* unknown_batch_cnt,
                             const float* known, const int* known_batch_cnt, float cell, float* dist2, int* idx,
                             void* ws, size_t ws_bytes, fv2p_stream_t stream);
int fv2p_three_interpolate_batch(int b, int c, int m, int n, const float* points, const int* idx,
                                 const float* weight, float* out, fv2p_stream_t stream);
int fv2p_three_interpolate_batch_grad(int b, int c, int n, int m, const float* grad_out, const int* idx,
                                      const float* weight, float* grad_points, fv2p_stream_t stream);
int fv2p_three_interpolate_stack(int n, int c, const float* features, const int* idx, const float* weight,
                                 float* out, fv2p_stream_t stream);
int fv2p_three_interpolate_stack_grad(int n, int c, const float* grad_out, const int* idx, const float* weight,
                                      float* grad_features, fv2p_stream_t stream);
/* The same gradient without float atomics, without the caller's zero fill and in a FIXED order: the entries e = 3 * query + slot of
 * idx [n, 3] are keyed (known row, e) and radix-sorted in the workspace; the sorted sequence is summed in segments of 32 entries (a lane
 * group each, one lane = four channels), runs that cross segment borders are closed from the segments' partial sums in segment order.
 * Bit-identical from run to run; values equal fv2p_three_interpolate_stack_grad's up to the association of each row's sum (the scatter
 * form adds a row's entries in whatever order its atomics land). */
size_t fv2p_three_interpolate_stack_grad_ws_bytes(int n, int c, int m);
int fv2p_three_interpolate_stack_grad_gather(int n, int c, int m, const float* grad_out, const int* idx, const float* weight,
                                             float* grad_features, void* ws, size_t ws_bytes, fv2p_stream_t stream);

/* ---- A11 consumer: fused grid set-abstraction (gather -> shared-MLP layer -> max over the samples) ---------------------
 * The part of PointnetSAModuleMSG.forward (pointnet2_batch/pointnet2_modules.py:30-62) that follows the ball query, for the
 * bn=False module of IoUGuidedRoIHead (iouguided_roi_head.py:52-76) after its first, linear layer has been applied per point
 * and per centre:  out[r, i, :] = max_s relu(W2 relu(per_point[r, idx[r, i, s], :] - per_centre[r, i, :])).
 * per_point [rois, n, c], per_centre [rois, m, c], idx [rois, m, s] i32 (ball query output), w2 [c, c] (Conv2d weight),
 * out / grad_out [rois, m, c]; c = 64, s in {16, 32}.  No grouped tensor exists in either direction.
 * arg [rois, m, c] u8 (forward: optional, may be NULL for inference; backward: required): the sample 0 .. s-1 that attained the
 * maximum of (centre, channel) — the first one in sample order, which is where F.max_pool2d sends the gradient
 * (pointnet2_modules.py:57-59) — or 255 where the maximum is 0 (ReLU inactive for every sample: no gradient).  Backward
 * recomputes relu(per_point - per_centre) only. */
int fv2p_sa_grid_supported(int n, int m, int s, int c);
int fv2p_sa_grid_fwd(const float* per_point, const float* per_centre, const int* idx, const float* w2, int rois, int n, int m,
                     int s, int c, float* out, unsigned char* arg, fv2p_stream_t stream);
size_t fv2p_sa_grid_bwd_ws_bytes(int rois);
int fv2p_sa_grid_bwd(const float* per_point, const float* per_centre, const int* idx, const float* w2, const unsigned char* arg,
                     const float* grad_out, int rois, int n, int m, int s, int c, float* grad_point, float* grad_centre,
                     float* grad_w2, void* ws, size_t ws_bytes, fv2p_stream_t stream);

/* ---- A13: modulated deformable convolution (DCNv2; DCNv1 = mask of ones) -------------------------
 * Replace DCN.modulated_deform_conv_forward / _backward and DCN.deform_conv_forward / _backward
 * (pcdet/ops/DeformableConvolutionV2PyTorch/src/vision.cpp:6-12 -> src/modulated_deform_conv.h:10-86 ->
 * src/cuda/modulated_deform_conv_cuda.cu:19-280, kernels src/cuda/modulated_deform_im2col_cuda.cuh:24-328).
 * Forward: fused implicit GEMM, no `columns` buffer.  Backward: the column GRADIENTS do pass through the workspace (below), at
 * most 1.5 GiB of them at a time.  Activations are NHWC on this side of the boundary:
 *   x_nhwc [B,H,W,Cin], y_nhwc / dy_nhwc [B*Ho*Wo, Cout]; the weight [Cout,Cin,kh,kw] arrives permuted: forward takes
 *   wt_oc = [kh*kw][Cout][Cin] (input channels contiguous), backward wt = [kh*kw][Cin][Cout] (output channels contiguous);
 *   offset [B, dg*2*kh*kw, Ho, Wo] ((2k, 2k+1) = (dh, dw)), mask [B, dg*kh*kw, Ho, Wo] — reference layouts.
 * groups == 1; Cin / deformable_group must be a multiple of 16; Cout <= 256 (backward: Cout a multiple of 4).
 * backward (modulated_deform_conv_cuda.cu:127-280): dx_nhwc, doffset, dmask, dwt are FULLY written, nothing to zero
 * (dwt [kh*kw][Cin][Cout]; the bias gradient is a plain column sum done by the caller).  No float atomics: the column
 * gradients go through the workspace ([B*Ho*Wo][kh*kw][Cin], the reference's `columns`), every input pixel then sums the
 * samples that touch it in ascending sample order (lists built with integer atomics), so all four gradients are
 * bit-identical from run to run (lists above 4096 samples on ONE input pixel keep an arbitrary order).
 * Any batch: both entry points cut the call into chunks of whole samples (the reference's im2col_step loop,
 * modulated_deform_conv_cuda.cu:85-118) so that x, y, offset and the column gradients of a chunk stay below the kernels' 32-bit
 * limits and the column gradients below 1.5 GiB; fv2p_dcn_backward_ws_bytes is the workspace of ONE chunk.  dx, doffset, dmask
 * are per sample (identical for every chunking); dwt adds the chunks in ascending order (fixed, run-to-run identical).  Only a
 * single sample above a limit is refused (FV2P_ELIMIT).
 */
int fv2p_dcn_forward(const float* x_nhwc, const float* wt_oc, const float* bias, const float* offset,
                     const float* mask, int batch, int height, int width, int c_in, int c_out, int h_out,
                     int w_out, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                     int deformable_group, float* y_nhwc, fv2p_stream_t stream);
/* Step order of the forward kernel: 0 = (tap, group, 16-channel chunk) - rounds 4 / 5 -, 1 = (group, chunk, tap): the nine taps of a chunk
 * back to back (the chunk's 64-byte segments stay in L1 / L2 across the taps), -1 = the library's choice.  Results are identical
 * term by term only within one order (the sum over (tap, channel) terms is taken in step order); both are within 1e-4 of the oracle. */
int fv2p_dcn_set_forward_order(int order);
/* Test hook: the cap on a chunk's column gradients in bytes (0 = back to the 1.5 GiB default). */
int fv2p_dcn_set_colg_cap(int64_t bytes);
size_t fv2p_dcn_backward_ws_bytes(int batch, int height, int width, int h_out, int w_out, int c_in, int c_out, int kh,
                                  int kw, int deformable_group);
int fv2p_dcn_backward(const float* x_nhwc, const float* wt, const float* offset, const float* mask,
                      const float* dy_nhwc, int batch, int height, int width, int c_in, int c_out, int h_out,
                      int w_out, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                      int deformable_group, float* dx_nhwc, float* doffset, float* dmask, float* dwt, void* ws,
                      size_t ws_bytes, fv2p_stream_t stream);

/* Layout copies around the NHWC entry points above (and fv2p_bev_interp_*): in [batch][rows][cols] -> out [batch][cols][rows],
 * e.g. NCHW -> NHWC with rows = C, cols = H*W.  The reference does the same copies with at::permute + contiguous
 * (modulated_deform_conv_cuda.cu:78,118). */
int fv2p_transpose_batched(const float* in, int batch, int64_t rows, int64_t cols, float* out, fv2p_stream_t stream);

/* ---- A14: deformable position-sensitive RoI pooling ----------------------------------------------
 * Replace DCN.deform_psroi_pooling_forward / _backward
 * (pcdet/ops/DeformableConvolutionV2PyTorch/src/vision.cpp:11-12 -> src/deform_psroi_pooling.h ->
 * src/cuda/deform_psroi_pooling_cuda.cu:264-418; kernels :59-147, :149-262; caller functions/deform_psroi_pooling_func.py:15-64).
 * Reference layouts: data / grad_data [B, C, H, W]; rois [R, 5] = (batch index, x1, y1, x2, y2) in image pixels;
 * trans / grad_trans [>=R, 2*num_classes, part, part] (NULL with no_trans, num_classes then counts as 1);
 * out / top_count / grad_out [R, output_dim, pooled, pooled] (top_count = samples that fell on the map, as float).
 * C must equal output_dim * group_size^2 (the reference asserts C == output_dim and reads beyond the map for
 * group_size > 1).  An RoI whose batch index is outside [0, B) pools to zeros with count 0.
 * backward: grad_data and grad_trans must be zeroed by the caller (accumulated with atomics). */
int fv2p_deform_psroi_pool_forward(const float* data, const float* rois, const float* trans, int batch, int channels,
                                   int height, int width, int num_rois, int no_trans, float spatial_scale,
                                   int output_dim, int group_size, int pooled_size, int part_size,
                                   int sample_per_part, float trans_std, int num_classes, float* out,
                                   float* top_count, fv2p_stream_t stream);
int fv2p_deform_psroi_pool_backward(const float* grad_out, const float* data, const float* rois, const float* trans,
                                    const float* top_count, int batch, int channels, int height, int width,
                                    int num_rois, int no_trans, float spatial_scale, int output_dim,
                                    int group_size, int pooled_size, int part_size, int sample_per_part,
                                    float trans_std, int num_classes, float* grad_data, float* grad_trans,
                                    fv2p_stream_t stream);

/* ---- (f).3: BatchNorm1d (+ReLU) over sparse-tensor features [N, C] --------------------------------
 * The pair every conv of the reference backbones is followed by: nn.BatchNorm1d(eps=1e-3, momentum=0.01) + nn.ReLU
 * (pcdet/models/backbones_3d/spconv_backbone.py:8-27, :75), applied to SparseConvTensor.features by
 * SparseSequential (pcdet/ops/spconv/modules.py:86-100).  torch semantics: biased batch variance for the
 * normalisation, unbiased for running_var; running = (1 - momentum) * running + momentum * batch; momentum < 0
 * means momentum=None (cumulative average over num_batches_tracked).  Sums are accumulated in fp64 and folded in a
 * fixed order (deterministic).
 *   fv2p_batchnorm_forward  : training-mode layer: batch mean / invstd of x[n,c] (also returned for the backward
 *                             pass), running_mean / running_var / num_batches_tracked updated in place when
 *                             running_mean != NULL, y = relu?((x - mean) * invstd * gamma + beta).
 *   fv2p_batchnorm_apply    : the same normalisation with given mean / invstd (eval mode: running statistics).
 *                             gamma / beta may be NULL; y may alias x.
 *   fv2p_batchnorm_backward : dz = dy * [y > 0] (mask recomputed from x); dgamma = sum dz * xhat, dbeta = sum dz,
 *                             dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat))  when batch_stats != 0,
 *                             dx = gamma * invstd * dz                                          otherwise (eval mode).
 * c <= 1024 (c % 4 == 0) or c <= 256.
 */
size_t fv2p_batchnorm_ws_bytes(int64_t n, int c);
int fv2p_batchnorm_forward(const float* x, int64_t n, int c, float eps, float momentum, const float* gamma,
                           const float* beta, int relu, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, float* mean, float* invstd, float* y, void* ws,
                           size_t ws_bytes, fv2p_stream_t stream);
/* fv2p_batchnorm_forward with the sums already taken (stats as left by fv2p_sparse_conv_rows_stats): one launch.
 * zero_next: NULL, or a second stats buffer whose first zero_count doubles are cleared for the next fused conv on this
 * stream (two buffers alternate: a launch reads one and clears what the other's last user left). */
int fv2p_batchnorm_forward_stats(const float* x, int64_t n, int c, float eps, float momentum, const float* gamma,
                                 const float* beta, int relu, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, float* mean, float* invstd, float* y,
                                 const double* stats, double* zero_next, int64_t zero_count, fv2p_stream_t stream);
/* fv2p_batchnorm_backward with (sum dz, sum dz * xhat) already in `stats` (fv2p_sparse_conv_rows_bnbwd): one launch. */
int fv2p_batchnorm_backward_stats(const float* x, const float* dy, int64_t n, int c, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta, int relu,
                                  int batch_stats, float* dx, float* dgamma, float* dbeta, const double* stats,
                                  double* zero_next, int64_t zero_count, fv2p_stream_t stream);
/* round 6: fv2p_batchnorm_apply with an optional residual term, y = relu?(bn(x) + residual) - the tail of a residual block
 * (spconv_backbone.py:63-66: out.features += identity; relu) in the normalisation's own pass; fv2p_batchnorm_backward_fin = the apply
 * half of the backward with c1 / c2 (coef) already finalised by fv2p_sparse_conv_rows_bnbwd_fin; fv2p_batchnorm_backward_res = backward
 * of out = relu(bn(x) + identity): the ReLU mask is read from `out`, dz = dout * [out > 0] (the identity branch's gradient) is written
 * beside dx. */
int fv2p_batchnorm_apply_res(const float* x, int64_t n, int c, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, int relu, const float* residual, float* y, fv2p_stream_t stream);
int fv2p_batchnorm_backward_fin(const float* x, const float* dy, int64_t n, int c, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, int relu, const float* coef, float* dx,
                                fv2p_stream_t stream);
int fv2p_batchnorm_backward_res(const float* x, const float* out, const float* dout, int64_t n, int c, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, int batch_stats, float* dx,
                                float* dz, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, fv2p_stream_t stream);
/* round 6: both passes of a BatchNorm whose sums no conv epilogue takes, as ONE launch each (reduce, grid barrier over <= 128 resident
 * workgroups, apply).  counters: two zeroed device words kept by the caller per stream (zero again when the launch ends); ws as
 * fv2p_batchnorm_one_ws_bytes(c).  residual / mask_y / dz_out as in fv2p_batchnorm_apply_res / _backward_res (NULL = plain layer). */
size_t fv2p_batchnorm_one_ws_bytes(int c);
int fv2p_batchnorm_one_pays(int64_t n, int c, int backward);   /* 1: the one-launch pass is the faster one at this size (measured: tools/bn_time.py) */
int fv2p_batchnorm_forward_one(const float* x, int64_t n, int c, float eps, float momentum, const float* gamma, const float* beta,
                               int relu, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean,
                               float* invstd, const float* residual, float* y, void* ws, size_t ws_bytes, unsigned* counters,
                               fv2p_stream_t stream);
int fv2p_batchnorm_backward_one(const float* x, const float* dy, int64_t n, int c, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, int relu, int batch_stats, const float* mask_y, float* dx,
                                float* dz_out, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, unsigned* counters,
                                fv2p_stream_t stream);
/* round 6: the two-launch passes for large tensors (the decoder's 49 152-row layers) with a WIDE reduce: up to 512 workgroups publish
 * their partial sums as rows, the reduce launch's last workgroups fold them in a fixed order (the protocol of
 * fv2p_sparse_conv_rows_bnfin) and the apply launch reads finished mean / invstd or c1 / c2.  counters:
 * fv2p_batchnorm_wide_counter_words(c) zeroed device words kept by the caller per stream. */
size_t fv2p_batchnorm_wide_ws_bytes(int c);
int fv2p_batchnorm_wide_counter_words(int c);
int fv2p_batchnorm_forward_wide(const float* x, int64_t n, int c, float eps, float momentum, const float* gamma, const float* beta,
                                int relu, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean,
                                float* invstd, const float* residual, float* y, void* ws, size_t ws_bytes, unsigned* counters,
                                fv2p_stream_t stream);
int fv2p_batchnorm_backward_wide(const float* x, const float* dy, int64_t n, int c, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int relu, int batch_stats, const float* mask_y, float* dx,
                                 float* dz_out, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, unsigned* counters,
                                 fv2p_stream_t stream);
int fv2p_batchnorm_apply(const float* x, int64_t n, int c, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, int relu, float* y, fv2p_stream_t stream);
int fv2p_batchnorm_backward(const float* x, const float* dy, int64_t n, int c, const float* mean,
                            const float* invstd, const float* gamma, const float* beta, int relu, int batch_stats,
                            float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                            fv2p_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FV2P_OPS_H_ */
