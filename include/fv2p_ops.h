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
 *                   indice_num [K]      pairs per offset (the reference's indiceNum).
 * begin and finish must be given the same workspace (>= fv2p_rulebook_ws_bytes) and arguments.
 * Kernel offset index k = (kz*Ky + ky)*Kx + kx with k_j = in_j - out_j*s_j + p_j (geometry.h:62-73).
 */
size_t fv2p_rulebook_ws_bytes(int64_t n_in, const int ksize[3], const int stride[3], const int dilation[3],
                              int subm, int transpose);
int fv2p_rulebook_begin(const int* indices, int64_t n_in, int batch, const int in_shape[3],
                        const int out_shape[3], const int ksize[3], const int stride[3], const int padding[3],
                        const int dilation[3], int subm, int transpose, int64_t* n_out_host, void* ws,
                        size_t ws_bytes, fv2p_stream_t stream);
int fv2p_rulebook_finish(const int* indices, int64_t n_in, int batch, const int in_shape[3],
                         const int out_shape[3], const int ksize[3], const int stride[3], const int padding[3],
                         const int dilation[3], int subm, int transpose, int64_t n_out, int* out_indices,
                         int* tab_in, int* tab_out, int* indice_num, void* ws, size_t ws_bytes,
                         fv2p_stream_t stream);
/* Reference-format pair lists indicePairs [K,2,n_in] (-1 padded, spconv_ops.h:55-57) from tab_in;
 * within one offset pairs are ordered by ascending input row (the CPU reference's order,
 * geometry.h:281-295; the GPU reference's slot order is atomic-order). */
size_t fv2p_rulebook_pairs_ws_bytes(int64_t n_in, int kvol);
int fv2p_rulebook_pairs(const int* tab_in, int64_t n_in, int kvol, int* pairs, void* ws, size_t ws_bytes,
                        fv2p_stream_t stream);
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
 * flip_k!=0 reads table row K-1-k for offset k (subm symmetry, lets subm reuse tab_in as tab_out).
 * fp32 in / fp32 accumulate on v_mfma_f32_16x16x4_f32; deterministic (fixed k order, no atomics).
 */
int fv2p_sparse_conv_rows(const float* src, int64_t n_src, int c_src, const float* weight, int kvol,
                          const int* tab, int64_t n_dst, int c_dst, int flip_k, int transpose_w,
                          const float* bias, float* dst, fv2p_stream_t stream);
/* dW_k[c_src][c_dst] = sum_r src[tab[k][r],:]^T grad[r,:]   (dweight [K][c_src][c_dst], zeroed here).
 *   forward conv's dW : src=features, grad=dOut [n_out,Cout], tab=tab_out, n_dst=n_out. */
int fv2p_sparse_conv_wgrad(const float* src, int64_t n_src, int c_src, const float* grad, const int* tab,
                           int64_t n_dst, int c_dst, int kvol, int flip_k, float* dweight,
                           fv2p_stream_t stream);

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

#ifdef __cplusplus
}
#endif
#endif /* FV2P_OPS_H_ */
