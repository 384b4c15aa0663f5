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

#ifdef __cplusplus
}
#endif
#endif /* FV2P_OPS_H_ */
