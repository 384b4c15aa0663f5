/*
 * ORACLE — test infrastructure only.  Not part of the product: only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library.
 *
 * CPU restatement of the reference voxeliser,
 *   pcdet/datasets/processor/voxel_generator.py:75-133 (points_to_voxel: buffers, dense
 *   coor_to_voxelidx map filled with -1) and :136-207 (_points_to_voxel_reverse_kernel).
 * Parity pin: tests/golden/voxel_*.npz were produced by importing that reference file in the
 * build container (oracle/gen_golden_voxel.py) and this restatement reproduces them bit-exactly
 * (tests/test_oracle_golden.py).
 *
 * Arithmetic mirrors the numba/numpy float32 scalar ops: fp32 subtract, fp32 divide, floor;
 * comparisons against the int grid size; build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Returns voxel_num.  voxels/coors/num_points_per_voxel must be zero-initialised by the caller
 * (voxel_generator.py:113-117).  use_dense_map=1 reproduces the reference's dense int32 map
 * ([Z,Y,X], alloc + fill with -1 on every call, :114); 0 uses the caller-supplied scratch map
 * that is assumed to be -1 everywhere and is restored before returning (for timing the loop
 * without the fill). */
int oracle_points_to_voxel(const float* points, int64_t n, int ndim, const float* voxel_size,
                           const float* coors_range, int max_points, int max_voxels, float* voxels,
                           int32_t* coors, int32_t* num_points_per_voxel, int32_t* scratch_map) {
  int32_t grid_size[3];
  for (int j = 0; j < 3; ++j) {
    float g = (coors_range[3 + j] - coors_range[j]) / voxel_size[j];
    grid_size[j] = (int32_t)rintf(g); /* np.round == round-half-even */
  }
  /* reversed map shape: [gz, gy, gx] */
  const int64_t vol = (int64_t)grid_size[0] * grid_size[1] * grid_size[2];
  int32_t* map = scratch_map;
  if (!map) {
    map = (int32_t*)malloc(sizeof(int32_t) * (size_t)vol);
    if (!map) return -1;
    memset(map, 0xFF, sizeof(int32_t) * (size_t)vol); /* -np.ones(...) */
  }
  int32_t coor[3];
  int voxel_num = 0;
  for (int64_t i = 0; i < n; ++i) {
    int failed = 0;
    for (int j = 0; j < 3; ++j) {
      float c = floorf((points[i * ndim + j] - coors_range[j]) / voxel_size[j]);
      if (c < 0 || c >= (float)grid_size[j]) { failed = 1; break; }
      coor[2 - j] = (int32_t)c;
    }
    if (failed) continue;
    const int64_t m = ((int64_t)coor[0] * grid_size[1] + coor[1]) * grid_size[0] + coor[2];
    int32_t voxelidx = map[m];
    if (voxelidx == -1) {
      voxelidx = voxel_num;
      if (voxel_num >= max_voxels) break;
      voxel_num += 1;
      map[m] = voxelidx;
      coors[voxelidx * 3 + 0] = coor[0];
      coors[voxelidx * 3 + 1] = coor[1];
      coors[voxelidx * 3 + 2] = coor[2];
    }
    int32_t num = num_points_per_voxel[voxelidx];
    if (num < max_points) {
      memcpy(voxels + ((int64_t)voxelidx * max_points + num) * ndim, points + i * ndim, sizeof(float) * ndim);
      num_points_per_voxel[voxelidx] += 1;
    }
  }
  if (!scratch_map) {
    free(map);
  } else {
    for (int v = 0; v < voxel_num; ++v) {
      const int64_t m = ((int64_t)coors[v * 3] * grid_size[1] + coors[v * 3 + 1]) * grid_size[0] + coors[v * 3 + 2];
      map[m] = -1;
    }
  }
  return voxel_num;
}
