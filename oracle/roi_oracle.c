/*
 * ORACLE — test infrastructure only (tests/, bench.py cpu_baseline, __graft_entry__.smoke()).
 *
 * CPU restatement of the reference point-in-box / RoI pooling ops:
 *   check_pt_in_box3d(+_cpu)      pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-36 (MARGIN 1e-5),
 *                                  roiaware_pool3d.cpp:121-140 (MARGIN 1e-2)
 *   points_in_boxes_kernel        roiaware_pool3d_kernel.cu:313-336        points_in_boxes_cpu  roiaware_pool3d.cpp:143-168
 *   generate_pts_mask / collect   roiaware_pool3d_kernel.cu:39-108         max/avg pool + backward  :111-286
 *   roipoint assign / pooled idx / forward   pcdet/ops/roipoint_pool3d/src/roipoint_pool3d_kernel.cu:38-130
 * Parity pin: UNPINNED by reference execution (GPU kernels; the .cpp needs THC/CUDA headers absent here; the
 * reference ships no vectors).  Anchors: geometric properties in tests/test_roi_*.py (axis-aligned boxes vs
 * closed-form interval tests, margin behaviour, first-box-wins, order preservation).
 * sin/cos: include/fv2p_math.h (deterministic fp32, shared with the device code).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/fv2p_math.h"

static int o_check_pt_in_box3d(const float* pt, const float* box3d, float MARGIN, float* local_x, float* local_y) {
  float x = pt[0], y = pt[1], z = pt[2];
  float cx = box3d[0], cy = box3d[1], cz = box3d[2];
  float dx = box3d[3], dy = box3d[4], dz = box3d[5], rz = box3d[6];
  if (fabsf(z - cz) > dz / 2.0) return 0;
  float shift_x = x - cx, shift_y = y - cy;
  float cosa = fv2p_cosf(-rz), sina = fv2p_sinf(-rz);
  *local_x = shift_x * cosa + shift_y * (-sina);
  *local_y = shift_x * sina + shift_y * cosa;
  float in_flag = (fabsf(*local_x) < dx / 2.0 + MARGIN) & (fabsf(*local_y) < dy / 2.0 + MARGIN);
  return (int)in_flag;
}

void oracle_points_in_boxes_gpu(const float* boxes, const float* pts, int batch, int boxes_num, int pts_num, int32_t* out) {
  float lx, ly;
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < pts_num; ++i) {
      int32_t idx = -1;
      for (int k = 0; k < boxes_num; ++k)
        if (o_check_pt_in_box3d(pts + ((int64_t)b * pts_num + i) * 3, boxes + ((int64_t)b * boxes_num + k) * 7, 1e-5f, &lx, &ly)) { idx = k; break; }
      out[(int64_t)b * pts_num + i] = idx;
    }
}

void oracle_points_in_boxes_cpu(const float* boxes, const float* pts, int boxes_num, int pts_num, int32_t* out) {
  float lx, ly;
  for (int i = 0; i < boxes_num; ++i)
    for (int j = 0; j < pts_num; ++j) out[(int64_t)i * pts_num + j] = o_check_pt_in_box3d(pts + j * 3, boxes + i * 7, 1e-2f, &lx, &ly);
}

void oracle_roipoint_pool3d(const float* xyz, const float* boxes3d, const float* feat, int batch, int pts_num, int boxes_num, int c,
                            int sampled, float* pooled, int32_t* empty_flag) {
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)sampled);
  float lx, ly;
  for (int b = 0; b < batch; ++b)
    for (int m = 0; m < boxes_num; ++m) {
      int cnt = 0;
      for (int k = 0; k < pts_num; ++k)
        if (o_check_pt_in_box3d(xyz + ((int64_t)b * pts_num + k) * 3, boxes3d + ((int64_t)b * boxes_num + m) * 7, 1e-5f, &lx, &ly)) {
          if (cnt < sampled) idx[cnt++] = k; else break;
        }
      empty_flag[(int64_t)b * boxes_num + m] = (cnt == 0);
      if (cnt == 0) continue;
      for (int k = cnt; k < sampled; ++k) idx[k] = idx[k % cnt];
      for (int s = 0; s < sampled; ++s) {
        float* dst = pooled + (((int64_t)b * boxes_num + m) * sampled + s) * (3 + c);
        memcpy(dst, xyz + ((int64_t)b * pts_num + idx[s]) * 3, sizeof(float) * 3);
        memcpy(dst + 3, feat + ((int64_t)b * pts_num + idx[s]) * c, sizeof(float) * c);
      }
    }
  free(idx);
}

/* forward of RoI-aware pooling; outputs must be zero-initialised by the caller (roiaware_pool3d_utils.py:84-86) */
void oracle_roiaware_pool3d(const float* rois, const float* pts, const float* feat, int boxes_num, int pts_num, int channels, int max_pts,
                            int out_x, int out_y, int out_z, int pool_method, int32_t* argmax, int32_t* pts_idx_of_voxels, float* pooled) {
  const int max_num = max_pts - 1;
  for (int bi = 0; bi < boxes_num; ++bi) {
    const float* roi = rois + bi * 7;
    int32_t* vox = pts_idx_of_voxels + (int64_t)bi * out_x * out_y * out_z * max_pts;
    for (int k = 0; k < pts_num; ++k) {
      float local_x = 0, local_y = 0;
      if (!o_check_pt_in_box3d(pts + k * 3, roi, 1e-5f, &local_x, &local_y)) continue;
      float local_z = pts[k * 3 + 2] - roi[2];
      float dx = roi[3], dy = roi[4], dz = roi[5];
      float x_res = dx / out_x, y_res = dy / out_y, z_res = dz / out_z;
      int x_idx = (int)((local_x + dx / 2) / x_res);
      int y_idx = (int)((local_y + dy / 2) / y_res);
      int z_idx = (int)((local_z + dz / 2) / z_res);
      x_idx = x_idx < 0 ? 0 : (x_idx > out_x - 1 ? out_x - 1 : x_idx);
      y_idx = y_idx < 0 ? 0 : (y_idx > out_y - 1 ? out_y - 1 : y_idx);
      z_idx = z_idx < 0 ? 0 : (z_idx > out_z - 1 ? out_z - 1 : z_idx);
      int64_t base = ((int64_t)x_idx * out_y * out_z + y_idx * out_z + z_idx) * max_pts;
      int cnt = vox[base];
      if (cnt < max_num) { vox[base + cnt + 1] = k; vox[base]++; }
    }
    for (int v = 0; v < out_x * out_y * out_z; ++v)
      for (int c = 0; c < channels; ++c) {
        const int32_t* list = vox + (int64_t)v * max_pts;
        int64_t o = ((int64_t)bi * out_x * out_y * out_z + v) * channels + c;
        int total = list[0];
        if (pool_method == 0) {
          int am = -1;
          float mv = -INFINITY;
          for (int k = 1; k <= total; ++k)
            if (feat[(int64_t)list[k] * channels + c] > mv) { mv = feat[(int64_t)list[k] * channels + c]; am = list[k]; }
          if (am != -1) pooled[o] = mv;
          argmax[o] = am;
        } else {
          float s = 0;
          for (int k = 1; k <= total; ++k) s += feat[(int64_t)list[k] * channels + c];
          if (total > 0) pooled[o] = s / total;
        }
      }
  }
}
