/*
 * ORACLE — test infrastructure only (tests/, bench.py cpu_baseline, __graft_entry__.smoke()).
 *
 * CPU restatement of the reference rotated-box IoU / NMS:
 *   box_overlap, iou_bev    pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-234 (== iou3d_cpu.cpp:104-230)
 *   iou_normal              iou3d_nms_kernel.cu:314-325
 *   NMS                     tile bitmask iou3d_nms_kernel.cu:267-311 + greedy loop iou3d_nms.cpp:121-135,
 *                           restated as the equivalent O(N^2) "suppressed[]" sweep
 *   boxes_iou_bev_cpu       iou3d_cpu.cpp:232-252
 *
 * Parity pin: UNPINNED by reference execution — iou3d_cpu.cpp includes <cuda.h>/<cuda_runtime_api.h>
 * (absent here), so it is unbuildable without stand-in headers, and the reference ships no test vectors.
 * Anchors used instead (tests/test_iou3d_*.py): closed-form overlaps of axis-aligned and 45-degree boxes,
 * an independent Sutherland-Hodgman polygon clip in float64, symmetry and IoU in [0,1].
 *
 * sin/cos/atan2 are include/fv2p_math.h (deterministic fp32, shared with the device code) so that the
 * integer NMS survivor lists are reproducible across host and device; the reference uses the CUDA libm,
 * whose last-ulp behaviour differs from any host libm by a similar amount.  Build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/fv2p_math.h"

typedef struct { float x, y; } pt_t;

static float o_cross2(pt_t a, pt_t b) { return a.x * b.y - a.y * b.x; }
static float o_cross3(pt_t p1, pt_t p2, pt_t p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
static float o_min(float a, float b) { return a < b ? a : b; }
static float o_max(float a, float b) { return a > b ? a : b; }

static int o_check_rect_cross(pt_t p1, pt_t p2, pt_t q1, pt_t q2) {
  return o_min(p1.x, p2.x) <= o_max(q1.x, q2.x) && o_min(q1.x, q2.x) <= o_max(p1.x, p2.x) &&
         o_min(p1.y, p2.y) <= o_max(q1.y, q2.y) && o_min(q1.y, q2.y) <= o_max(p1.y, p2.y);
}

static int o_check_in_box2d(const float* box, pt_t p) {
  const float MARGIN = 1e-2f;
  float center_x = box[0], center_y = box[1];
  float angle_cos = fv2p_cosf(-box[6]), angle_sin = fv2p_sinf(-box[6]);
  float rot_x = (p.x - center_x) * angle_cos + (p.y - center_y) * (-angle_sin);
  float rot_y = (p.x - center_x) * angle_sin + (p.y - center_y) * angle_cos;
  return (fabsf(rot_x) < box[3] / 2 + MARGIN && fabsf(rot_y) < box[4] / 2 + MARGIN);
}

static int o_intersection(pt_t p1, pt_t p0, pt_t q1, pt_t q0, pt_t* ans) {
  const float EPS = 1e-8f;
  if (o_check_rect_cross(p0, p1, q0, q1) == 0) return 0;
  float s1 = o_cross3(q0, p1, p0);
  float s2 = o_cross3(p1, q1, p0);
  float s3 = o_cross3(p0, q1, q0);
  float s4 = o_cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = o_cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static void o_rotate_around_center(pt_t center, float angle_cos, float angle_sin, pt_t* p) {
  float new_x = (p->x - center.x) * angle_cos + (p->y - center.y) * (-angle_sin) + center.x;
  float new_y = (p->x - center.x) * angle_sin + (p->y - center.y) * angle_cos + center.y;
  p->x = new_x;
  p->y = new_y;
}

static int o_point_cmp(pt_t a, pt_t b, pt_t center) {
  return fv2p_atan2f(a.y - center.y, a.x - center.x) > fv2p_atan2f(b.y - center.y, b.x - center.x);
}

float oracle_box_overlap(const float* box_a, const float* box_b) {
  float a_angle = box_a[6], b_angle = box_b[6];
  float a_dx_half = box_a[3] / 2, b_dx_half = box_b[3] / 2, a_dy_half = box_a[4] / 2, b_dy_half = box_b[4] / 2;
  float a_x1 = box_a[0] - a_dx_half, a_y1 = box_a[1] - a_dy_half;
  float a_x2 = box_a[0] + a_dx_half, a_y2 = box_a[1] + a_dy_half;
  float b_x1 = box_b[0] - b_dx_half, b_y1 = box_b[1] - b_dy_half;
  float b_x2 = box_b[0] + b_dx_half, b_y2 = box_b[1] + b_dy_half;
  pt_t center_a = {box_a[0], box_a[1]}, center_b = {box_b[0], box_b[1]};
  pt_t ca[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  pt_t cb[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};
  float a_angle_cos = fv2p_cosf(a_angle), a_angle_sin = fv2p_sinf(a_angle);
  float b_angle_cos = fv2p_cosf(b_angle), b_angle_sin = fv2p_sinf(b_angle);
  for (int k = 0; k < 4; k++) {
    o_rotate_around_center(center_a, a_angle_cos, a_angle_sin, &ca[k]);
    o_rotate_around_center(center_b, b_angle_cos, b_angle_sin, &cb[k]);
  }
  ca[4] = ca[0];
  cb[4] = cb[0];
  pt_t cross_points[16];
  pt_t poly_center = {0, 0};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      int flag = o_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &cross_points[cnt]);
      if (flag) {
        poly_center.x = poly_center.x + cross_points[cnt].x;
        poly_center.y = poly_center.y + cross_points[cnt].y;
        cnt++;
      }
    }
  for (int k = 0; k < 4; k++) {
    if (o_check_in_box2d(box_a, cb[k])) {
      poly_center.x = poly_center.x + cb[k].x;
      poly_center.y = poly_center.y + cb[k].y;
      cross_points[cnt] = cb[k];
      cnt++;
    }
    if (o_check_in_box2d(box_b, ca[k])) {
      poly_center.x = poly_center.x + ca[k].x;
      poly_center.y = poly_center.y + ca[k].y;
      cross_points[cnt] = ca[k];
      cnt++;
    }
  }
  if (cnt == 0) return 0.0f; /* reference: 0/0 centre, empty polygon, area 0 */
  poly_center.x /= cnt;
  poly_center.y /= cnt;
  for (int j = 0; j < cnt - 1; j++)
    for (int i = 0; i < cnt - j - 1; i++)
      if (o_point_cmp(cross_points[i], cross_points[i + 1], poly_center)) {
        pt_t temp = cross_points[i];
        cross_points[i] = cross_points[i + 1];
        cross_points[i + 1] = temp;
      }
  float area = 0;
  for (int k = 0; k < cnt - 1; k++) {
    pt_t u = {cross_points[k].x - cross_points[0].x, cross_points[k].y - cross_points[0].y};
    pt_t v = {cross_points[k + 1].x - cross_points[0].x, cross_points[k + 1].y - cross_points[0].y};
    area += o_cross2(u, v);
  }
  return (float)(fabs((double)area) / 2.0);
}

float oracle_iou_bev(const float* box_a, const float* box_b) {
  float sa = box_a[3] * box_a[4];
  float sb = box_b[3] * box_b[4];
  float s_overlap = oracle_box_overlap(box_a, box_b);
  return s_overlap / fmaxf(sa + sb - s_overlap, 1e-8f);
}

float oracle_iou_normal(const float* a, const float* b) {
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  float interS = width * height;
  float Sa = a[3] * a[4];
  float Sb = b[3] * b[4];
  return interS / fmaxf(Sa + Sb - interS, 1e-8f);
}

/* mode 0: overlap area, 1: IoU */
void oracle_boxes_bev(const float* a, int na, const float* b, int nb, int mode, float* out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      out[(int64_t)i * nb + j] = mode ? oracle_iou_bev(a + i * 7, b + j * 7) : oracle_box_overlap(a + i * 7, b + j * 7);
}

/* NMS over score-sorted boxes: box i survives iff no surviving j < i has iou(j, i) > thresh
 * (bit (j-row, i-col) of the reference mask, consumed in index order by iou3d_nms.cpp:121-135). */
int oracle_nms(const float* boxes, int n, float thresh, int normal, int64_t* keep) {
  unsigned char* removed = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  int num = 0;
  for (int i = 0; i < n; ++i) {
    if (removed[i]) continue;
    keep[num++] = i;
    for (int j = i + 1; j < n; ++j) {
      if (removed[j]) continue;
      float v = normal ? oracle_iou_normal(boxes + i * 7, boxes + j * 7) : oracle_iou_bev(boxes + i * 7, boxes + j * 7);
      if (v > thresh) removed[j] = 1;
    }
  }
  free(removed);
  return num;
}

/* ---- all-core forms (BASELINE.md section 2, B2 / B3 "1 thread and all cores (OpenMP)"): the same per-pair arithmetic, rows in parallel */
void oracle_boxes_bev_mt(const float* a, int na, const float* b, int nb, int mode, float* out, int threads) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      out[(int64_t)i * nb + j] = mode ? oracle_iou_bev(a + i * 7, b + j * 7) : oracle_box_overlap(a + i * 7, b + j * 7);
}

/* The reference's two phases: the suppression mask of every pair j > i (what nms_kernel computes on the GPU,
 * iou3d_nms_kernel.cu:268-316; here the rows in parallel), then the serial greedy pass over it (iou3d_nms.cpp:121-135).
 * Same survivors as oracle_nms, which evaluates only the pairs the greedy pass reaches. */
int oracle_nms_mt(const float* boxes, int n, float thresh, int normal, int64_t* keep, int threads) {
  const int64_t words = ((int64_t)n + 63) / 64;
  uint64_t* mask = (uint64_t*)calloc((size_t)(n > 0 ? n : 1) * (size_t)(words > 0 ? words : 1), sizeof(uint64_t));
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      float v = normal ? oracle_iou_normal(boxes + i * 7, boxes + j * 7) : oracle_iou_bev(boxes + i * 7, boxes + j * 7);
      if (v > thresh) mask[(int64_t)i * words + (j >> 6)] |= (uint64_t)1 << (j & 63);
    }
  uint64_t* removed = (uint64_t*)calloc((size_t)(words > 0 ? words : 1), sizeof(uint64_t));
  int num = 0;
  for (int i = 0; i < n; ++i) {
    if (removed[i >> 6] & ((uint64_t)1 << (i & 63))) continue;
    keep[num++] = i;
    for (int64_t w = i >> 6; w < words; ++w) removed[w] |= mask[(int64_t)i * words + w];
  }
  free(removed);
  free(mask);
  return num;
}
