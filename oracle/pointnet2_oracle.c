/*
 * ORACLE — test infrastructure only (tests/, bench.py cpu_baseline, __graft_entry__.smoke()).
 *
 * CPU restatement of the reference pointnet2 kernels (thread loops unrolled into plain loops, same comparison
 * operators, index tie-breaks and fp32 operand order; build with -ffp-contract=off):
 *   batch : ball_query_kernel_fast      pointnet2_batch/src/ball_query_gpu.cu:15-51
 *           group / gather (+grad)      group_points_gpu.cu:14-72, sampling_gpu.cu:15-70
 *           furthest_point_sampling     sampling_gpu.cu:93-216 with opt_n_threads (cuda_utils.h:10-14): the strided
 *                                       thread ownership and the tree reduction's "lower slot wins ties" are emulated
 *           three_nn / interpolate      interpolate_gpu.cu:16-149
 *   stack : ball_query / group / three_nn / interpolate / voxel_query
 *                                       pointnet2_stack/src/{ball_query_gpu.cu:16-66, group_points_gpu.cu:15-102,
 *                                       interpolate_gpu.cu:16-172, voxel_query_gpu.cu:10-89}
 * Parity pin: UNPINNED by reference execution (GPU-only kernels, wrappers need <THC/THC.h>; no reference vectors).
 * Anchors: brute-force numpy properties in tests/test_pointnet2_oracle.py (ball members within radius and in index
 * order, 3-NN == stable sort, FPS min-distance sequence non-increasing and first index 0).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float o_d2(const float* a, const float* b) {
  return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

/* idx (B,M,nsample) pre-zeroed by the caller (pointnet2_utils.py:218) */
void oracle_ball_query_batch(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int32_t* idx) {
  float radius2 = radius * radius;
  for (int bs = 0; bs < b; ++bs)
    for (int pt = 0; pt < m; ++pt) {
      const float* q = new_xyz + ((int64_t)bs * m + pt) * 3;
      int32_t* out = idx + ((int64_t)bs * m + pt) * nsample;
      int cnt = 0;
      for (int k = 0; k < n; ++k) {
        float d2 = o_d2(q, xyz + ((int64_t)bs * n + k) * 3);
        if (d2 < radius2) {
          if (cnt == 0) for (int l = 0; l < nsample; ++l) out[l] = k;
          out[cnt] = k;
          ++cnt;
          if (cnt >= nsample) break;
        }
      }
    }
}

void oracle_ball_query_stack(int B, int M, float radius, int nsample, const float* new_xyz, const int32_t* new_cnt, const float* xyz,
                             const int32_t* xyz_cnt, int32_t* idx) {
  float radius2 = radius * radius;
  for (int pt = 0; pt < M; ++pt) {
    int bs_idx = 0, pt_cnt = new_cnt[0];
    for (int k = 1; k < B; k++) { if (pt < pt_cnt) break; pt_cnt += new_cnt[k]; bs_idx = k; }
    int start = 0;
    for (int k = 0; k < bs_idx; k++) start += xyz_cnt[k];
    int n = xyz_cnt[bs_idx], cnt = 0;
    int32_t* out = idx + (int64_t)pt * nsample;
    for (int k = 0; k < n; ++k) {
      float d2 = o_d2(new_xyz + pt * 3, xyz + ((int64_t)start + k) * 3);
      if (d2 < radius2) {
        if (cnt == 0) for (int l = 0; l < nsample; ++l) out[l] = k;
        out[cnt] = k;
        ++cnt;
        if (cnt >= nsample) break;
      }
    }
    if (cnt == 0) out[0] = -1;
  }
}

void oracle_voxel_query_stack(int M, int R1, int R2, int R3, int nsample, float radius, int z_range, int y_range, int x_range,
                              const float* new_xyz, const float* xyz, const int32_t* new_coords, const int32_t* point_indices, int32_t* idx) {
  float radius2 = radius * radius;
  for (int pt = 0; pt < M; ++pt) {
    const float* q = new_xyz + pt * 3;
    const int32_t* c = new_coords + pt * 4;
    int32_t* out = idx + (int64_t)pt * nsample;
    int cnt = 0;
    for (int dz = -z_range; dz <= z_range; ++dz) {
      int z = c[1] + dz;
      if (z < 0 || z >= R1) continue;
      for (int dy = -y_range; dy <= y_range; ++dy) {
        int y = c[2] + dy;
        if (y < 0 || y >= R2) continue;
        for (int dx = -x_range; dx <= x_range; ++dx) {
          int x = c[3] + dx;
          if (x < 0 || x >= R3) continue;
          int32_t nb = point_indices[(((int64_t)c[0] * R1 + z) * R2 + y) * R3 + x];
          if (nb < 0) continue;
          float dist2 = o_d2(xyz + (int64_t)nb * 3, q);
          if (dist2 > radius2) continue;
          if (cnt < nsample) {
            if (cnt == 0) for (int l = 0; l < nsample; ++l) out[l] = nb;
            out[cnt] = nb;
            ++cnt;
          }
        }
      }
    }
    if (cnt == 0) out[0] = -1;
  }
}

/* sampling_gpu.cu:100-216 emulated: bs "threads", thread t owns t, t+bs, ...; per-thread first max (strict >),
 * tree reduction keeps the lower slot unless the upper value is strictly larger. temp (B,N) initialised by caller. */
void oracle_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp, int32_t* idxs) {
  int pow_2 = (int)(log((double)n) / log(2.0));
  int bs = 1 << pow_2;
  if (bs > 1024) bs = 1024;
  if (bs < 1) bs = 1;
  float* dists = (float*)malloc(sizeof(float) * bs);
  int32_t* dists_i = (int32_t*)malloc(sizeof(int32_t) * bs);
  for (int bi = 0; bi < b; ++bi) {
    const float* data = dataset + (int64_t)bi * n * 3;
    float* tp = temp + (int64_t)bi * n;
    int32_t* out = idxs + (int64_t)bi * m;
    if (m <= 0) continue;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1;
        for (int k = tid; k < n; k += bs) {
          float d = o_d2(data + k * 3, data + old * 3);
          float d2 = d < tp[k] ? d : tp[k];
          tp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = bs / 2; s >= 1; s /= 2)
        for (int tid = 0; tid < s; ++tid) {
          float v1 = dists[tid], v2 = dists[tid + s];
          int32_t i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = v1 > v2 ? v1 : v2;
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(dists);
  free(dists_i);
}

static void o_three_nn_one(const float* u, const float* known, int m, float* dist2, int32_t* idx, int idx_off) {
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k = 0; k < m; ++k) {
    float d = o_d2(u, known + k * 3);
    if (d < best1) { best3 = best2; besti3 = besti2; best2 = best1; besti2 = besti1; best1 = d; besti1 = k; }
    else if (d < best2) { best3 = best2; besti3 = besti2; best2 = d; besti2 = k; }
    else if (d < best3) { best3 = d; besti3 = k; }
  }
  dist2[0] = (float)best1; dist2[1] = (float)best2; dist2[2] = (float)best3;
  idx[0] = besti1 + idx_off; idx[1] = besti2 + idx_off; idx[2] = besti3 + idx_off;
}

void oracle_three_nn_batch(int b, int n, int m, const float* unknown, const float* known, float* dist2, int32_t* idx) {
  for (int bs = 0; bs < b; ++bs)
    for (int pt = 0; pt < n; ++pt)
      o_three_nn_one(unknown + ((int64_t)bs * n + pt) * 3, known + (int64_t)bs * m * 3, m, dist2 + ((int64_t)bs * n + pt) * 3,
                     idx + ((int64_t)bs * n + pt) * 3, 0);
}

void oracle_three_nn_stack(int B, int N, const float* unknown, const int32_t* unk_cnt, const float* known, const int32_t* known_cnt,
                           float* dist2, int32_t* idx) {
  for (int pt = 0; pt < N; ++pt) {
    int bs_idx = 0, pt_cnt = unk_cnt[0];
    for (int k = 1; k < B; k++) { if (pt < pt_cnt) break; pt_cnt += unk_cnt[k]; bs_idx = k; }
    int start = 0;
    for (int k = 0; k < bs_idx; k++) start += known_cnt[k];
    o_three_nn_one(unknown + pt * 3, known + (int64_t)start * 3, known_cnt[bs_idx], dist2 + pt * 3, idx + pt * 3, start);
  }
}
