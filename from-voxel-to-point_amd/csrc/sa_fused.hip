// Fused grid set-abstraction of the RoI head (A11 consumers: PointnetSAModuleMSG as IoUGuidedRoIHead builds it,
// pcdet/models/roi_heads/iouguided_roi_head.py:52-76, 258-275; pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:30-62).
//
// The reference materialises, per radius, the grouped tensor (R, 3 + C, M, S) — 1.39 GB at R = 384 RoIs, M = 216 grid centres,
// S = 32 samples — and runs its shared MLP (1x1 convs + ReLU) over it as plain GEMMs, then max-pools over the S samples.  With
// the first (linear) layer applied per POINT and per CENTRE beforehand (fv2p_harness/fv2p_model.py: sa_msg_grid), what is left
// per (centre i, sample s) is
//        h1 = relu(P[idx[i, s], :] - Q[i, :])          64 channels, P = per-point, Q = per-centre first-layer products
//        h2 = relu(W2 h1)                               second shared-MLP layer, 64 x 64
//        out[i, :] = max_s h2
// and this file computes that without any grouped tensor: a wave gathers 16 sample rows (256 B each, one float4 per lane and
// 16-channel block), runs the 64 x 64 layer on fp32 MFMA against W2 fragments resident in LDS, and reduces the maximum over
// the rows in registers.  The product is formed TRANSPOSED (h2^T = W2 h1^T): then a lane that supplied the channels
// 16j + 4g .. + 3 of sample row r as the B operand receives the channels 16b + 4g .. + 3 of the same row in its accumulators,
// i.e. results have the operand layout and chain into the next product (backward: dh1^T = W2^T dh2^T) without any shuffle.
//
// The forward pass also stores, per centre and channel, WHICH sample attained the maximum (one byte; the first one in sample
// order, as the reference's F.max_pool2d does, pointnet2_modules.py:57-59; 255 = the maximum is 0, no gradient).  Backward
// recomputes h1 only (a gather and a subtraction), routes d out to that sample, and produces dP (scatter-add into an LDS tile
// per RoI and 32-channel half, flushed once), dQ (row sums, direct stores) and dW2 (MFMA over the sample rows after an LDS
// transpose; one partial tile per workgroup, summed by a second launch).  Rounds 1-2 stored nothing but `out` and recomputed
// h2 to find the maxima again: half of the backward pass's MFMA work (64 of 128 per 16-sample tile), twice — once per
// channel half.
#include "common.hpp"

namespace fv2p {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSaC = 64;          // channels of both layers
constexpr int kSaCentres = 16;    // centres per workgroup (forward)

// Reductions over the 16 lanes of a DPP row (= the 16 sample rows of a tile; lanes sharing l / 16 hold the same channels), 8 or 16
// values at a time, each as ONE asm statement in step-major order: the DPP read of a value is then N instructions behind the
// write it depends on (the two wait states a DPP operand needs are covered by the other values' instructions) and the
// compiler cannot slip a register copy in front of a step.  quad_perm xor-1, xor-2, row_half_mirror, row_mirror: every lane
// of the row ends with the row's result.
__device__ __forceinline__ void row16_max_n(float (&v)[16]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %9, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %10, %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %11, %11, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %12, %12, %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %13, %13, %13 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %14, %14, %14 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %15, %15, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %9, %9, %9 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %10, %10, %10 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %11, %11, %11 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %12, %12, %12 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %13, %13, %13 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %14, %14, %14 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %15, %15, %15 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %8, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %9, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %10, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %11, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %12, %12, %12 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %13, %13, %13 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %14, %14, %14 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %15, %15, %15 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %4, %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %8, %8, %8 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %9, %9, %9 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %10, %10, %10 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %11, %11, %11 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %12, %12, %12 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %13, %13, %13 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %14, %14, %14 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %15, %15, %15 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}
__device__ __forceinline__ void row16_sum_n(float (&v)[16]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %10, %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %11, %11, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %12, %12, %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %13, %13, %13 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %14, %14, %14 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %15, %15, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %10, %10, %10 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %11, %11, %11 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %12, %12, %12 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %13, %13, %13 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %14, %14, %14 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %15, %15, %15 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %10, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %11, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %12, %12, %12 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %13, %13, %13 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %14, %14, %14 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %15, %15, %15 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %10, %10, %10 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %11, %11, %11 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %12, %12, %12 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %13, %13, %13 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %14, %14, %14 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %15, %15, %15 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}
__device__ __forceinline__ void row16_sum_n(float (&v)[8]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}

// LDS image of a 64 x 64 row-major matrix A for the role "A operand of D = A * X^T": frag[b][j][lane] (float4) =
// A[16 b + lane % 16][16 j + 4 (lane / 16) .. + 3]
__device__ __forceinline__ void stage_frag(const float* __restrict__ a, bool transpose, float* __restrict__ lds) {
  for (int e = threadIdx.x; e < 16 * 64; e += blockDim.x) {
    const int lane = e & 63, bj = e >> 6, b = bj >> 2, j = bj & 3;
    const int row = 16 * b + (lane & 15), col = 16 * j + 4 * (lane >> 4);
    float4 v;
    if (!transpose) v = *reinterpret_cast<const float4*>(a + row * kSaC + col);
    else v = make_float4(a[(col + 0) * kSaC + row], a[(col + 1) * kSaC + row], a[(col + 2) * kSaC + row], a[(col + 3) * kSaC + row]);
    *reinterpret_cast<float4*>(lds + e * 4) = v;
  }
}
// acc[b] += A-fragments(lds)[b][j] * x[j]   (x[j] = this lane's float4 of channel block j of its row)
__device__ __forceinline__ void mma64(const float* __restrict__ frag, int lane, const float4 (&x)[4], f32x4 (&acc)[4]) {
  // j outer, the four accumulator blocks inner: consecutive MFMAs belong to different dependency chains
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 a[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) a[b] = *reinterpret_cast<const float4*>(frag + ((b * 4 + j) * 64 + lane) * 4);
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b].x, x[j].x, acc[b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b].y, x[j].y, acc[b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b].z, x[j].z, acc[b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b].w, x[j].w, acc[b], 0, 0, 0);
  }
}
__device__ __forceinline__ float4 relu_sub(float4 p, float4 q) {
  return make_float4(fmaxf(p.x - q.x, 0.f), fmaxf(p.y - q.y, 0.f), fmaxf(p.z - q.z, 0.f), fmaxf(p.w - q.w, 0.f));
}

// ------------------------------------------------------------------ forward ---------------------------------------------------
// grid (ceil(M / 16), R), block 256: wave w handles centres 16 * blockIdx.x + w, w + 4, ...
__global__ __launch_bounds__(256) void sa_grid_fwd_k(int n, int m, int s, const float* __restrict__ P, const float* __restrict__ Q,
                                                     const int* __restrict__ idx, const float* __restrict__ W2, float* __restrict__ out,
                                                     unsigned char* __restrict__ arg) {
  __shared__ __attribute__((aligned(16))) float wfrag[16 * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, g = lane >> 4;
  const int r = blockIdx.y;
  stage_frag(W2, false, wfrag);
  __syncthreads();
  const float* Pr = P + static_cast<long long>(r) * n * kSaC;
  for (int ci = wave; ci < kSaCentres; ci += 4) {
    const int i = blockIdx.x * kSaCentres + ci;
    if (i >= m) break;
    const float* q = Q + (static_cast<long long>(r) * m + i) * kSaC + 4 * g;
    float4 qv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) qv[j] = *reinterpret_cast<const float4*>(q + 16 * j);
    const int* id = idx + (static_cast<long long>(r) * m + i) * s;
    float best[4][4];
    uint32_t late = 0;   // bit 4 b + e: this lane's best value of that channel came from the second tile (s <= 32)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) best[b][e] = 0.f;   // relu output: the maximum is >= 0
    for (int t = 0; t < s; t += 16) {
      const float* p = Pr + static_cast<long long>(id[t + row]) * kSaC + 4 * g;
      float4 x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = relu_sub(*reinterpret_cast<const float4*>(p + 16 * j), qv[j]);
      f32x4 acc[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
      mma64(wfrag, lane, x, acc);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (t != 0 && acc[b][e] > best[b][e]) late |= 1u << (4 * b + e);   // strict: the earlier sample keeps a tie
          best[b][e] = fmaxf(best[b][e], acc[b][e]);
        }
    }
    // lane (row, g) holds channels 16 b + 4 g + e of sample row `row`: maximum over the 16 lanes of the DPP row
    float* o = out + (static_cast<long long>(r) * m + i) * kSaC + 4 * g;
    float flat[16];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) flat[4 * b + e] = best[b][e];
    row16_max_n(flat);
    if (row == 0) {
#pragma unroll
      for (int b = 0; b < 4; ++b) *reinterpret_cast<float4*>(o + 16 * b) = make_float4(flat[4 * b], flat[4 * b + 1], flat[4 * b + 2], flat[4 * b + 3]);
    }
    if (arg) {
      // first sample (in sample order) that attains the maximum: lanes holding it offer -(their sample number), the row maximum of
      // that is the smallest number; a lane's own first is its first tile unless the second was strictly larger
      float cand[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const bool holds = best[k >> 2][k & 3] == flat[k] && flat[k] > 0.f;
        cand[k] = holds ? -static_cast<float>(((late >> k) & 1u) * 16u + static_cast<uint32_t>(row)) : -255.f;
      }
      row16_max_n(cand);
      if (row == 0) {
        unsigned char* a = arg + (static_cast<long long>(r) * m + i) * kSaC + 4 * g;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const uint32_t packed = static_cast<uint32_t>(-cand[4 * b]) | (static_cast<uint32_t>(-cand[4 * b + 1]) << 8) |
                                  (static_cast<uint32_t>(-cand[4 * b + 2]) << 16) | (static_cast<uint32_t>(-cand[4 * b + 3]) << 24);
          *reinterpret_cast<uint32_t*>(a + 16 * b) = packed;
        }
      }
    }
  }
}

// ------------------------------------------------------------------ backward --------------------------------------------------
// grid (2, R): workgroup (h, r) owns the channel half h of dP / dQ (channels 32 h .. 32 h + 31) and the row half h of dW2
// (output channels c' in the same range) of RoI r; block 512 (256), wave w takes centres w, w + 8 (4), ...  S <= 32.
constexpr int kSaTs = 72;   // row stride of the transpose tiles (floats): 64 + 8, conflict-light both ways
constexpr int kSaDp = 33;   // row stride of the dP tile (floats): odd, so the scatter's rows (arbitrary points) spread over all LDS banks —
                            // with 32 every row starts on bank 0 or 32 and a wave's 64 adds fall on 8 banks
// kSaBwdWaves = 8 where the dP tile leaves room for eight waves' transpose tiles (n <= 589), two per SIMD: one wave's gathers, LDS
// atomics and transposes run beside the other's MFMAs (one workgroup per CU either way: the dP tile); 4 above that.
template <int TILES, int kSaBwdWaves>   // TILES = samples per centre / 16: register arrays below are indexed by compile-time tile numbers only
__global__ __launch_bounds__(kSaBwdWaves * 64) void sa_grid_bwd_k(int n, int m, const float* __restrict__ P, const float* __restrict__ Q,
                                                     const int* __restrict__ idx, const float* __restrict__ W2, const unsigned char* __restrict__ arg,
                                                     const float* __restrict__ dout, float* __restrict__ dP, float* __restrict__ dQ,
                                                     float* __restrict__ dW2_partial) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wtfrag = lds;                         // W2^T    as A operand (dh1^T = W2^T dh2^T), this half's two row blocks only: 8 KB
  float* tile = wtfrag + 8 * 64 * 4;           // per wave: two 16 x 72 transpose tiles (dh2, h1)
  float* dpt = tile + kSaBwdWaves * 2 * 16 * kSaTs;      // dP accumulator of this RoI and channel half: [n][kSaDp]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, g = lane >> 4;
  const int half = blockIdx.x, r = blockIdx.y;
  // fragments [bb][j][lane] of W2^T rows 32 half + 16 bb + lane % 16, columns 16 j + 4 (lane / 16) .. + 3
  for (int e = threadIdx.x; e < 8 * 64; e += kSaBwdWaves * 64) {
    const int l = e & 63, bj = e >> 6, bb = bj >> 2, j = bj & 3;
    const int rw = 32 * static_cast<int>(blockIdx.x) + 16 * bb + (l & 15), col = 16 * j + 4 * (l >> 4);
    *reinterpret_cast<float4*>(wtfrag + e * 4) = make_float4(W2[(col + 0) * kSaC + rw], W2[(col + 1) * kSaC + rw], W2[(col + 2) * kSaC + rw], W2[(col + 3) * kSaC + rw]);
  }
  for (int e = threadIdx.x; e < n * kSaDp; e += kSaBwdWaves * 64) dpt[e] = 0.f;
  __syncthreads();
  float* t_dh2 = tile + wave * 2 * 16 * kSaTs;
  float* t_h1 = t_dh2 + 16 * kSaTs;
  const float* Pr = P + static_cast<long long>(r) * n * kSaC;
  // dW2 rows c' = 32 half + 16 bb + ..., all 64 columns: 2 x 4 accumulator tiles, summed over every sample row this wave sees
  f32x4 dw[2][4];
#pragma unroll
  for (int bb = 0; bb < 2; ++bb)
#pragma unroll
    for (int bc = 0; bc < 4; ++bc) dw[bb][bc] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int s = TILES * 16;
  // A centre's inputs come from two dependent rounds of global loads (its sample list, then the rows the list names): ~2 x 2 us that
  // nothing covered while the centre's own work is 2 - 4 k clocks of matrix pipe - the round-3 kernel spent 13.8 k clocks per centre and
  // wave.  Now they are two centres deep in flight: while centre i multiplies, the rows / centre vector / arg bytes / gradient of centre
  // i + W and the sample list of centre i + 2 W are on their way (W = waves of the workgroup).  The prefetches are issued AFTER the
  // current centre's operands have been consumed into x[][], so that the wait in front of that use has only loads of the iteration
  // before to wait for.
  struct Staged { float4 qv[4], gd[4], praw[TILES][4]; uint32_t who[4]; int src[TILES]; };
  auto centre_of = [&](int i) { return static_cast<long long>(r) * m + (i < m ? i : m - 1); };   // past the end: the last centre again (never used)
  auto load_list = [&](int i, int (&src)[TILES]) {
    const int* id = idx + centre_of(i) * s;
#pragma unroll
    for (int t = 0; t < TILES; ++t) src[t] = id[16 * t + row];
  };
  auto load_rows = [&](int i, Staged& c) {   // c.src holds the list already
    const long long ctr = centre_of(i);
    const float* q = Q + ctr * kSaC + 4 * g;
    const unsigned char* am = arg + ctr * kSaC + 4 * g;
    const float* go = dout + ctr * kSaC + 4 * g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c.qv[j] = *reinterpret_cast<const float4*>(q + 16 * j);
      c.who[j] = *reinterpret_cast<const uint32_t*>(am + 16 * j);
      c.gd[j] = *reinterpret_cast<const float4*>(go + 16 * j);
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const float* p = Pr + static_cast<long long>(c.src[t]) * kSaC + 4 * g;
#pragma unroll
      for (int j = 0; j < 4; ++j) c.praw[t][j] = *reinterpret_cast<const float4*>(p + 16 * j);
    }
  };
  Staged cur, nxt;
  int list2[TILES];
  load_list(wave, cur.src);
  load_rows(wave, cur);
  load_list(wave + kSaBwdWaves, nxt.src);
  for (int i = wave; i < m; i += kSaBwdWaves) {
    uint32_t who[4];   // byte e of who[b]: the sample that attained the maximum of channel 16 b + 4 g + e (255: none)
    float4 gd[4];
    float4 x[TILES][4];
    int src[TILES];
#pragma unroll
    for (int b = 0; b < 4; ++b) { who[b] = cur.who[b]; gd[b] = cur.gd[b]; }
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      src[t] = cur.src[t];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[t][j] = relu_sub(cur.praw[t][j], cur.qv[j]);
    }
    __builtin_amdgcn_sched_barrier(0);   // the prefetches stay behind the use above
    load_rows(i + kSaBwdWaves, nxt);
    load_list(i + 2 * kSaBwdWaves, list2);
    __builtin_amdgcn_sched_barrier(0);
    // pass 2 per tile: dh2, dh1 = (W2^T dh2) * [h1 > 0], scatter / reduce, dW2 += dh2^T h1
    float dq[2][4];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int e = 0; e < 4; ++e) dq[bb][e] = 0.f;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      float4 dh2[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float gv[4] = {gd[b].x, gd[b].y, gd[b].z, gd[b].w};
        float d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = ((who[b] >> (8 * e)) & 0xffu) == static_cast<uint32_t>(16 * t + row) ? gv[e] : 0.f;
        dh2[b] = make_float4(d[0], d[1], d[2], d[3]);
      }
      // dh1^T block bb of this half: channels 32 half + 16 bb + 4 g + e
      f32x4 d1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 a0 = *reinterpret_cast<const float4*>(wtfrag + ((0 * 4 + j) * 64 + lane) * 4);
        const float4 a1 = *reinterpret_cast<const float4*>(wtfrag + ((1 * 4 + j) * 64 + lane) * 4);
        d1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, dh2[j].x, d1[0], 0, 0, 0);
        d1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, dh2[j].x, d1[1], 0, 0, 0);
        d1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, dh2[j].y, d1[0], 0, 0, 0);
        d1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, dh2[j].y, d1[1], 0, 0, 0);
        d1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, dh2[j].z, d1[0], 0, 0, 0);
        d1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, dh2[j].z, d1[1], 0, 0, 0);
        d1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, dh2[j].w, d1[0], 0, 0, 0);
        d1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, dh2[j].w, d1[1], 0, 0, 0);
      }
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const float4 h = half ? x[t][2 + bb] : x[t][bb];   // (a runtime index would push the array into scratch)
        const float hv[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = hv[e] > 0.f ? d1[bb][e] : 0.f;
          dq[bb][e] += v;
          if (v != 0.f) atomicAdd(&dpt[src[t] * kSaDp + 16 * bb + 4 * g + e], v);
        }
      }
      // dW2[c'][c] += sum_rows dh2[row][c'] h1[row][c]: rows become the K dimension -> transpose both tiles through LDS
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        *reinterpret_cast<float4*>(t_h1 + row * kSaTs + 16 * b + 4 * g) = x[t][b];
        if ((b >> 1) == half) *reinterpret_cast<float4*>(t_dh2 + row * kSaTs + 16 * (b & 1) + 4 * g) = dh2[b];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's own LDS writes (tiles are private to the wave)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {       // K step: sample rows 4 ks + g
        float a[2], bcol[4];
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) a[bb] = t_dh2[(4 * ks + g) * kSaTs + 16 * bb + row];       // A[m = c' % 16][k = row]
#pragma unroll
        for (int bc = 0; bc < 4; ++bc) bcol[bc] = t_h1[(4 * ks + g) * kSaTs + 16 * bc + row];      // B[k = row][n = c % 16]
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int bc = 0; bc < 4; ++bc) dw[bb][bc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[bb], bcol[bc], dw[bb][bc], 0, 0, 0);
      }
    }
    // dQ[i][c] = - sum over the centre's samples of dh1 (the 16 lanes of the DPP row), this half's 32 channels
    float* dqo = dQ + (static_cast<long long>(r) * m + i) * kSaC + 32 * half + 4 * g;
    float flat[8];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int e = 0; e < 4; ++e) flat[4 * bb + e] = dq[bb][e];
    row16_sum_n(flat);
    if (row == 0) {
      *reinterpret_cast<float4*>(dqo) = make_float4(-flat[0], -flat[1], -flat[2], -flat[3]);
      *reinterpret_cast<float4*>(dqo + 16) = make_float4(-flat[4], -flat[5], -flat[6], -flat[7]);
    }
    cur = nxt;
#pragma unroll
    for (int t = 0; t < TILES; ++t) nxt.src[t] = list2[t];
  }
  __syncthreads();
  // flush dP (each (RoI, half) tile is owned by this workgroup: plain stores)
  float* dpo = dP + static_cast<long long>(r) * n * kSaC + 32 * half;
  for (int e = threadIdx.x; e < n * 8; e += kSaBwdWaves * 64) {
    const int pt = e >> 3, c4 = (e & 7) * 4;
    const float* t4 = dpt + pt * kSaDp + c4;
    *reinterpret_cast<float4*>(dpo + static_cast<long long>(pt) * kSaC + c4) = make_float4(t4[0], t4[1], t4[2], t4[3]);
  }
  // dW2 partial of this workgroup: the waves' accumulators summed through LDS (the transpose tiles are free now)
  __syncthreads();
  float* buf = lds;    // fragments + tiles (8 + 74 KB, all waves are past them): room for 8 waves x 2 x 4 x 64 x 4 floats = 64 KB
#pragma unroll
  for (int bb = 0; bb < 2; ++bb)
#pragma unroll
    for (int bc = 0; bc < 4; ++bc)
      *reinterpret_cast<f32x4*>(buf + (((wave * 2 + bb) * 4 + bc) * 64 + lane) * 4) = dw[bb][bc];
  __syncthreads();
  // accumulator (bb, bc) of lane l holds dW2[32 half + 16 bb + 4 (l / 16) + e][16 bc + l % 16]
  float* dwo = dW2_partial + (static_cast<long long>(r) * 2 + half) * 32 * kSaC;
  for (int e = threadIdx.x; e < 2 * 4 * 64 * 4; e += kSaBwdWaves * 64) {
    const int comp = e & 3, l = (e >> 2) & 63, bc = (e >> 8) & 3, bb = e >> 10;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kSaBwdWaves; ++w) v += buf[(((w * 2 + bb) * 4 + bc) * 64 + l) * 4 + comp];
    dwo[(16 * bb + 4 * (l >> 4) + comp) * kSaC + 16 * bc + (l & 15)] = v;
  }
}

// dW2[c'][c] = sum over RoIs of the partial tiles: the tiles are one [rois][64 * 64] matrix (row r = the two half tiles of RoI r), the
// result its column sums.  A workgroup owns 16 columns; its 256 threads are 16 row groups x 16 columns, a thread adds the rows
// rg, rg + 16, ... in ascending order and the 16 group sums are added in group order: a fixed association (deterministic).  (One
// thread per column walking all the rows, the round-3 form, took 98 us at 384 RoIs - a fifth of the backward kernel beside it.)
__global__ __launch_bounds__(256) void sa_grid_dw_reduce_k(int rois, const float* __restrict__ partial, float* __restrict__ dW2) {
  __shared__ float part[16][17];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15), rg = threadIdx.x >> 4;
  float s = 0.f;
  for (int r = rg; r < rois; r += 16) s += partial[static_cast<long long>(r) * (kSaC * kSaC) + col];
  part[rg][threadIdx.x & 15] = s;
  __syncthreads();
  if (threadIdx.x < 16) {
    float v = part[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < 16; ++q) v += part[q][threadIdx.x];
    dW2[blockIdx.x * 16 + threadIdx.x] = v;
  }
}

}  // namespace fv2p
using namespace fv2p;

static bool sa_shapes_ok(int rois, int n, int m, int s, int c) {
  return rois >= 1 && n >= 1 && m >= 1 && c == kSaC && (s == 16 || s == 32) && rois <= 65535;
}

extern "C" int fv2p_sa_grid_supported(int n, int m, int s, int c) {
  return (c == kSaC && (s == 16 || s == 32) && n >= 1 && m >= 1 && static_cast<size_t>(n) * kSaDp * 4 <= 114 * 1024) ? 1 : 0;   // backward: 45 KB of fragments / tiles (four waves) + the dP tile within 160 KB of LDS
}

extern "C" int fv2p_sa_grid_fwd(const float* per_point, const float* per_centre, const int* idx, const float* w2, int rois, int n, int m,
                                int s, int c, float* out, unsigned char* arg, fv2p_stream_t stream) {
  FV2P_REQUIRE(sa_shapes_ok(rois, n, m, s, c), FV2P_EINVAL, "sa_grid: needs 64 channels and 16 or 32 samples per centre");
  FV2P_REQUIRE(per_point && per_centre && idx && w2 && out, FV2P_EINVAL, "sa_grid_fwd: null pointer");
  hipLaunchKernelGGL(sa_grid_fwd_k, dim3(static_cast<unsigned>(ceil_div(m, kSaCentres)), rois), dim3(256), 0, static_cast<hipStream_t>(stream),
                     n, m, s, per_point, per_centre, idx, w2, out, arg);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t fv2p_sa_grid_bwd_ws_bytes(int rois) {
  Sizer sz;
  sz.take<float>(static_cast<size_t>(rois > 0 ? rois : 1) * kSaC * kSaC);
  return sz.bytes();
}

extern "C" int fv2p_sa_grid_bwd(const float* per_point, const float* per_centre, const int* idx, const float* w2, const unsigned char* arg,
                                const float* grad_out, int rois, int n, int m, int s, int c, float* grad_point, float* grad_centre,
                                float* grad_w2, void* ws, size_t ws_bytes, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FV2P_REQUIRE(sa_shapes_ok(rois, n, m, s, c) && fv2p_sa_grid_supported(n, m, s, c), FV2P_EINVAL, "sa_grid_bwd: unsupported shape");
  FV2P_REQUIRE(per_point && per_centre && idx && w2 && arg && grad_out && grad_point && grad_centre && grad_w2, FV2P_EINVAL, "sa_grid_bwd: null pointer");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_sa_grid_bwd_ws_bytes(rois), FV2P_EWORKSPACE, "sa_grid_bwd: workspace too small");
  Carver cv(ws, ws_bytes);
  float* partial = cv.take<float>(static_cast<size_t>(rois) * kSaC * kSaC);
  auto lds_of = [&](int waves) { return (8 * 64 * 4 + static_cast<size_t>(waves) * 2 * 16 * kSaTs + static_cast<size_t>(n) * kSaDp) * sizeof(float); };
  const int waves = lds_of(8) <= 160 * 1024 ? 8 : 4;
  const size_t lds = lds_of(waves);
  static bool roomy[4] = {false, false, false, false};   // per kernel instance: the dynamic-LDS limit is raised once
  bool& raised = roomy[(s == 32 ? 2 : 0) + (waves == 8 ? 1 : 0)];
  auto launch = [&](auto kernel) -> int {
    if (lds > 48 * 1024 && !raised) {
      FV2P_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised = true;
    }
    hipLaunchKernelGGL(kernel, dim3(2, rois), dim3(waves * 64), lds, stream, n, m, per_point, per_centre, idx, w2, arg, grad_out, grad_point, grad_centre, partial);
    return 0;
  };
  int rc;
  if (s == 16) rc = waves == 8 ? launch(&sa_grid_bwd_k<1, 8>) : launch(&sa_grid_bwd_k<1, 4>);
  else rc = waves == 8 ? launch(&sa_grid_bwd_k<2, 8>) : launch(&sa_grid_bwd_k<2, 4>);
  if (rc) return rc;
  hipLaunchKernelGGL(sa_grid_dw_reduce_k, dim3(kSaC * kSaC / 16), dim3(256), 0, stream, rois, partial, grad_w2);
  FV2P_LAUNCH_CHECK();
  return 0;
}
