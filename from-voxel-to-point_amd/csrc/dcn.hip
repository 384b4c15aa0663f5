// A13 — modulated deformable convolution (DCNv2) as fused implicit GEMMs on fp32 MFMA.
//
// Replaces DCN.modulated_deform_conv_{forward,backward} (pcdet/ops/DeformableConvolutionV2PyTorch/src/vision.cpp:6-12,
// src/cuda/modulated_deform_conv_cuda.cu:19-280, kernels modulated_deform_im2col_cuda.cuh:24-328).  The reference
// writes a [Cin*kh*kw, B*Ho*Wo] `columns` buffer (1.3 GB for the MGAF head) and runs cuBLAS on it, and its backward
// scatters with atomics from three element-wise kernels.  Here the bilinear samples are produced inside the GEMM
// operand fetch, so no columns buffer exists in either direction:
//
//   forward   y[p, co]        = b[co] + sum_{k,ci} m[p,k] * bilin(x, p, k)[ci] * W[k][ci][co]
//   backward  dcol[p, k, ci]  = sum_co dy[p, co] * W[k][ci][co]           (MFMA, A = dy rows, B = W_k^T)
//             dx    += scatter(dcol * m * tap weights)                     (epilogue, atomics on the 4 taps)
//             dm[p,k], doff[p,k] = reductions of dcol * bilin / d bilin    (epilogue, wave shuffles over ci)
//             dW[k][ci][co] = sum_p m*bilin(x,p,k)[ci] * dy[p, co]         (MFMA over pixel chunks, deterministic reduce)
//
// Activations are NHWC here (16-byte channel vectors per tap); the Python layer permutes once on entry/exit, as the
// reference itself computes NHWC and permutes (modulated_deform_conv_cuda.cu:78,118).  Sampling follows
// mdmcn_im2col_bilinear (:24-54) and the validity test of :176 exactly; offsets/masks keep the reference layout
// [B, dg*2*K, Ho, Wo] / [B, dg*K, Ho, Wo] with (2*(i*kw+j), +1) = (dh, dw).
#include "common.hpp"
#include <stdlib.h>
#include <stdio.h>
#include <algorithm>

namespace fv2p {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DcnGeom {
  int B, H, W, Cin, Cout, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg;
};

struct Taps {
  int o[4];     // element offsets (pixel index in [0, H*W)) of the 4 taps, -1 if outside
  float w[4];   // bilinear weights hh*hw, hh*lw, lh*hw, lh*lw
  float lh, lw; // fractional parts
  int valid;    // the reference's whole-sample validity (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
};

__device__ __forceinline__ Taps make_taps(const DcnGeom& g, float h_im, float w_im) {
  Taps t;
  t.valid = (h_im > -1.f && w_im > -1.f && h_im < static_cast<float>(g.H) && w_im < static_cast<float>(g.W));
  const int h_low = static_cast<int>(floorf(h_im)), w_low = static_cast<int>(floorf(w_im));
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - h_low, lw = w_im - w_low, hh = 1.f - lh, hw = 1.f - lw;
  t.lh = lh; t.lw = lw;
  t.w[0] = hh * hw; t.w[1] = hh * lw; t.w[2] = lh * hw; t.w[3] = lh * lw;
  const bool hl = h_low >= 0, hhi = h_high <= g.H - 1, wl = w_low >= 0, whi = w_high <= g.W - 1;
  t.o[0] = (t.valid && hl && wl) ? h_low * g.W + w_low : -1;
  t.o[1] = (t.valid && hl && whi) ? h_low * g.W + w_high : -1;
  t.o[2] = (t.valid && hhi && wl) ? h_high * g.W + w_low : -1;
  t.o[3] = (t.valid && hhi && whi) ? h_high * g.W + w_high : -1;
  return t;
}

// sample position of output pixel (b, ho, wo), kernel tap k = i*kw + j, deformable group dgi
__device__ __forceinline__ Taps pixel_taps(const DcnGeom& g, const float* __restrict__ offset, const float* __restrict__ mask, int b, int ho,
                                           int wo, int k, int dgi, float* m) {
  const int K = g.kh * g.kw, i = k / g.kw, j = k % g.kw;
  const long long plane = static_cast<long long>(g.Ho) * g.Wo, pos = static_cast<long long>(ho) * g.Wo + wo;
  const float* ob = offset + (static_cast<long long>(b) * g.dg + dgi) * 2 * K * plane;
  const float off_h = ob[(2 * k) * plane + pos], off_w = ob[(2 * k + 1) * plane + pos];
  *m = mask[((static_cast<long long>(b) * g.dg + dgi) * K + k) * plane + pos];
  const float h_im = static_cast<float>(ho * g.sh - g.ph + i * g.dh) + off_h;
  const float w_im = static_cast<float>(wo * g.sw - g.pw + j * g.dw) + off_w;
  return make_taps(g, h_im, w_im);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// B fragment staging: 16 source rows x COLS columns of a row-major [rows][ld] matrix -> fragment order (see sparse_conv.hip)
template <int NB>
__device__ __forceinline__ void stage16(const float* __restrict__ src, int ld, int rows_valid, int cols_valid, float* __restrict__ lds) {
  constexpr int COLS = NB * 16;
  for (int e = threadIdx.x; e < 16 * (COLS / 4); e += 256) {
    const int c = e / (COLS / 4), col = (e % (COLS / 4)) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < rows_valid) {
      const float* p = src + static_cast<long long>(c) * ld + col;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (col + u < cols_valid) v[u] = p[u];
    }
    const int gq = (c >> 2) & 3, t = c & 3;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int cc = col + u;
      lds[(((cc >> 4) * 64) + gq * 16 + (cc & 15)) * 4 + t] = v[u];
    }
  }
}

// ---------------------------------------------------------------- forward ----------------------------
// x NHWC [B,H,W,Cin], wt [K][Cin][Cout], y NHWC [B*Ho*Wo, Cout]; block = 64 output pixels, wave = 16 pixels x all Cout.
// The reduction runs over steps (kernel tap k, 16 input channels): the weight chunk of a step is a [16][Cout] MFMA B fragment in LDS,
// double buffered — while the MFMAs of step s run, the weight chunk and the bilinear taps of step s + 1 are already in flight
// (global loads into registers), and one barrier per step hands the buffers over.  Round 1-2 staged every chunk between two barriers
// and gathered the taps right before their MFMAs: 36 - 41 TFLOP/s (0.23 - 0.26 of the fp32-MFMA peak).
template <int NB>
__global__ __launch_bounds__(256) void dcn_fwd_k(DcnGeom g, const float* __restrict__ x, const float* __restrict__ wt,
                                                 const float* __restrict__ bias, const float* __restrict__ offset,
                                                 const float* __restrict__ mask, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 x (16 x NB*16) fragments
  constexpr int COLS = NB * 16, FRAG = 16 * COLS, IT = (NB + 3) / 4;   // IT float4 of the chunk per thread
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, gq = lane >> 4;
  const long long npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  const long long pix = static_cast<long long>(blockIdx.x) * 64 + wave * 16 + r;
  const bool live = pix < npix;
  const int b = live ? static_cast<int>(pix / (g.Ho * g.Wo)) : 0;
  const int rem = live ? static_cast<int>(pix % (g.Ho * g.Wo)) : 0;
  const int ho = rem / g.Wo, wo = rem % g.Wo;
  const int K = g.kh * g.kw, cpg = g.Cin / g.dg, chunks = g.Cin / 16, steps = K * chunks;   // Cin % 16 == 0 (dcn_check)
  const float* xb = x + static_cast<long long>(b) * g.H * g.W * g.Cin;
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // weight chunk of a step: rows c0 .. c0 + 15 of wt[k], this thread's float4s (row c, columns col .. col + 3)
  auto load_w = [&](int step, float4 (&v)[IT]) {
    const float* src = wt + (static_cast<long long>(step / chunks) * g.Cin + (step % chunks) * 16) * g.Cout;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = threadIdx.x + i * 256;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < 16 * (COLS / 4)) {
        const int c = e / (COLS / 4), col = (e % (COLS / 4)) * 4;
        const float* p = src + static_cast<long long>(c) * g.Cout + col;
        if (col + 3 < g.Cout) v[i] = ld4(p);
        else {
          if (col < g.Cout) v[i].x = p[0];
          if (col + 1 < g.Cout) v[i].y = p[1];
          if (col + 2 < g.Cout) v[i].z = p[2];
        }
      }
    }
  };
  auto store_w = [&](const float4 (&v)[IT], float* buf) {   // -> fragment order (see sparse_conv.hip)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = threadIdx.x + i * 256;
      if (e < 16 * (COLS / 4)) {
        const int c = e / (COLS / 4), col = (e % (COLS / 4)) * 4;
        const int fq = (c >> 2) & 3, t = c & 3;
        const float u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          const int cc = col + k4;
          buf[(((cc >> 4) * 64) + fq * 16 + (cc & 15)) * 4 + t] = u[k4];
        }
      }
    }
  };
  // modulated bilinear sample of this lane's pixel, channels c0 + 4 gq .. + 3 of the step
  Taps t;
  float m = 0.f;
  int t_of = -1;   // the (k, dgi) the taps belong to
  auto gather = [&](int step) -> float4 {
    const int k = step / chunks, c = (step % chunks) * 16 + 4 * gq, dgi = c / cpg;
    if (k * g.dg + dgi != t_of) {
      t_of = k * g.dg + dgi;
      m = 0.f;
      t.valid = 0;
      t.o[0] = t.o[1] = t.o[2] = t.o[3] = -1;
      t.w[0] = t.w[1] = t.w[2] = t.w[3] = 0.f;
      if (live) t = pixel_taps(g, offset, mask, b, ho, wo, k, dgi, &m);
    }
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (t.o[q] >= 0) {
        const float4 v = ld4(xb + static_cast<long long>(t.o[q]) * g.Cin + c);
        a.x += t.w[q] * v.x; a.y += t.w[q] * v.y; a.z += t.w[q] * v.z; a.w += t.w[q] * v.w;
      }
    a.x *= m; a.y *= m; a.z *= m; a.w *= m;
    return a;
  };
  float4 wv[IT];
  load_w(0, wv);
  float4 a = gather(0);
  store_w(wv, lds);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    float* cur = lds + (s & 1) * FRAG;
    float4 a_next = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s + 1 < steps) {
      load_w(s + 1, wv);
      a_next = gather(s + 1);
    }
    float4 bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bv[nb] = *reinterpret_cast<const float4*>(&cur[(nb * 64 + lane) * 4]);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv[nb].x, acc[nb], 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv[nb].y, acc[nb], 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bv[nb].z, acc[nb], 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bv[nb].w, acc[nb], 0, 0, 0);
    if (s + 1 < steps) store_w(wv, lds + ((s + 1) & 1) * FRAG);   // the other buffer: every wave left it at the barrier of step s - 1
    __syncthreads();
    a = a_next;
  }
  const int n = lane & 15, q = lane >> 4;
  const long long row0 = static_cast<long long>(blockIdx.x) * 64 + wave * 16;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int col = nb * 16 + n;
    if (col >= g.Cout) continue;
    const float bb = bias ? bias[col] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const long long row = row0 + q * 4 + reg;
      if (row < npix) y[row * g.Cout + col] = acc[nb][reg] + bb;
    }
  }
}

// ---------------------------------------------------------------- backward: data / offset / mask -----
// One wave = 16 output pixels.  dy fragment (A operand, Cout values per pixel) stays in registers; for every tap k and
// 16-channel chunk the wave forms dcol[16 pixels][16 ci] with Cout/4 MFMAs against W_k^T (LDS), then turns it into
// grad_x (atomics on the 4 taps), grad_mask and grad_offset (reduced over ci with shuffles, accumulated over chunks).
template <int JO>  // Cout padded to 16*JO
__global__ __launch_bounds__(256) void dcn_bwd_data_k(DcnGeom g, const float* __restrict__ x, const float* __restrict__ wt,
                                                      const float* __restrict__ offset, const float* __restrict__ mask,
                                                      const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ doff,
                                                      float* __restrict__ dmask) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // B fragment: [JO][64 lanes][4]: B[co][ci]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, gq = lane >> 4;
  const long long npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  const long long row0 = static_cast<long long>(blockIdx.x) * 64 + wave * 16;
  const long long pix = row0 + r;
  const bool live = pix < npix;
  const int K = g.kh * g.kw, cpg = g.Cin / g.dg;
  const long long plane = static_cast<long long>(g.Ho) * g.Wo;
  // A fragment: lane (r, gq) holds dy[pix_r][16j + 4gq .. +3]
  float4 av[JO];
#pragma unroll
  for (int j = 0; j < JO; ++j) {
    av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = 16 * j + 4 * gq;
    if (live && c < g.Cout) av[j] = ld4(dy + pix * g.Cout + c);
  }
  // epilogue view: lane (n = ci within chunk, q) owns pixels row0 + 4q + reg
  const int n = lane & 15, q = lane >> 4;
  for (int k = 0; k < K; ++k) {
    for (int dgi = 0; dgi < g.dg; ++dgi) {
      // taps of the 4 pixels this lane post-processes
      Taps tp[4];
      float mk[4];
      int pb[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const long long p = row0 + q * 4 + reg;
        mk[reg] = 0.f;
        pb[reg] = 0;
        tp[reg].valid = 0;
        tp[reg].o[0] = tp[reg].o[1] = tp[reg].o[2] = tp[reg].o[3] = -1;
        tp[reg].w[0] = tp[reg].w[1] = tp[reg].w[2] = tp[reg].w[3] = 0.f;
        tp[reg].lh = tp[reg].lw = 0.f;
        if (p < npix) {
          const int b = static_cast<int>(p / plane), rem = static_cast<int>(p % plane);
          pb[reg] = b;
          tp[reg] = pixel_taps(g, offset, mask, b, rem / g.Wo, rem % g.Wo, k, dgi, &mk[reg]);
        }
      }
      float gm[4] = {0.f, 0.f, 0.f, 0.f}, gh[4] = {0.f, 0.f, 0.f, 0.f}, gw[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c0 = dgi * cpg; c0 < (dgi + 1) * cpg; c0 += 16) {
        __syncthreads();
        // B[co][ci] = wt[k][c0 + ci][co]  -> fragment order with source index co: element at ((co>>4)*64 + ((co>>2)&3)*16 + ci)*4 + (co&3)
        for (int e = threadIdx.x; e < 16 * (JO * 4); e += 256) {
          const int ci = e / (JO * 4), co = (e % (JO * 4)) * 4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c0 + ci < (dgi + 1) * cpg) {
            const float* p = wt + (static_cast<long long>(k) * g.Cin + c0 + ci) * g.Cout + co;
            if (co + 3 < g.Cout) v = ld4(p);
            else {
              if (co < g.Cout) v.x = p[0];
              if (co + 1 < g.Cout) v.y = p[1];
              if (co + 2 < g.Cout) v.z = p[2];
            }
          }
          *reinterpret_cast<float4*>(&lds[(((co >> 4) * 64) + ((co >> 2) & 3) * 16 + ci) * 4]) = v;
        }
        __syncthreads();
        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < JO; ++j) {
          const float4 bv = *reinterpret_cast<const float4*>(&lds[(j * 64 + lane) * 4]);
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv.x, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv.y, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv.z, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv.w, d, 0, 0, 0);
        }
        // d[reg] = dcol[pixel row0+4q+reg][ci = c0 + n]
        const int ci = c0 + n;
        const bool cok = ci < (dgi + 1) * cpg;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const Taps& t = tp[reg];
          if (!t.valid || !cok) continue;
          const float gcol = d[reg];
          const float* xb = x + static_cast<long long>(pb[reg]) * g.H * g.W * g.Cin + ci;
          float* dxb = dx + static_cast<long long>(pb[reg]) * g.H * g.W * g.Cin + ci;
          float v[4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            v[qq] = t.o[qq] >= 0 ? xb[static_cast<long long>(t.o[qq]) * g.Cin] : 0.f;
            if (t.o[qq] >= 0) atomicAdd(&dxb[static_cast<long long>(t.o[qq]) * g.Cin], gcol * mk[reg] * t.w[qq]);
          }
          const float val = t.w[0] * v[0] + t.w[1] * v[1] + t.w[2] * v[2] + t.w[3] * v[3];
          gm[reg] += gcol * val;
          // d val / d h = -hw*v1 - lw*v2 + hw*v3 + lw*v4 ; d val / d w = -hh*v1 + hh*v2 - lh*v3 + lh*v4 (mdmcn_get_coordinate_weight)
          const float hw = 1.f - t.lw, hh = 1.f - t.lh;
          gh[reg] += gcol * mk[reg] * (-hw * v[0] - t.lw * v[1] + hw * v[2] + t.lw * v[3]);
          gw[reg] += gcol * mk[reg] * (-hh * v[0] + hh * v[1] - t.lh * v[2] + t.lh * v[3]);
        }
      }
      // reduce over the 16 ci lanes (n) and store: (pixel, k, dgi) is owned by exactly one wave
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float a = gm[reg], bh = gh[reg], bw = gw[reg];
#pragma unroll
        for (int s = 1; s < 16; s <<= 1) {
          a += __shfl_xor(a, s, 64); bh += __shfl_xor(bh, s, 64); bw += __shfl_xor(bw, s, 64);
        }
        const long long p = row0 + q * 4 + reg;
        if (n == 0 && p < npix) {
          const int b = static_cast<int>(p / plane);
          const long long pos = p % plane;
          dmask[((static_cast<long long>(b) * g.dg + dgi) * K + k) * plane + pos] = a;
          float* ob = doff + (static_cast<long long>(b) * g.dg + dgi) * 2 * K * plane;
          ob[(2 * k) * plane + pos] = bh;
          ob[(2 * k + 1) * plane + pos] = bw;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- backward: weight ---------------------
// grid (pixel chunks, K, Cin/16); block 256 = 4 waves splitting the chunk's pixels; A_mfma[m = ci][kk = pixel] =
// modulated sample, B_mfma[kk = pixel][n = co] = dy; partial [chunk][k][ci][co] tiles, reduced by wgrad-style sum.
template <int NB>
__global__ __launch_bounds__(256) void dcn_bwd_weight_k(DcnGeom g, const float* __restrict__ x, const float* __restrict__ offset,
                                                        const float* __restrict__ mask, const float* __restrict__ dy, int pix_per_chunk,
                                                        float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [16][NB*16]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, m = lane & 15, gq = lane >> 4;
  const int k = blockIdx.y, c0 = blockIdx.z * 16, K = g.kh * g.kw, cpg = g.Cin / g.dg;
  const int dgi = c0 / cpg;
  const long long npix = static_cast<long long>(g.B) * g.Ho * g.Wo, plane = static_cast<long long>(g.Ho) * g.Wo;
  constexpr int TILE = 16 * NB * 16;
  for (int e = threadIdx.x; e < TILE; e += 256) red[e] = 0.f;
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long p_begin = static_cast<long long>(blockIdx.x) * pix_per_chunk + static_cast<long long>(w) * (pix_per_chunk / 4);
  const long long p_end = min(p_begin + pix_per_chunk / 4, npix);
  const int ci = c0 + m;
  for (long long p0 = p_begin; p0 < p_end; p0 += 4) {
    const long long p = p0 + gq;
    float a = 0.f;
    float bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bv[nb] = 0.f;
    if (p < p_end) {
      const int b = static_cast<int>(p / plane), rem = static_cast<int>(p % plane);
      float mk;
      const Taps t = pixel_taps(g, offset, mask, b, rem / g.Wo, rem % g.Wo, k, dgi, &mk);
      if (ci < g.Cin) {
        const float* xb = x + static_cast<long long>(b) * g.H * g.W * g.Cin + ci;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          if (t.o[qq] >= 0) a += t.w[qq] * xb[static_cast<long long>(t.o[qq]) * g.Cin];
        a *= mk;
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int co = nb * 16 + m;
        if (co < g.Cout) bv[nb] = dy[p * g.Cout + co];
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[nb], acc[nb], 0, 0, 0);
  }
  __syncthreads();
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) atomicAdd(&red[(gq * 4 + reg) * (NB * 16) + nb * 16 + m], acc[nb][reg]);
  __syncthreads();
  float* out = partial + ((static_cast<long long>(blockIdx.x) * K + k) * g.Cin + c0) * g.Cout;
  for (int e = threadIdx.x; e < TILE; e += 256) {
    const int cr = e / (NB * 16), cc = e % (NB * 16);
    if (c0 + cr < g.Cin && cc < g.Cout) out[static_cast<long long>(cr) * g.Cout + cc] = red[e];
  }
}

__global__ void dcn_reduce_k(const float* __restrict__ partial, int chunks, long long per_chunk, float* __restrict__ out) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= per_chunk) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += partial[c * per_chunk + t];
  out[t] = s;
}

// ================================================================ round 4 kernels ==================

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_dst) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds_dst)));
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

struct Corner4 {
  unsigned o[4];
  float w[4];
  float m;
};

__device__ __forceinline__ void make_corners(const DcnGeom& g, bool live, int b, float h_im, float w_im, float m, unsigned lane_bytes, Corner4& t) {
  const bool valid = live && (h_im > -1.f && w_im > -1.f && h_im < static_cast<float>(g.H) && w_im < static_cast<float>(g.W));
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h_low = static_cast<int>(hf), w_low = static_cast<int>(wf);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
  const bool hl = valid && h_low >= 0, hhi = valid && h_high <= g.H - 1, wl = w_low >= 0, whi = w_high <= g.W - 1;
  t.w[0] = (hl && wl) ? hh * hw : 0.f;
  t.w[1] = (hl && whi) ? hh * lw : 0.f;
  t.w[2] = (hhi && wl) ? lh * hw : 0.f;
  t.w[3] = (hhi && whi) ? lh * lw : 0.f;
  t.m = valid ? m : 0.f;
  const int h0 = min(max(h_low, 0), g.H - 1), h1 = min(max(h_high, 0), g.H - 1);
  const int w0 = min(max(w_low, 0), g.W - 1), w1 = min(max(w_high, 0), g.W - 1);
  const unsigned row0 = static_cast<unsigned>((b * g.H + h0) * g.W), row1 = static_cast<unsigned>((b * g.H + h1) * g.W);
  const unsigned pitch = static_cast<unsigned>(g.Cin) * 4u;
  t.o[0] = (row0 + w0) * pitch + lane_bytes;
  t.o[1] = (row0 + w1) * pitch + lane_bytes;
  t.o[2] = (row1 + w0) * pitch + lane_bytes;
  t.o[3] = (row1 + w1) * pitch + lane_bytes;
}

__device__ __forceinline__ f32x4 ldx4(const char* base, unsigned off) { return *reinterpret_cast<const f32x4*>(base + off); }

// x NHWC [B,H,W,Cin]; wt_oc [K][Cout][Cin]; y NHWC [npix][Cout].
// Block = 4 waves, wave = MB*16 pixels x NB*16 output channels (grid.y = column blocks of NB*16).  Product formed transposed:
// A = weights (rows = output channels, LDS), B = modulated bilinear samples (columns = pixels, registers of the lane that gathered them).
template <int NB, int MB>
__global__ __launch_bounds__(256, 2) void dcn_fwd2_k(DcnGeom g, const float* __restrict__ x, const float* __restrict__ wt_oc,
                                                     const float* __restrict__ bias, const float* __restrict__ offset,
                                                     const float* __restrict__ mask, float* __restrict__ y, long long pix_base) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x NB x 256 floats
  constexpr int FRAG = NB * 256;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 15, gq = lane >> 4;
  const long long npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  const int plane = g.Ho * g.Wo;
  const int K = g.kh * g.kw, cpg = g.Cin / g.dg, cps = cpg / 16, segs = K * g.dg, steps = segs * cps;
  const int col0 = blockIdx.y * (NB * 16);
  const long long tile0 = pix_base + static_cast<long long>(blockIdx.x) * (64 * MB) + wave * (16 * MB);
  // this lane's pixels
  bool live[MB];
  int pb[MB], pho[MB], pwo[MB], ppos[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const long long pix = tile0 + mb * 16 + n;
    live[mb] = pix < npix;
    const long long pp = live[mb] ? pix : 0;
    pb[mb] = static_cast<int>(pp / plane);
    ppos[mb] = static_cast<int>(pp % plane);
    pho[mb] = ppos[mb] / g.Wo;
    pwo[mb] = ppos[mb] % g.Wo;
  }
  // weight rows this lane feeds to the LDS-DMA: piece nb (1 KB = 64 lanes x 16 B), lane (n, gq) <- wt_oc[k][col0 + nb*16 + n][c0 + 4 gq ..]
  constexpr int NDMA = (NB + 3) / 4;
  const float* wrow[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int nb = wave + 4 * i;
    const int co = min(col0 + nb * 16 + n, g.Cout - 1);
    wrow[i] = wt_oc + static_cast<long long>(co) * g.Cin + 4 * gq;
  }
  auto dma = [&](int k, int c0, float* buf) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int nb = wave + 4 * i;
      if (nb < NB) glds16(wrow[i] + (static_cast<long long>(k) * g.Cout * g.Cin + c0), buf + nb * 256);
    }
  };
  Corner4 t[MB];
  float roh[MB], row_[MB], rom[MB];   // raw offsets / mask of the NEXT segment
  auto load_offsets = [&](int k, int dgi) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const float* ob = offset + (static_cast<long long>(pb[mb]) * g.dg + dgi) * 2 * K * plane + ppos[mb];
      roh[mb] = ob[static_cast<long long>(2 * k) * plane];
      row_[mb] = ob[static_cast<long long>(2 * k + 1) * plane];
      rom[mb] = mask[((static_cast<long long>(pb[mb]) * g.dg + dgi) * K + k) * plane + ppos[mb]];
    }
  };
  auto set_taps = [&](int k) {
    const int i = k / g.kw, j = k % g.kw;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const float h_im = static_cast<float>(pho[mb] * g.sh - g.ph + i * g.dh) + roh[mb];
      const float w_im = static_cast<float>(pwo[mb] * g.sw - g.pw + j * g.dw) + row_[mb];
      make_corners(g, live[mb], pb[mb], h_im, w_im, rom[mb], 16u * gq, t[mb]);
    }
  };
  f32x4 raw[MB][4];
  auto gather = [&](int c0) {
    const char* xc = reinterpret_cast<const char*>(x + c0);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int q = 0; q < 4; ++q) raw[mb][q] = ldx4(xc, t[mb].o[q]);
  };
  auto combine = [&](f32x4 (&bs)[MB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      f32x4 a;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = t[mb].w[0] * raw[mb][0][e];
        v = __builtin_fmaf(t[mb].w[1], raw[mb][1][e], v);
        v = __builtin_fmaf(t[mb].w[2], raw[mb][2][e], v);
        v = __builtin_fmaf(t[mb].w[3], raw[mb][3][e], v);
        a[e] = v * t[mb].m;
      }
      bs[mb] = a;
    }
  };
  f32x4 acc[NB][MB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: segment 0 taps, step 0 operands
  load_offsets(0, 0);
  set_taps(0);
  if (segs > 1) load_offsets(g.dg > 1 ? 0 : 1, g.dg > 1 ? 1 : 0);   // segment 1 = (tap, group) after (0, 0)
  dma(0, 0, lds);
  gather(0);
  f32x4 bs[MB];
  combine(bs);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int k = 0, dgi = 0, c = 0;   // position of step s
  for (int s = 0; s < steps; ++s) {
    float* cur = lds + (s & 1) * FRAG;
    // position of step s + 1
    int c1 = c + 1, dg1 = dgi, k1 = k;
    if (c1 == cps) { c1 = 0; ++dg1; if (dg1 == g.dg) { dg1 = 0; ++k1; } }
    const bool more = s + 1 < steps;
    if (more) {
      if (c1 == 0) set_taps(k1);                    // offsets were fetched at the first step of this segment
      dma(k1, (dg1 * cps + c1) * 16, lds + ((s + 1) & 1) * FRAG);
      gather((dg1 * cps + c1) * 16);
      if (c1 == 0) {                                // fetch the offsets of the segment after that
        int dg2 = dg1 + 1, k2 = k1;
        if (dg2 == g.dg) { dg2 = 0; ++k2; }
        if (k2 < K) load_offsets(k2, dg2);
      }
    }
    f32x4 af[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) af[nb] = *reinterpret_cast<const f32x4*>(&cur[(nb * 64 + lane) * 4]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[nb][j], bs[mb][j], acc[nb][mb], 0, 0, 0);
    if (more) combine(bs);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    c = c1; dgi = dg1; k = k1;
  }
  // epilogue: acc[nb][mb][reg] = y[pixel mb*16 + n][col0 + nb*16 + 4 gq + reg]
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const long long pix = tile0 + mb * 16 + n;
    if (pix >= npix) continue;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int col = col0 + nb * 16 + 4 * gq;
      if (col >= g.Cout) continue;
      f32x4 v = acc[nb][mb];
      float* dst = y + pix * g.Cout + col;
      if (col + 3 < g.Cout && (g.Cout & 3) == 0) {
        if (bias) { const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + col); v += bb; }
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < g.Cout) dst[e] = v[e] + (bias ? bias[col + e] : 0.f);
      }
    }
  }
}



// ---------------------------------------------------------------- backward, pass 1: column gradients -------------
// dcol[p, k, ci] = sum_co dy[p, co] * W[k][ci][co]  (the reference's gemm into `columns`, modulated_deform_conv_cuda.cu:196-204),
// formed transposed (A = W rows ci from LDS, B = dy rows of the wave's pixels, resident in registers for the whole kernel), so that a
// lane ends up with FOUR ADJACENT CHANNELS of ONE pixel: exactly the float4 it gathers from x at the four bilinear corners.  Epilogue per
// (tap, group, 32-channel block):  D_q += <dcol, x[corner q]>  (-> grad_mask, grad_offset at the end of the (tap, group) segment,
// modulated_deform_im2col_cuda.cuh:256-328) and  colg[p, k, ci] = dcol * mask  (what pass 2 sums into grad_input, :196-254).
struct CornerB {
  unsigned o[4];
  float w[4];
  float m, lh, lw;
  unsigned vb;   // bit q: corner q is inside the map (and the sample valid)
};

__device__ __forceinline__ void make_corners_b(const DcnGeom& g, bool live, int b, float h_im, float w_im, float m, unsigned lane_bytes, CornerB& t) {
  const bool valid = live && (h_im > -1.f && w_im > -1.f && h_im < static_cast<float>(g.H) && w_im < static_cast<float>(g.W));
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h_low = static_cast<int>(hf), w_low = static_cast<int>(wf);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
  const bool hl = valid && h_low >= 0, hhi = valid && h_high <= g.H - 1, wl = w_low >= 0, whi = w_high <= g.W - 1;
  t.w[0] = (hl && wl) ? hh * hw : 0.f;
  t.w[1] = (hl && whi) ? hh * lw : 0.f;
  t.w[2] = (hhi && wl) ? lh * hw : 0.f;
  t.w[3] = (hhi && whi) ? lh * lw : 0.f;
  t.vb = (hl && wl ? 1u : 0u) | (hl && whi ? 2u : 0u) | (hhi && wl ? 4u : 0u) | (hhi && whi ? 8u : 0u);
  t.m = valid ? m : 0.f;
  t.lh = lh; t.lw = lw;
  const int h0 = min(max(h_low, 0), g.H - 1), h1 = min(max(h_high, 0), g.H - 1);
  const int w0 = min(max(w_low, 0), g.W - 1), w1 = min(max(w_high, 0), g.W - 1);
  const unsigned row0 = static_cast<unsigned>((b * g.H + h0) * g.W), row1 = static_cast<unsigned>((b * g.H + h1) * g.W);
  const unsigned pitch = static_cast<unsigned>(g.Cin) * 4u;
  t.o[0] = (row0 + w0) * pitch + lane_bytes;
  t.o[1] = (row0 + w1) * pitch + lane_bytes;
  t.o[2] = (row1 + w0) * pitch + lane_bytes;
  t.o[3] = (row1 + w1) * pitch + lane_bytes;
}

// JO = ceil(Cout / 16); NBP = 16-pixel blocks per wave; block = 4 waves = 64 * NBP pixels; step = (tap, group, 16 * MC input channels)
template <int JO, int NBP, int MC>
__global__ __launch_bounds__(256, 2) void dcn_bwd_col_k(DcnGeom g, const float* __restrict__ x, const float* __restrict__ wt,
                                                        const float* __restrict__ offset, const float* __restrict__ mask,
                                                        const float* __restrict__ dy, float* __restrict__ colg, float* __restrict__ doff,
                                                        float* __restrict__ dmask, long long pix_base) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x (2 x JO) pieces of 256 floats
  constexpr int FRAG = MC * JO * 256, CS = 16 * MC;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 15, gq = lane >> 4;
  const long long npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  const int plane = g.Ho * g.Wo;
  const int K = g.kh * g.kw, cpg = g.Cin / g.dg, cps = cpg / CS, segs = K * g.dg, steps = segs * cps;   // cpg % CS == 0
  const long long tile0 = pix_base + static_cast<long long>(blockIdx.x) * (64 * NBP) + wave * (16 * NBP);
  bool live[NBP];
  int pb[NBP], pho[NBP], pwo[NBP], ppos[NBP];
  unsigned cbase[NBP];   // byte offset of colg[p][0][4 gq]
  f32x4 dyr[NBP][JO];
#pragma unroll
  for (int np = 0; np < NBP; ++np) {
    const long long pix = tile0 + np * 16 + n;
    live[np] = pix < npix;
    const long long pp = live[np] ? pix : 0;
    pb[np] = static_cast<int>(pp / plane);
    ppos[np] = static_cast<int>(pp % plane);
    pho[np] = ppos[np] / g.Wo;
    pwo[np] = ppos[np] % g.Wo;
    cbase[np] = static_cast<unsigned>((pp * K * g.Cin + 4 * gq) * 4);
#pragma unroll
    for (int jb = 0; jb < JO; ++jb) {
      const int c = 16 * jb + 4 * gq;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live[np]) {
        const float* p = dy + pp * g.Cout + c;
        if (c + 3 < g.Cout && (g.Cout & 3) == 0) v = *reinterpret_cast<const f32x4*>(p);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c + e < g.Cout) v[e] = p[e];
        }
      }
      dyr[np][jb] = v;
    }
  }
  // LDS-DMA pieces: piece (mc, jb), lane (n, gq) <- wt[k][ci0 + mc*16 + n][16 jb + 4 gq ..]; pieces dealt to the waves round robin
  constexpr int NP = MC * JO, NDMA = (NP + 3) / 4;
  auto dma = [&](int k, int ci0, float* buf) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int pc = wave + 4 * i;
      if (pc < NP) {
        const int mc = pc / JO, jb = pc % JO;
        const int co = min(16 * jb + 4 * gq, g.Cout - 4);
        glds16(wt + (static_cast<long long>(k) * g.Cin + ci0 + mc * 16 + n) * g.Cout + co, buf + pc * 256);
      }
    }
  };
  CornerB t[NBP];
  float roh[NBP], row_[NBP], rom[NBP];
  auto load_offsets = [&](int k, int dgi) {
#pragma unroll
    for (int np = 0; np < NBP; ++np) {
      const float* ob = offset + (static_cast<long long>(pb[np]) * g.dg + dgi) * 2 * K * plane + ppos[np];
      roh[np] = ob[static_cast<long long>(2 * k) * plane];
      row_[np] = ob[static_cast<long long>(2 * k + 1) * plane];
      rom[np] = mask[((static_cast<long long>(pb[np]) * g.dg + dgi) * K + k) * plane + ppos[np]];
    }
  };
  auto set_taps = [&](int k) {
    const int i = k / g.kw, j = k % g.kw;
#pragma unroll
    for (int np = 0; np < NBP; ++np) {
      const float h_im = static_cast<float>(pho[np] * g.sh - g.ph + i * g.dh) + roh[np];
      const float w_im = static_cast<float>(pwo[np] * g.sw - g.pw + j * g.dw) + row_[np];
      make_corners_b(g, live[np], pb[np], h_im, w_im, rom[np], 16u * gq, t[np]);
    }
  };
  float Dq[NBP][4];
#pragma unroll
  for (int np = 0; np < NBP; ++np) Dq[np][0] = Dq[np][1] = Dq[np][2] = Dq[np][3] = 0.f;
  load_offsets(0, 0);
  set_taps(0);
  if (segs > 1) load_offsets(g.dg > 1 ? 0 : 1, g.dg > 1 ? 1 : 0);
  dma(0, 0, lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int k = 0, dgi = 0, c = 0;
  for (int s = 0; s < steps; ++s) {
    float* cur = lds + (s & 1) * FRAG;
    int c1 = c + 1, dg1 = dgi, k1 = k;
    if (c1 == cps) { c1 = 0; ++dg1; if (dg1 == g.dg) { dg1 = 0; ++k1; } }
    const bool more = s + 1 < steps;
    const int ci0 = (dgi * cps + c) * CS;
    if (more) dma(k1, (dg1 * cps + c1) * CS, lds + ((s + 1) & 1) * FRAG);
    // x at the four corners of this step's channels (consumed by the epilogue, after the MFMAs)
    f32x4 raw[NBP][MC][4];
    {
      const char* xc = reinterpret_cast<const char*>(x + ci0);
#pragma unroll
      for (int np = 0; np < NBP; ++np)
#pragma unroll
        for (int mc = 0; mc < MC; ++mc)
#pragma unroll
          for (int q = 0; q < 4; ++q) raw[np][mc][q] = ldx4(xc + mc * 64, t[np].o[q]);
    }
    f32x4 acc[MC][NBP];
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int np = 0; np < NBP; ++np) acc[mc][np] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jb = 0; jb < JO; ++jb) {
      f32x4 af[MC];
#pragma unroll
      for (int mc = 0; mc < MC; ++mc) af[mc] = *reinterpret_cast<const f32x4*>(&cur[((mc * JO + jb) * 64 + lane) * 4]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mc = 0; mc < MC; ++mc)
#pragma unroll
          for (int np = 0; np < NBP; ++np) acc[mc][np] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mc][j], dyr[np][jb][j], acc[mc][np], 0, 0, 0);
    }
    // epilogue: acc[mc][np][e] = dcol[pixel np*16 + n][ci0 + mc*16 + 4 gq + e]
    {
      char* cg = reinterpret_cast<char*>(colg + (static_cast<long long>(k) * g.Cin + ci0));
#pragma unroll
      for (int np = 0; np < NBP; ++np)
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) {
          const f32x4 d = acc[mc][np];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float a = Dq[np][q];
#pragma unroll
            for (int e = 0; e < 4; ++e) a = __builtin_fmaf(d[e], raw[np][mc][q][e], a);
            Dq[np][q] = a;
          }
          if (live[np]) *reinterpret_cast<f32x4*>(cg + mc * 64 + cbase[np]) = d * t[np].m;
        }
    }
    if (c1 == 0 || !more) {
      // end of the (tap, group) segment: reduce D_q over the four channel quarters (lanes n, n+16, n+32, n+48), write grad_mask / grad_offset
#pragma unroll
      for (int np = 0; np < NBP; ++np) {
        float D[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = Dq[np][q];
          a += __shfl_xor(a, 16, 64);
          a += __shfl_xor(a, 32, 64);
          D[q] = (t[np].vb >> q) & 1u ? a : 0.f;
          Dq[np][q] = 0.f;
        }
        if (gq == 0 && live[np]) {
          const float lh = t[np].lh, lw = t[np].lw, hh = 1.f - lh, hw = 1.f - lw, m = t[np].m;
          const float gm = t[np].w[0] * D[0] + t[np].w[1] * D[1] + t[np].w[2] * D[2] + t[np].w[3] * D[3];
          const float gh = m * (-hw * D[0] - lw * D[1] + hw * D[2] + lw * D[3]);
          const float gw = m * (-hh * D[0] + hh * D[1] - lh * D[2] + lh * D[3]);
          dmask[((static_cast<long long>(pb[np]) * g.dg + dgi) * K + k) * plane + ppos[np]] = gm;
          float* ob = doff + (static_cast<long long>(pb[np]) * g.dg + dgi) * 2 * K * plane + ppos[np];
          ob[static_cast<long long>(2 * k) * plane] = gh;
          ob[static_cast<long long>(2 * k + 1) * plane] = gw;
        }
      }
      if (more) {
        set_taps(k1);
        int dg2 = dg1 + 1, k2 = k1;
        if (dg2 == g.dg) { dg2 = 0; ++k2; }
        if (k2 < K) load_offsets(k2, dg2);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    c = c1; dgi = dg1; k = k1;
  }
}


// ---------------------------------------------------------------- backward, pass 2: grad_input in gather form -------
// The reference scatters every column gradient to its four bilinear corners with atomicAdd (modulated_deform_im2col_cuda.cuh:196-254):
// 1.3 G float atomics at the MGAF head's shape and a sum whose order changes from run to run.  Here every input pixel (and deformable
// group) owns a list of the samples that touch it - built with INTEGER atomics (count, scan, fill: the totals are exact whatever the
// order), put into ascending sample order inside the gathering wave - and sums them itself: no float atomics, no zero fill, and the
// same bits on every run.
//   entry = {src = ((pixel * K + tap) << 2) | corner, bilinear weight of that corner}
__device__ __forceinline__ void sample_corners(const DcnGeom& g, int ho, int wo, int k, float off_h, float off_w, int (&ty)[4], int (&tx)[4],
                                               float (&wq)[4]) {
  const int i = k / g.kw, j = k % g.kw;
  const float h_im = static_cast<float>(ho * g.sh - g.ph + i * g.dh) + off_h;
  const float w_im = static_cast<float>(wo * g.sw - g.pw + j * g.dw) + off_w;
  const bool valid = (h_im > -1.f && w_im > -1.f && h_im < static_cast<float>(g.H) && w_im < static_cast<float>(g.W));
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h_low = static_cast<int>(hf), w_low = static_cast<int>(wf);
  const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
  ty[0] = ty[1] = h_low; ty[2] = ty[3] = h_low + 1;
  tx[0] = tx[2] = w_low; tx[1] = tx[3] = w_low + 1;
  wq[0] = hh * hw; wq[1] = hh * lw; wq[2] = lh * hw; wq[3] = lh * lw;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (!(valid && ty[q] >= 0 && ty[q] <= g.H - 1 && tx[q] >= 0 && tx[q] <= g.W - 1)) wq[q] = 0.f;   // weight 0 = no entry
}

// one thread per sample (b, group, tap, output pixel): FILL = 0 counts the entries of every target, FILL = 1 writes them
// (cursor[t] starts at the list's first slot and ends at its end: list t = [cursor[t - 1], cursor[t]) afterwards)
template <int FILL>
__global__ __launch_bounds__(256) void dcn_index_k(DcnGeom g, const float* __restrict__ offset, int* __restrict__ cursor, uint2* __restrict__ entries) {
  const int plane = g.Ho * g.Wo, K = g.kh * g.kw;
  const long long total = static_cast<long long>(g.B) * g.dg * K * plane;
  const long long id = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (id >= total) return;
  const int pos = static_cast<int>(id % plane);
  const int k = static_cast<int>((id / plane) % K);
  const int bd = static_cast<int>(id / (static_cast<long long>(plane) * K));   // b * dg + dgi
  const int b = bd / g.dg;
  const float* ob = offset + static_cast<long long>(bd) * 2 * K * plane + pos;
  const float off_h = ob[static_cast<long long>(2 * k) * plane], off_w = ob[static_cast<long long>(2 * k + 1) * plane];
  int ty[4], tx[4];
  float wq[4];
  sample_corners(g, pos / g.Wo, pos % g.Wo, k, off_h, off_w, ty, tx, wq);
  const long long pix = static_cast<long long>(b) * plane + pos;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (wq[q] == 0.f) continue;
    const long long t = (static_cast<long long>(bd) * g.H + ty[q]) * g.W + tx[q];
    const int slot = atomicAdd(&cursor[t], 1);
    if (FILL) entries[slot] = make_uint2(static_cast<unsigned>((pix * K + k) * 4 + q), __float_as_uint(wq[q]));
  }
}

// Lists longer than a wave (more than 64 samples on one input pixel) are put in order here, one workgroup per list, up to 4096 entries;
// beyond that the list keeps its fill order: the sum is still complete, only its rounding may differ between runs.
constexpr int kDcnLongList = 4096;
__global__ __launch_bounds__(256) void dcn_index_sort_long_k(const int* __restrict__ ends, long long ntargets, uint2* __restrict__ entries) {
  __shared__ uint2 buf[kDcnLongList];
  __shared__ int queue[256], qn;
  const long long per = (ntargets + gridDim.x - 1) / gridDim.x;
  const long long t0 = static_cast<long long>(blockIdx.x) * per, t1 = min(t0 + per, ntargets);
  for (long long base = t0; base < t1; base += 256) {
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    const long long t = base + threadIdx.x;
    if (t < t1) {
      const int n = ends[t] - (t ? ends[t - 1] : 0);
      if (n > 64 && n <= kDcnLongList) queue[atomicAdd(&qn, 1)] = static_cast<int>(t - base);
    }
    __syncthreads();
    const int nq = qn;
    for (int qi = 0; qi < nq; ++qi) {   // which list comes first does not matter: lists are disjoint
      const long long tt = base + queue[qi];
      const int start = tt ? ends[tt - 1] : 0, n = ends[tt] - start;
      for (int i = threadIdx.x; i < n; i += 256) buf[i] = entries[start + i];
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += 256) {
        const uint2 e = buf[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += buf[j].x < e.x;
        entries[start + rank] = e;
      }
      __syncthreads();
    }
  }
}

// one wave per (target pixel, deformable group); lane = VEC adjacent channels of the group
template <int VEC>
__global__ __launch_bounds__(256) void dcn_col2im_k(DcnGeom g, const float* __restrict__ colg, const int* __restrict__ ends,
                                                    const uint2* __restrict__ entries, float* __restrict__ dx) {
  __shared__ uint2 sorted[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long HW = static_cast<long long>(g.H) * g.W, ntargets = static_cast<long long>(g.B) * g.dg * HW;
  const long long t = static_cast<long long>(blockIdx.x) * 4 + wave;
  if (t >= ntargets) return;
  const int K = g.kh * g.kw, cpg = g.Cin / g.dg;
  const int bd = static_cast<int>(t / HW), b = bd / g.dg, dgi = bd % g.dg;
  const long long tp = static_cast<long long>(b) * HW + t % HW;   // pixel of x / dx
  const int start = t ? ends[t - 1] : 0, n = ends[t] - start;
  const uint2* list = entries + start;
  const bool in_lds = n <= 64;
  if (in_lds) {
    uint2 e = make_uint2(0xffffffffu, 0u);
    if (lane < n) e = list[lane];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += static_cast<unsigned>(__shfl(static_cast<int>(e.x), j, 64)) < e.x;
    if (lane < n) sorted[wave][rank] = e;
    // the wave reads its own slots only: LDS operations of one wave execute in order, no barrier
  }
  for (int c0 = 0; c0 < cpg; c0 += 64 * VEC) {
    const int c = c0 + lane * VEC;
    const bool act = c < cpg;
    const float* col = colg + static_cast<long long>(dgi) * cpg + (act ? c : 0);
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    int i = 0;
    for (; i + 4 <= n; i += 4) {
      uint2 e[4];
      float vals[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) e[u] = in_lds ? sorted[wave][i + u] : list[i + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* p = col + static_cast<long long>(e[u].x >> 2) * g.Cin;
        if constexpr (VEC == 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(p); vals[u][0] = v[0]; vals[u][1] = v[1]; vals[u][2] = v[2]; vals[u][3] = v[3]; }
        else if constexpr (VEC == 2) { const float2 v = *reinterpret_cast<const float2*>(p); vals[u][0] = v.x; vals[u][1] = v.y; }
        else vals[u][0] = p[0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = __builtin_fmaf(__uint_as_float(e[u].y), vals[u][v], acc[v]);
    }
    for (; i < n; ++i) {
      const uint2 e = in_lds ? sorted[wave][i] : list[i];
      const float* p = col + static_cast<long long>(e.x >> 2) * g.Cin;
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = __builtin_fmaf(__uint_as_float(e.y), p[v], acc[v]);
    }
    if (act) {
      float* out = dx + tp * g.Cin + dgi * cpg + c;
      if constexpr (VEC == 4) *reinterpret_cast<f32x4*>(out) = f32x4{acc[0], acc[1], acc[2], acc[3]};
      else if constexpr (VEC == 2) *reinterpret_cast<float2*>(out) = make_float2(acc[0], acc[1]);
      else out[0] = acc[0];
    }
  }
}

// ---------------------------------------------------------------- backward: weight gradient ------------------------
// dW[k][ci][co] = sum_p col[p, k, ci] * dy[p, co], col = mask * bilinear sample of x (the forward's operand).  One workgroup =
// (pixel range, tap, 128 input channels, 128 output channels); per 16-pixel step all 256 threads build the col tile [16 px][128 ci]
// (a thread = one pixel, four adjacent channels: four 16-byte corner loads, as in the forward) and copy the dy tile [16 px][128 co] into
// LDS, pixel-major; a wave then multiplies its 64 x 64 corner of the tile (A[ci][px] and B[px][co] read one float per lane and MFMA).
// Loads of step s + 1 are in flight during the MFMAs of step s.  Partial tiles of the pixel ranges are summed in a fixed order (dcn_reduce_k).
constexpr int kDwPitch = 132;   // floats per pixel row of an LDS tile: rows 4 pixels apart land 16 banks apart
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void dcn_bwd_weight2_k(DcnGeom g, const float* __restrict__ x, const float* __restrict__ offset,
                                                            const float* __restrict__ mask, const float* __restrict__ dy, int pix_per_block,
                                                            float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 buffers x (col tile 16 x 132 + dy tile 16 x 132)
  constexpr int TILE = 16 * kDwPitch, BUF = 2 * TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, gq = lane >> 4;
  const int K = g.kh * g.kw, cpg = g.Cin / g.dg, plane = g.Ho * g.Wo;
  const int ci_tiles = static_cast<int>((g.Cin + 127) / 128);
  const int k = blockIdx.y, ci0 = (blockIdx.z % ci_tiles) * 128, co0 = (blockIdx.z / ci_tiles) * 128;
  const long long npix = static_cast<long long>(g.B) * plane;
  const long long p_begin = static_cast<long long>(blockIdx.x) * pix_per_block, p_end = min(p_begin + pix_per_block, npix);
  // staging role: pixel tid >> 4 of the step, channel quads (tid & 15) and (tid & 15) + 16
  const int spx = tid >> 4, cq = tid & 15;
  const int i_k = k / g.kw, j_k = k % g.kw;
  f32x4 raw[2][4], dyv[2];
  float wq[2][4], mk[2];
  float roh[2], row_[2], rom[2];
  auto load_offsets = [&](long long p) {   // raw offsets / mask of pixel p for this thread's two channel quads
    const bool ok = p < p_end;
    const long long pp = ok ? p : p_begin;
    const int b = static_cast<int>(pp / plane), pos = static_cast<int>(pp % plane);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ci = ci0 + 64 * h + 4 * cq;
      const int dgi = min(ci, g.Cin - 1) / cpg;
      const float* ob = offset + (static_cast<long long>(b) * g.dg + dgi) * 2 * K * plane + pos;
      roh[h] = ob[static_cast<long long>(2 * k) * plane];
      row_[h] = ob[static_cast<long long>(2 * k + 1) * plane];
      rom[h] = mask[((static_cast<long long>(b) * g.dg + dgi) * K + k) * plane + pos];
    }
  };
  auto issue = [&](long long p) {   // corner loads + dy loads of pixel p (taps from the raw offsets fetched one step earlier)
    const bool ok = p < p_end;
    const long long pp = ok ? p : p_begin;
    const int b = static_cast<int>(pp / plane), pos = static_cast<int>(pp % plane);
    const int ho = pos / g.Wo, wo = pos % g.Wo;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ci = ci0 + 64 * h + 4 * cq;
      const bool cok = ok && ci < g.Cin;
      Corner4 t;
      const float h_im = static_cast<float>(ho * g.sh - g.ph + i_k * g.dh) + roh[h];
      const float w_im = static_cast<float>(wo * g.sw - g.pw + j_k * g.dw) + row_[h];
      make_corners(g, cok, b, h_im, w_im, rom[h], static_cast<unsigned>(min(ci, g.Cin - 4)) * 4u, t);
#pragma unroll
      for (int q = 0; q < 4; ++q) { raw[h][q] = ldx4(reinterpret_cast<const char*>(x), t.o[q]); wq[h][q] = t.w[q]; }
      mk[h] = t.m;
      const int co = co0 + 64 * h + 4 * cq;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok && co < g.Cout) v = *reinterpret_cast<const f32x4*>(dy + pp * g.Cout + co);   // Cout % 4 == 0
      dyv[h] = v;
    }
  };
  auto stage = [&](float* buf) {   // combine + write both tiles, pixel-major
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 a;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = wq[h][0] * raw[h][0][e];
        v = __builtin_fmaf(wq[h][1], raw[h][1][e], v);
        v = __builtin_fmaf(wq[h][2], raw[h][2][e], v);
        v = __builtin_fmaf(wq[h][3], raw[h][3][e], v);
        a[e] = v * mk[h];
      }
      *reinterpret_cast<f32x4*>(&buf[spx * kDwPitch + 64 * h + 4 * cq]) = a;
      *reinterpret_cast<f32x4*>(&buf[TILE + spx * kDwPitch + 64 * h + 4 * cq]) = dyv[h];
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wi = wave >> 1, wj = wave & 1;
  const long long steps = (p_end - p_begin + 15) / 16;
  if (steps > 0) {
    load_offsets(p_begin + spx);
    issue(p_begin + spx);
    if (steps > 1) load_offsets(p_begin + 16 + spx);
    stage(lds);
  }
  __syncthreads();
  for (long long s = 0; s < steps; ++s) {
    const float* cur = lds + (s & 1) * BUF;
    const bool more = s + 1 < steps;
    if (more) {
      issue(p_begin + (s + 1) * 16 + spx);
      if (s + 2 < steps) load_offsets(p_begin + (s + 2) * 16 + spx);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = cur[(4 * gq + j) * kDwPitch + (wi * 4 + a) * 16 + r];
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = cur[TILE + (4 * gq + j) * kDwPitch + (wj * 4 + c) * 16 + r];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[c], acc[a][c], 0, 0, 0);
    }
    if (more) stage(lds + ((s + 1) & 1) * BUF);
    __syncthreads();
  }
  // acc[a][c][e] = dW[k][ci0 + (wi*4 + a)*16 + 4 gq + e][co0 + (wj*4 + c)*16 + r]
  float* out = partial + (static_cast<long long>(blockIdx.x) * K + k) * g.Cin * g.Cout;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ci = ci0 + (wi * 4 + a) * 16 + 4 * gq + e;
      if (ci >= g.Cin) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int co = co0 + (wj * 4 + c) * 16 + r;
        if (co < g.Cout) out[static_cast<long long>(ci) * g.Cout + co] = acc[a][c][e];
      }
    }
}

static int dcn_check(const DcnGeom& g) {
  FV2P_REQUIRE(g.B >= 0 && g.H >= 1 && g.W >= 1 && g.Cin >= 1 && g.Cout >= 1 && g.kh >= 1 && g.kw >= 1 && g.sh >= 1 && g.sw >= 1 &&
                   g.dh >= 1 && g.dw >= 1 && g.dg >= 1 && g.Ho >= 1 && g.Wo >= 1,
               FV2P_EINVAL, "dcn: bad geometry");
  FV2P_REQUIRE(g.Cin % g.dg == 0 && (g.Cin / g.dg) % 16 == 0, FV2P_ELIMIT,
               "dcn: channels per deformable group (%d/%d) must be a multiple of 16", g.Cin, g.dg);
  FV2P_REQUIRE(g.Cout <= 256, FV2P_ELIMIT, "dcn: more than 256 output channels per launch (split on the host)");
  return 0;
}

static const int kDcnPixChunk = 2048;

}  // namespace fv2p
using namespace fv2p;

#define DCN_GEOM_ARGS int batch, int height, int width, int c_in, int c_out, int h_out, int w_out, int kh, int kw, int sh, int sw, \
                      int ph, int pw, int dh, int dw, int deformable_group
#define DCN_GEOM_INIT DcnGeom g = {batch, height, width, c_in, c_out, h_out, w_out, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group}


static int dcn_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    else cus = 256;
  }
  return cus;
}

template <int NB, int MB>
static void dcn_fwd_launch(const DcnGeom& g, const float* x, const float* wt_oc, const float* bias, const float* offset, const float* mask,
                           float* y, long long pix_base, long long npix_part, int col_blocks, hipStream_t stream) {
  if (npix_part <= 0) return;
  const dim3 grid(static_cast<unsigned>(ceil_div(npix_part, 64 * MB)), static_cast<unsigned>(col_blocks));
  hipLaunchKernelGGL((dcn_fwd2_k<NB, MB>), grid, dim3(256), 2 * NB * 256 * sizeof(float), stream, g, x, wt_oc, bias, offset, mask, y, pix_base);
}

// wt_oc: weight permuted to [kh*kw][Cout][Cin] (input channels contiguous: what the LDS-DMA of the forward kernel fetches 16 bytes at a time)
extern "C" int fv2p_dcn_forward(const float* x_nhwc, const float* wt_oc, const float* bias, const float* offset, const float* mask,
                                DCN_GEOM_ARGS, float* y_nhwc, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DCN_GEOM_INIT;
  if (int rc = dcn_check(g)) return rc;
  const long long npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  if (npix == 0) return 0;
  FV2P_REQUIRE(x_nhwc && wt_oc && offset && mask && y_nhwc, FV2P_EINVAL, "dcn_forward: null pointer");
  FV2P_REQUIRE(static_cast<long long>(g.B) * g.H * g.W * g.Cin * 4 < (1ll << 32), FV2P_ELIMIT, "dcn_forward: input above 4 GiB (split the batch)");
  // Tile plan.  A workgroup computes 64*MB pixels x 16*NB columns; per-pixel arithmetic does not depend on the plan (every output element
  // is one fused multiply-add chain over (tap, channel) in a fixed order), so results are identical for every batch size.
  // Whole rounds of 128-pixel tiles, then the remainder as 64-pixel tiles: the last round costs half a tile per CU.
  const int cus = dcn_cu_count();
  const int nb_all = static_cast<int>(ceil_div(g.Cout, 16));
  const char* force = getenv("FV2P_DCN_FWD_PLAN");   // development: "<NB>,<MB>"
  int fnb = 0, fmb = 0;
  if (force) sscanf(force, "%d,%d", &fnb, &fmb);
  int NBsel = nb_all <= 4 ? 4 : (nb_all <= 8 ? 8 : 16);
  if (fnb) NBsel = fnb;
  const int col_blocks = static_cast<int>(ceil_div(nb_all, NBsel));
  const long long tiles128 = npix / 128;
  long long big = (tiles128 / cus) * cus;   // whole rounds
  if (NBsel == 16) big = 0;                 // <16,2> does not fit the register file
  if (fmb == 1) big = 0;
  if (fmb == 2) big = tiles128;
  const long long big_pix = big * 128;
#define DCN_F(NB) do { dcn_fwd_launch<NB, 2>(g, x_nhwc, wt_oc, bias, offset, mask, y_nhwc, 0, big_pix, col_blocks, stream); \
                       dcn_fwd_launch<NB, 1>(g, x_nhwc, wt_oc, bias, offset, mask, y_nhwc, big_pix, npix - big_pix, col_blocks, stream); } while (0)
  if (NBsel == 4) DCN_F(4);
  else if (NBsel == 8) DCN_F(8);
  else dcn_fwd_launch<16, 1>(g, x_nhwc, wt_oc, bias, offset, mask, y_nhwc, 0, npix, col_blocks, stream);
#undef DCN_F
  FV2P_LAUNCH_CHECK();
  return 0;
}


// ---- backward: workspace carving shared by the size query and the call
struct DcnBwdPlan {
  long long npix, ntargets, max_entries;
  int splits, pix_per_block, ci_tiles, co_tiles;
};
static DcnBwdPlan dcn_bwd_plan(const DcnGeom& g) {
  DcnBwdPlan p;
  const int K = g.kh * g.kw;
  p.npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  p.ntargets = static_cast<long long>(g.B) * g.dg * g.H * g.W;
  p.max_entries = p.npix * g.dg * K * 4;
  p.ci_tiles = static_cast<int>(ceil_div(g.Cin, 128));
  p.co_tiles = static_cast<int>(ceil_div(g.Cout, 128));
  const long long cols = static_cast<long long>(K) * p.ci_tiles * p.co_tiles;
  long long s = (2ll * dcn_cu_count()) / cols;            // one resident round of workgroups (two per CU)
  const long long smax = ceil_div(p.npix > 0 ? p.npix : 1, 64);
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  long long ppb = ceil_div(ceil_div(p.npix > 0 ? p.npix : 1, s), 16) * 16;
  p.pix_per_block = static_cast<int>(ppb);
  p.splits = static_cast<int>(ceil_div(p.npix > 0 ? p.npix : 1, ppb));
  return p;
}
template <typename C>
static void dcn_bwd_carve(C& c, const DcnGeom& g, const DcnBwdPlan& p, float** colg, int** cursor, uint2** entries, void** scan_ws, float** partial) {
  const int K = g.kh * g.kw;
  *colg = c.template take<float>(static_cast<size_t>(p.npix) * K * g.Cin);
  *cursor = c.template take<int>(static_cast<size_t>(p.ntargets) + 1);
  *entries = c.template take<uint2>(static_cast<size_t>(p.max_entries));
  *scan_ws = c.template take<char>(scan_ws_bytes(p.ntargets));
  *partial = c.template take<float>(static_cast<size_t>(p.splits) * K * g.Cin * g.Cout);
}
struct SizerC : Sizer {
  template <typename T> T* take(size_t n) { Sizer::take<T>(n); return nullptr; }
};

extern "C" size_t fv2p_dcn_backward_ws_bytes(int batch, int height, int width, int h_out, int w_out, int c_in, int c_out, int kh, int kw,
                                             int deformable_group) {
  DcnGeom g = {batch, height, width, c_in, c_out, h_out, w_out, kh, kw, 1, 1, 0, 0, 1, 1, deformable_group > 0 ? deformable_group : 1};
  const DcnBwdPlan p = dcn_bwd_plan(g);
  SizerC s;
  float *a, *e; int* b; uint2* c; void* d;
  dcn_bwd_carve(s, g, p, &a, &b, &c, &d, &e);
  return s.bytes();
}

template <int JO, int NBP, int MC>
static void dcn_col_launch(const DcnGeom& g, const float* x, const float* wt, const float* offset, const float* mask, const float* dy, float* colg,
                           float* doff, float* dmask, long long pix_base, long long npix_part, hipStream_t stream) {
  if (npix_part <= 0) return;
  hipLaunchKernelGGL((dcn_bwd_col_k<JO, NBP, MC>), dim3(static_cast<unsigned>(ceil_div(npix_part, 64 * NBP))), dim3(256),
                     2 * MC * JO * 256 * sizeof(float), stream, g, x, wt, offset, mask, dy, colg, doff, dmask, pix_base);
}

// dx_nhwc, doffset, dmask, dwt are fully written (nothing to zero).  wt = [kh*kw][Cin][Cout].
extern "C" int fv2p_dcn_backward(const float* x_nhwc, const float* wt, const float* offset, const float* mask, const float* dy_nhwc,
                                 DCN_GEOM_ARGS, float* dx_nhwc, float* doffset, float* dmask, float* dwt, void* ws, size_t ws_bytes,
                                 fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DCN_GEOM_INIT;
  if (int rc = dcn_check(g)) return rc;
  FV2P_REQUIRE((g.Cout & 3) == 0, FV2P_ELIMIT, "dcn_backward: output channels must be a multiple of 4 (pad the weight)");
  const long long npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  const int K = g.kh * g.kw;
  FV2P_REQUIRE(dwt, FV2P_EINVAL, "dcn_backward: null dwt");
  const long long nin = static_cast<long long>(g.B) * g.H * g.W;
  if (npix == 0) {
    FV2P_HIP(hipMemsetAsync(dwt, 0, sizeof(float) * (size_t)K * g.Cin * g.Cout, stream));
    if (nin > 0 && dx_nhwc) FV2P_HIP(hipMemsetAsync(dx_nhwc, 0, sizeof(float) * (size_t)nin * g.Cin, stream));
    return 0;
  }
  FV2P_REQUIRE(x_nhwc && wt && offset && mask && dy_nhwc && dx_nhwc && doffset && dmask, FV2P_EINVAL, "dcn_backward: null pointer");
  FV2P_REQUIRE(nin * g.Cin * 4 < (1ll << 32) && npix * K * g.Cin * 4 < (1ll << 32), FV2P_ELIMIT,
               "dcn_backward: input or column gradients above 4 GiB (split the batch)");
  const DcnBwdPlan p = dcn_bwd_plan(g);
  FV2P_REQUIRE(p.max_entries < (1ll << 31) && npix * K * 4 < (1ll << 32), FV2P_ELIMIT, "dcn_backward: too many samples (split the batch)");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_dcn_backward_ws_bytes(g.B, g.H, g.W, g.Ho, g.Wo, g.Cin, g.Cout, g.kh, g.kw, g.dg), FV2P_EWORKSPACE,
               "dcn_backward: workspace too small");
  Carver c(ws, ws_bytes);
  float *colg, *partial; int* cursor; uint2* entries; void* sws;
  dcn_bwd_carve(c, g, p, &colg, &cursor, &entries, &sws, &partial);
  // 1. per-target sample lists
  FV2P_HIP(hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)(p.ntargets + 1), stream));
  const long long nsamples = npix * g.dg * K;
  const unsigned iblocks = static_cast<unsigned>(ceil_div(nsamples, 256));
  hipLaunchKernelGGL((dcn_index_k<0>), dim3(iblocks), dim3(256), 0, stream, g, offset, cursor, entries);
  if (int rc = exclusive_scan_i32(cursor, cursor, p.ntargets, nullptr, sws, scan_ws_bytes(p.ntargets), stream)) return rc;
  hipLaunchKernelGGL((dcn_index_k<1>), dim3(iblocks), dim3(256), 0, stream, g, offset, cursor, entries);
  hipLaunchKernelGGL(dcn_index_sort_long_k, dim3(static_cast<unsigned>(std::min<long long>(1024, ceil_div(p.ntargets, 256)))), dim3(256), 0, stream,
                     cursor, p.ntargets, entries);
  // 2. column gradients, grad_mask, grad_offset
  const int cus = dcn_cu_count();
  const int jo = static_cast<int>(ceil_div(g.Cout, 16));
  const bool mc2 = (g.Cin / g.dg) % 32 == 0;
  const long long tiles128 = npix / 128;
  long long big = (tiles128 / cus) * cus;
  const char* force = getenv("FV2P_DCN_BWD_PLAN");   // development: "<NBP>"
  if (force && atoi(force) == 1) big = 0;
  if (force && atoi(force) == 2) big = tiles128;
  const long long big_pix = big * 128;
#define DCN_C(JO, MC) do { if (JO <= 8) dcn_col_launch<JO, (JO <= 8 ? 2 : 1), MC>(g, x_nhwc, wt, offset, mask, dy_nhwc, colg, doffset, dmask, 0, big_pix, stream); \
                           dcn_col_launch<JO, 1, MC>(g, x_nhwc, wt, offset, mask, dy_nhwc, colg, doffset, dmask, JO <= 8 ? big_pix : 0, JO <= 8 ? npix - big_pix : npix, stream); } while (0)
#define DCN_CJ(JO) do { if (mc2) DCN_C(JO, 2); else DCN_C(JO, 1); } while (0)
  if (jo <= 1) DCN_CJ(1); else if (jo <= 2) DCN_CJ(2); else if (jo <= 4) DCN_CJ(4); else if (jo <= 8) DCN_CJ(8); else DCN_CJ(16);
#undef DCN_CJ
#undef DCN_C
  // 3. grad_input
  const int cpg = g.Cin / g.dg;
  const unsigned gblocks = static_cast<unsigned>(ceil_div(p.ntargets, 4));
  if (cpg % 4 == 0 && cpg > 128) hipLaunchKernelGGL((dcn_col2im_k<4>), dim3(gblocks), dim3(256), 0, stream, g, colg, cursor, entries, dx_nhwc);
  else if (cpg > 64) hipLaunchKernelGGL((dcn_col2im_k<2>), dim3(gblocks), dim3(256), 0, stream, g, colg, cursor, entries, dx_nhwc);
  else hipLaunchKernelGGL((dcn_col2im_k<1>), dim3(gblocks), dim3(256), 0, stream, g, colg, cursor, entries, dx_nhwc);
  // 4. weight gradient
  const dim3 wgrid(static_cast<unsigned>(p.splits), static_cast<unsigned>(K), static_cast<unsigned>(p.ci_tiles * p.co_tiles));
  hipLaunchKernelGGL((dcn_bwd_weight2_k<0>), wgrid, dim3(256), 2 * 2 * 16 * kDwPitch * sizeof(float), stream, g, x_nhwc, offset, mask, dy_nhwc,
                     p.pix_per_block, partial);
  const long long per_chunk = static_cast<long long>(K) * g.Cin * g.Cout;
  hipLaunchKernelGGL(dcn_reduce_k, dim3(static_cast<unsigned>(ceil_div(per_chunk, 256))), dim3(256), 0, stream, partial, p.splits, per_chunk, dwt);
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" int fv2p_dcn_forward_v1(const float* x_nhwc, const float* wt, const float* bias, const float* offset, const float* mask,
                                DCN_GEOM_ARGS, float* y_nhwc, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DCN_GEOM_INIT;
  if (int rc = dcn_check(g)) return rc;
  const long long npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  if (npix == 0) return 0;
  FV2P_REQUIRE(x_nhwc && wt && offset && mask && y_nhwc, FV2P_EINVAL, "dcn_forward: null pointer");
  const unsigned blocks = static_cast<unsigned>(ceil_div(npix, 64));
  const int nb = static_cast<int>(ceil_div(g.Cout, 16));
#define DCN_FWD(NB) hipLaunchKernelGGL((dcn_fwd_k<NB>), dim3(blocks), dim3(256), 2 * 16 * NB * 16 * sizeof(float), stream, g, x_nhwc, wt, bias, offset, mask, y_nhwc)
  if (nb <= 1) DCN_FWD(1); else if (nb <= 2) DCN_FWD(2); else if (nb <= 4) DCN_FWD(4); else if (nb <= 8) DCN_FWD(8); else DCN_FWD(16);
#undef DCN_FWD
  FV2P_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t fv2p_dcn_backward_ws_bytes_v1(int batch, int h_out, int w_out, int c_in, int c_out, int kh, int kw) {
  const long long npix = static_cast<long long>(batch) * h_out * w_out;
  const long long chunks = ceil_div(npix > 0 ? npix : 1, kDcnPixChunk);
  Sizer s;
  s.take<float>(static_cast<size_t>(chunks) * kh * kw * c_in * c_out);
  return s.bytes();
}

// dx_nhwc must be zeroed by the caller (atomically accumulated); doff / dmask / dwt are fully written.
extern "C" int fv2p_dcn_backward_v1(const float* x_nhwc, const float* wt, const float* offset, const float* mask, const float* dy_nhwc,
                                 DCN_GEOM_ARGS, float* dx_nhwc, float* doffset, float* dmask, float* dwt, void* ws, size_t ws_bytes,
                                 fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DCN_GEOM_INIT;
  if (int rc = dcn_check(g)) return rc;
  const long long npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  const int K = g.kh * g.kw;
  FV2P_REQUIRE(dwt, FV2P_EINVAL, "dcn_backward: null dwt");
  if (npix == 0) {
    FV2P_HIP(hipMemsetAsync(dwt, 0, sizeof(float) * (size_t)K * g.Cin * g.Cout, stream));
    return 0;
  }
  FV2P_REQUIRE(x_nhwc && wt && offset && mask && dy_nhwc && dx_nhwc && doffset && dmask, FV2P_EINVAL, "dcn_backward: null pointer");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_dcn_backward_ws_bytes_v1(g.B, g.Ho, g.Wo, g.Cin, g.Cout, g.kh, g.kw), FV2P_EWORKSPACE,
               "dcn_backward: workspace too small");
  const unsigned blocks = static_cast<unsigned>(ceil_div(npix, 64));
  const int jo = static_cast<int>(ceil_div(g.Cout, 16));
#define DCN_BD(JO) hipLaunchKernelGGL((dcn_bwd_data_k<JO>), dim3(blocks), dim3(256), JO * 64 * 4 * sizeof(float), stream, g, x_nhwc, wt, offset, mask, dy_nhwc, dx_nhwc, doffset, dmask)
  if (jo <= 1) DCN_BD(1); else if (jo <= 2) DCN_BD(2); else if (jo <= 4) DCN_BD(4); else if (jo <= 8) DCN_BD(8); else DCN_BD(16);
#undef DCN_BD
  const unsigned chunks = static_cast<unsigned>(ceil_div(npix, kDcnPixChunk));
  float* partial = static_cast<float*>(ws);
  const dim3 grid(chunks, K, static_cast<unsigned>(ceil_div(g.Cin, 16)));
#define DCN_BW(NB) hipLaunchKernelGGL((dcn_bwd_weight_k<NB>), grid, dim3(256), 16 * NB * 16 * sizeof(float), stream, g, x_nhwc, offset, mask, dy_nhwc, kDcnPixChunk, partial)
  if (jo <= 1) DCN_BW(1); else if (jo <= 2) DCN_BW(2); else if (jo <= 4) DCN_BW(4); else if (jo <= 8) DCN_BW(8); else DCN_BW(16);
#undef DCN_BW
  const long long per_chunk = static_cast<long long>(K) * g.Cin * g.Cout;
  hipLaunchKernelGGL(dcn_reduce_k, dim3(static_cast<unsigned>(ceil_div(per_chunk, 256))), dim3(256), 0, stream, partial, (int)chunks, per_chunk, dwt);
  FV2P_LAUNCH_CHECK();
  return 0;
}

