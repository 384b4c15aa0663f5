// A13 — modulated deformable convolution (DCNv2) as implicit GEMMs on fp32 MFMA.
//
// Replaces DCN.modulated_deform_conv_{forward,backward} (pcdet/ops/DeformableConvolutionV2PyTorch/src/vision.cpp:6-12,
// src/cuda/modulated_deform_conv_cuda.cu:19-280, kernels modulated_deform_im2col_cuda.cuh:24-328).  The reference writes a
// [Cin*kh*kw, B*Ho*Wo] `columns` buffer (1.3 GB for the MGAF head), runs cuBLAS on it, and scatters grad_input with atomicAdd.  Here:
//
//   forward   y[p, co]       = b[co] + sum_{k,ci} m[p,k] * bilin(x, p, k)[ci] * W[k][ci][co]        dcn_fwd_k: samples built in the operand fetch
//   backward  dcol[p, k, ci] = sum_co dy[p, co] * W[k][ci][co]                                      dcn_bwd_col_k (dy resident in registers)
//             dm[p,k], doff[p,k] = <dcol, bilin> , <dcol, d bilin / d pos>                           its epilogue (x corners gathered like the forward)
//             dx[t, ci]      = sum over the samples touching pixel t of  weight * m * dcol           dcn_index_k + dcn_col2im_k: a GATHER, no float atomics
//             dW[k][ci][co]  = sum_p m * bilin(x, p, k)[ci] * dy[p, co]                             dcn_bwd_weight_k, partial tiles summed in a fixed order
//
// All four products are formed TRANSPOSED (weights or dy as the MFMA's A operand): a lane then owns four adjacent channels of one pixel,
// which is both what it gathers from x (one 16-byte load per bilinear corner) and what it stores.  Every output element is one fused
// multiply-add chain in a fixed order, independent of tile shapes and batch size; the backward is bit-identical from run to run.
// Activations are NHWC here; the Python layer permutes once on entry/exit, as the reference itself computes NHWC and permutes
// (modulated_deform_conv_cuda.cu:78,118).  Sampling follows mdmcn_im2col_bilinear (:24-54) and the validity test of :176; offsets / masks
// keep the reference layout [B, dg*2*K, Ho, Wo] / [B, dg*K, Ho, Wo] with (2*(i*kw+j), +1) = (dh, dw).  Out-of-map corners are read at the
// nearest in-map pixel with weight 0 (loads are unconditional).
#include "common.hpp"
#include <stdlib.h>
#include <stdio.h>
#include <algorithm>
#include <type_traits>

namespace fv2p {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DcnGeom {
  int B, H, W, Cin, Cout, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg;
};

// out = (accumulate ? out : 0) + partial[0] + partial[1] + ...: a fixed order, also across the batch chunks of one call
__global__ void dcn_reduce_k(const float* __restrict__ partial, int chunks, long long per_chunk, float* __restrict__ out, int accumulate) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= per_chunk) return;
  float s = accumulate ? out[t] : 0.f;
  for (int c = 0; c < chunks; ++c) s += partial[c * per_chunk + t];
  out[t] = s;
}


// Workgroup b of a launch runs on XCD b % 8, each XCD with an L2 of its own.  Handing out tiles in launch order gives every XCD every
// eighth tile: the eight L2s then all hold the same band of the map (a tile's bilinear samples reach into its neighbours' pixels) and
// each line is fetched from the fabric eight times - 815 MB of L2 fills for 83 MB of input at [4,128,200,176], rocprofv3 FETCH_SIZE.
// XCD-major order gives every XCD one contiguous eighth of the tiles instead.  (tile, sub): sub is the fast index (column block / share).
__device__ __forceinline__ void xcd_tile(int n_sub, long long& tile, int& sub) {
  const long long total = gridDim.x, b = blockIdx.x;
  const long long x = b & 7, slot = b >> 3;
  const long long lp = x * (total >> 3) + (x < (total & 7) ? x : (total & 7)) + slot;
  tile = lp / n_sub;
  sub = static_cast<int>(lp % n_sub);
}

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
// TAPIN (round 6): step order (group, 16-channel chunk, tap) instead of (tap, group, chunk).  The nine taps of one chunk gather from the same
// 4 x 4-pixel neighbourhoods of 64-byte segments back to back (L1 / L2 hits), where the tap-outer order walks a pixel's whole channel
// range per tap and comes back to it a tap later, after the concurrent tiles of the XCD have pushed it out of the L2.  A step's tap state
// (corner offsets, bilinear weights, mask) then changes at every step: the K states of the current group are built once per group by the
// wave's own lanes (lane (n, gq): taps gq, gq + 4, ... of pixel n) into LDS, 32 bytes per (tap, pixel), and a step reads its state back
// with two 16-byte reads.  (First form: rebuilt at every step from offsets fetched two steps ahead - ~45 VALU in front of every gather:
// 2 080 against 1 730 us at the MGAF head.)
template <int NB, int MB, bool TAPIN = false>
__global__ __launch_bounds__(256, 2) void dcn_fwd_k(DcnGeom g, const float* __restrict__ x, const float* __restrict__ wt_oc,
                                                     const float* __restrict__ bias, const float* __restrict__ offset,
                                                     const float* __restrict__ mask, float* __restrict__ y, long long pix_base, int n_sub) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x NB x 256 floats
  constexpr int FRAG = NB * 256;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 15, gq = lane >> 4;
  const long long npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  const int plane = g.Ho * g.Wo;
  const int K = g.kh * g.kw, cpg = g.Cin / g.dg, cps = cpg / 16, segs = K * g.dg, steps = segs * cps;
  long long tile;
  int cb;
  xcd_tile(n_sub, tile, cb);
  const int col0 = cb * (NB * 16);
  const long long tile0 = pix_base + tile * (64 * MB) + wave * (16 * MB);
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

  // position of a step: tap k, group dgi, chunk c of the group
  auto advance = [&](int& k_, int& dg_, int& c_) {
    if constexpr (TAPIN) { if (++k_ == K) { k_ = 0; if (++c_ == cps) { c_ = 0; ++dg_; } } }
    else { if (++c_ == cps) { c_ = 0; if (++dg_ == g.dg) { dg_ = 0; ++k_; } } }
  };
  // TAPIN: this wave's tap states of the current group, [K][16 pixels][8]: {offset of corner 0, + column, + row, mask} {four bilinear weights}
  float* taps = lds + 2 * FRAG + wave * (K * 16 * 8);
  auto build_group = [&](int dg_) {
    static_assert(!TAPIN || MB == 1, "tap-inner order: one 16-pixel block per wave");
    for (int kk = gq; kk < K; kk += 4) {
      const float* ob = offset + (static_cast<long long>(pb[0]) * g.dg + dg_) * 2 * K * plane + ppos[0];
      const float oh = ob[static_cast<long long>(2 * kk) * plane], ow = ob[static_cast<long long>(2 * kk + 1) * plane];
      const float mm = mask[((static_cast<long long>(pb[0]) * g.dg + dg_) * K + kk) * plane + ppos[0]];
      const int i = kk / g.kw, j = kk % g.kw;
      Corner4 cn;
      make_corners(g, live[0], pb[0], static_cast<float>(pho[0] * g.sh - g.ph + i * g.dh) + oh, static_cast<float>(pwo[0] * g.sw - g.pw + j * g.dw) + ow, mm, 0u, cn);
      float* d = taps + (kk * 16 + n) * 8;
      *reinterpret_cast<uint4*>(d) = make_uint4(cn.o[0], cn.o[1] - cn.o[0], cn.o[2] - cn.o[0], __float_as_uint(cn.m));
      *reinterpret_cast<float4*>(d + 4) = make_float4(cn.w[0], cn.w[1], cn.w[2], cn.w[3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a wave's LDS operations complete in order: its own reads below see these writes
  };
  uint4 ta = make_uint4(0u, 0u, 0u, 0u);   // TAPIN: a step's tap state as it lies in LDS
  float4 tw = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch_taps = [&](int k_) {
    const float* d = taps + (k_ * 16 + n) * 8;
    ta = *reinterpret_cast<const uint4*>(d);
    tw = *reinterpret_cast<const float4*>(d + 4);
  };
  auto expand_taps = [&]() {
    const unsigned lb = 16u * gq;
    t[0].o[0] = ta.x + lb; t[0].o[1] = ta.x + ta.y + lb; t[0].o[2] = ta.x + ta.z + lb; t[0].o[3] = ta.x + ta.y + ta.z + lb;
    t[0].w[0] = tw.x; t[0].w[1] = tw.y; t[0].w[2] = tw.z; t[0].w[3] = tw.w;
    t[0].m = __uint_as_float(ta.w);
  };
  // prologue: step 0 taps and operands; the offsets of the next tap state that will be needed
  if constexpr (TAPIN) {
    build_group(0);
    fetch_taps(0);
    expand_taps();
  } else {
    load_offsets(0, 0);
    set_taps(0);
    if (segs > 1) load_offsets(g.dg > 1 ? 0 : 1, g.dg > 1 ? 1 : 0);   // segment 1 = (tap, group) after (0, 0)
  }
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
    int c1 = c, dg1 = dgi, k1 = k;
    advance(k1, dg1, c1);
    const bool more = s + 1 < steps;
    if (more) {
      if constexpr (TAPIN) {
        if (dg1 != dgi) build_group(dg1);           // (the state of step s is in registers already)
        fetch_taps(k1);                             // (read two steps ahead into a second register set: 2 014 against 1 856 us at the head)
        expand_taps();
      } else if (c1 == 0) set_taps(k1);             // its offsets were fetched a segment ago
      dma(k1, (dg1 * cps + c1) * 16, lds + ((s + 1) & 1) * FRAG);
      gather((dg1 * cps + c1) * 16);
      if constexpr (TAPIN) {
      } else if (c1 == 0) {                         // fetch the offsets of the segment after that
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
                                                        float* __restrict__ dmask, long long pix_base, int n_sub) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x (MC x JO) pieces of 256 floats
  constexpr int FRAG = MC * JO * 256, CS = 16 * MC;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 15, gq = lane >> 4;
  const long long npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  const int plane = g.Ho * g.Wo;
  const int K = g.kh * g.kw, cpg = g.Cin / g.dg, cps = cpg / CS, segs = K * g.dg;   // cpg % CS == 0
  // n_sub shares of the (tap, group) segments per pixel tile: small maps do not fill the chip with pixel tiles alone
  long long tile;
  int share;
  xcd_tile(n_sub, tile, share);
  const int seg_begin = static_cast<int>(static_cast<long long>(segs) * share / n_sub);
  const int seg_end = static_cast<int>(static_cast<long long>(segs) * (share + 1) / n_sub);
  const int steps = (seg_end - seg_begin) * cps;
  const long long tile0 = pix_base + tile * (64 * NBP) + wave * (16 * NBP);
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
  if (steps <= 0) return;
  int k = seg_begin / g.dg, dgi = seg_begin % g.dg, c = 0;
  load_offsets(k, dgi);
  set_taps(k);
  if (seg_begin + 1 < seg_end) load_offsets((seg_begin + 1) / g.dg, (seg_begin + 1) % g.dg);
  dma(k, dgi * cps * CS, lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
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
        if (k2 * g.dg + dg2 < seg_end) load_offsets(k2, dg2);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    c = c1; dgi = dg1; k = k1;
  }
}


// ---------------------------------------------------------------- backward, pass 2: grad_input in gather form -------
// The reference scatters every column gradient to its four bilinear corners with atomicAdd (modulated_deform_im2col_cuda.cuh:196-254):
// 1.3 G float atomics at the MGAF head's shape and a sum whose order changes from run to run.  Here every sample is filed under its
// TOP-LEFT corner pixel ("key": one integer atomic per sample for the count, one for the slot; totals are exact whatever the order), and
// a wave that owns a 2 x 4 tile of input pixels walks the 3 x 5 keys around it: an entry of key (r, c) adds to the pixels (r, c),
// (r, c+1), (r+1, c), (r+1, c+1) of the tile with its four bilinear weights.  Keys are visited in a fixed order and every list in
// ascending sample order (ranked inside the wave), so each pixel's sum has one order: no float atomics, no zero fill, same bits every run.
//   key grid per (sample, deformable group): (H + 1) x (W + 1), key (h_low + 1, w_low + 1); entry = {src = pixel * K + tap, lh, lw}
struct DcnEntries {
  unsigned* src;
  float* lh;
  float* lw;
};

// one thread per sample (b, group, tap, output pixel): FILL = 0 counts the entries of every key, FILL = 1 writes them
// (cursor[key] starts at the list's first slot and ends at its end: list = [cursor[key - 1], cursor[key]) afterwards)
template <int FILL>
__global__ __launch_bounds__(256) void dcn_index_k(DcnGeom g, const float* __restrict__ offset, int* __restrict__ cursor, DcnEntries en) {
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
  const int ho = pos / g.Wo, wo = pos % g.Wo, i = k / g.kw, j = k % g.kw;
  const float h_im = static_cast<float>(ho * g.sh - g.ph + i * g.dh) + off_h;
  const float w_im = static_cast<float>(wo * g.sw - g.pw + j * g.dw) + off_w;
  if (!(h_im > -1.f && w_im > -1.f && h_im < static_cast<float>(g.H) && w_im < static_cast<float>(g.W))) return;   // :237 of the reference
  const float hf = floorf(h_im), wf = floorf(w_im);
  const long long key = (static_cast<long long>(bd) * (g.H + 1) + static_cast<int>(hf) + 1) * (g.W + 1) + static_cast<int>(wf) + 1;
  const int slot = atomicAdd(&cursor[key], 1);
  if (FILL) {
    en.src[slot] = static_cast<unsigned>((static_cast<long long>(b) * plane + pos) * K + k);
    en.lh[slot] = h_im - hf;
    en.lw[slot] = w_im - wf;
  }
}

// Lists longer than a wave (more than 64 samples under one key) are put in order here, up to 4096 entries; beyond that a list keeps
// its fill order: the sum is still complete, only its rounding may differ between runs.
constexpr int kDcnLongList = 4096;
__global__ __launch_bounds__(256) void dcn_index_sort_long_k(const int* __restrict__ ends, long long nkeys, DcnEntries en) {
  __shared__ unsigned bs[kDcnLongList];
  __shared__ float bh[kDcnLongList], bw[kDcnLongList];
  __shared__ int queue[256], qn;
  const long long per = (nkeys + gridDim.x - 1) / gridDim.x;
  const long long t0 = static_cast<long long>(blockIdx.x) * per, t1 = min(t0 + per, nkeys);
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
      for (int i = threadIdx.x; i < n; i += 256) { bs[i] = en.src[start + i]; bh[i] = en.lh[start + i]; bw[i] = en.lw[start + i]; }
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += 256) {
        const unsigned s = bs[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += bs[j] < s;
        en.src[start + rank] = s; en.lh[start + rank] = bh[i]; en.lw[start + rank] = bw[i];
      }
      __syncthreads();
    }
  }
}

template <int VEC>
struct ColVec { float v[VEC]; };
template <int VEC>
__device__ __forceinline__ ColVec<VEC> col_load(const float* p) {
  ColVec<VEC> r;
  if constexpr (VEC == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(p); r.v[0] = t[0]; r.v[1] = t[1]; r.v[2] = t[2]; r.v[3] = t[3]; }
  else if constexpr (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
  else r.v[0] = p[0];
  return r;
}

// one wave per (2 x 4 tile of input pixels, sample, deformable group); lane = VEC adjacent channels of the group
template <int VEC>
__global__ __launch_bounds__(256) void dcn_col2im_k(DcnGeom g, const float* __restrict__ colg, const int* __restrict__ ends, DcnEntries en,
                                                    float* __restrict__ dx) {
  __shared__ unsigned s_src[4][64];
  __shared__ float s_lh[4][64], s_lw[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_c = (g.W + 3) / 4, tiles_r = (g.H + 1) / 2;
  const long long ntiles = static_cast<long long>(g.B) * g.dg * tiles_r * tiles_c;
  long long blk;
  int unused;
  xcd_tile(1, blk, unused);
  const long long tile = blk * 4 + wave;
  if (tile >= ntiles) return;
  const int cpg = g.Cin / g.dg;
  const int bd = static_cast<int>(tile / (tiles_r * tiles_c)), b = bd / g.dg, dgi = bd % g.dg;
  const int tr0 = static_cast<int>((tile / tiles_c) % tiles_r) * 2, tc0 = static_cast<int>(tile % tiles_c) * 4;
  const int nl = min(5, g.W + 1 - tc0);   // key columns tc0 .. tc0 + nl - 1 exist
  unsigned* ssrc = s_src[wave];
  float *slh = s_lh[wave], *slw = s_lw[wave];
  for (int c0 = 0; c0 < cpg; c0 += 64 * VEC) {
    const int c = c0 + lane * VEC;
    const bool act = c < cpg;
    const float* col = colg + static_cast<long long>(dgi) * cpg + (act ? c : 0);
    float acc[2][4][VEC];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[a][e][v] = 0.f;
    auto add = [&](auto I_, auto J_, unsigned src, float lh, float lw, const ColVec<VEC>& row) {
      constexpr int I = decltype(I_)::value, J = decltype(J_)::value;
      const float hh = 1.f - lh, hw = 1.f - lw;
#pragma unroll
      for (int dr = 0; dr < 2; ++dr)
#pragma unroll
        for (int dc = 0; dc < 2; ++dc) {
          constexpr int dummy = 0; (void)dummy;
          const int tr = I - 1 + dr, tc = J - 1 + dc;
          if (tr < 0 || tr > 1 || tc < 0 || tc > 3) continue;
          const float w = (dr ? lh : hh) * (dc ? lw : hw);
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[tr][tc][v] = __builtin_fmaf(w, row.v[v], acc[tr][tc][v]);
        }
    };
    // entries [lo, hi) of the wave's LDS stage belong to key (I, J)
    auto walk = [&](auto I_, auto J_, int lo, int hi) {
      int e = lo;
      for (; e + 4 <= hi; e += 4) {
        unsigned s[4]; float a[4], bq[4]; ColVec<VEC> r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { s[u] = ssrc[e + u]; a[u] = slh[e + u]; bq[u] = slw[e + u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = col_load<VEC>(col + static_cast<long long>(s[u]) * g.Cin);
#pragma unroll
        for (int u = 0; u < 4; ++u) add(I_, J_, s[u], a[u], bq[u], r[u]);
      }
      for (; e < hi; ++e) add(I_, J_, ssrc[e], slh[e], slw[e], col_load<VEC>(col + static_cast<long long>(ssrc[e]) * g.Cin));
    };
    auto key_row = [&](auto I_) {
      constexpr int I = decltype(I_)::value;
      const int kr = tr0 + I;   // key row; exists for kr <= H
      if (kr > g.H) return;
      const long long kbase = (static_cast<long long>(bd) * (g.H + 1) + kr) * (g.W + 1) + tc0;
      // list bounds of the nl keys: s[j] = end of key kbase + j - 1 (= start of key kbase + j)
      int bound = 0;
      if (lane <= nl) { const long long kk = kbase + lane - 1; bound = kk >= 0 ? ends[kk] : 0; }
      int s[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) s[j] = __shfl(bound, min(j, nl), 64);
      const int run0 = s[0], nrun = s[5] - s[0];
      if (nrun <= 64) {
        // the five lists are one contiguous run: rank every entry by (list, sample) inside the wave
        unsigned src = 0xffffffffu; float lh = 0.f, lw = 0.f;
        int mylist = 8;
        if (lane < nrun) {
          src = en.src[run0 + lane]; lh = en.lh[run0 + lane]; lw = en.lw[run0 + lane];
          mylist = 0;
#pragma unroll
          for (int j = 1; j < 5; ++j) mylist += (run0 + lane >= s[j]);
        }
        const unsigned long long mine = (static_cast<unsigned long long>(mylist) << 32) | src;
        int rank = 0;
        for (int j = 0; j < nrun; ++j) {
          const unsigned long long o = (static_cast<unsigned long long>(static_cast<unsigned>(__shfl(mylist, j, 64))) << 32) |
                                       static_cast<unsigned>(__shfl(static_cast<int>(src), j, 64));
          rank += o < mine;
        }
        if (lane < nrun) { ssrc[rank] = src; slh[rank] = lh; slw[rank] = lw; }
        // (one wave, in-order LDS: no barrier)
        walk(I_, std::integral_constant<int, 0>{}, s[0] - run0, s[1] - run0);
        walk(I_, std::integral_constant<int, 1>{}, s[1] - run0, s[2] - run0);
        walk(I_, std::integral_constant<int, 2>{}, s[2] - run0, s[3] - run0);
        walk(I_, std::integral_constant<int, 3>{}, s[3] - run0, s[4] - run0);
        walk(I_, std::integral_constant<int, 4>{}, s[4] - run0, s[5] - run0);
      } else {
        auto one = [&](auto J_, int lo, int hi) {
          const int n = hi - lo;
          for (int base = 0; base < n; base += 64) {
            const int cnt = min(64, n - base);
            unsigned src = 0xffffffffu; float lh = 0.f, lw = 0.f;
            if (lane < cnt) { src = en.src[lo + base + lane]; lh = en.lh[lo + base + lane]; lw = en.lw[lo + base + lane]; }
            int rank = lane;                                   // lists above 64 entries were sorted by dcn_index_sort_long_k
            if (n <= 64) {
              rank = 0;
              for (int j = 0; j < cnt; ++j) rank += static_cast<unsigned>(__shfl(static_cast<int>(src), j, 64)) < src;
            }
            if (lane < cnt) { ssrc[rank] = src; slh[rank] = lh; slw[rank] = lw; }
            walk(I_, J_, 0, cnt);
          }
        };
        one(std::integral_constant<int, 0>{}, s[0], s[1]);
        one(std::integral_constant<int, 1>{}, s[1], s[2]);
        one(std::integral_constant<int, 2>{}, s[2], s[3]);
        one(std::integral_constant<int, 3>{}, s[3], s[4]);
        one(std::integral_constant<int, 4>{}, s[4], s[5]);
      }
    };
    key_row(std::integral_constant<int, 0>{});
    key_row(std::integral_constant<int, 1>{});
    key_row(std::integral_constant<int, 2>{});
    if (act) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = tr0 + a, cc = tc0 + e;
          if (r >= g.H || cc >= g.W) continue;
          float* out = dx + ((static_cast<long long>(b) * g.H + r) * g.W + cc) * g.Cin + dgi * cpg + c;
          if constexpr (VEC == 4) *reinterpret_cast<f32x4*>(out) = f32x4{acc[a][e][0], acc[a][e][1], acc[a][e][2], acc[a][e][3]};
          else if constexpr (VEC == 2) *reinterpret_cast<float2*>(out) = make_float2(acc[a][e][0], acc[a][e][1]);
          else out[0] = acc[a][e][0];
        }
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
__global__ __launch_bounds__(256, 2) void dcn_bwd_weight_k(DcnGeom g, const float* __restrict__ x, const float* __restrict__ offset,
                                                            const float* __restrict__ mask, const float* __restrict__ dy, int pix_per_block,
                                                            int n_ztiles, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 buffers x (col tile 16 x 132 + dy tile 16 x 132)
  constexpr int TILE = 16 * kDwPitch, BUF = 2 * TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, gq = lane >> 4;
  const int K = g.kh * g.kw, cpg = g.Cin / g.dg, plane = g.Ho * g.Wo;
  const int ci_tiles = static_cast<int>((g.Cin + 127) / 128);
  // one grid dimension, decoded XCD-major: the kh*kw taps (and channel tiles) of one pixel range run on ONE XCD and share its x and dy
  // lines in that L2 (in launch order they went to all eight: 1 131 MB of L2 fills for 144 MB of x + dy at [4,128,200,176])
  long long lp;
  int zt;
  xcd_tile(n_ztiles, lp, zt);
  const int k = static_cast<int>(lp % K), split = static_cast<int>(lp / K);
  const int ci0 = (zt % ci_tiles) * 128, co0 = (zt / ci_tiles) * 128;
  const long long npix = static_cast<long long>(g.B) * plane;
  const long long p_begin = static_cast<long long>(split) * pix_per_block, p_end = min(p_begin + pix_per_block, npix);
  // staging role: pixel tid >> 4 of the step, channel quads (tid & 15) and (tid & 15) + 16
  const int spx = tid >> 4, cq = tid & 15;
  const int i_k = k / g.kw, j_k = k % g.kw;
  f32x4 raw[2][4], dyv[2];
  float wq[2][4], mk[2];
  float roh[2], row_[2], rom[2];
  unsigned off0[4] = {0u, 0u, 0u, 0u};
  const bool one_group = cpg % 128 == 0;   // the 128 input channels of the tile belong to one deformable group: one sampling position per pixel
  auto load_offsets = [&](long long p) {   // raw offsets / mask of pixel p for this thread's two channel quads
    const bool ok = p < p_end;
    const long long pp = ok ? p : p_begin;
    const int b = static_cast<int>(pp / plane), pos = static_cast<int>(pp % plane);
#pragma unroll
    for (int h = 0; h < (one_group ? 1 : 2); ++h) {
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
      if (h == 1 && one_group) {   // (uniform) both channel quads sample at the same position: the second one is 64 channels further
#pragma unroll
        for (int q = 0; q < 4; ++q) { raw[1][q] = ldx4(reinterpret_cast<const char*>(x) + 256, off0[q]); wq[1][q] = wq[0][q]; }
        mk[1] = mk[0];
      } else {
        Corner4 t;
        const float h_im = static_cast<float>(ho * g.sh - g.ph + i_k * g.dh) + roh[h];
        const float w_im = static_cast<float>(wo * g.sw - g.pw + j_k * g.dw) + row_[h];
        make_corners(g, cok, b, h_im, w_im, rom[h], static_cast<unsigned>(min(ci, g.Cin - 4)) * 4u, t);
#pragma unroll
        for (int q = 0; q < 4; ++q) { raw[h][q] = ldx4(reinterpret_cast<const char*>(x), t.o[q]); wq[h][q] = t.w[q]; off0[q] = t.o[q]; }
        mk[h] = t.m;
      }
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
  float* out = partial + (static_cast<long long>(split) * K + k) * g.Cin * g.Cout;
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

// forward step order: -1 by shape, 0 (tap, group, chunk), 1 (group, chunk, tap) - fv2p_dcn_set_forward_order (tests, A/B runs)
constexpr int kDcnTapInnerDefault = 0;   // forward step order when nobody chose one (measured: see DESIGN 3.6)
static int g_dcn_fwd_order = -1;
static bool dcn_forward_tap_inner(const DcnGeom& g) {
  if (g_dcn_fwd_order >= 0) return g_dcn_fwd_order == 1;
  static const int dev = [] { const char* e = FV2P_DEV_ENV("FV2P_DCN_FWD_ORDER"); return e ? atoi(e) : -1; }();
  if (dev >= 0) return dev == 1;
  return kDcnTapInnerDefault != 0;
}

template <int NB>
static void dcn_fwd_launch(const DcnGeom& g, const float* x, const float* wt_oc, const float* bias, const float* offset, const float* mask,
                           float* y, long long npix, int col_blocks, hipStream_t stream) {
  const dim3 grid(static_cast<unsigned>(ceil_div(npix, 64) * col_blocks));   // one dimension: (tile, column block) decoded XCD-major in the kernel
  const size_t lds_tapin = (2 * NB * 256 + 4 * static_cast<size_t>(g.kh) * g.kw * 16 * 8) * sizeof(float);
  static const bool big_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&dcn_fwd_k<NB, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
  if (dcn_forward_tap_inner(g) && big_ok && lds_tapin <= 96 * 1024)
    hipLaunchKernelGGL((dcn_fwd_k<NB, 1, true>), grid, dim3(256), lds_tapin, stream, g, x, wt_oc, bias, offset, mask, y, 0ll, col_blocks);
  else
    hipLaunchKernelGGL((dcn_fwd_k<NB, 1>), grid, dim3(256), 2 * NB * 256 * sizeof(float), stream, g, x, wt_oc, bias, offset, mask, y, 0ll, col_blocks);
}

// Samples per launch sequence: the kernels address x, the column gradients and the sample lists with 32-bit offsets, so a call is
// cut into chunks of whole samples that stay below those limits (the reference's im2col_step chunking, modulated_deform_conv_cuda.cu:
// 85-118, serves the same purpose) - and, for the backward, below kColgCapBytes of column gradients, which bounds the workspace
// whatever the batch.  Per-pixel arithmetic does not depend on the chunking; the weight gradient adds the chunks in ascending order.
constexpr long long kColgCapBytes = 3ll << 29;   // 1.5 GiB: the MGAF head at batch 4 (1.30 GB) is still one chunk
static long long g_colg_cap = kColgCapBytes;     // fv2p_dcn_set_colg_cap (tests: several chunks at shapes the oracle can answer)
static int dcn_chunk_samples(const DcnGeom& g, bool backward) {
  const long long K = static_cast<long long>(g.kh) * g.kw;
  const long long x_bytes = static_cast<long long>(g.H) * g.W * g.Cin * 4;
  const long long pix = static_cast<long long>(g.Ho) * g.Wo;
  long long bs = g.B > 0 ? g.B : 1;
  bs = std::min(bs, ((1ll << 32) - 1) / std::max(x_bytes, 1ll));
  bs = std::min(bs, ((1ll << 32) - 1) / std::max(pix * std::max<long long>(g.Cout, g.dg * 2 * K) * 4, 1ll));   // y / dy, offset
  if (backward) {
    bs = std::min(bs, std::max(1ll, g_colg_cap / std::max(pix * K * g.Cin * 4, 1ll)));   // the cap is a preference: never below one sample
    bs = std::min(bs, ((1ll << 32) - 1) / std::max(pix * K * g.Cin * 4, 1ll));             // ... the 32-bit limit on its column gradients is hard
    bs = std::min(bs, ((1ll << 31) - 1) / std::max(pix * g.dg * K, 1ll));
  }
  return static_cast<int>(bs);   // 0: one sample alone is above a limit
}

extern "C" int fv2p_dcn_set_forward_order(int order) {
  FV2P_REQUIRE(order >= -1 && order <= 1, FV2P_EINVAL, "dcn_set_forward_order: -1 (by shape), 0 (tap outer) or 1 (tap inner)");
  g_dcn_fwd_order = order;
  return 0;
}

extern "C" int fv2p_dcn_set_colg_cap(int64_t bytes) {
  FV2P_REQUIRE(bytes >= 0, FV2P_EINVAL, "dcn_set_colg_cap: bytes >= 0 (0 = the default)");
  g_colg_cap = bytes > 0 ? bytes : kColgCapBytes;
  return 0;
}

// wt_oc: weight permuted to [kh*kw][Cout][Cin] (input channels contiguous: what the LDS-DMA of the forward kernel fetches 16 bytes at a time)
extern "C" int fv2p_dcn_forward(const float* x_nhwc, const float* wt_oc, const float* bias, const float* offset, const float* mask,
                                DCN_GEOM_ARGS, float* y_nhwc, fv2p_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DCN_GEOM_INIT;
  if (int rc = dcn_check(g)) return rc;
  if (static_cast<long long>(g.B) * g.Ho * g.Wo == 0) return 0;
  FV2P_REQUIRE(x_nhwc && wt_oc && offset && mask && y_nhwc, FV2P_EINVAL, "dcn_forward: null pointer");
  const int bs = dcn_chunk_samples(g, false);
  FV2P_REQUIRE(bs >= 1, FV2P_ELIMIT, "dcn_forward: one sample's input is above 4 GiB");
  const long long K = static_cast<long long>(g.kh) * g.kw, pix = static_cast<long long>(g.Ho) * g.Wo;
  const int nb_all = static_cast<int>(ceil_div(g.Cout, 16));
  for (int s0 = 0; s0 < g.B; s0 += bs) {
    DcnGeom gc = g;
    gc.B = std::min(bs, g.B - s0);
    const long long npix = static_cast<long long>(gc.B) * pix;
    const float* xc = x_nhwc + static_cast<long long>(s0) * g.H * g.W * g.Cin;
    const float* oc = offset + static_cast<long long>(s0) * g.dg * 2 * K * pix;
    const float* mc = mask + static_cast<long long>(s0) * g.dg * K * pix;
    float* yc = y_nhwc + static_cast<long long>(s0) * pix * g.Cout;
    // A workgroup computes 64 pixels x 16*NB columns (32-pixel wave tiles measured 3 - 8 % slower: half the waves per SIMD).  Per-pixel
    // arithmetic does not depend on the plan, so results are identical for every batch size.  256 columns go to one workgroup unless the
    // map is too small to fill the chip: then two column halves, each gathering for itself.
    int nb = nb_all <= 4 ? 4 : (nb_all <= 8 ? 8 : 16);
    if (nb == 16 && ceil_div(npix, 64) < 2 * dcn_cu_count()) nb = 8;
    if (const char* force = FV2P_DEV_ENV("FV2P_DCN_FWD_NB")) {   // development: only the instantiated tiles, anything else is ignored
      const int f = atoi(force);
      if (f == 4 || f == 8 || f == 16) nb = std::max(f, nb_all <= 4 ? 4 : (nb_all <= 8 ? 8 : f));
    }
    const int col_blocks = static_cast<int>(ceil_div(nb_all, nb));
    if (nb == 4) dcn_fwd_launch<4>(gc, xc, wt_oc, bias, oc, mc, yc, npix, col_blocks, stream);
    else if (nb == 8) dcn_fwd_launch<8>(gc, xc, wt_oc, bias, oc, mc, yc, npix, col_blocks, stream);
    else dcn_fwd_launch<16>(gc, xc, wt_oc, bias, oc, mc, yc, npix, col_blocks, stream);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}

// ---- backward: workspace carving shared by the size query and the call
static long long kDwBlocksPerCu = 3;   // workgroups of the weight-gradient kernel per CU (156 VGPRs: three fit; measured 2 / 3 / 4: 1390 / 1368 / 1399 us backward at [4,128,200,176]); FV2P_DCN_DW_BPC overrides
struct DcnBwdPlan {
  long long npix, nkeys, max_entries;
  int splits, pix_per_block, ci_tiles, co_tiles;
  int splits_cap;   // the most splits any chunk of at most this many samples can get: what `partial` is carved for
};
static DcnBwdPlan dcn_bwd_plan(const DcnGeom& g) {
  DcnBwdPlan p;
  { static bool once = false; if (!once) { once = true; if (const char* e = FV2P_DEV_ENV("FV2P_DCN_DW_BPC")) kDwBlocksPerCu = std::max(1, atoi(e)); } }
  const int K = g.kh * g.kw;
  p.npix = static_cast<long long>(g.B) * g.Ho * g.Wo;
  p.nkeys = static_cast<long long>(g.B) * g.dg * (g.H + 1) * (g.W + 1);
  p.max_entries = p.npix * g.dg * K;
  p.ci_tiles = static_cast<int>(ceil_div(g.Cin, 128));
  p.co_tiles = static_cast<int>(ceil_div(g.Cout, 128));
  const long long cols = static_cast<long long>(K) * p.ci_tiles * p.co_tiles;
  long long s = (kDwBlocksPerCu * dcn_cu_count()) / cols;   // one resident round of workgroups
  const long long smax = ceil_div(p.npix > 0 ? p.npix : 1, 64);
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  long long ppb = ceil_div(ceil_div(p.npix > 0 ? p.npix : 1, s), 16) * 16;
  p.pix_per_block = static_cast<int>(ppb);
  p.splits = static_cast<int>(ceil_div(p.npix > 0 ? p.npix : 1, ppb));
  // A SMALLER chunk (the ragged tail of a batch) can get MORE splits than the full one: pix_per_block is rounded up to 16, so
  // e.g. 100 x 88 px at 3 samples gives 83 splits and at 2 samples 85.  Whatever the sample count, splits <= s <= the resident
  // round, and the workspace of the largest chunk serves every chunk: carve for that bound.
  p.splits_cap = static_cast<int>(std::max<long long>(1, std::min<long long>((kDwBlocksPerCu * dcn_cu_count()) / cols, smax)));
  return p;
}
template <typename C>
static void dcn_bwd_carve(C& c, const DcnGeom& g, const DcnBwdPlan& p, float** colg, int** cursor, DcnEntries* entries, void** scan_ws, float** partial) {
  const int K = g.kh * g.kw;
  *colg = c.template take<float>(static_cast<size_t>(p.npix) * K * g.Cin);
  *cursor = c.template take<int>(static_cast<size_t>(p.nkeys) + 1);
  entries->src = c.template take<unsigned>(static_cast<size_t>(p.max_entries));
  entries->lh = c.template take<float>(static_cast<size_t>(p.max_entries));
  entries->lw = c.template take<float>(static_cast<size_t>(p.max_entries));
  *scan_ws = c.template take<char>(scan_ws_bytes(p.nkeys));
  *partial = c.template take<float>(static_cast<size_t>(p.splits_cap) * K * g.Cin * g.Cout);
}
struct SizerC : Sizer {
  template <typename T> T* take(size_t n) { Sizer::take<T>(n); return nullptr; }
};

extern "C" size_t fv2p_dcn_backward_ws_bytes(int batch, int height, int width, int h_out, int w_out, int c_in, int c_out, int kh, int kw,
                                             int deformable_group) {
  DcnGeom g = {batch, height, width, c_in, c_out, h_out, w_out, kh, kw, 1, 1, 0, 0, 1, 1, deformable_group > 0 ? deformable_group : 1};
  g.B = std::max(1, std::min(g.B, dcn_chunk_samples(g, true)));   // the workspace of one chunk serves every chunk
  const DcnBwdPlan p = dcn_bwd_plan(g);
  SizerC s;
  float *a, *e; int* b; DcnEntries c; void* d;
  dcn_bwd_carve(s, g, p, &a, &b, &c, &d, &e);
  return s.bytes();
}

template <int JO, int MC>
static void dcn_col_launch(const DcnGeom& g, const float* x, const float* wt, const float* offset, const float* mask, const float* dy, float* colg,
                           float* doff, float* dmask, long long npix, int seg_split, hipStream_t stream) {
  hipLaunchKernelGGL((dcn_bwd_col_k<JO, 1, MC>), dim3(static_cast<unsigned>(ceil_div(npix, 64) * seg_split)), dim3(256),
                     2 * MC * JO * 256 * sizeof(float), stream, g, x, wt, offset, mask, dy, colg, doff, dmask, 0ll, seg_split);
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
  const int bs = dcn_chunk_samples(g, true);
  FV2P_REQUIRE(bs >= 1, FV2P_ELIMIT, "dcn_backward: one sample's input, column gradients or sample list is above the 32-bit limits");
  FV2P_REQUIRE(ws && ws_bytes >= fv2p_dcn_backward_ws_bytes(g.B, g.H, g.W, g.Ho, g.Wo, g.Cin, g.Cout, g.kh, g.kw, g.dg), FV2P_EWORKSPACE,
               "dcn_backward: workspace too small");
  DcnGeom gmax = g;
  gmax.B = std::min(bs, g.B);
  const DcnBwdPlan pmax = dcn_bwd_plan(gmax);
  Carver c(ws, ws_bytes);
  float *colg, *partial; int* cursor; DcnEntries entries; void* sws;
  dcn_bwd_carve(c, gmax, pmax, &colg, &cursor, &entries, &sws, &partial);
  const long long pix = static_cast<long long>(g.Ho) * g.Wo;
  for (int s0 = 0; s0 < g.B; s0 += bs) {
    DcnGeom gc = g;
    gc.B = std::min(bs, g.B - s0);
    const DcnBwdPlan p = dcn_bwd_plan(gc);
    FV2P_REQUIRE(p.splits <= pmax.splits_cap && p.npix <= pmax.npix && p.nkeys <= pmax.nkeys, FV2P_EWORKSPACE,
                 "dcn_backward: chunk of %d samples does not fit the workspace carved for %d", gc.B, gmax.B);
    const long long cpix = static_cast<long long>(gc.B) * pix;
    const float* xc = x_nhwc + static_cast<long long>(s0) * g.H * g.W * g.Cin;
    const float* oc = offset + static_cast<long long>(s0) * g.dg * 2 * K * pix;
    const float* mc = mask + static_cast<long long>(s0) * g.dg * K * pix;
    const float* dyc = dy_nhwc + static_cast<long long>(s0) * pix * g.Cout;
    float* dxc = dx_nhwc + static_cast<long long>(s0) * g.H * g.W * g.Cin;
    float* doc = doffset + static_cast<long long>(s0) * g.dg * 2 * K * pix;
    float* dmc = dmask + static_cast<long long>(s0) * g.dg * K * pix;
    // 1. per-target sample lists
    FV2P_HIP(hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)(p.nkeys + 1), stream));
    const long long nsamples = cpix * g.dg * K;
    const unsigned iblocks = static_cast<unsigned>(ceil_div(nsamples, 256));
    hipLaunchKernelGGL((dcn_index_k<0>), dim3(iblocks), dim3(256), 0, stream, gc, oc, cursor, entries);
    if (int rc = exclusive_scan_i32(cursor, cursor, p.nkeys, nullptr, sws, scan_ws_bytes(p.nkeys), stream)) return rc;
    hipLaunchKernelGGL((dcn_index_k<1>), dim3(iblocks), dim3(256), 0, stream, gc, oc, cursor, entries);
    hipLaunchKernelGGL(dcn_index_sort_long_k, dim3(static_cast<unsigned>(std::min<long long>(1024, ceil_div(p.nkeys, 256)))), dim3(256), 0, stream,
                       cursor, p.nkeys, entries);
    // 2. column gradients, grad_mask, grad_offset
    const int jo = static_cast<int>(ceil_div(g.Cout, 16));
    const bool mc2 = (g.Cin / g.dg) % 32 == 0;
    // small maps: the (tap, group) segments of a pixel tile go to several workgroups, in equal shares (550 tiles on 256 CUs are two
    // rounds with the second almost empty: [4,256,100,88] 1 327 -> 1 182 us with three shares; at 2 200 tiles shares only cost)
    int seg_split = 1;
    for (int sp = 1; sp <= K * g.dg; ++sp)
      if ((K * g.dg) % sp == 0) { seg_split = sp; if (ceil_div(cpix, 64) * sp >= 6 * dcn_cu_count()) break; }
    if (const char* force = FV2P_DEV_ENV("FV2P_DCN_BWD_SPLIT")) {   // development: a divisor of the segment count or nothing
      const int f = atoi(force);
      if (f >= 1 && f <= K * g.dg && (K * g.dg) % f == 0) seg_split = f;
    }
#define DCN_CJ(JO) do { if (mc2) dcn_col_launch<JO, 2>(gc, xc, wt, oc, mc, dyc, colg, doc, dmc, cpix, seg_split, stream); \
                        else dcn_col_launch<JO, 1>(gc, xc, wt, oc, mc, dyc, colg, doc, dmc, cpix, seg_split, stream); } while (0)
    if (jo <= 1) DCN_CJ(1); else if (jo <= 2) DCN_CJ(2); else if (jo <= 4) DCN_CJ(4); else if (jo <= 8) DCN_CJ(8); else DCN_CJ(16);
#undef DCN_CJ
    // 3. grad_input
    const int cpg = g.Cin / g.dg;
    const long long ntiles = static_cast<long long>(gc.B) * g.dg * ((g.H + 1) / 2) * ((g.W + 3) / 4);
    const unsigned gblocks = static_cast<unsigned>(ceil_div(ntiles, 4));
    if (cpg % 4 == 0 && cpg > 128) hipLaunchKernelGGL((dcn_col2im_k<4>), dim3(gblocks), dim3(256), 0, stream, gc, colg, cursor, entries, dxc);
    else if (cpg > 64) hipLaunchKernelGGL((dcn_col2im_k<2>), dim3(gblocks), dim3(256), 0, stream, gc, colg, cursor, entries, dxc);
    else hipLaunchKernelGGL((dcn_col2im_k<1>), dim3(gblocks), dim3(256), 0, stream, gc, colg, cursor, entries, dxc);
    // 4. weight gradient: the chunk's pixel splits, added to dwt in ascending chunk order
    const dim3 wgrid(static_cast<unsigned>(static_cast<long long>(p.splits) * K * p.ci_tiles * p.co_tiles));
    hipLaunchKernelGGL(dcn_bwd_weight_k, wgrid, dim3(256), 2 * 2 * 16 * kDwPitch * sizeof(float), stream, gc, xc, oc, mc, dyc,
                       p.pix_per_block, p.ci_tiles * p.co_tiles, partial);
    const long long per_chunk = static_cast<long long>(K) * g.Cin * g.Cout;
    hipLaunchKernelGGL(dcn_reduce_k, dim3(static_cast<unsigned>(ceil_div(per_chunk, 256))), dim3(256), 0, stream, partial, p.splits, per_chunk, dwt,
                       s0 > 0 ? 1 : 0);
  }
  FV2P_LAUNCH_CHECK();
  return 0;
}
