// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 with NACC rotating accumulators, optionally fed by ds_read_b128.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/mfma_rate.hip -o gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float w[4096];
  for (int e = threadIdx.x; e < 4096; e += 256) w[e] = 1.0f / (1 + e);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = lane * 0.001f;
  float4 b = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = LDS ? *reinterpret_cast<const float4*>(&w[((j * 4 + q) * 64 + lane) * 4]) : b;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[(0) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[q].x, acc[(0) % NACC], 0, 0, 0);
        acc[(1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[q].y, acc[(1) % NACC], 0, 0, 0);
        acc[(2) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[q].z, acc[(2) % NACC], 0, 0, 0);
        acc[(3) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[q].w, acc[(3) % NACC], 0, 0, 0);
      }
    }
    asm volatile("" : "+v"(a));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks, int iters) {
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_wave = 64.0 * iters;
  const double flops = mfma_per_wave * 2048 * blocks * 4;
  printf("%-28s blocks=%4d iters=%d  %.1f us  %.1f TF/s  -> %.1f ns per MFMA per wave-slot (waves/SIMD=%.1f)\n", name, blocks, iters, ms * 1e3,
         flops / ms / 1e9, ms * 1e6 / mfma_per_wave / (blocks / 256.0 > 1 ? blocks / 256.0 : 1), blocks / 256.0);
  hipFree(out);
}

int main() {
  for (int blocks : {256, 512, 1024}) {
    run<4, false>("4 acc, reg operands", blocks, 400);
    run<4, true>("4 acc, ds_read_b128 operands", blocks, 400);
    run<2, false>("2 acc, reg operands", blocks, 400);
    run<1, false>("1 acc (dependent chain)", blocks, 400);
  }
  return 0;
}
