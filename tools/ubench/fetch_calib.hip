// Calibration of rocprofv3 FETCH_SIZE on gfx950 for the access patterns of this repo's kernels (MI355X_MICROARCH.md, HBM section: "a wide
// coalesced streaming read reports exactly half of its bytes ... other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern").  Every kernel reads each byte of a buffer exactly ONCE (buffer >> L2 + Infinity Cache is not required: the
// buffer is written by the host copy long before, and each kernel touches a buffer of its own):
//   stream16   lane l of a wave reads 16 B at consecutive addresses (1 KiB per wave instruction)
//   rows512    a 512-byte row per 32 lanes, rows in a random order (the sparse conv's row gather at 128 channels)
//   rows256    a 256-byte row per 16 lanes, random order (64 channels)
//   seg64      a 64-byte segment per 4 lanes, random order (the DCN kernels' corner gather: 16 channels of one pixel)
//   dword      4 B per lane, consecutive
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/ubench/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(e) do { hipError_t ck_ = (e); if (ck_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(ck_), __LINE__); exit(1); } } while (0)

__global__ void stream16(const float4* __restrict__ p, size_t n, float* __restrict__ sink) {
  const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 v = p[i];
  if (v.x == 12345.678f) sink[0] = v.y;
}
__global__ void dword(const float* __restrict__ p, size_t n, float* __restrict__ sink) {
  const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = p[i];
  if (v == 12345.678f) sink[0] = v;
}
// SEG = 16-byte pieces per segment (32: 512-B rows, 16: 256 B, 4: 64 B); segment order from perm
template <int SEG>
__global__ void gather(const float4* __restrict__ p, const int* __restrict__ perm, size_t nseg, float* __restrict__ sink) {
  const size_t t = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  const size_t s = t / SEG;
  if (s >= nseg) return;
  const float4 v = p[static_cast<size_t>(perm[s]) * SEG + t % SEG];
  if (v.x == 12345.678f) sink[0] = v.y;
}

int main() {
  const size_t bytes = 64ull << 20;   // per kernel
  float* sink;
  CK(hipMalloc(&sink, 16));
  std::vector<float> host(bytes / 4, 1.0f);
  auto fresh = [&]() { float* d; CK(hipMalloc(&d, bytes)); CK(hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice)); return d; };
  auto perm_of = [&](size_t n) { std::vector<int> h(n); std::iota(h.begin(), h.end(), 0); std::mt19937 r(7); std::shuffle(h.begin(), h.end(), r);
                                 int* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d; };
  float* a = fresh(); float* b = fresh(); float* c = fresh(); float* e = fresh(); float* f = fresh();
  int* p512 = perm_of(bytes / 512); int* p256 = perm_of(bytes / 256); int* p64 = perm_of(bytes / 64);
  CK(hipDeviceSynchronize());
  const size_t n16 = bytes / 16;
  hipLaunchKernelGGL(stream16, dim3((n16 + 255) / 256), dim3(256), 0, 0, reinterpret_cast<const float4*>(a), n16, sink);
  hipLaunchKernelGGL(dword, dim3((bytes / 4 + 255) / 256), dim3(256), 0, 0, b, bytes / 4, sink);
  hipLaunchKernelGGL((gather<32>), dim3((n16 + 255) / 256), dim3(256), 0, 0, reinterpret_cast<const float4*>(c), p512, bytes / 512, sink);
  hipLaunchKernelGGL((gather<16>), dim3((n16 + 255) / 256), dim3(256), 0, 0, reinterpret_cast<const float4*>(e), p256, bytes / 256, sink);
  hipLaunchKernelGGL((gather<4>), dim3((n16 + 255) / 256), dim3(256), 0, 0, reinterpret_cast<const float4*>(f), p64, bytes / 64, sink);
  CK(hipDeviceSynchronize());
  printf("each kernel read %zu bytes once (gather kernels: + their index array, %zu / %zu / %zu bytes)\n", bytes, bytes / 512 * 4, bytes / 256 * 4, bytes / 64 * 4);
  return 0;
}
