// How fast are agent-scope atomics on ONE address from the workgroups of one launch (conv_stats_done's counter), compared with spreading them?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void arrive_k(unsigned* c, int spread, int work) {
  // a little work so the launch looks like a real one: all workgroups finish at about the same time
  float v = threadIdx.x;
  for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) c[1000] = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* p = c + (spread ? (blockIdx.x % spread) * 32 : 0);
    __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void arrive_noret_k(unsigned* c, int spread, int work) {
  float v = threadIdx.x;
  for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) c[1000] = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* p = c + (spread ? (blockIdx.x % spread) * 32 : 0);
    __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
__global__ void none_k(unsigned* c, int spread, int work) {
  float v = threadIdx.x;
  for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) c[1000] = 1;
}
__global__ void fold_k(const double* s, int c, double* out, int depth) {   // one workgroup reads 64 x 2 x c doubles with agent-scope loads
  const int tid = threadIdx.x, L = 256 / c, e = tid % c, lq = tid / c;
  double a = 0, b = 0;
  if (depth == 0) {
#pragma unroll 8
    for (int q = lq; q < 64; q += L) { a += __hip_atomic_load(s + (q * 2 + 0) * c + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b += __hip_atomic_load(s + (q * 2 + 1) * c + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  } else {
    for (int q0 = lq; q0 < 64; q0 += L * 16) {
      double va[16], vb[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { int q = q0 + L * i; bool on = q < 64; va[i] = on ? __hip_atomic_load(s + (q * 2 + 0) * c + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0; vb[i] = on ? __hip_atomic_load(s + (q * 2 + 1) * c + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0; }
#pragma unroll
      for (int i = 0; i < 16; ++i) { a += va[i]; b += vb[i]; }
    }
  }
  out[tid] = a + b;
}
template <class F> float time_us(F f, int reps = 200) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}
int main() {
  unsigned* c; hipMalloc(&c, 1 << 20); hipMemset(c, 0, 1 << 20);
  double* s; hipMalloc(&s, 64 * 2 * 128 * 8); hipMemset(s, 0, 64 * 2 * 128 * 8);
  double* o; hipMalloc(&o, 256 * 8);
  for (int wgs : {128, 488, 1024, 2200}) {
    for (int work : {2000, 20000}) {
      float t0 = time_us([&] { hipLaunchKernelGGL(none_k, dim3(wgs), dim3(256), 0, 0, c, 0, work); });
      printf("wgs %4d work %5d: no atomic %.2f us", wgs, work, t0);
      for (int spread : {0, 8, 64}) {
        float t1 = time_us([&] { hipLaunchKernelGGL(arrive_k, dim3(wgs), dim3(256), 0, 0, c, spread, work); });
        printf(" | spread %2d: +%.2f us", spread, t1 - t0);
      }
      float t2 = time_us([&] { hipLaunchKernelGGL(arrive_noret_k, dim3(wgs), dim3(256), 0, 0, c, 0, work); });
      printf(" | +wait %.2f\n", t2 - t0);
    }
  }
  for (int ch : {16, 32, 64, 128})
    for (int depth : {0, 1})
      printf("fold c=%3d depth %d: %.2f us\n", ch, depth, time_us([&] { hipLaunchKernelGGL(fold_k, dim3(1), dim3(256), 0, 0, s, ch, o, depth); }));
  return 0;
}
