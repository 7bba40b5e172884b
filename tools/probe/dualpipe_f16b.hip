// Probe (round 3, second part): WHO gets the vector issue port when a v_mfma_f32_32x32x16_f16 stream and an fp32 VALU stream share a SIMD?
// dualpipe_f16.hip showed t(both) = t(matrix) + t(vector) when the matrix waves are the OLDER ones.  This probe stamps every wave's own
// duration (s_memrealtime, 100 MHz) so that starvation (one class runs at full speed, the other waits) can be told from a shared pipe
// (both slowed), and varies age and priority:
//   roles: 'M' matrix waves have the low wave indices (dispatched first = older), 'V' vector waves have the low indices
//   prio:  0 none, 1 s_setprio(3) on the vector waves, 2 s_setprio(3) on the matrix waves
//   gap:   number of s_nop 7 (8 idle cycles each) the matrix stream inserts after every MFMA (does a non-greedy matrix stream let VALU through?)
//   build: hipcc -O3 --offload-arch=gfx950 tools/probe/dualpipe_f16b.hip -o tools/probe/dualpipe_f16b
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NW, int GAP>
__global__ __launch_bounds__(64 * NW) void k(float* out, unsigned long long* stamps, int iters, int vfirst, int prio, int run_m, int run_v, float seed) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool matrix = vfirst ? wave >= NW - 4 : wave < 4;
  float r = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (matrix) {
    if (!run_m) return;
    if (prio == 2) __builtin_amdgcn_s_setprio(3);
    floatx16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(seed + 0.001f * (threadIdx.x + j)); y[j] = (_Float16)(seed * 0.5f + 0.002f * j); }
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
      if (GAP >= 1) __builtin_amdgcn_sched_barrier(0);
      if (GAP >= 1) asm volatile("s_nop 7");
      if (GAP >= 2) asm volatile("s_nop 7");
      if (GAP >= 3) asm volatile("s_nop 7");
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
      if (GAP >= 1) __builtin_amdgcn_sched_barrier(0);
      if (GAP >= 1) asm volatile("s_nop 7");
      if (GAP >= 2) asm volatile("s_nop 7");
      if (GAP >= 3) asm volatile("s_nop 7");
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a2, 0, 0, 0);
      if (GAP >= 1) __builtin_amdgcn_sched_barrier(0);
      if (GAP >= 1) asm volatile("s_nop 7");
      if (GAP >= 2) asm volatile("s_nop 7");
      if (GAP >= 3) asm volatile("s_nop 7");
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, y, a3, 0, 0, 0);
      if (GAP >= 1) __builtin_amdgcn_sched_barrier(0);
      if (GAP >= 1) asm volatile("s_nop 7");
      if (GAP >= 2) asm volatile("s_nop 7");
      if (GAP >= 3) asm volatile("s_nop 7");
    }
    for (int j = 0; j < 16; ++j) r += a0[j] + a1[j] + a2[j] + a3[j];
  } else {
    if (!run_v) return;
    if (prio == 1) __builtin_amdgcn_s_setprio(3);
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = j * seed;
    const float x = seed + threadIdx.x * 1e-3f, c = seed * 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = fmaf(acc[j], x, c);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) r += acc[j];
  }
  out[blockIdx.x * 64 * NW + threadIdx.x] = r;
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * NW + wave] = t1 - t0;
}

template <int NW, int GAP> static void run(float* d, unsigned long long* st, int iters) {
  std::vector<unsigned long long> h(256 * NW);
  struct Cfg { int vfirst, prio, rm, rv; const char* name; };
  const Cfg cfgs[] = {
      {0, 0, 1, 0, "matrix only                      "}, {0, 0, 0, 1, "vector only                      "},
      {0, 0, 1, 1, "both, matrix older               "}, {1, 0, 1, 1, "both, vector older               "},
      {0, 1, 1, 1, "both, matrix older, vector prio 3"}, {1, 2, 1, 1, "both, vector older, matrix prio 3"},
      {0, 2, 1, 1, "both, matrix older, matrix prio 3"}, {1, 1, 1, 1, "both, vector older, vector prio 3"},
  };
  for (const Cfg& c : cfgs) {
    hipLaunchKernelGGL((k<NW, GAP>), dim3(256), dim3(64 * NW), 0, 0, d, st, 100, c.vfirst, c.prio, c.rm, c.rv, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipMemset(st, 0, 256 * NW * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NW, GAP>), dim3(256), dim3(64 * NW), 0, 0, d, st, iters, c.vfirst, c.prio, c.rm, c.rv, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h.data(), st, 256 * NW * 8, hipMemcpyDeviceToHost);
    double tm = 0, tv = 0;
    int nm = 0, nv = 0;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < NW; ++w) {
        const bool matrix = c.vfirst ? w >= NW - 4 : w < 4;
        const double us = h[b * NW + w] * 0.01;
        if (us == 0) continue;
        if (matrix) { tm += us; ++nm; } else { tv += us; ++nv; }
      }
    printf("NW %2d gap %d  %s: kernel %7.3f ms   matrix waves %8.1f us   vector waves %8.1f us\n", NW, GAP, c.name, ms, nm ? tm / nm : 0.0, nv ? tv / nv : 0.0);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
}

int main() {
  float* d;
  unsigned long long* st;
  (void)hipMalloc(&d, 256 * 64 * 16 * 4);
  (void)hipMalloc(&st, 256 * 16 * 8);
  const int iters = 40000;
  run<8, 0>(d, st, iters);
  run<12, 0>(d, st, iters);
  run<12, 1>(d, st, iters);
  run<12, 2>(d, st, iters);
  (void)hipFree(d);
  (void)hipFree(st);
  return 0;
}
