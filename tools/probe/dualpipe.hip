// Probe: do fp32 MFMA waves and fp32 VALU-FMA waves on the same SIMD overlap?  (MI355X: both pipes peak at 64 FLOP/clk/SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

// mode 0: every wave MFMA; 1: every wave VALU; 2: even waves MFMA, odd waves VALU (same SIMD hosts both kinds)
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, float seed) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || (mode == 2 && (wave & 4) == 0);   // waves 0-3 / 4-7 land on SIMDs 0-3 twice
  float r = 0.f;
  if (do_mfma) {
    floatx16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = seed + threadIdx.x, y = seed * 0.5f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    for (int j = 0; j < 16; ++j) r += a0[j] + a1[j] + a2[j] + a3[j];
  } else {
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = j * seed;
    float x = seed + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = fmaf(acc[j], x, 1.0001f);   // 128 FMAs per iteration = 4 MFMAs worth of FLOPs
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) r += acc[j];
  }
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  const int iters = 20000;
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, 100, mode, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // every wave does iters * 4 * 4096 FLOP (MFMA) or iters * 128 * 64 * 2 = iters * 16384 FLOP (VALU): same
    double flop = 256.0 * 8 * iters * 16384.0;
    printf("mode %d: %.3f ms  %.1f TFLOP/s\n", mode, ms, flop / ms / 1e9);
  }
  return 0;
}
