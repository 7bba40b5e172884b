// Probe (round 3): do v_mfma_f32_32x32x16_f16 waves and fp32 VALU waves on the same SIMD overlap?  tools/probe/dualpipe.hip asked the
// same of the fp32-input MFMA (answer: no, it runs on the vector FMA lanes).  The f16 MFMA is what conv_bf16_kernel<NP=2> issues, and the
// question decides whether a convolution kernel whose PRODUCER waves compute the anti-aliased activation (VALU) while its CONSUMER waves
// run the MFMA loop costs max(VALU, MFMA) or their sum.
//   build: hipcc -O3 --offload-arch=gfx950 tools/probe/dualpipe_f16.hip -o tools/probe/dualpipe_f16
// Workgroup = NW waves.  Roles by wave index: waves [0, 4) are "matrix" waves, waves [4, NW) are "vector" waves (a workgroup's waves go
// to SIMDs round-robin, so every SIMD hosts one matrix wave and (NW - 4) / 4 vector waves).
//   mode 0: matrix waves run, vector waves exit           -> t_m
//   mode 1: vector waves run, matrix waves exit           -> t_v
//   mode 2: both run                                      -> t_mv   (co-issue: ~max(t_m, t_v); shared pipe: ~t_m + t_v)
//   mode 3: every wave runs an interleaved stream: 1 MFMA + F independent v_fma_f32 per loop step (same-wave co-issue)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NW, int F>
__global__ __launch_bounds__(64 * NW) void k(float* out, int iters, int mode, float seed) {
  const int wave = threadIdx.x >> 6;
  const bool matrix = wave < 4;
  float r = 0.f;
  if (mode == 3) {
    floatx16 a0 = {0}, a1 = {0};
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(seed + 0.001f * (threadIdx.x + j)); y[j] = (_Float16)(seed * 0.5f + 0.002f * j); }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = j * seed;
    const float xs = seed + threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < F; ++j) acc[j % 8] = fmaf(acc[j % 8], xs, 1.0001f);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < F; ++j) acc[(j + 4) % 8] = fmaf(acc[(j + 4) % 8], xs, 1.0001f);
      }
    }
    for (int j = 0; j < 16; ++j) r += a0[j] + a1[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) r += acc[j];
  } else if (matrix) {
    if (mode == 1) return;
    floatx16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(seed + 0.001f * (threadIdx.x + j)); y[j] = (_Float16)(seed * 0.5f + 0.002f * j); }
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, y, a3, 0, 0, 0);
    }
    for (int j = 0; j < 16; ++j) r += a0[j] + a1[j] + a2[j] + a3[j];
  } else {
    if (mode == 0) return;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = j * seed;
    const float x = seed + threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = fmaf(acc[j], x, 1.0001f);   // 32 FMAs per iteration: 128 issue cycles for a lone wave
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) r += acc[j];
  }
  out[blockIdx.x * 64 * NW + threadIdx.x] = r;
}

template <int NW, int F> static void run(float* d, int iters) {
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL((k<NW, F>), dim3(256), dim3(64 * NW), 0, 0, d, 100, mode, 1.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<NW, F>), dim3(256), dim3(64 * NW), 0, 0, d, iters, mode, 1.0f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
      hipEventDestroy(e0); hipEventDestroy(e1);
    }
    // per CU and iteration: mode 0/2: 4 matrix waves x 4 MFMAs; mode 3: NW waves x 4 MFMAs
    const double mfmas = (mode == 1 ? 0.0 : mode == 3 ? (double)NW * 4 : 16.0) * iters * 256.0;
    const double fmas = (mode == 0 ? 0.0 : mode == 3 ? (double)NW * 4 * F : (double)(NW - 4) * 32) * iters * 256.0;   // wave-instructions
    printf("NW %2d F %d mode %d: %8.3f ms   MFMA %7.1f TF/s (f16 32x32x16)   VALU %6.2f T wave-fma/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", NW, F, mode,
           best, mfmas * 32768.0 / best / 1e9, fmas / best / 1e9, mfmas > 0 ? best * 1e-3 * 2.4e9 / (mfmas / 1024.0) : 0.0);
  }
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 64 * 16 * 4);
  const int iters = 40000;
  run<8, 4>(d, iters);     // one matrix + one vector wave per SIMD
  run<12, 5>(d, iters);    // one matrix + two vector waves per SIMD (the fused convolution's 4 consumers + 8 producers)
  run<16, 6>(d, iters);    // one matrix + three vector waves per SIMD
  hipFree(d);
  return 0;
}
