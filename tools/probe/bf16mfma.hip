// Probe: sustained rate of v_mfma_f32_32x32x16_bf16 on the whole chip, operands in registers.
//   mode 0: 4 accumulators x 6 MFMAs (the split-fp32 inner step), small-integer operands
//   mode 1: same, random-bit operands (toggle-heavy: shows the power-limited clock)
//   mode 2: like 1 but with 6 ds_read_b128 per 24 MFMAs in front (LDS-fed)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k(float* out, const uint4* src, int iters, int mode) {
  __shared__ uint4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = src[i];
  __syncthreads();
  floatx16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  uint4 a[6], b[6];
  for (int j = 0; j < 6; ++j) { a[j] = src[(threadIdx.x * 7 + j * 64) & 1023]; b[j] = src[(threadIdx.x * 3 + j * 64 + 17) & 1023]; }
  for (int i = 0; i < iters; ++i) {
    if (mode == 2) {
      const uint4* p = lds + ((threadIdx.x + i) & 63);
#pragma unroll
      for (int j = 0; j < 6; ++j) b[j] = p[j * 64];
    }
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(t + (j >> 1)) % 6]), __builtin_bit_cast(bf16x8, b[(t + (j & 1)) % 6]),
                                                         acc[j], 0, 0, 0);
  }
  float r = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int q = 0; q < 16; ++q) r += acc[j][q];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  float* d;
  uint4* s;
  hipMalloc(&d, 2048 * 256 * 4);
  hipMalloc(&s, 1024 * 16);
  uint32_t* h = (uint32_t*)malloc(1024 * 16);
  for (int mode = 0; mode < 3; ++mode) {
    uint32_t st = 12345;
    for (int i = 0; i < 4096; ++i) {
      st = st * 1664525u + 1013904223u;
      // small integers 0..3 as bf16 pairs (mode 0) or random mantissas with exponent near 1.0 (modes 1, 2)
      if (mode == 0) h[i] = 0x3f803f80u;
      else h[i] = ((0x3f00u | ((st >> 8) & 0xff)) << 16) | (0x3f00u | ((st >> 20) & 0xff)) | ((st & 1) << 31) | ((st & 2) << 14);
    }
    hipMemcpy(s, h, 1024 * 16, hipMemcpyHostToDevice);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
      const int iters = 4000, grid = 256 * blocks_per_cu;
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, s, 10, mode);
      hipDeviceSynchronize();
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, s, iters, mode);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)grid * 4 * iters * 24 * 32768.0;
      double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 24 * blocks_per_cu);
      printf("mode %d, %d wave(s)/SIMD: %.3f ms  %.0f TFLOP/s executed  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", mode, blocks_per_cu, ms,
             flop / ms / 1e9, cyc);
    }
  }
  return 0;
}
