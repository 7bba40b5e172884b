// Debug probe (round 3): kernels that leave garbage behind -- in ~250 VGPRs per lane (mode 0) or in 64 KB of LDS (mode 1) -- to find out whether
// another kernel's result depends on what ran on its CU before (an uninitialised register or LDS read).  Build: hipcc -O3 --offload-arch=gfx950
// -shared -fPIC -o tools/probe/libpoison.so tools/probe/poison.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(256) void poison_vgpr(float* sink, int n) {
  float a[240];
#pragma unroll
  for (int i = 0; i < 240; ++i) a[i] = __uint_as_float(0x7fc00000u + i + threadIdx.x);      // NaNs with distinct payloads
  // keep them all live across a loop the compiler cannot fold
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 240; ++i) a[i] = __uint_as_float(__float_as_uint(a[i]) ^ (uint32_t)(it + i));
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 240; ++i) acc ^= __float_as_uint(a[i]);
  if (acc == 0x12345678u) sink[threadIdx.x] = 1.f;
}

__global__ __launch_bounds__(256) void poison_lds(float* sink, int n) {
  extern __shared__ float sm[];
  for (int it = 0; it < n; ++it)
    for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = __uint_as_float(0x7fc00000u + i + it);
  __syncthreads();
  if (sm[(threadIdx.x * 7) & 16383] == 1.2345f) sink[threadIdx.x] = 1.f;
}

extern "C" int poison_launch(int mode, int blocks, int n, float* sink, void* stream) {
  if (mode == 0) hipLaunchKernelGGL(poison_vgpr, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, n);
  else hipLaunchKernelGGL(poison_lds, dim3(blocks), dim3(256), 65536, (hipStream_t)stream, sink, n);
  return (int)hipGetLastError();
}
