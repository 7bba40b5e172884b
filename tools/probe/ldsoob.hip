// Debug probe (round 3): does the hardware keep a workgroup's out-of-range LDS writes away from ANOTHER workgroup's LDS on the same CU?
// victim: fills 32 KB of LDS with a pattern, spins, counts changed words.  rogue: allocates 1 KB of dynamic LDS and writes 0xDEADBEEF to word
// offsets up to 64 KB.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/probe/ldsoob tools/probe/ldsoob.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void victim(unsigned* bad, int spins) {
  extern __shared__ unsigned sm[];
  for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = 0x1000000u + i;
  __syncthreads();
  unsigned n = 0;
  for (int s = 0; s < spins; ++s) {
    for (int i = threadIdx.x; i < 8192; i += 256) n += (sm[i] != 0x1000000u + i);
    __builtin_amdgcn_s_sleep(8);
  }
  if (n) atomicAdd(bad, n);
}
__global__ __launch_bounds__(256) void rogue(unsigned* sink, int reps) {
  extern __shared__ unsigned sm[];
  volatile unsigned* p = sm;
  for (int r = 0; r < reps; ++r)
    for (int i = threadIdx.x; i < 16384; i += 256) p[i] = 0xDEADBEEFu;      // 64 KB of words; only the first 256 are ours
  if (p[threadIdx.x] == 0x12345u) sink[0] = 1;
}
int main() {
  unsigned *bad, *sink;
  hipMalloc(&bad, 4); hipMalloc(&sink, 4); hipMemset(bad, 0, 4);
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  for (int it = 0; it < 20; ++it) {
    hipLaunchKernelGGL(victim, dim3(512), dim3(256), 32768, a, bad, 200);
    hipLaunchKernelGGL(rogue, dim3(2048), dim3(256), 1024, b, sink, 50);
  }
  hipDeviceSynchronize();
  unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("victim words seen changed by a co-resident workgroup's out-of-range LDS writes: %u (0 = the hardware isolates LDS)\n", h);
  return 0;
}
