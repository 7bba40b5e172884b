// Stand-in for the HIP runtime, for the ADDRESS-SANITIZER build of the HOST side of libdmel_hip.so (tools/asan/run.sh): device memory is
// host memory (so that ASan sees every host-side write into an upload buffer), copies are memcpy, streams / events are dummies and kernel
// launches do nothing.  What runs for real is everything the library does on the host: weight packers and re-tilers, mel / chunk
// tables, workspace planners, argument checks, tile selection, launch-argument assembly.  GPU sanitizers are not available on this
// pool; SURVEY.md section 5 asks for an ASan build of the CPU side.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

extern "C" {
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) std::memmove((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "fake HIP runtime"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }

static thread_local dim3 g_grid, g_block;
static thread_local size_t g_shared;
static thread_local hipStream_t g_stream;
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shared, hipStream_t stream) {
  g_grid = grid; g_block = block; g_shared = shared; g_stream = stream;
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shared, hipStream_t* stream) {
  *grid = g_grid; *block = g_block; *shared = g_shared; *stream = g_stream;
  return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
}
