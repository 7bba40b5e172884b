// Shared host-side helpers of libdmel_hip.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dmel_hip.h"

namespace dmel {

inline bool valid_precision(int p) { return p == DMEL_PRECISION_FP32 || p == DMEL_PRECISION_BF16 || p == DMEL_PRECISION_FP32_MFMA; }


void set_error(const char* fmt, ...);

#define DMEL_CHECK_ARG(cond, ...)         \
  do {                                    \
    if (!(cond)) {                        \
      ::dmel::set_error(__VA_ARGS__);     \
      return DMEL_EINVAL;                 \
    }                                     \
  } while (0)

#define DMEL_HIP(call)                                                                   \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess) {                                                             \
      ::dmel::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return (int)e__;                                                                   \
    }                                                                                    \
  } while (0)

#define DMEL_TRY(call)       \
  do {                       \
    int r__ = (call);        \
    if (r__ != DMEL_OK) return r__; \
  } while (0)

// Device buffer owned by a handle (weights, tables).  Not used for activations (caller-owned).
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  int upload(const void* host, size_t n) {
    release();
    if (n == 0) return DMEL_OK;
    DMEL_HIP(hipMalloc(&p, n));
    bytes = n;
    DMEL_HIP(hipMemcpy(p, host, n, hipMemcpyHostToDevice));
    return DMEL_OK;
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Per-family launch timing (bench.py's roofline leg): hipEvents on the launch stream.
struct ProfScope {
  ProfScope(const char* family, hipStream_t s, double flops, double bytes);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};

// Training precision of the calling thread for the duration of one training entry point (dmel_*_forward_train / _backward):
// DMEL_PRECISION_BF16 makes every convolution launched inside -- forward, backward-data and the long-row weight gradients -- run with
// bf16-rounded operands and fp32 accumulation (what `precision: bf16-mixed` autocast does to the reference's conv1d / conv2d); -1 = no
// override (each launch's own precision).  Thread-local: handles stay re-entrant.
int& train_precision_override();
struct TrainPrecisionScope {
  int saved;
  explicit TrainPrecisionScope(int precision) : saved(train_precision_override()) {
    train_precision_override() = precision == DMEL_PRECISION_BF16 ? DMEL_PRECISION_BF16 : -1;
  }
  ~TrainPrecisionScope() { train_precision_override() = saved; }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
  char* base;
  size_t cap, off = 0;
  bool dry;  // dry run: only measure
  Arena(void* b, size_t c) : base((char*)b), cap(c), dry(b == nullptr) {}
  template <class T> T* take(size_t n) {
    size_t o = align_up(off, 256);
    off = o + n * sizeof(T);
    if (dry) return nullptr;
    return reinterpret_cast<T*>(base + o);
  }
  bool ok() const { return dry || off <= cap; }
};

}  // namespace dmel
