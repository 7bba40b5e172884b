// Shared host-side helpers of libdmel_hip.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dmel_hip.h"

namespace dmel {

inline bool valid_precision(int p) { return p >= DMEL_PRECISION_FP32 && p <= DMEL_PRECISION_FP32_BF16X3; }
// precision of a launch inside a training entry point / a backward pass: gradients span too many binades for the fp16 split, and the
// bf16 inference mode is not a training mode (dmel_*_set_train_precision is)
inline int exact_precision(int p) {
  return (p == DMEL_PRECISION_BF16 || p == DMEL_PRECISION_FP32_F16X2 || p == DMEL_PRECISION_FP32_BF16X3) ? DMEL_PRECISION_FP32 : p;
}

// fp32 -> fp16 bits, round to nearest even, subnormals kept, overflow to inf: what v_cvt_f16_f32 does in the default mode (the host packer
// and the device re-pack kernel must produce identical images)
inline uint16_t f32_to_f16_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                   // NaN
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                  // >= 65520 rounds to inf
  if (a < 0x33000001u) return sign;                                         // <= 2^-25 rounds to zero (ties to even)
  const int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u;                                 // 24-bit significand
  int shift = e >= -14 ? 13 : 13 + (-14 - e);                               // bits dropped
  const uint32_t half = 1u << (shift - 1), mask = (1u << shift) - 1u;
  uint32_t q = m >> shift;
  const uint32_t rem = m & mask;
  if (rem > half || (rem == half && (q & 1u))) ++q;
  // normal: q in [2^10, 2^11]; exponent field e + 15, the implicit bit adds into it (a carry to 2^11 bumps the exponent)
  const uint32_t bits = e >= -14 ? ((uint32_t)(e + 14) << 10) + q : q;
  return (uint16_t)(sign | bits);
}
inline float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
  float f;
  if (e == 0) f = std::ldexp((float)m, -24);
  else if (e == 31) f = m ? NAN : INFINITY;
  else f = std::ldexp((float)(m | 0x400u), (int)e - 25);
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u |= sign;
  std::memcpy(&f, &u, 4);
  return f;
}
constexpr float kF16XScale = 0.015625f;      // staged activations x 2^-6 ...
constexpr float kF16WScale = 64.f;           // ... weight image x 2^6 (exact powers of two: the product is unchanged)
constexpr float kF16LoScale = 2048.f;        // second piece is stored x 2^11


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

// Library-owned scratch of the calling thread, per (current device, stream, purpose): two streams driven by one thread -- two models, a
// side-stream backward, the lanes of dmel_codec_amd/pipeline.py -- must not share partial tiles or folded weights (launches are ordered
// within a stream only).  Never freed: thread_local destructors would call hipFree at thread / process exit, possibly after the HIP
// runtime has shut down; the buffers (a few MB per stream that ever trained) go with the process.
DevBuf* thread_scratch(int which, hipStream_t stream);      // which: 0 absmax ring, 1 weight-gradient partial tiles, 2 folded discriminator weights

// A backward pass clears its whole flat gradient buffer with ONE memset and declares the range here; launches inside it that would clear
// their own slot first (weight / bias gradients that accumulate with atomics) ask zero_unless_cleared and skip theirs -- round 2 issued
// ~480 memsets per training step that way.  Per thread, nested scopes restore the outer range.
struct ClearedRange {
  ClearedRange(void* p, size_t bytes, hipStream_t s);      // error(): result of the memset
  ~ClearedRange();
  int error() const { return err; }
  const char *lo, *hi, *prev_lo, *prev_hi;
  int err;
};
int zero_unless_cleared(void* p, size_t bytes, hipStream_t s);

// Per-family launch timing (bench.py's roofline leg): hipEvents on the launch stream.
struct ProfScope {
  // issue_flops: matrix-core flops actually issued for `flops` algorithmic ones (x6 for the bf16 split, x3 for the fp16 split, ...)
  ProfScope(const char* family, hipStream_t s, double flops, double bytes, double issue_flops = 0.0);
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
