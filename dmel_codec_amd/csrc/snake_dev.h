// Device-side pieces of the anti-aliased Snake shared by aa_snake.hip and conv_snake.hip.  Internal to the library.
#pragma once
#include "ops.h"

namespace dmel {

struct Taps12 {
  float f[12];
};

// sin^2(x) for the Snake term.  Reduce x to r = x - n*(pi/2), |r| <= pi/4 (three-term Cody-Waite, exact products for
// |n| < 2^13), evaluate the odd degree-9 polynomial of sin(r) and use sin^2(x) = s^2 (n even) or 1 - s^2 (n odd): the
// sign of sin never matters.  Measured against fp64 on 1e7 points, |x| up to 3e4: max abs error 1.2e-7, rms 2.2e-8 --
// slightly better than squaring a correctly rounded sinf -- at a third of the instructions of the library path.
__device__ __forceinline__ float sin_sq(float x) {
  if (fabsf(x) > 8192.f) {
    const float s = sinf(x);
    return s * s;
  }
  const float n = rintf(x * 0.63661977236758134308f);
  float r = fmaf(n, -1.5703125f, x);
  r = fmaf(n, -4.837512969970703125e-4f, r);
  r = fmaf(n, -7.54978995489188216e-8f, r);
  const float r2 = r * r;
  float p = 2.7557314e-06f;
  p = fmaf(p, r2, -1.9841270e-04f);
  p = fmaf(p, r2, 8.3333333e-03f);
  p = fmaf(p, r2, -1.6666667e-01f);
  const float s = fmaf(r * r2, p, r);
  const float s2 = s * s;
  return ((int)n & 1) ? 1.0f - s2 : s2;
}

// Branch-free body of sin_sq for |x| <= 8192, and the whole function out of line.  Callers that evaluate several values test ALL of them
// with one wave-uniform branch (ballot) and run either the straight-line bodies -- which the compiler can then interleave: with the
// branch inside, every evaluation is a basic block of its own and four independent 20-deep dependency chains run one after the other --
// or, when any lane of the wave holds a large argument (never, for a trained network), the out-of-line copy.  Same values, bit for bit.
__device__ __forceinline__ float sin_sq_small(float x) {
  // sin^2 has period pi: reduce x to r = x - n pi, |r| <= pi / 2 (three-term Cody-Waite: pi = 3.140625 + 9.67502593994140625e-4 +
  // 1.509957990978376432e-7, the products n * piece exact for |n| < 2^13), then (r + r^3 P(r^2))^2 with a degree-11 odd near-minimax
  // sine.  No quadrant bookkeeping (convert, and, compare, 1 - s^2, select: 5 instructions of the pi/2 form): 13 instructions instead of
  // 19.  Against fp64 on 1e7 points, |x| <= 8e3: max abs error 2.5e-7, rms 4e-8 (the pi/2 form: 1.2e-7 / 2.2e-8); the activation tests
  // hold the kernel to 1e-5 of the oracle.
  const float n = rintf(x * 0.318309886183790671538f);
  float r = fmaf(n, -3.140625f, x);
  r = fmaf(n, -9.67502593994140625e-4f, r);
  r = fmaf(n, -1.509957990978376432e-7f, r);
  const float r2 = r * r;
  float p = -2.3841987939476894e-08f;
  p = fmaf(p, r2, 2.7522303298610495e-06f);
  p = fmaf(p, r2, -1.9840797176584601e-04f);
  p = fmaf(p, r2, 8.333330042660236e-03f);
  p = fmaf(p, r2, -1.666666716337204e-01f);
  const float s = fmaf(r * r2, p, r);
  return s * s;
}
static __device__ __attribute__((noinline)) float sin_sq_call(float x) { return sin_sq(x); }
// v[i] <- v[i] + inv_b[i] * sin^2(a[i] * v[i]), N values at once
template <int N> __device__ __forceinline__ void snake_n(float (&v)[N], const float (&a)[N], const float (&inv_b)[N]) {
  float w[N];
  bool big = false;
#pragma unroll
  for (int i = 0; i < N; ++i) { w[i] = v[i] * a[i]; big = big || fabsf(w[i]) > 8192.f; }
  if (__builtin_amdgcn_ballot_w64(big) != 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = fmaf(inv_b[i], sin_sq_call(w[i]), v[i]);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = fmaf(inv_b[i], sin_sq_small(w[i]), v[i]);
  }
}

// Seven consecutive float2 from LDS as seven ds_read_b64.  Left to itself the compiler fuses the 12 dwords the low-pass filter uses
// (p0.y .. p6.x) into six ds_read2_b32, whose two dword accesses each have an 8-byte lane stride: two-way bank conflicts on every one
// of them -- rocprofv3 showed the activation kernel's LDS array busy 88 % of the time, a third of it conflict cycles.  ds_read_b64 at
// an 8-byte lane stride is conflict-free and moves the same bytes in a quarter of the LDS cycles.
__device__ __forceinline__ void lds_read7_b64(const float2* p, float2 (&v)[7]) {
  const uint32_t a = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
  asm volatile(
      "ds_read_b64 %0, %7\n\tds_read_b64 %1, %7 offset:8\n\tds_read_b64 %2, %7 offset:16\n\tds_read_b64 %3, %7 offset:24\n\t"
      "ds_read_b64 %4, %7 offset:32\n\tds_read_b64 %5, %7 offset:40\n\tds_read_b64 %6, %7 offset:48\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6])
      : "v"(a)
      : "memory");
}

}  // namespace dmel
