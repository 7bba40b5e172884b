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

}  // namespace dmel
