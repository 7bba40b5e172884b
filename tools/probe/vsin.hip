// Probe: accuracy of the hardware v_sin_f32 (input in revolutions) against fp64, for the snake's sin^2 term.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    float r = x[i] * 0.15915494309189535f;   // revolutions
    r = r - rintf(r);                         // v_sin_f32 domain is limited: reduce to [-0.5, 0.5]
    y[i] = __builtin_amdgcn_sinf(r);
  }
}
int main() {
  const int n = 1 << 22;
  std::vector<float> h(n), o(n);
  for (int i = 0; i < n; ++i) h[i] = -40.f + 80.f * (float)i / n;
  float *dx, *dy;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
  hipMemcpy(o.data(), dy, n * 4, hipMemcpyDeviceToHost);
  double me = 0, me2 = 0, rms = 0;
  for (int i = 0; i < n; ++i) {
    double s = std::sin((double)h[i]);
    double e = std::fabs(o[i] - s), e2 = std::fabs((double)o[i] * o[i] - s * s);
    me = std::fmax(me, e); me2 = std::fmax(me2, e2); rms += e2 * e2;
  }
  printf("v_sin_f32: max |sin err| %.3e  max |sin^2 err| %.3e  rms sin^2 err %.3e\n", me, me2, std::sqrt(rms / n));
  return 0;
}
