// Elementwise pieces of the WaveNet training path (forward with saved activations, and backward).  All tensors are
// contiguous (N, C, T) fp32; every kernel is a grid-stride pass bound by its HBM traffic.
// What autograd differentiates in the reference: wavenet.py:116-135 (ResidualBlock.forward), :204-225 (WaveNet.forward).
#include "ops.h"

namespace dmel {

namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// z = sigmoid(pre[:, :C]) * tanh(pre[:, C:])                                        wavenet.py:128-129
__global__ void gate_fwd_kernel(const float* __restrict__ pre, float* __restrict__ z, int64_t CT, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / CT, r = i - n * CT;
    const float a = pre[n * 2 * CT + r], b = pre[n * 2 * CT + CT + r];
    z[i] = sigmoidf_(a) * tanhf(b);
  }
}

// dpre[:, :C] = dz * tanh(b) * s (1 - s),  dpre[:, C:] = dz * s * (1 - tanh(b)^2),  s = sigmoid(a)
__global__ void gate_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ pre, float* __restrict__ dpre, int64_t CT,
                                int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / CT, r = i - n * CT;
    const float a = pre[n * 2 * CT + r], b = pre[n * 2 * CT + CT + r];
    const float s = sigmoidf_(a), t = tanhf(b), g = dz[i];
    dpre[n * 2 * CT + r] = g * t * s * (1.f - s);
    dpre[n * 2 * CT + CT + r] = g * s * (1.f - t * t);
  }
}

// x_next = (x + o[:, :C]) / sqrt(2);  skipsum (+)= o[:, C:]                             wavenet.py:131-135, :218
__global__ void resskip_fwd_kernel(const float* __restrict__ x, const float* __restrict__ o, float* __restrict__ xn,
                                   float* __restrict__ skipsum, int first, int64_t CT, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / CT, r = i - n * CT;
    xn[i] = (x[i] + o[n * 2 * CT + r]) / 1.41421356237309504880f;
    const float s = o[n * 2 * CT + CT + r];
    skipsum[i] = first ? s : skipsum[i] + s;
  }
}

// go = concat(gx / sqrt(2), gs) (gradient of the block's 1x1 output), gxs = gx / sqrt(2) (the identity path)
__global__ void resskip_bwd_kernel(const float* __restrict__ gx, const float* __restrict__ gs, float* __restrict__ go,
                                   float* __restrict__ gxs, int64_t CT, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / CT, r = i - n * CT;
    const float v = gx[i] / 1.41421356237309504880f;
    gxs[i] = v;
    go[n * 2 * CT + r] = v;
    go[n * 2 * CT + CT + r] = gs[i];
  }
}

__global__ void silu_fwd_kernel(const float* __restrict__ u, float* __restrict__ y, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = u[i];
    y[i] = v * sigmoidf_(v);
  }
}

// d silu(u) / du = s (1 + u (1 - s))
__global__ void silu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u, float* __restrict__ du, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = u[i], s = sigmoidf_(v);
    du[i] = g[i] * s * (1.f + v * (1.f - s));
  }
}

__global__ void scale_kernel(const float* __restrict__ x, float* __restrict__ y, float k, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) y[i] = x[i] * k;
}

unsigned blocks_for(int64_t total) { return (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32); }

}  // namespace

#define DMEL_EW_LAUNCH(kernel, total, bytes_per_elem, ...)                                  \
  do {                                                                                      \
    ProfScope ps("train_elementwise", s, 0.0, (double)(bytes_per_elem) * (double)(total));  \
    hipLaunchKernelGGL(kernel, dim3(blocks_for(total)), dim3(256), 0, s, __VA_ARGS__);      \
  } while (0);                                                                              \
  DMEL_HIP(hipGetLastError());                                                              \
  return DMEL_OK

int launch_gate_fwd(const float* pre, float* z, int N, int C, int64_t T, hipStream_t s) {
  const int64_t CT = (int64_t)C * T, total = N * CT;
  DMEL_EW_LAUNCH(gate_fwd_kernel, total, 12, pre, z, CT, total);
}
int launch_gate_bwd(const float* dz, const float* pre, float* dpre, int N, int C, int64_t T, hipStream_t s) {
  const int64_t CT = (int64_t)C * T, total = N * CT;
  DMEL_EW_LAUNCH(gate_bwd_kernel, total, 20, dz, pre, dpre, CT, total);
}
int launch_resskip_fwd(const float* x, const float* o, float* xn, float* skipsum, int first, int N, int C, int64_t T, hipStream_t s) {
  const int64_t CT = (int64_t)C * T, total = N * CT;
  DMEL_EW_LAUNCH(resskip_fwd_kernel, total, 24, x, o, xn, skipsum, first, CT, total);
}
int launch_resskip_bwd(const float* gx, const float* gs, float* go, float* gxs, int N, int C, int64_t T, hipStream_t s) {
  const int64_t CT = (int64_t)C * T, total = N * CT;
  DMEL_EW_LAUNCH(resskip_bwd_kernel, total, 20, gx, gs, go, gxs, CT, total);
}
int launch_silu_fwd(const float* u, float* y, int64_t total, hipStream_t s) { DMEL_EW_LAUNCH(silu_fwd_kernel, total, 8, u, y, total); }
int launch_silu_bwd(const float* g, const float* u, float* du, int64_t total, hipStream_t s) {
  DMEL_EW_LAUNCH(silu_bwd_kernel, total, 12, g, u, du, total);
}
int launch_scale(const float* x, float* y, float k, int64_t total, hipStream_t s) { DMEL_EW_LAUNCH(scale_kernel, total, 8, x, y, k, total); }

}  // namespace dmel
