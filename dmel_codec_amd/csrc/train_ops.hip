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

// ---- device-side weight re-pack: one thread per (packed row m, K step, k in 0..15) element ------------------------------
struct RepackArgs {
  RepackSeg seg[2];
  const float* b0;
  const float* b1;
  float* w;
  uint16_t* w16;
  uint16_t* w48;
  float* bias;
  int mode, C, RP, phases, nseg, Mpad, steps;
  int Cin[2], taps[2], nchunk[2];
};

__device__ __forceinline__ int repack_src_row(const RepackArgs& a, int m) {      // conv.h: pack_conv::src_row
  if (a.mode != EPI_LINEAR) {
    const int q32 = m >> 5, rho = m & 31, grp = rho >> 3, i = rho & 7;
    const int c = q32 * 16 + (grp >> 1) * 8 + i;
    return c >= a.C ? -1 : (grp & 1) * a.C + c;
  }
  if (m >= a.RP * a.phases) return -1;
  const int ph = m / a.RP, co = m % a.RP;
  return co >= a.C ? -1 : ph * a.C + co;
}

__global__ void repack_kernel(RepackArgs a) {
  const int64_t total = (int64_t)a.Mpad * a.steps * kCK;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e % kCK);
    const int step = (int)((e / kCK) % a.steps);
    const int m = (int)(e / ((int64_t)kCK * a.steps));
    int sg = 0, ls = step;
    if (a.nseg > 1 && ls >= a.nchunk[0] * a.taps[0]) { ls -= a.nchunk[0] * a.taps[0]; sg = 1; }
    const int ch = ls / a.taps[sg], tp = ls - ch * a.taps[sg];
    const int ci = ch * kCK + k;
    const int sr = repack_src_row(a, m);
    float v = 0.f;
    if (sr >= 0 && ci < a.Cin[sg]) {
      const RepackSeg& s = a.seg[sg];
      const int t = s.rev ? a.taps[sg] - 1 - tp : tp;
      v = s.w[sr * s.rs + ci * s.cs + t * s.ts];
    }
    const int tile = m >> 5, r = m & 31;
    {  // fp32 image: [tile][step][half][lane][4], lane = 32*(k&1) + r
      const int h = k & 1, kk = k >> 1, hf = kk >> 2, j = kk & 3, lane = 32 * h + r;
      a.w[((((int64_t)tile * a.steps + step) * 2 + hf) * 64 + lane) * 4 + j] = v;
    }
    {  // bf16 images: [tile][step][(piece)][lane][8], lane = 32*(k>>3) + r
      const int h = k >> 3, j = k & 7, lane = 32 * h + r;
      const uint32_t u = __float_as_uint(v);
      uint16_t rne;
      if ((u & 0x7fffffffu) > 0x7f800000u) rne = (uint16_t)((u >> 16) | 0x40);
      else rne = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
      a.w16[(((int64_t)tile * a.steps + step) * 64 + lane) * 8 + j] = rne;
      const float p1 = __uint_as_float(u & 0xffff0000u), r1 = v - p1;
      const float p2 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u), r2 = r1 - p2;
      const int64_t base = ((int64_t)tile * a.steps + step) * 3;
      a.w48[((base + 0) * 64 + lane) * 8 + j] = (uint16_t)(__float_as_uint(p1) >> 16);
      a.w48[((base + 1) * 64 + lane) * 8 + j] = (uint16_t)(__float_as_uint(p2) >> 16);
      a.w48[((base + 2) * 64 + lane) * 8 + j] = (uint16_t)(__float_as_uint(r2) >> 16);
    }
    if (step == 0 && k == 0) {
      float b = 0.f;
      if (sr >= 0) {
        if (a.b0) b = a.b0[sr];
        if (a.b1) b += a.b1[sr];
      }
      a.bias[m] = b;
    }
  }
}

int launch_repack(PackedConv& pc, const RepackSrc& src, hipStream_t s) {
  const PackDesc& d = pc.d;
  DMEL_CHECK_ARG(pc.w.p && pc.w16.p && pc.w48.p && pc.bias.p, "repack: the convolution was never packed");
  DMEL_CHECK_ARG(d.nseg >= 1 && d.nseg <= 2 && src.seg[0].w && (d.nseg == 1 || src.seg[1].w), "repack: missing source tensor");
  RepackArgs a{};
  a.seg[0] = src.seg[0]; a.seg[1] = src.seg[1];
  a.b0 = src.b0; a.b1 = src.b1;
  a.w = pc.w.as<float>(); a.w16 = pc.w16.as<uint16_t>(); a.w48 = pc.w48.as<uint16_t>(); a.bias = pc.bias.as<float>();
  a.mode = d.mode; a.C = d.C; a.RP = pc.RP; a.phases = d.phases; a.nseg = d.nseg; a.Mpad = pc.Mpad; a.steps = pc.steps;
  for (int i = 0; i < d.nseg; ++i) {
    a.Cin[i] = d.seg[i].Cin; a.taps[i] = d.seg[i].taps; a.nchunk[i] = (d.seg[i].Cin + kCK - 1) / kCK;
  }
  if (d.nseg == 1) { a.Cin[1] = 0; a.taps[1] = 1; a.nchunk[1] = 0; }
  const int64_t total = (int64_t)pc.Mpad * pc.steps * kCK;
  hipLaunchKernelGGL(repack_kernel, dim3(blocks_for(total)), dim3(256), 0, s, a);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

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
