// Small HBM/launch-bound kernels of the quantiser path (a6-a8 of SURVEY.md section 8): depthwise conv + LayerNorm,
// FSQ encode / decode, masked copy, mask + quality add.  Their total work is ~0.02 GFLOP per utterance-second.
#include "ops.h"

namespace dmel {

// ---- ConvNeXt front half: depthwise k7 conv (zero pad 3) then LayerNorm over channels (eps, biased var).
// firefly.py:386-388 (dwconv, permute, norm).  x, y: (N, C, T).
constexpr int kDwTile = 32;

__global__ __launch_bounds__(256) void dwconv_ln_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ dw_w, const float* __restrict__ dw_b,
                                                        const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                        float* __restrict__ h0 /*nullable: pre-norm output, kept for training*/,
                                                        int C, int64_t T, float eps) {
  extern __shared__ float sm[];
  float* hbuf = sm;                         // [C][kDwTile+1]
  float* stat = sm + (size_t)C * (kDwTile + 1);  // [2][kDwTile]
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * kDwTile;
  const float* xn = x + (int64_t)n * C * T;
  for (int idx = tid; idx < C * kDwTile; idx += 256) {
    const int c = idx / kDwTile, j = idx % kDwTile;
    const int64_t t = t0 + j;
    float acc = 0.f;
    if (t < T) {
      acc = dw_b[c];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int64_t s = t + k - 3;
        if (s >= 0 && s < T) acc = fmaf(dw_w[c * 7 + k], xn[(int64_t)c * T + s], acc);
      }
    }
    hbuf[c * (kDwTile + 1) + j] = acc;
  }
  __syncthreads();
  if (tid < kDwTile) {
    float mean = 0.f;
    for (int c = 0; c < C; ++c) mean += hbuf[c * (kDwTile + 1) + tid];
    mean /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; ++c) {
      const float d = hbuf[c * (kDwTile + 1) + tid] - mean;
      var = fmaf(d, d, var);
    }
    var /= (float)C;
    stat[tid] = mean;
    stat[kDwTile + tid] = 1.0f / sqrtf(var + eps);
  }
  __syncthreads();
  float* yn = y + (int64_t)n * C * T;
  for (int idx = tid; idx < C * kDwTile; idx += 256) {
    const int c = idx / kDwTile, j = idx % kDwTile;
    const int64_t t = t0 + j;
    if (t < T) {
      yn[(int64_t)c * T + t] = (hbuf[c * (kDwTile + 1) + j] - stat[j]) * stat[kDwTile + j] * ln_w[c] + ln_b[c];
      if (h0) h0[((int64_t)n * C + c) * T + t] = hbuf[c * (kDwTile + 1) + j];
    }
  }
}

int launch_dwconv_ln(const float* x, float* y, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                     int N, int C, int64_t T, hipStream_t s, float* h0) {
  DMEL_CHECK_ARG(N > 0 && N <= 65535 && C > 0 && T > 0, "dwconv_ln: bad shape");
  const size_t lds = ((size_t)C * (kDwTile + 1) + 2 * kDwTile) * sizeof(float);
  DMEL_CHECK_ARG(lds <= 64 * 1024, "dwconv_ln: %d channels exceed the LDS tile", C);
  dim3 grid((unsigned)((T + kDwTile - 1) / kDwTile), (unsigned)N);
  {
    ProfScope ps("small", s, 0.0, 8.0 * N * C * (double)T);
    hipLaunchKernelGGL(dwconv_ln_kernel, grid, dim3(256), lds, s, x, y, dw_w, dw_b, ln_w, ln_b, h0, C, T, 1e-6f);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

// ---- FSQ ---------------------------------------------------------------------------------------

__device__ __forceinline__ float fsq_bound(float z, const FsqConst& k, int j) {
  return tanhf(z + k.shift[j]) * k.half_l[j] - k.offset[j];
}

// z: (B*G, C, T4) channel-major rows of group g of item b at row b*G+g.  w_in: (G, D, C), b_in: (G, D).
// ids: (B, G, T4) int32.  prequant (optional): (G, B, T4, D).
__global__ __launch_bounds__(256) void fsq_encode_kernel(const float* __restrict__ z, const float* __restrict__ w_in,
                                                         const float* __restrict__ b_in, int32_t* __restrict__ ids,
                                                         float* __restrict__ prequant, FsqConst k, int B, int G, int C,
                                                         int64_t T4) {
  const int64_t total = (int64_t)B * G * T4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t l = i % T4;
  const int g = (int)((i / T4) % G);
  const int b = (int)(i / (T4 * G));
  const float* zr = z + ((int64_t)(b * G + g) * C) * T4 + l;
  float acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = 0.f;
  for (int c = 0; c < C; ++c) {
    const float v = zr[(int64_t)c * T4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < k.n_levels) acc[j] = fmaf(w_in[((int64_t)g * k.n_levels + j) * C + c], v, acc[j]);
  }
  if (k.strict) {
    // strict mode: the last Linear and the bound(s) in float64, so that the value that is rounded does not depend on a summation
    // order or on a tanhf implementation -- two implementations fed the same features produce the same ids (the CPU oracle's strict
    // restatement does exactly this).  70 x 3 double FMAs per token: free.
    double dacc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < C; ++c) {
      const double v = (double)zr[(int64_t)c * T4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < k.n_levels) dacc[j] = fma((double)w_in[((int64_t)g * k.n_levels + j) * C + c], v, dacc[j]);
    }
    int sid = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < k.n_levels) {
        double v = dacc[j] + (double)b_in[g * k.n_levels + j];
        const double hl = (double)k.half_l[j], of = (double)k.offset[j], sh = atanh(of / hl);   // shift in float64 too
        if (k.prebound) v = tanh(v + sh) * hl - of;
        v = tanh(v + sh) * hl - of;
        const float vf = (float)v;
        if (prequant) prequant[(((int64_t)g * B + b) * T4 + l) * k.n_levels + j] = vf;
        sid += ((int)rintf(vf) + k.half_width[j]) * k.basis[j];
      }
    }
    ids[i] = sid;
    return;
  }
  int id = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < k.n_levels) {
      float v = acc[j] + b_in[g * k.n_levels + j];
      if (k.prebound) v = fsq_bound(v, k, j);
      v = fsq_bound(v, k, j);
      if (prequant) prequant[(((int64_t)g * B + b) * T4 + l) * k.n_levels + j] = v;
      const int q = (int)rintf(v);  // round half to even, as torch.round
      id += (q + k.half_width[j]) * k.basis[j];
    }
  }
  ids[i] = id;
}

// ids (B, G, T4) -> z (B*G, C, T4):  code_j = (digit_j - hw_j) / hw_j ; z = W_out code + b_out.  w_out: (G, C, D).
__global__ __launch_bounds__(256) void fsq_decode_kernel(const int32_t* __restrict__ ids, const float* __restrict__ w_out,
                                                         const float* __restrict__ b_out, float* __restrict__ z, FsqConst k,
                                                         int B, int G, int C, int64_t T4) {
  const int64_t total = (int64_t)B * G * C * T4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t l = i % T4;
  const int c = (int)((i / T4) % C);
  const int64_t bg = i / (T4 * C);
  const int g = (int)(bg % G);
  const int id = ids[bg * T4 + l];
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < k.n_levels) {
      const int digit = (id / k.basis[j]) % k.levels[j];
      const float code = ((float)digit - (float)k.half_width[j]) / (float)k.half_width[j];
      acc = fmaf(w_out[((int64_t)g * C + c) * k.n_levels + j], code, acc);
    }
  }
  z[i] = acc + b_out[g * C + c];
}

// ---- FSQ backward (straight-through estimator) -----------------------------------------------------------------------------
// forward (vector_quantize_pytorch FSQ.forward, restated): z3 = W_in x + b_in; [zz = bound(z3)]; b2 = bound(zz);
// code = round_ste(b2) / half_width; out = W_out code + b_out.  round_ste passes the gradient through unchanged, so
//   d b2 = d code / half_width;  d zz = d b2 * half_l (1 - tanh^2(zz + shift));  d z3 = the same factor once more when pre-bounded.
// Point kernel: one thread per (b, g, l): d x (70 channel rows) plus the per-point d z3 and code for the parameter kernel.
__global__ __launch_bounds__(256) void fsq_bwd_point_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                            const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                            const float* __restrict__ w_out, float* __restrict__ dx,
                                                            float* __restrict__ dz3s, float* __restrict__ codes, FsqConst k, int B, int G,
                                                            int C, int64_t T4) {
  const int64_t total = (int64_t)B * G * T4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t l = i % T4;
  const int g = (int)((i / T4) % G);
  const int b = (int)(i / (T4 * G));
  const int64_t row0 = ((int64_t)(b * G + g) * C) * T4 + l;
  float z3[4], dcode[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { z3[j] = 0.f; dcode[j] = 0.f; }
  for (int c = 0; c < C; ++c) {
    const float v = x[row0 + (int64_t)c * T4], d = dout[row0 + (int64_t)c * T4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < k.n_levels) {
        z3[j] = fmaf(w_in[((int64_t)g * k.n_levels + j) * C + c], v, z3[j]);
        dcode[j] = fmaf(w_out[((int64_t)g * C + c) * k.n_levels + j], d, dcode[j]);
      }
  }
  float dz3[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dz3[j] = 0.f;
    if (j < k.n_levels) {
      const float z = z3[j] + b_in[g * k.n_levels + j];
      float zz = z, f0 = 1.f;
      if (k.prebound) {
        const float t0 = tanhf(z + k.shift[j]);
        zz = t0 * k.half_l[j] - k.offset[j];
        f0 = k.half_l[j] * (1.f - t0 * t0);
      }
      const float t1 = tanhf(zz + k.shift[j]);
      const float b2 = t1 * k.half_l[j] - k.offset[j];
      const float code = rintf(b2) / (float)k.half_width[j];
      dz3[j] = dcode[j] / (float)k.half_width[j] * k.half_l[j] * (1.f - t1 * t1) * f0;
      const int64_t o = (((int64_t)g * B + b) * T4 + l) * k.n_levels + j;
      dz3s[o] = dz3[j];
      codes[o] = code;
    }
  }
  for (int c = 0; c < C; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < k.n_levels) acc = fmaf(w_in[((int64_t)g * k.n_levels + j) * C + c], dz3[j], acc);
    dx[row0 + (int64_t)c * T4] = acc;
  }
}

// grid (C, G): d W_out[g][c][:], d b_out[g][c], d W_in[g][:][c] (and d b_in[g][:] from the c == 0 workgroup), reduced over (b, l)
__global__ __launch_bounds__(256) void fsq_bwd_param_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                            const float* __restrict__ dz3s, const float* __restrict__ codes,
                                                            float* __restrict__ dw_in, float* __restrict__ db_in, float* __restrict__ dw_out,
                                                            float* __restrict__ db_out, int D, int B, int G, int C, int64_t T4) {
  __shared__ float part[13][4];
  const int c = blockIdx.x, g = blockIdx.y;
  float acc[13];        // [0..3] dW_out, [4] db_out, [5..8] dW_in, [9..12] db_in
#pragma unroll
  for (int q = 0; q < 13; ++q) acc[q] = 0.f;
  const int64_t n_pts = (int64_t)B * T4;
  for (int64_t n = threadIdx.x; n < n_pts; n += 256) {
    const int b = (int)(n / T4);
    const int64_t l = n - (int64_t)b * T4;
    const int64_t e = ((int64_t)(b * G + g) * C + c) * T4 + l;
    const float d = dout[e], v = x[e];
    const int64_t o = (((int64_t)g * B + b) * T4 + l) * D;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < D) {
        acc[j] = fmaf(d, codes[o + j], acc[j]);
        acc[5 + j] = fmaf(dz3s[o + j], v, acc[5 + j]);
        acc[9 + j] += dz3s[o + j];
      }
    acc[4] += d;
  }
#pragma unroll
  for (int q = 0; q < 13; ++q) {
    float sacc = acc[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
    if ((threadIdx.x & 63) == 0) part[q][threadIdx.x >> 6] = sacc;
  }
  __syncthreads();
  if (threadIdx.x < 13) {
    const int q = threadIdx.x;
    const float v = part[q][0] + part[q][1] + part[q][2] + part[q][3];
    if (q < 4) { if (q < D) dw_out[((int64_t)g * C + c) * D + q] = v; }
    else if (q == 4) db_out[g * C + c] = v;
    else if (q < 9) { if (q - 5 < D) dw_in[((int64_t)g * D + (q - 5)) * C + c] = v; }
    else if (c == 0 && q - 9 < D) db_in[g * D + (q - 9)] = v;
  }
}

// dx (B*G, C, T4); parameter gradients in the packed (G, ...) layouts of the handle's FSQ buffers; scratch: 2 * G*B*T4*D floats
int launch_fsq_backward(const float* x, const float* dout, const float* w_in, const float* b_in, const float* w_out, float* dx,
                        float* dw_in, float* db_in, float* dw_out, float* db_out, float* scratch, const FsqConst& k, int B, int G, int C,
                        int64_t T4, hipStream_t s) {
  const int64_t total = (int64_t)B * G * T4;
  float* dz3s = scratch;
  float* codes = scratch + total * k.n_levels;
  hipLaunchKernelGGL(fsq_bwd_point_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, dout, w_in, b_in, w_out, dx, dz3s,
                     codes, k, B, G, C, T4);
  DMEL_HIP(hipGetLastError());
  hipLaunchKernelGGL(fsq_bwd_param_kernel, dim3((unsigned)C, (unsigned)G), dim3(256), 0, s, x, dout, dz3s, codes, dw_in, db_in, dw_out,
                     db_out, k.n_levels, B, G, C, T4);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

int make_fsq_const(FsqConst& k, const int* levels, int n, int prebound) {
  DMEL_CHECK_ARG(n >= 1 && n <= 4, "FSQ: 1..4 levels supported, got %d", n);
  k.n_levels = n;
  k.prebound = prebound;
  k.strict = 0;
  int basis = 1;
  for (int j = 0; j < 4; ++j) {
    if (j < n) {
      const int L = levels[j];
      DMEL_CHECK_ARG(L >= 2, "FSQ: level %d < 2", L);
      k.levels[j] = L;
      k.half_width[j] = L / 2;
      k.basis[j] = basis;
      basis *= L;
      k.half_l[j] = (float)(L - 1) * 1.001f / 2.0f;   // (levels - 1) * (1 + eps) / 2, eps = 1e-3, in fp32
      k.offset[j] = (L % 2 == 0) ? 0.5f : 0.0f;
      k.shift[j] = atanhf(k.offset[j] / k.half_l[j]);
    } else {
      k.levels[j] = 1; k.half_width[j] = 1; k.basis[j] = 0; k.half_l[j] = 0; k.offset[j] = 0; k.shift[j] = 0;
    }
  }
  return DMEL_OK;
}

int launch_fsq_encode(const float* z, const float* w_in, const float* b_in, int32_t* ids, float* prequant,
                      const FsqConst& k, int B, int G, int C, int64_t T4, hipStream_t s) {
  const int64_t total = (int64_t)B * G * T4;
  {
    ProfScope ps("small", s, 0.0, 4.0 * B * G * C * (double)T4);
    hipLaunchKernelGGL(fsq_encode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, w_in, b_in, ids,
                       prequant, k, B, G, C, T4);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

int launch_fsq_decode(const int32_t* ids, const float* w_out, const float* b_out, float* z, const FsqConst& k, int B,
                      int G, int C, int64_t T4, hipStream_t s) {
  const int64_t total = (int64_t)B * G * C * T4;
  {
    ProfScope ps("small", s, 0.0, 4.0 * (double)total);
    hipLaunchKernelGGL(fsq_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ids, w_out, b_out, z, k,
                       B, G, C, T4);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

// ---- y[n,c,t] = x[n,c,t] * (t < len[n / div])   (noise * mask of codec_lit_modules.py:473) ----------
__global__ __launch_bounds__(256) void masked_copy_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const int64_t* __restrict__ len, int div, int64_t CT, int64_t T,
                                                          int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float v = x[i];
  if (len) {
    const int64_t n = i / CT, t = i % T;
    if (t >= len[n / div]) v = 0.f;
  }
  y[i] = v;
}

int launch_masked_copy(const float* x, float* y, const int64_t* len, int div, int N, int C, int64_t T, hipStream_t s) {
  const int64_t total = (int64_t)N * C * T;
  {
    ProfScope ps("small", s, 0.0, 8.0 * (double)total);
    hipLaunchKernelGGL(masked_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, len,
                       div > 0 ? div : 1, (int64_t)C * T, T, total);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

// ---- folded batch: (N, C, T) <-> (C, N * P) with zero gaps (ops.h) -------------------------------------------------
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ x, float* __restrict__ xf, const int64_t* __restrict__ len,
                                                   int div, int N, int C, int T, int P, int64_t pitch) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;      // column of the folded row
  const int c = blockIdx.y;
  if (j >= pitch) return;
  const int n = (int)(j / P), t = (int)(j - (int64_t)n * P);
  float v = 0.f;
  if (n < N && t < T && (!len || t < len[n / div])) v = x[((int64_t)n * C + c) * T + t];
  xf[(int64_t)c * pitch + j] = v;
}
__global__ __launch_bounds__(256) void unfold_kernel(const float* __restrict__ xf, float* __restrict__ y, const int64_t* __restrict__ len,
                                                     int div, int C, int T, int P, int64_t pitch, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int t = (int)(i % T);
  const int64_t nc = i / T;
  const int c = (int)(nc % C), n = (int)(nc / C);
  y[i] = (!len || t < len[n / div]) ? xf[(int64_t)c * pitch + (int64_t)n * P + t] : 0.f;
}

int launch_fold(const float* x, float* xf, const int64_t* len, int div, int N, int C, int64_t T, int P, int64_t pitch, hipStream_t s) {
  {
    ProfScope ps("small", s, 0.0, 4.0 * (double)N * C * T + 4.0 * (double)C * pitch);
    hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((pitch + 255) / 256), (unsigned)C), dim3(256), 0, s, x, xf, len, div > 0 ? div : 1, N, C,
                       (int)T, P, pitch);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}
int launch_unfold(const float* xf, float* y, const int64_t* len, int div, int N, int C, int64_t T, int P, int64_t pitch, hipStream_t s) {
  const int64_t total = (int64_t)N * C * T;
  {
    ProfScope ps("small", s, 0.0, 8.0 * (double)total);
    hipLaunchKernelGGL(unfold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xf, y, len, div > 0 ? div : 1, C, (int)T, P,
                       pitch, total);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

// ---- polyphase sinc resampling (torchaudio.functional.resample, the call of utils/spectrogram.py:122-123) ------------------------
// y[b, n * up + p] = sum_k bank[p][k] * xpad[b, n * down + k],  xpad = x padded with `width` zeros on the left and width + down on the
// right, k < kw = 2 * width + down.  One thread per output sample; the (up x kw) filter bank sits in LDS when it fits (16 kHz -> 24 kHz:
// 3 x 16 taps; 44.1 -> 24 kHz: 80 x 171), the input window comes through L1 (neighbouring outputs share all but `down` samples).
// HBM-bound: 4 bytes read per input + 4 written per output sample.
constexpr int kResampleLdsFloats = 15 * 1024;
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ bank,
                                                       int64_t L, int64_t Lout, int down, int up, int width, int kw, int bank_in_lds) {
  extern __shared__ float bsm[];
  if (bank_in_lds) {
    for (int i = threadIdx.x; i < up * kw; i += 256) bsm[i] = bank[i];
    __syncthreads();
  }
  const float* bk = bank_in_lds ? bsm : bank;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= Lout) return;
  const int b = blockIdx.y;
  const int64_t n = o / up;
  const int p = (int)(o - n * up);
  const float* xb = x + (int64_t)b * L;
  const float* w = bk + (int64_t)p * kw;
  const int64_t s0 = n * down - width;
  float acc = 0.f;
  for (int k = 0; k < kw; ++k) {
    const int64_t s = s0 + k;
    const float v = (s >= 0 && s < L) ? xb[s] : 0.f;
    acc = fmaf(w[k], v, acc);
  }
  y[(int64_t)b * Lout + o] = acc;
}

int launch_resample(const float* x, float* y, const float* bank_dev, int B, int64_t L, int64_t Lout, int down, int up, int width,
                    hipStream_t s) {
  const int kw = 2 * width + down;
  const int in_lds = up * kw <= kResampleLdsFloats;
  {
    ProfScope ps("small", s, 0.0, 4.0 * (double)B * ((double)L + (double)Lout));
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((Lout + 255) / 256), (unsigned)B), dim3(256),
                       in_lds ? (size_t)up * kw * sizeof(float) : 0, s, x, y, bank_dev, L, Lout, down, up, width, kw, in_lds);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

// ---- conv_post: C -> 1 channel, k taps, zero "same" padding, then tanh | clamp   (bigvgan.py:386-391) -----------
// One output row: a 32-row MFMA tile would be 97 % zeros, so this is a plain reduction over (channel, tap) -- HBM
// bound (C*T*4 bytes read per item, each input row read once per workgroup through L1).
constexpr int kPostTile = 1024;
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ w, float bias, int C, int K, int T,
                                                        int act) {
  extern __shared__ float wsm[];   // [C][K]
  for (int i = threadIdx.x; i < C * K; i += 256) wsm[i] = w[i];
  __syncthreads();
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * kPostTile;
  const float* xb = x + (int64_t)b * C * T;
  const int pad = (K - 1) / 2;
  float acc[4];
  int tt[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { acc[e] = bias; tt[e] = t0 + threadIdx.x + 256 * e; }
  for (int c = 0; c < C; ++c) {
    const float* xr = xb + (int64_t)c * T;
    for (int k = 0; k < K; ++k) {
      const float wv = wsm[c * K + k];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = tt[e] + k - pad;
        const float v = (s >= 0 && s < T) ? xr[s] : 0.f;
        acc[e] = fmaf(wv, v, acc[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (tt[e] < T) y[(int64_t)b * T + tt[e]] = act == 2 ? tanhf(acc[e]) : (act == 3 ? fminf(fmaxf(acc[e], -1.f), 1.f) : acc[e]);
}

int launch_conv_post(const float* x, float* y, const float* w_dev, float bias, int act, int B, int C, int K, int64_t T,
                     hipStream_t s) {
  DMEL_CHECK_ARG(B > 0 && B <= 65535 && C > 0 && K > 0 && (K % 2) == 1 && T > 0 && T < ((int64_t)1 << 30), "conv_post: bad shape");
  const size_t lds = (size_t)C * K * sizeof(float);
  DMEL_CHECK_ARG(lds <= 48 * 1024, "conv_post: weight table too large");
  dim3 grid((unsigned)((T + kPostTile - 1) / kPostTile), (unsigned)B);
  {
    ProfScope ps("small", s, 0.0, 4.0 * B * (double)T * (C + 1));
    hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), lds, s, x, y, w_dev, bias, C, K, (int)T, act);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

// ---- z = z * mask + (w * value + bias)      codec_lit_modules.py:520-526 --------------------------
__global__ __launch_bounds__(256) void mask_add_quality_kernel(float* __restrict__ z, const int64_t* __restrict__ len,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               float value, int C, int64_t T, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t t = i % T;
  const int c = (int)((i / T) % C);
  const int64_t b = i / (T * C);
  float v = z[i];
  if (len && t >= len[b]) v = 0.f;
  z[i] = v + (w[c] * value + bias[c]);
}

// ---- data front end: peak-normalise + right-pad collate      dataset/lhotse_tts_dataset.py:29-32 (librosa.util.normalize(audio) * 0.95),
// :46-65 (right pad to the longest clip, stack to (B, 1, L), lengths (1, B) int32) ---------------------------------------------------------
// Clips stay where the decoder / resampler left them (B separate device buffers): pass 1 reduces max |x| per clip (slices of a clip on the
// grid, one atomic max on the bit pattern of a non-negative float per workgroup), pass 2 writes audios[b, 0, t] = x[t] / peak * 0.95 for
// t < len and 0 behind it, plus the lengths row.  librosa leaves a clip whose peak is below the smallest normal float unscaled.
constexpr int kCollateTile = 4096;
__global__ __launch_bounds__(256) void collate_absmax_kernel(const float* const* __restrict__ clips, const int64_t* __restrict__ lens,
                                                              const int32_t* __restrict__ order, uint32_t* __restrict__ peaks) {
  const int b = blockIdx.y;
  const int src = order ? order[b] : b;
  const int64_t n = lens[src];
  const int64_t t0 = (int64_t)blockIdx.x * kCollateTile;
  if (t0 >= n) return;
  const float* x = clips[src];
  float m = 0.f;
  for (int64_t t = t0 + threadIdx.x; t < min(n, t0 + kCollateTile); t += 256) m = fmaxf(m, fabsf(x[t]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(peaks + b, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
__global__ __launch_bounds__(256) void collate_scale_pad_kernel(const float* const* __restrict__ clips, const int64_t* __restrict__ lens,
                                                                 const int32_t* __restrict__ order, const uint32_t* __restrict__ peaks,
                                                                 float* __restrict__ audios, int32_t* __restrict__ lengths, int64_t Lmax,
                                                                 float peak) {
  const int b = blockIdx.y;
  const int src = order ? order[b] : b;
  const int64_t n = min(lens[src], Lmax);
  if (blockIdx.x == 0 && threadIdx.x == 0) lengths[b] = (int32_t)n;
  const float* x = clips[src];
  const float m = __uint_as_float(peaks[b]);
  const float length = m < 1.17549435e-38f ? 1.0f : m;      // librosa.util.normalize: norms below `tiny` are replaced by 1 (fill = None)
  float* y = audios + (int64_t)b * Lmax;
  const int64_t t0 = (int64_t)blockIdx.x * kCollateTile;
  for (int64_t t = t0 + threadIdx.x; t < min(Lmax, t0 + kCollateTile); t += 256) y[t] = t < n ? (x[t] / length) * peak : 0.f;
}

}  // namespace dmel

extern "C" int dmel_collate_peak_f32(const float* const* clips_dev, const int64_t* lengths_dev, const int32_t* order_dev, float* audios,
                                     int32_t* audio_lengths, uint32_t* peaks_scratch, int B, int64_t Lmax, float peak, void* stream) {
  using namespace dmel;
  DMEL_CHECK_ARG(clips_dev && lengths_dev && audios && audio_lengths && peaks_scratch, "collate_peak: NULL argument");
  DMEL_CHECK_ARG(B > 0 && B <= 65535 && Lmax > 0 && Lmax < ((int64_t)1 << 40), "collate_peak: bad shape");
  hipStream_t s = (hipStream_t)stream;
  DMEL_HIP(hipMemsetAsync(peaks_scratch, 0, (size_t)B * sizeof(uint32_t), s));
  const dim3 grid((unsigned)((Lmax + kCollateTile - 1) / kCollateTile), (unsigned)B);
  {
    ProfScope ps("small", s, 0.0, 12.0 * B * (double)Lmax);
    hipLaunchKernelGGL(collate_absmax_kernel, grid, dim3(256), 0, s, clips_dev, lengths_dev, order_dev, peaks_scratch);
    hipLaunchKernelGGL(collate_scale_pad_kernel, grid, dim3(256), 0, s, clips_dev, lengths_dev, order_dev, peaks_scratch, audios,
                       audio_lengths, Lmax, peak);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

namespace dmel {
}  // namespace dmel

extern "C" int dmel_mask_add_quality_f32(float* z, const int64_t* lengths, const float* w, const float* bias, float value,
                                         int B, int C, int64_t T, void* stream) {
  using namespace dmel;
  DMEL_CHECK_ARG(z && w && bias && B > 0 && C > 0 && T > 0, "mask_add_quality: bad argument");
  const int64_t total = (int64_t)B * C * T;
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope ps("small", s, 0.0, 8.0 * (double)total);
    hipLaunchKernelGGL(mask_add_quality_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, lengths, w,
                       bias, value, C, T, total);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

extern "C" int dmel_resample_f32(const float* x, float* y, const float* filter_bank_dev, int B, int64_t L, int64_t Lout, int orig_freq,
                                 int new_freq, int width, void* stream) {
  DMEL_CHECK_ARG(x && y && filter_bank_dev, "resample: NULL argument");
  DMEL_CHECK_ARG(B > 0 && B <= 65535 && L > 0 && Lout > 0 && orig_freq > 0 && new_freq > 0 && width >= 0, "resample: bad shape");
  DMEL_CHECK_ARG(Lout <= (L * new_freq + orig_freq - 1) / orig_freq, "resample: Lout exceeds ceil(new_freq * L / orig_freq)");
  return dmel::launch_resample(x, y, filter_bank_dev, B, L, Lout, orig_freq, new_freq, width, (hipStream_t)stream);
}
