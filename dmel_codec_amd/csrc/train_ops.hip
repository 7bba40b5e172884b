// Elementwise pieces of the WaveNet training path (forward with saved activations, and backward).  All tensors are
// contiguous (N, C, T) fp32; every kernel is a grid-stride pass bound by its HBM traffic.
// What autograd differentiates in the reference: wavenet.py:116-135 (ResidualBlock.forward), :204-225 (WaveNet.forward).
#include "ops.h"

#include <cstring>
#include <memory>
#include <vector>

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

__global__ void silu_fwd_kernel(const float* u, float* y, int64_t total) {   // y may alias u
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = u[i];
    y[i] = v * sigmoidf_(v);
  }
}

// d silu(u) / du = s (1 + u (1 - s))
__global__ void silu_bwd_kernel(const float* g, const float* __restrict__ u, float* du, int64_t total) {   // du may alias g
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = u[i], s = sigmoidf_(v);
    du[i] = g[i] * s * (1.f + v * (1.f - s));
  }
}

__global__ void scale_kernel(const float* __restrict__ x, float* __restrict__ y, float k, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) y[i] = x[i] * k;
}

// ---- ConvNeXt block pieces ------------------------------------------------------------------------------------------
__global__ void gelu_fwd_kernel(const float* __restrict__ u, float* __restrict__ g, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = u[i];
    g[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  }
}
// d gelu(u) / du = Phi(u) + u * phi(u)
__global__ void gelu_bwd_kernel(const float* dg, const float* __restrict__ u, float* du, int64_t total) {   // du may alias dg
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = u[i];
    const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
    du[i] = dg[i] * (cdf + v * pdf);
  }
}
__global__ void layerscale_res_fwd_kernel(const float* __restrict__ x, const float* __restrict__ v, const float* __restrict__ gamma,
                                          float* __restrict__ y, int C, int64_t T, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / T) % C);
    y[i] = x[i] + gamma[c] * v[i];
  }
}
// one workgroup per channel: dv = dy * gamma[c]; dgamma[c] = sum dy * v
__global__ __launch_bounds__(256) void layerscale_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ v,
                                                             const float* __restrict__ gamma, float* __restrict__ dv,
                                                             float* __restrict__ dgamma, int N, int C, int T) {
  __shared__ float part[4];
  const int c = blockIdx.x;
  const float g = gamma[c];
  float s = 0.f;
  const int64_t total = (int64_t)N * T;                 // flattened (item, frame) index: rows are 23-46 frames long
  for (int64_t i = threadIdx.x; i < total; i += 256) {
    const int n = (int)(i / T);
    const int64_t e = ((int64_t)n * C + c) * T + (i - (int64_t)n * T);
    const float d = dy[e];
    s = fmaf(d, v[e], s);
    dv[e] = d * g;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dgamma[c] = part[0] + part[1] + part[2] + part[3];
}

constexpr int kLnTile = 32;
// grid (ceil(T/32), N): a [C][32] tile of h0 and dh1 in LDS; statistics per column by one lane, then the per-element update;
// per-channel parameter sums over the tile's columns go out as one atomic per channel and workgroup.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dh1, const float* __restrict__ h0,
                                                     const float* __restrict__ ln_w, float* __restrict__ dh0, float* __restrict__ dln_w,
                                                     float* __restrict__ dln_b, int C, int64_t T, float eps) {
  extern __shared__ float lsm[];
  float* hb = lsm;                                  // [C][33]  h0, then xhat
  float* gb = lsm + (size_t)C * (kLnTile + 1);      // [C][33]  dh1
  float* st = gb + (size_t)C * (kLnTile + 1);       // [4][32]  mean, rstd, s1, s2
  const int tid = threadIdx.x, n = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * kLnTile;
  const int64_t base = (int64_t)n * C * T;
  for (int idx = tid; idx < C * kLnTile; idx += 256) {
    const int c = idx / kLnTile, j = idx % kLnTile;
    const int64_t t = t0 + j;
    const bool ok = t < T;
    hb[c * (kLnTile + 1) + j] = ok ? h0[base + (int64_t)c * T + t] : 0.f;
    gb[c * (kLnTile + 1) + j] = ok ? dh1[base + (int64_t)c * T + t] : 0.f;
  }
  __syncthreads();
  if (tid < kLnTile) {
    float mean = 0.f;
    for (int c = 0; c < C; ++c) mean += hb[c * (kLnTile + 1) + tid];
    mean /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; ++c) {
      const float d = hb[c * (kLnTile + 1) + tid] - mean;
      var = fmaf(d, d, var);
    }
    var /= (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float xh = (hb[c * (kLnTile + 1) + tid] - mean) * rstd;
      const float dxh = gb[c * (kLnTile + 1) + tid] * ln_w[c];
      s1 += dxh;
      s2 = fmaf(dxh, xh, s2);
    }
    st[tid] = mean; st[kLnTile + tid] = rstd; st[2 * kLnTile + tid] = s1 / (float)C; st[3 * kLnTile + tid] = s2 / (float)C;
  }
  __syncthreads();
  for (int idx = tid; idx < C * kLnTile; idx += 256) {
    const int c = idx / kLnTile, j = idx % kLnTile;
    const int64_t t = t0 + j;
    const float xh = (hb[c * (kLnTile + 1) + j] - st[j]) * st[kLnTile + j];
    hb[c * (kLnTile + 1) + j] = xh;                  // keep xhat for the parameter sums
    if (t < T) dh0[base + (int64_t)c * T + t] = st[kLnTile + j] * (gb[c * (kLnTile + 1) + j] * ln_w[c] - st[2 * kLnTile + j] - xh * st[3 * kLnTile + j]);
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float sw = 0.f, sb = 0.f;
    for (int j = 0; j < kLnTile; ++j) {              // out-of-range columns hold dh1 = 0
      const float g = gb[c * (kLnTile + 1) + j];
      sw = fmaf(g, hb[c * (kLnTile + 1) + j], sw);
      sb += g;
    }
    atomicAdd(&dln_w[c], sw);
    atomicAdd(&dln_b[c], sb);
  }
}

// one workgroup per channel: dx[c, t] = dres + sum_k w[c, k] dh0[c, t - k + 3];  ddw[c, k] = sum dh0[c, t] x[c, t + k - 3];  ddb = sum dh0
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const float* __restrict__ dh0, const float* __restrict__ x,
                                                         const float* __restrict__ dw_w, const float* __restrict__ dres,
                                                         float* __restrict__ dx, float* __restrict__ ddw, float* __restrict__ ddb, int N,
                                                         int C, int T) {
  __shared__ float part[8][4];
  const int c = blockIdx.x;
  float w[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) w[k] = dw_w[c * 7 + k];
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  // threads walk the flattened (item, frame) index: the quantiser's rows are 23-46 frames long, one item per pass would leave 80 % of the
  // workgroup idle (4 launches x 354 us per training step before; per-thread partial sums, so the summation order differs from a loop over items)
  const int64_t total = (int64_t)N * T;
  for (int64_t i = threadIdx.x; i < total; i += 256) {
    const int n = (int)(i / T), t = (int)(i - (int64_t)n * T);
    const int64_t base = ((int64_t)n * C + c) * T;
    const float g = dh0[base + t];
    float d = dres ? dres[base + t] : 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int td = t - k + 3;                       // y[td] used x[td + k - 3] = x[t]
      if (td >= 0 && td < T) d = fmaf(w[k], dh0[base + td], d);
      const int tx = t + k - 3;
      if (tx >= 0 && tx < T) acc[k] = fmaf(g, x[base + tx], acc[k]);
    }
    acc[7] += g;
    dx[base + t] = d;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float s = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const float s = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
    if (threadIdx.x < 7) ddw[c * 7 + threadIdx.x] = s;
    else ddb[c] = s;
  }
}

unsigned blocks_for(int64_t total) { return (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32); }

}  // namespace

// ---- device-side weight re-pack: one thread per (packed row m, K step, k in 0..15) element ------------------------------
struct RepackArgs {
  RepackSeg seg[2];
  const float* b0;
  const float* b1;
  int bias_mod;
  float* w;
  uint16_t* w16;
  uint16_t* w48;
  uint16_t* w32h;
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

// One thread per (packed row m, K step, half h): the eight weights k = 8 h .. 8 h + 7 of that row and step, i.e. one 16-byte unit of each
// 16-bit image (lane 32 h + r of the A fragment) and two of the fp32 image (lanes r and 32 + r, half-step h).  A wave covers the 64 lanes
// of one (tile, step): 1 KiB contiguous per image and piece.  (Round 2 had one thread per weight writing 2-byte units: 0.6 TB/s.)
// Jobs of one refresh are batched into ONE launch: the table lives in device memory, `first` holds the first block of every job.
struct RepackJob {
  RepackArgs a;
  int first_block;      // blocks [first_block, next job's first_block) work on this job, 256 units each
};

__device__ __forceinline__ void repack_unit(const RepackArgs& a, int64_t u) {
  const int r = (int)(u & 31), h = (int)((u >> 5) & 1);
  const int64_t q = u >> 6;
  const int step = (int)(q % a.steps), tile = (int)(q / a.steps);
  const int m = tile * 32 + r;
  int sg = 0, ls = step;
  if (a.nseg > 1 && ls >= a.nchunk[0] * a.taps[0]) { ls -= a.nchunk[0] * a.taps[0]; sg = 1; }
  const int ch = ls / a.taps[sg], tp = ls - ch * a.taps[sg];
  const int sr = repack_src_row(a, m);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (sr >= 0) {
    const RepackSeg& s = a.seg[sg];
    const int t = s.rev ? a.taps[sg] - 1 - tp : tp;
    const int64_t ro = (s.pC > 0 ? (int64_t)(sr % s.pC) * s.rs + (int64_t)(sr / s.pC) * s.ps : (int64_t)sr * s.rs) + t * s.ts;
    const int ci0 = ch * kCK + 8 * h;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (ci0 + j < a.Cin[sg]) v[j] = s.w[ro + (int64_t)(ci0 + j) * s.cs];
  }
  const int64_t ts = (int64_t)tile * a.steps + step;
  {  // fp32 image: [tile][step][half][lane][4]; weight k sits in lane 32 (k & 1) + r, half (k >> 3), word (k >> 1) & 3
    float4* w4 = reinterpret_cast<float4*>(a.w) + (ts * 2 + h) * 64;
    w4[r] = make_float4(v[0], v[2], v[4], v[6]);
    w4[32 + r] = make_float4(v[1], v[3], v[5], v[7]);
  }
  // 16-bit images: [tile][step][(piece)][lane][8], lane = 32 h + r
  const int lane = 32 * h + r;
  uint16_t b16[8], p1[8], p2[8], p3[8], hi[8], lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t u32 = __float_as_uint(v[j]);
    if ((u32 & 0x7fffffffu) > 0x7f800000u) b16[j] = (uint16_t)((u32 >> 16) | 0x40);
    else b16[j] = (uint16_t)((u32 + 0x7fffu + ((u32 >> 16) & 1u)) >> 16);
    const float f1 = __uint_as_float(u32 & 0xffff0000u), r1 = v[j] - f1;
    const float f2 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u), r2 = r1 - f2;
    p1[j] = (uint16_t)(__float_as_uint(f1) >> 16);
    p2[j] = (uint16_t)(__float_as_uint(f2) >> 16);
    p3[j] = (uint16_t)(__float_as_uint(r2) >> 16);
    // fp16 pieces of 2^6 w (conv.h pack_conv; v_cvt_f16_f32 rounds to nearest even and keeps subnormals, like f32_to_f16_bits)
    const float vs = v[j] * kF16WScale;
    const _Float16 fh = (_Float16)vs;
    const _Float16 fl = (_Float16)((vs - (float)fh) * kF16LoScale);
    hi[j] = __builtin_bit_cast(uint16_t, fh);
    lo[j] = __builtin_bit_cast(uint16_t, fl);
  }
  auto pack8 = [](const uint16_t (&x)[8]) {
    return make_uint4((uint32_t)x[0] | ((uint32_t)x[1] << 16), (uint32_t)x[2] | ((uint32_t)x[3] << 16), (uint32_t)x[4] | ((uint32_t)x[5] << 16),
                      (uint32_t)x[6] | ((uint32_t)x[7] << 16));
  };
  reinterpret_cast<uint4*>(a.w16)[ts * 64 + lane] = pack8(b16);
  uint4* w48 = reinterpret_cast<uint4*>(a.w48) + ts * 3 * 64;
  w48[lane] = pack8(p1);
  w48[64 + lane] = pack8(p2);
  w48[128 + lane] = pack8(p3);
  uint4* w32h = reinterpret_cast<uint4*>(a.w32h) + ts * 2 * 64;
  w32h[lane] = pack8(hi);
  w32h[64 + lane] = pack8(lo);
  if (step == 0 && h == 0) {
    float b = 0.f;
    if (sr >= 0) {
      const int bi = a.bias_mod > 0 ? sr % a.bias_mod : sr;
      if (a.b0) b = a.b0[bi];
      if (a.b1) b += a.b1[bi];
    }
    a.bias[m] = b;
  }
}

__global__ __launch_bounds__(256) void repack_kernel(const RepackJob* __restrict__ jobs, int njobs) {
  // the job of this block: last entry whose first_block <= blockIdx.x (uniform: scalar loads)
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const RepackJob& jb = jobs[lo];
  const int64_t units = (int64_t)jb.a.Mpad * jb.a.steps * 2;
  const int64_t u = (int64_t)((int)blockIdx.x - jb.first_block) * 256 + threadIdx.x;
  if (u < units) repack_unit(jb.a, u);
}

namespace {
// Jobs collected between RepackBatch::begin and ::flush on this thread (a refresh of one module: 2-120 convolutions) go out as one launch.
struct RepackBatchState {
  bool open = false;
  std::vector<RepackJob> jobs;
  int blocks = 0;
  // device copies of the tables this thread has launched, by content: the arguments of a module's refresh are the same every optimiser step
  // (parameter storages do not move), so the upload -- a pageable-memory copy that would make the host wait for the stream -- happens once
  struct Cached { std::vector<char> bytes; DevBuf dev; hipStream_t stream; };      // uploaded ON that stream: another stream uploads its own
  std::vector<std::unique_ptr<Cached>> cache;
};
RepackBatchState& repack_state() {
  static thread_local RepackBatchState* s = new RepackBatchState();      // leaked on purpose: see thread_scratch (common.h)
  return *s;
}
int repack_launch_jobs(RepackBatchState& st, hipStream_t s) {
  if (st.jobs.empty()) return DMEL_OK;
  const size_t nbytes = st.jobs.size() * sizeof(RepackJob);
  const char* raw = reinterpret_cast<const char*>(st.jobs.data());
  RepackBatchState::Cached* hit = nullptr;
  for (auto& c : st.cache)
    if (c->stream == s && c->bytes.size() == nbytes && std::memcmp(c->bytes.data(), raw, nbytes) == 0) { hit = c.get(); break; }
  if (!hit) {
    if (st.cache.size() >= 64) {      // parameters were re-allocated again and again: drop the oldest table once the stream is done with it
      DMEL_HIP(hipStreamSynchronize(s));
      st.cache.erase(st.cache.begin());
    }
    auto c = std::make_unique<RepackBatchState::Cached>();
    c->bytes.assign(raw, raw + nbytes);
    c->stream = s;
    DMEL_HIP(hipMalloc(&c->dev.p, nbytes));
    c->dev.bytes = nbytes;
    DMEL_HIP(hipMemcpyAsync(c->dev.p, c->bytes.data(), nbytes, hipMemcpyHostToDevice, s));
    hit = c.get();
    st.cache.push_back(std::move(c));
  }
  hipLaunchKernelGGL(repack_kernel, dim3((unsigned)st.blocks), dim3(256), 0, s, reinterpret_cast<const RepackJob*>(hit->dev.p), (int)st.jobs.size());
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}
}  // namespace

RepackBatch::RepackBatch(hipStream_t s) : stream(s) {
  RepackBatchState& st = repack_state();
  owner = !st.open;
  if (owner) { st.open = true; st.jobs.clear(); st.blocks = 0; }
}
int RepackBatch::flush() {
  RepackBatchState& st = repack_state();
  if (!owner || !st.open) return DMEL_OK;
  const int rc = repack_launch_jobs(st, stream);
  st.jobs.clear();
  st.blocks = 0;
  return rc;
}
RepackBatch::~RepackBatch() {
  if (!owner) return;
  RepackBatchState& st = repack_state();
  st.open = false;      // an error path left jobs behind: they are dropped with the error the caller already reports
  st.jobs.clear();
  st.blocks = 0;
}

int launch_repack(PackedConv& pc, const RepackSrc& src, hipStream_t s) {
  const PackDesc& d = pc.d;
  DMEL_CHECK_ARG(pc.w.p && pc.w16.p && pc.w48.p && pc.w32h.p && pc.bias.p, "repack: the convolution was never packed");
  DMEL_CHECK_ARG(d.nseg >= 1 && d.nseg <= 2 && src.seg[0].w && (d.nseg == 1 || src.seg[1].w), "repack: missing source tensor");
  RepackJob jb{};
  RepackArgs& a = jb.a;
  a.seg[0] = src.seg[0]; a.seg[1] = src.seg[1];
  a.b0 = src.b0; a.b1 = src.b1; a.bias_mod = src.bias_mod;
  a.w = pc.w.as<float>(); a.w16 = pc.w16.as<uint16_t>(); a.w48 = pc.w48.as<uint16_t>(); a.w32h = pc.w32h.as<uint16_t>(); a.bias = pc.bias.as<float>();
  a.mode = d.mode; a.C = d.C; a.RP = pc.RP; a.phases = d.phases; a.nseg = d.nseg; a.Mpad = pc.Mpad; a.steps = pc.steps;
  for (int i = 0; i < d.nseg; ++i) {
    a.Cin[i] = d.seg[i].Cin; a.taps[i] = d.seg[i].taps; a.nchunk[i] = (d.seg[i].Cin + kCK - 1) / kCK;
  }
  if (d.nseg == 1) { a.Cin[1] = 0; a.taps[1] = 1; a.nchunk[1] = 0; }
  const int64_t units = (int64_t)pc.Mpad * pc.steps * 2;
  const int64_t nb = (units + 255) / 256;
  RepackBatchState& st = repack_state();
  DMEL_CHECK_ARG(nb > 0 && (int64_t)st.blocks + nb < ((int64_t)1 << 30), "repack: image too large");
  const bool batched = st.open;
  if (!batched) { st.jobs.clear(); st.blocks = 0; }
  jb.first_block = st.blocks;
  st.jobs.push_back(jb);
  st.blocks += (int)nb;
  if (batched) return DMEL_OK;      // RepackBatch::flush launches
  const int rc = repack_launch_jobs(st, s);
  st.jobs.clear();
  st.blocks = 0;
  return rc;
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
int launch_gelu_fwd(const float* u, float* g, int64_t total, hipStream_t s) { DMEL_EW_LAUNCH(gelu_fwd_kernel, total, 8, u, g, total); }
int launch_gelu_bwd(const float* dg, const float* u, float* du, int64_t total, hipStream_t s) {
  DMEL_EW_LAUNCH(gelu_bwd_kernel, total, 12, dg, u, du, total);
}
int launch_layerscale_res_fwd(const float* x, const float* v, const float* gamma, float* y, int N, int C, int64_t T, hipStream_t s) {
  const int64_t total = (int64_t)N * C * T;
  DMEL_EW_LAUNCH(layerscale_res_fwd_kernel, total, 12, x, v, gamma, y, C, T, total);
}
int launch_layerscale_bwd(const float* dy, const float* v, const float* gamma, float* dv, float* dgamma, int N, int C, int64_t T,
                          hipStream_t s) {
  DMEL_CHECK_ARG(T < ((int64_t)1 << 31), "layerscale_bwd: T too large");
  hipLaunchKernelGGL(layerscale_bwd_kernel, dim3((unsigned)C), dim3(256), 0, s, dy, v, gamma, dv, dgamma, N, C, (int)T);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}
int launch_ln_bwd(const float* dh1, const float* h0, const float* ln_w, float* dh0, float* dln_w, float* dln_b, int N, int C, int64_t T,
                  hipStream_t s) {
  DMEL_CHECK_ARG(N > 0 && N <= 65535 && C > 0 && T > 0, "ln_bwd: bad shape");
  const size_t lds = ((size_t)2 * C * (kLnTile + 1) + 4 * kLnTile) * sizeof(float);
  DMEL_CHECK_ARG(lds <= 64 * 1024, "ln_bwd: %d channels exceed the LDS tile", C);
  DMEL_TRY(zero_unless_cleared(dln_w, (size_t)C * sizeof(float), s));
  DMEL_TRY(zero_unless_cleared(dln_b, (size_t)C * sizeof(float), s));
  dim3 grid((unsigned)((T + kLnTile - 1) / kLnTile), (unsigned)N);
  hipLaunchKernelGGL(ln_bwd_kernel, grid, dim3(256), lds, s, dh1, h0, ln_w, dh0, dln_w, dln_b, C, T, 1e-6f);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}
int launch_dwconv_bwd(const float* dh0, const float* x, const float* dw_w, const float* dres, float* dx, float* ddw, float* ddb, int N, int C,
                      int64_t T, hipStream_t s) {
  DMEL_CHECK_ARG(T < ((int64_t)1 << 31), "dwconv_bwd: T too large");
  hipLaunchKernelGGL(dwconv_bwd_kernel, dim3((unsigned)C), dim3(256), 0, s, dh0, x, dw_w, dres, dx, ddw, ddb, N, C, (int)T);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}
int launch_scale(const float* x, float* y, float k, int64_t total, hipStream_t s) { DMEL_EW_LAUNCH(scale_kernel, total, 8, x, y, k, total); }

}  // namespace dmel
