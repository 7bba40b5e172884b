// Anti-aliased Snake / SnakeBeta activation, fused, for gfx950.
//
// Replaces the reference's CUDA kernel (alias_free_activation/cuda/anti_alias_activation_cuda.cu:44-179) and the
// six-launch torch path (alias_free_activation/torch/act.py:25-30, resample.py:29-38, filter.py:94-101):
//   x2 upsample (replicate pad 5/5, zero-stuffed 12-tap kaiser-sinc, x2 gain, crop 15/15)
//   -> v + 1/(beta+1e-9) * sin^2(alpha v) -> replicate pad 5/6 -> 12-tap low-pass, stride 2.
//
// Polyphase form (no zero stuffing): with f the 12 taps,
//   up[2m]   = 2 * sum_j f[2j+1] * x[clamp(m+2-j)]      j = 0..5
//   up[2m+1] = 2 * sum_j f[2j]   * x[clamp(m+3-j)]
//   out[t]   = sum_k f[k] * v[clamp(2t+k-5, 0, 2T-1)],  v = snake(up)
// One workgroup owns up to 1024 consecutive outputs of one (b, c) row: the row segment (+6/+6 halo) is read
// from HBM once with coalesced dword loads into LDS, the activated 2x signal lives only in LDS as
// (even, odd) pairs, and the store is coalesced.  HBM traffic = 1 read + 1 write per element, which is the
// algorithmic minimum (8 B/element); the reference kernel's stride-32-per-thread addressing is uncoalesced.
// Measured (rocprofv3 --pmc, round 1): ~120 vector instructions per output and VALU active ~100 % of the kernel, i.e.
// the kernel is VALU-issue bound at 2.3-2.6 TB/s, not HBM bound; the two sin^2 per input sample are a third of it.
#include "ops.h"
#include "snake_dev.h"

#include <algorithm>
#include <cstdlib>


namespace dmel {

constexpr int kSnakeTile = 1024;

// 32-bit indices, and tiles that do not touch a sequence edge (block-uniform test) skip every clamp and select.
// (A v_pk_fma_f32 version of this kernel measured 0.6x: packed fp32 issues at half rate on gfx950.  Four consecutive samples
// per thread with ds_read_b128 windows -- 8 LDS reads instead of 56 per four outputs -- measured 0.84x: 2.26 vs 2.74 TB/s.)
// tu: the up-sampling taps times 2 (the x2 gain of UpSample1d, resample.py:37; exact in fp32), td: the low-pass taps of DownSample1d
// STAGED: xs was filled (and the workgroup synchronised) by the caller -- the software-pipelined kernel below
template <bool EDGE, bool STAGED = false>
__device__ __forceinline__ void aa_snake_tile(const float* __restrict__ xr, float* __restrict__ yr, const Taps12& tu, const Taps12& td,
                                              float a, float inv_b, int t0, int len, int T, float* xs, float2* vs, int tid) {
  if (!STAGED) {
    for (int i = tid; i < len + 12; i += 256) {
      int s = t0 - 6 + i;
      if (EDGE) s = min(max(s, 0), T - 1);
      xs[i] = xr[s];
    }
    __syncthreads();
  }
  for (int p = tid; p < len + 6; p += 256) {
    const int m = t0 - 3 + p;
    const int mc = EDGE ? min(max(m, 0), T - 1) : m;
    const float* xp = xs + (mc - t0 + 6);          // xp[d] = x[clamp(mc + d)]
    float ue = 0.f, uo = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      ue = fmaf(tu.f[2 * j + 1], xp[2 - j], ue);
      uo = fmaf(tu.f[2 * j], xp[3 - j], uo);
    }
    // both evaluations of sin^2 behind ONE wave-uniform test of the large-argument case: as two calls of sin_sq each is a basic block of
    // its own and the two 20-deep dependency chains run one after the other (snake_dev.h)
    float uu[2] = {ue, uo};
    const float aa[2] = {a, a}, bb[2] = {inv_b, inv_b};
    snake_n(uu, aa, bb);
    float ve = uu[0], vo = uu[1];
    if (EDGE) {
      if (m < 0) vo = ve;          // replicate pad of the 2x signal: v[0] on the left ...
      if (m > T - 1) ve = vo;      // ... v[2T-1] on the right
    }
    vs[p] = make_float2(ve, vo);
  }
  __syncthreads();
  for (int o = tid; o < len; o += 256) {
    float2 pv[7];
    lds_read7_b64(vs + o, pv);          // seven conflict-free ds_read_b64 (the compiler's six ds_read2_b32 were two-way conflicts each)
    const float2 p0 = pv[0], p1 = pv[1], p2 = pv[2], p3 = pv[3], p4 = pv[4], p5 = pv[5], p6 = pv[6];
    float acc = td.f[0] * p0.y;
    acc = fmaf(td.f[1], p1.x, acc);
    acc = fmaf(td.f[2], p1.y, acc);
    acc = fmaf(td.f[3], p2.x, acc);
    acc = fmaf(td.f[4], p2.y, acc);
    acc = fmaf(td.f[5], p3.x, acc);
    acc = fmaf(td.f[6], p3.y, acc);
    acc = fmaf(td.f[7], p4.x, acc);
    acc = fmaf(td.f[8], p4.y, acc);
    acc = fmaf(td.f[9], p5.x, acc);
    acc = fmaf(td.f[10], p5.y, acc);
    acc = fmaf(td.f[11], p6.x, acc);
    yr[t0 + o] = acc;
  }
}

// ---- round-2 forward kernel -------------------------------------------------------------------------------------------------
// Same algorithm and data flow as aa_snake_tile above, re-cut to spend fewer instructions per output (the kernel is VALU-issue bound:
// ~120 vector + scalar instructions per output in the version above, profiles/r01_pmc_traffic.md):
//   * 1018 outputs per workgroup, so that the activated 2x signal the tile needs is exactly 1024 pairs: the pair loop is two full
//     iterations of 256 threads x 2 pairs instead of five iterations with the last one 2 % full;
//   * two adjacent pairs per thread share their x window (8 LDS words for 24 FMAs instead of 14) and are stored as one 16-byte
//     unit; two adjacent outputs per thread share their window of the 2x signal (four 16-byte LDS reads instead of fourteen 8-byte
//     ones) and are stored as one 8-byte unit;
//   * sin^2 by reduction modulo PI (sin^2 has period pi, so the quadrant bookkeeping of the pi/2 form -- convert, and, compare, select,
//     1 - s^2 -- disappears) and an odd degree-11 polynomial on |r| <= pi/2: max abs error 2.5e-7, rms 4e-8 against fp64 on 1e7
//     points with |x| up to 8e3 (the pi/2 form: 1.2e-7 / 2.2e-8; both far inside the 1e-5 bar of the activation tests);
//   * the large-argument fallback (|a u| > 8192, never taken by a trained network) is one WAVE-uniform branch per four values
//     (ballot) instead of a divergent exec-mask region around every value.
constexpr int kFwdTile = 1018;

__device__ __forceinline__ float sin_sq_pi(float x) {      // |x| <= 8192
  const float n = rintf(x * 0.318309886183790671538f);
  float r = fmaf(n, -3.140625f, x);                          // pi = 3.140625 + 9.67502593994140625e-4 + 1.509957990978376432e-7: the
  r = fmaf(n, -9.67502593994140625e-4f, r);                  // products n * piece are exact for |n| < 2^13
  r = fmaf(n, -1.509957990978376432e-7f, r);
  const float r2 = r * r;
  float p = -2.3841987939476894e-08f;                        // near-minimax fit of sin(r) / r - 1 in r^2 on [0, (pi/2)^2]
  p = fmaf(p, r2, 2.7522303298610495e-06f);
  p = fmaf(p, r2, -1.9840797176584601e-04f);
  p = fmaf(p, r2, 8.333330042660236e-03f);
  p = fmaf(p, r2, -1.666666716337204e-01f);
  const float s = fmaf(r * r2, p, r);
  return s * s;
}
__device__ __forceinline__ void snake4(float (&u)[4], float a, float inv_b) {     // u -> u + inv_b sin^2(a u), four values at a time
  float w[4];
  bool big = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) { w[i] = u[i] * a; big = big || fabsf(w[i]) > 8192.f; }
  if (__builtin_amdgcn_ballot_w64(big) != 0) {              // wave-uniform
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = fmaf(inv_b, sin_sq(w[i]), u[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = fmaf(inv_b, sin_sq_pi(w[i]), u[i]);
  }
}

template <bool EDGE>
__device__ __forceinline__ void aa_snake_tile2(const float* __restrict__ xr, float* __restrict__ yr, const Taps12& tu, const Taps12& td,
                                               float a, float inv_b, int t0, int len, int T, float* xs, float4* vs, int tid) {
  // xs[i] = x[clamp(t0 - 6 + i)], i < len + 12 (+ 2 slack words read by the last pair's window)
  for (int i = tid; i < len + 14; i += 256) {
    int s = t0 - 6 + i;
    if (EDGE || i >= len + 12) s = min(max(s, 0), T - 1);
    xs[i] = xr[s];
  }
  __syncthreads();
  // pairs p = 0 .. len + 5 hold (v[2m], v[2m+1]) for m = t0 - 3 + p; thread handles p = 2i, 2i + 1
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = tid + 256 * it;
    if (2 * i < len + 6) {
      const int m0 = t0 - 3 + 2 * i;
      float u[4];          // ue(m0), uo(m0), ue(m0 + 1), uo(m0 + 1)
      if (!EDGE) {
        // x[m0 - 3 .. m0 + 4] = xs[2 i .. 2 i + 7]: four 8-byte reads (lane stride 8 bytes: conflict-free; four-byte reads at a
        // two-word lane stride would be two-way bank conflicts)
        const float2* xp2 = reinterpret_cast<const float2*>(xs + 2 * i);
        float w[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) { const float2 t = xp2[d]; w[2 * d] = t.x; w[2 * d + 1] = t.y; }
        u[0] = u[1] = u[2] = u[3] = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          u[0] = fmaf(tu.f[2 * j + 1], w[5 - j], u[0]);        // x[m0 + 2 - j]
          u[1] = fmaf(tu.f[2 * j], w[6 - j], u[1]);            // x[m0 + 3 - j]
          u[2] = fmaf(tu.f[2 * j + 1], w[6 - j], u[2]);        // x[m0 + 1 + 2 - j]
          u[3] = fmaf(tu.f[2 * j], w[7 - j], u[3]);            // x[m0 + 1 + 3 - j]
        }
        snake4(u, a, inv_b);
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int m = m0 + q;
          const int mc = min(max(m, 0), T - 1);
          const float* xp = xs + (mc - t0 + 6);
          float ue = 0.f, uo = 0.f;
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            ue = fmaf(tu.f[2 * j + 1], xp[2 - j], ue);
            uo = fmaf(tu.f[2 * j], xp[3 - j], uo);
          }
          u[2 * q] = ue; u[2 * q + 1] = uo;
        }
        snake4(u, a, inv_b);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int m = m0 + q;
          if (m < 0) u[2 * q + 1] = u[2 * q];          // replicate pad of the 2x signal: v[0] on the left ...
          if (m > T - 1) u[2 * q] = u[2 * q + 1];      // ... v[2T-1] on the right
        }
      }
      vs[i] = make_float4(u[0], u[1], u[2], u[3]);
    }
  }
  __syncthreads();
  // outputs o = 2i, 2i + 1: out[t0 + o] = sum_k td[k] v[2 (t0 + o) + k - 5]; pair index of v[2 t + k - 5] is (o + (k + 1) / 2) rel. to p
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = tid + 256 * it;
    const int o = 2 * i;
    if (o < len) {
      const float4 q0 = vs[i], q1 = vs[i + 1], q2 = vs[i + 2], q3 = vs[i + 3];
      // pairs o .. o+7 as (even, odd): P0 = (q0.x, q0.y), P1 = (q0.z, q0.w), P2 = (q1.x, q1.y), ...
      float acc0 = td.f[0] * q0.y;
      acc0 = fmaf(td.f[1], q0.z, acc0);
      acc0 = fmaf(td.f[2], q0.w, acc0);
      acc0 = fmaf(td.f[3], q1.x, acc0);
      acc0 = fmaf(td.f[4], q1.y, acc0);
      acc0 = fmaf(td.f[5], q1.z, acc0);
      acc0 = fmaf(td.f[6], q1.w, acc0);
      acc0 = fmaf(td.f[7], q2.x, acc0);
      acc0 = fmaf(td.f[8], q2.y, acc0);
      acc0 = fmaf(td.f[9], q2.z, acc0);
      acc0 = fmaf(td.f[10], q2.w, acc0);
      acc0 = fmaf(td.f[11], q3.x, acc0);
      float acc1 = td.f[0] * q0.w;
      acc1 = fmaf(td.f[1], q1.x, acc1);
      acc1 = fmaf(td.f[2], q1.y, acc1);
      acc1 = fmaf(td.f[3], q1.z, acc1);
      acc1 = fmaf(td.f[4], q1.w, acc1);
      acc1 = fmaf(td.f[5], q2.x, acc1);
      acc1 = fmaf(td.f[6], q2.y, acc1);
      acc1 = fmaf(td.f[7], q2.z, acc1);
      acc1 = fmaf(td.f[8], q2.w, acc1);
      acc1 = fmaf(td.f[9], q3.x, acc1);
      acc1 = fmaf(td.f[10], q3.y, acc1);
      acc1 = fmaf(td.f[11], q3.z, acc1);
      if (o + 1 < len) *reinterpret_cast<float2*>(yr + t0 + o) = make_float2(acc0, acc1);
      else yr[t0 + o] = acc0;
    }
  }
}

__global__ __launch_bounds__(256) void aa_snake_kernel2(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ alpha, const float* __restrict__ beta,
                                                        Taps12 tu, Taps12 td, int logscale, int C, int T) {
  __shared__ __attribute__((aligned(16))) float xs[kFwdTile + 14];
  __shared__ __attribute__((aligned(16))) float4 vs[516];          // 512 units of two pairs + the three the last outputs reach into
  const int c = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * kFwdTile;
  const int len = min(kFwdTile, T - t0);
  const float* xr = x + ((int64_t)b * C + c) * T;
  float* yr = y + ((int64_t)b * C + c) * T;
  float a = alpha[c], bt = beta ? beta[c] : a;
  if (logscale) {
    bt = beta ? expf(bt) : expf(a);
    a = expf(a);
  }
  const float inv_b = 1.0f / (bt + 1e-9f);
  if (t0 >= 6 && t0 + len + 8 <= T) aa_snake_tile2<false>(xr, yr, tu, td, a, inv_b, t0, len, T, xs, vs, threadIdx.x);
  else aa_snake_tile2<true>(xr, yr, tu, td, a, inv_b, t0, len, T, xs, vs, threadIdx.x);
}

__global__ __launch_bounds__(256) void aa_snake_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const float* __restrict__ alpha, const float* __restrict__ beta,
                                                          Taps12 tu, Taps12 td, int logscale, int C, int T, int nsub) {
  __shared__ float xs[kSnakeTile + 12];
  __shared__ float2 vs[kSnakeTile + 6];
  const int c = blockIdx.y, b = blockIdx.z;
  const float* xr = x + ((int64_t)b * C + c) * T;
  float* yr = y + ((int64_t)b * C + c) * T;
  float a = alpha[c], bt = beta ? beta[c] : a;
  if (logscale) {
    bt = beta ? expf(bt) : expf(a);
    a = expf(a);
  }
  const float inv_b = 1.0f / (bt + 1e-9f);
  // nsub consecutive tiles per workgroup, SOFTWARE-PIPELINED: the row segment of tile s + 1 is fetched into registers before tile s is
  // computed.  The kernel reads 4 KB per workgroup and then computes for ~4 us without touching memory: with 8 workgroups per CU only
  // ~10 KB per CU were in flight at any time, and bytes-in-flight / latency (Little's law: 10 KB / 2 us x 256 CUs = 1.3 TB/s) was
  // exactly the read rate the kernel achieved -- it was bound by memory-level parallelism, not by instruction issue.
  const int tid = threadIdx.x;
  constexpr int NR = (kSnakeTile + 12 + 255) / 256;
  float pre[NR];
  auto fetch = [&](int t0) {
    const int len = min(kSnakeTile, T - t0);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int i = tid + 256 * k;
      const int sidx = min(max(t0 - 6 + min(i, len + 11), 0), T - 1);
      pre[k] = xr[sidx];
    }
  };
  const int first = blockIdx.x * nsub * kSnakeTile;
  if (first < T) fetch(first);
  for (int sub = 0; sub < nsub; ++sub) {
    const int t0 = first + sub * kSnakeTile;
    if (t0 >= T) break;
    const int len = min(kSnakeTile, T - t0);
    if (sub) __syncthreads();                  // the previous tile's reads of xs / vs are done
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int i = tid + 256 * k;
      if (i < len + 12) xs[i] = pre[k];
    }
    if (sub + 1 < nsub && t0 + kSnakeTile < T) fetch(t0 + kSnakeTile);     // in flight while this tile is computed
    __syncthreads();
    if (t0 >= 6 && t0 + len + 6 <= T) aa_snake_tile<false, true>(xr, yr, tu, td, a, inv_b, t0, len, T, xs, vs, tid);
    else aa_snake_tile<true, true>(xr, yr, tu, td, a, inv_b, t0, len, T, xs, vs, tid);
  }
}

// ---- backward ------------------------------------------------------------------------------------------------------
// The reference has no backward for its fused kernel (alias_free_activation/cuda/activation1d.py:29-32 raises
// NotImplementedError; training goes through the six-op torch path and autograd).  Reverse mode of the forward above:
//   dv[i]   = sum_{t,k : clamp(2t+k-5) = i} f[k] dy[t]                      (transposed low-pass, replicate pads folded into the ends)
//   du      = dv * (1 + a * inv_b * sin(2 a u))                              (d/du [u + inv_b sin^2(a u)])
//   dx[s]   = sum_{m,j : clamp(m+2-j) = s} 2 f[2j+1] du_even[m] + sum_{m,j : clamp(m+3-j) = s} 2 f[2j] du_odd[m]
//   d a     = sum dv * inv_b * u * sin(2 a u),   d inv_b = sum dv * sin^2(a u)
// One workgroup owns the dx of up to 1024 samples of one (b, c) row, recomputes u from x (+6/+6 halo) and forms the two
// transposed filters by scatter-adding into LDS (ds_add_f32), which handles the replicate padding at the row ends with the same
// code as the interior.  Parameter gradients: block reduction, one atomic per workgroup.  Correctness first: this kernel has not
// been tuned (the vocoder is frozen in the reference's training_step; it matters for vocoder fine-tuning only).
__global__ __launch_bounds__(256) void aa_snake_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dx, const float* __restrict__ alpha,
                                                            const float* __restrict__ beta, float* __restrict__ dalpha,
                                                            float* __restrict__ dbeta, Taps12 tu, Taps12 td, int logscale, int C, int T) {
  __shared__ float xs[kSnakeTile + 12];
  __shared__ float dys[kSnakeTile + 12];
  __shared__ float dvs[2 * (kSnakeTile + 6)];
  __shared__ float dxs[kSnakeTile];
  __shared__ float red[2][4];
  const int c = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int t0 = blockIdx.x * kSnakeTile;
  const int len = min(kSnakeTile, T - t0);
  const float* xr = x + ((int64_t)b * C + c) * T;
  const float* dyr = dy + ((int64_t)b * C + c) * T;
  float* dxr = dx + ((int64_t)b * C + c) * T;
  float a = alpha[c], bt = beta ? beta[c] : a;
  if (logscale) {
    bt = beta ? expf(bt) : expf(a);
    a = expf(a);
  }
  const float inv_b = 1.0f / (bt + 1e-9f);
  for (int i = tid; i < len + 12; i += 256) {
    const int s = t0 - 6 + i;
    xs[i] = xr[min(max(s, 0), T - 1)];
    dys[i] = (s >= 0 && s < T) ? dyr[s] : 0.f;
  }
  for (int i = tid; i < 2 * (len + 6); i += 256) dvs[i] = 0.f;
  for (int i = tid; i < len; i += 256) dxs[i] = 0.f;
  __syncthreads();
  // transposed low-pass: every dy sample in reach scatters its 12 taps into the pairs this tile owns (m in [t0-3, t0+len+3))
  for (int i = tid; i < len + 12; i += 256) {
    const int t = t0 - 6 + i;
    if (t < 0 || t >= T) continue;
    const float g = dys[i];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int vi = min(max(2 * t + k - 5, 0), 2 * T - 1);
      const int p = (vi >> 1) - (t0 - 3);
      if (p >= 0 && p < len + 6) atomicAdd(&dvs[2 * p + (vi & 1)], td.f[k] * g);
    }
  }
  __syncthreads();
  float sa = 0.f, sb = 0.f;
  for (int p = tid; p < len + 6; p += 256) {
    const int m = t0 - 3 + p;
    if (m < 0 || m > T - 1) continue;
    const float* xp = xs + (m - t0 + 6);          // xp[d] = x[clamp(m + d)] (xs holds the clamped row)
    float ue = 0.f, uo = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      ue = fmaf(tu.f[2 * j + 1], xp[2 - j], ue);
      uo = fmaf(tu.f[2 * j], xp[3 - j], uo);
    }
    const float dve = dvs[2 * p], dvo = dvs[2 * p + 1];
    const float s2e = sinf(2.f * a * ue), s2o = sinf(2.f * a * uo);
    const float due = dve * fmaf(a * inv_b, s2e, 1.f), duo = dvo * fmaf(a * inv_b, s2o, 1.f);
    if (m >= t0 && m < t0 + len) {                // parameter gradients: every pair is counted by exactly one tile
      sa += inv_b * (dve * ue * s2e + dvo * uo * s2o);
      sb += dve * sin_sq(a * ue) + dvo * sin_sq(a * uo);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int se = min(max(m + 2 - j, 0), T - 1) - t0, so = min(max(m + 3 - j, 0), T - 1) - t0;
      if (se >= 0 && se < len) atomicAdd(&dxs[se], tu.f[2 * j + 1] * due);
      if (so >= 0 && so < len) atomicAdd(&dxs[so], tu.f[2 * j] * duo);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sa += __shfl_xor(sa, o, 64);
    sb += __shfl_xor(sb, o, 64);
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = sa; red[1][tid >> 6] = sb; }
  __syncthreads();
  for (int i = tid; i < len; i += 256) dxr[t0 + i] = dxs[i];
  if (tid == 0) {
    const float ga = red[0][0] + red[0][1] + red[0][2] + red[0][3];        // d loss / d a'      (a' = effective alpha)
    const float gib = red[1][0] + red[1][1] + red[1][2] + red[1][3];       // d loss / d inv_b
    const float gb = -inv_b * inv_b * gib;                                 // d loss / d b'      (inv_b = 1 / (b' + 1e-9))
    if (beta) {
      atomicAdd(&dalpha[c], logscale ? ga * a : ga);
      atomicAdd(&dbeta[c], logscale ? gb * bt : gb);
    } else {
      atomicAdd(&dalpha[c], logscale ? (ga + gb) * a : ga + gb);           // Snake: b' = a'
    }
  }
}

int launch_aa_snake_bwd(const float* x, const float* dy, float* dx, const float* alpha, const float* beta, float* dalpha, float* dbeta,
                        const float* up_taps_host, const float* down_taps_host, int logscale, int B, int C, int64_t T, hipStream_t s) {
  DMEL_CHECK_ARG(x && dy && dx && alpha && dalpha && up_taps_host && down_taps_host, "aa_snake_backward: NULL argument");
  DMEL_CHECK_ARG((beta != nullptr) == (dbeta != nullptr), "aa_snake_backward: dbeta must be given exactly when beta is");
  DMEL_CHECK_ARG(B > 0 && C > 0 && T > 0 && B <= 65535 && C <= 65535 && T < ((int64_t)1 << 29), "aa_snake_backward: bad shape");
  Taps12 tu, td;
  for (int i = 0; i < 12; ++i) { tu.f[i] = 2.f * up_taps_host[i]; td.f[i] = down_taps_host[i]; }
  DMEL_HIP(hipMemsetAsync(dalpha, 0, (size_t)C * sizeof(float), s));
  if (dbeta) DMEL_HIP(hipMemsetAsync(dbeta, 0, (size_t)C * sizeof(float), s));
  dim3 grid((unsigned)((T + kSnakeTile - 1) / kSnakeTile), (unsigned)C, (unsigned)B);
  {
    ProfScope ps("aa_snake_bwd", s, 0.0, 12.0 * (double)B * C * (double)T);
    hipLaunchKernelGGL(aa_snake_bwd_kernel, grid, dim3(256), 0, s, x, dy, dx, alpha, beta, dalpha, dbeta, tu, td, logscale, C, (int)T);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

int launch_aa_snake(const float* x, float* y, const float* alpha, const float* beta, const float* up_taps_host,
                    const float* down_taps_host, int logscale, int B, int C, int64_t T, hipStream_t s) {
  DMEL_CHECK_ARG(x && y && alpha && up_taps_host && down_taps_host, "aa_snake: NULL argument");
  DMEL_CHECK_ARG(B > 0 && C > 0 && T > 0 && B <= 65535 && C <= 65535 && T < ((int64_t)1 << 30), "aa_snake: bad shape");
  Taps12 tu, td;
  for (int i = 0; i < 12; ++i) { tu.f[i] = 2.f * up_taps_host[i]; td.f[i] = down_taps_host[i]; }
  const char* e = getenv("DMEL_SNAKE_V2");                 // A/B: the re-cut kernel (aa_snake_kernel2), measured 5-15 % slower
  const bool v2 = e && e[0] == '1';
  const char* e2 = getenv("DMEL_SNAKE_NSUB");
  // two tiles per workgroup (the second one's row segment is in flight while the first is computed): +4-8 % over one; four or eight
  // leave the last workgroup of a 5888- or 11776-sample row with a fraction of the work (measured 15-25 % slower on those rows)
  const int nsub = v2 ? 1 : std::max(1, e2 ? atoi(e2) : 2);
  const int tile = v2 ? kFwdTile : kSnakeTile * nsub;
  dim3 grid((unsigned)((T + tile - 1) / tile), (unsigned)C, (unsigned)B);
  {
    ProfScope ps("aa_snake", s, 0.0, 8.0 * (double)B * C * (double)T);
    if (!v2) hipLaunchKernelGGL(aa_snake_kernel, grid, dim3(256), 0, s, x, y, alpha, beta, tu, td, logscale, C, (int)T, nsub);
    else hipLaunchKernelGGL(aa_snake_kernel2, grid, dim3(256), 0, s, x, y, alpha, beta, tu, td, logscale, C, (int)T);
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

}  // namespace dmel

extern "C" int dmel_aa_snake_backward_f32(const float* x, const float* dy, float* dx, const float* alpha, const float* beta,
                                          float* dalpha, float* dbeta, const float* up_filter12_host, const float* down_filter12_host,
                                          int logscale, int B, int C, int64_t T, void* stream) {
  return dmel::launch_aa_snake_bwd(x, dy, dx, alpha, beta, dalpha, dbeta, up_filter12_host, down_filter12_host, logscale, B, C, T,
                                   (hipStream_t)stream);
}

extern "C" int dmel_aa_snake_f32(const float* x, float* y, const float* alpha, const float* beta, const float* up_filter12_host,
                                 const float* down_filter12_host, int logscale, int B, int C, int64_t T, void* stream) {
  return dmel::launch_aa_snake(x, y, alpha, beta, up_filter12_host, down_filter12_host, logscale, B, C, T, (hipStream_t)stream);
}
