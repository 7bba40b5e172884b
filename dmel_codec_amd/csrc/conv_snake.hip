// Anti-aliased Snake fused into the convolution that consumes it (gfx950).
//
// Every convolution of an AMP block reads the output of an Activation1d (bigvgan.py:132-141: xt = a1(x); xt = c1(xt); xt = a2(xt);
// xt = c2(xt); x = xt + x -- alias_free_activation/torch/act.py:25-30).  As two kernels the activated tensor makes a round trip
// through HBM (11.8 GB per bench step) and the two kernels never meet on a CU: the activation is VALU-bound, the convolution
// matrix-pipe-bound, and each one's stalls (HBM / L2 / LDS latency, barriers) are idle issue slots.  Here the workgroup has two kinds
// of waves:
//
//   * NPROD PRODUCER waves compute the activation of the next 16-channel chunk of the x tile -- row segment (+6/+6) -> wave-private LDS,
//     polyphase x2 up-sampling, u + 1/b sin^2(a u), 12-tap low-pass back to the input rate: the arithmetic of aa_snake_tile
//     (aa_snake.hip), expression for expression -- split the result into the two fp16 pieces of the three-product matrix-core form
//     and write them into the B-operand image conv_bf16_kernel<NP = 2> stages ([piece][8-channel group][column][8 x fp16]);
//   * WM x WN CONSUMER waves run the K loop of conv_bf16_kernel<NP = 2> on the chunk staged before (weights straight from L2 into
//     registers in A-fragment order, B fragments by ds_read_b128, 3 MFMAs per 32x32x16 block) and nothing else;
//   * one workgroup barrier per chunk swaps the two B buffers.
//
// tools/probe/dualpipe_f16*.hip (profiles/r03_dualpipe_f16.txt) measured what the two kinds of waves get from a SIMD they share: a dense
// MFMA stream keeps the vector issue port (matrix waves run at full speed whatever the age or priority of the others), the VALU
// waves get the cycles in which the matrix wave is NOT presenting an MFMA -- its waits for weights, LDS and barriers.  So the fused kernel
// costs about (MFMA issue time) + (VALU issue time) with every wait of one role covered by the other, instead of two kernels that each
// pay their own waits plus an HBM round trip in between.
//
// The outputs are BIT-IDENTICAL to launch_aa_snake + launch_conv (fp16 split): same activation expressions, same operand split, same
// K order (chunk-major, tap-minor), same MFMA order per block, same epilogue (tests/test_gpu_parity.py holds it to torch.equal).
#include "conv_dev.h"
#include "snake_dev.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace dmel {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct SnakeArgs {
  const float* alpha;
  const float* beta;      // Snake (beta = alpha): the alpha pointer again
  int logscale;
  Taps12 tu, td;          // up-sampling taps x 2, low-pass taps
};

constexpr int kSegW = 122;    // output columns per producer item: 122 + 6 = 128 pairs of the 2x signal = two full passes of 64 lanes
constexpr int kXRow = 136;    // floats per staged x row (kSegW + 12 = 134) and float2 per activated row (128 + the 6 idle lanes read ahead)

// ACT = false: the same workgroup without the activation -- the producers only stage x (load, scale, split, LDS), the consumers run the
// same lean MFMA loop.  A plain convolution whose matrix waves carry no staging arithmetic (round-3 experiment against conv_bf16_kernel).
template <int WM, int WN, int NT, int NPROD, int HALO, bool ACT = true>
__global__ __launch_bounds__(64 * (WM * WN + NPROD)) void conv_snake_kernel(KArgs a, SnakeArgs sa) {
  constexpr int NC = WM * WN;
  constexpr int BN = WN * NT * 32;
  constexpr int XS = BN + HALO;
  constexpr int KG = 2;                                   // 8-channel groups per staged chunk: one 16-channel K step per tap
  constexpr int PSZ = KG * XS;                            // uint4 per piece
  constexpr int NSEGMAX = (XS + kSegW - 1) / kSegW;
  constexpr int IPP = (8 * NSEGMAX + NPROD - 1) / NPROD;  // items (channel pair, column segment) per producer and chunk, at most
  constexpr int SCR = IPP * 2 * kXRow + 4 * kXRow;        // floats of private LDS per producer: xs[IPP][2][kXRow], vs[2][kXRow] float2
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint4* Xb = reinterpret_cast<uint4*>(smem);             // [2 buffers][2 pieces][KG][XS] x 16 bytes

  const int lane = threadIdx.x & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tile_n = blockIdx.x, mblk = blockIdx.y, b = blockIdx.z;
  const int q0 = tile_n * BN;
  const int taps = a.seg[0].taps, dil = a.seg[0].dil, nchunk = a.seg[0].nchunk;
  const int steps = a.steps;

  if (wave_u < NC) {
    // ------------------------------------------------------------------------------------------------ consumer: the MFMA loop
    const int wave_m = wave_u / WN, wave_n = wave_u % WN;
    const int h = lane >> 5, l31 = lane & 31;
    floatx16 acc[1][NT], acl[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][ni][r] = 0.f; acl[ni][r] = 0.f; }
    const int tile = min(mblk * WM + wave_m, a.mtiles - 1);
    const char* wT = reinterpret_cast<const char*>(a.w32h) + (size_t)tile * steps * 2048;
    const uint32_t lane16 = lane * 16;
    auto load_w = [&](uint4 (&dst)[2], int step) __attribute__((always_inline)) {
      const char* sp = wT + (size_t)step * 2048;
      dst[0] = *reinterpret_cast<const uint4*>(sp + lane16);
      dst[1] = *reinterpret_cast<const uint4*>(sp + 1024 + lane16);
    };
    constexpr int PD = 2;
    uint4 wa[PD + 1][2];
    load_w(wa[0], 0);
    load_w(wa[1], min(1, steps - 1));
    __syncthreads();                                      // chunk 0 is staged
    int tap = 0, xbuf = 0;
    constexpr int kWaitW = (2 & 15) | (7 << 4) | (15 << 8);      // s_waitcnt vmcnt(2): the weights of the next step have landed
    auto k_step = [&](auto R, int s) __attribute__((always_inline)) {
      constexpr int r = decltype(R)::value;
      uint4 (&use)[2] = wa[r % (PD + 1)];
      load_w(wa[(r + PD) % (PD + 1)], min(s + PD, steps - 1));   // unconditional: see conv_bf16_kernel (a branch here costs a vmcnt(0))
      const uint4* xp = Xb + xbuf * (2 * PSZ) + h * XS + wave_n * (NT * 32) + l31 + tap * dil;
      f16x8 bh[NT], bl[NT];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        bh[ni] = __builtin_bit_cast(f16x8, xp[ni * 32]);
        bl[ni] = __builtin_bit_cast(f16x8, xp[PSZ + ni * 32]);
      }
      const f16x8 ah = __builtin_bit_cast(f16x8, use[0]), al = __builtin_bit_cast(f16x8, use[1]);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        acl[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ni], acl[ni], 0, 0, 0);
        acl[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ni], acl[ni], 0, 0, 0);
        acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ni], acc[0][ni], 0, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(kWaitW);
      if (++tap == taps) {
        tap = 0;
        if (s + 1 < steps) {
          __syncthreads();                                // the producers have staged the next chunk; this one may be overwritten
          xbuf ^= 1;
        }
      }
    };
    for (int s = 0; s < steps; s += 3) {
      k_step(std::integral_constant<int, 0>{}, s);
      if (s + 1 < steps) k_step(std::integral_constant<int, 1>{}, s + 1);
      if (s + 2 < steps) k_step(std::integral_constant<int, 2>{}, s + 2);
    }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][ni][r] = fmaf(acl[ni][r], 1.f / kF16LoScale, acc[0][ni][r]);
    conv_epilogue<1, NT, EPI_LINEAR, 8, true>(a, acc, (mblk * WM + wave_m) * 32, q0 + wave_n * (NT * 32) + l31, b, b, h);
    return;
  }

  // -------------------------------------------------------------------------------------------------- producer: the activation
  // Work of a chunk: 8 channel pairs x nseg column segments of <= kSegW columns = "items"; producer p takes items p, p + NPROD, ...
  // (at most IPP).  Per chunk: (1) the x rows of all its items, fetched into registers one chunk ahead, go to private LDS rows;
  // (2) the loads of the next chunk are issued; (3) the items are computed one after the other (a run-time loop: one copy of the code).
  const int p = wave_u - NC;
  if constexpr (!ACT) {
    // plain staging: an item is a channel pair x up to 128 columns (two passes of 64 lanes); loads of the next chunk in flight while this one
    // is converted
    constexpr int SEG2 = 128, NS2 = (XS + SEG2 - 1) / SEG2, IP2 = (8 * NS2 + NPROD - 1) / NPROD;
    const int T = (int)a.seg[0].Tin, Cin = a.seg[0].Cin;
    const int wx = BN + (taps - 1) * dil;
    const int tau0 = q0 - a.seg[0].pad_left;
    const int nitems = 8 * ((wx + SEG2 - 1) / SEG2);
    const float* xb = a.seg[0].x + (int64_t)b * a.seg[0].bstride;
    const int cs = (int)a.seg[0].cstride;
    float pre[IP2][2][2];
    auto fetch2 = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < IP2; ++r) {
        const int it = p + NPROD * r;
        if (it >= nitems) break;
        const int cp = it & 7, tb = tau0 + (it >> 3) * SEG2;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float* xr = xb + (int64_t)min(chunk * 16 + 2 * cp + c, Cin - 1) * cs;
#pragma unroll
          for (int k = 0; k < 2; ++k) pre[r][c][k] = xr[min(max(tb + lane + 64 * k, 0), T - 1)];
        }
      }
    };
    fetch2(0);
    for (int ck = 0; ck < nchunk; ++ck) {
      uint4* dst = Xb + (ck & 1) * (2 * PSZ);
      float cur[IP2][2][2];
#pragma unroll
      for (int r = 0; r < IP2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int k = 0; k < 2; ++k) cur[r][c][k] = pre[r][c][k];
      if (ck + 1 < nchunk) fetch2(ck + 1);
#pragma unroll
      for (int r = 0; r < IP2; ++r) {
        const int it = p + NPROD * r;
        if (it >= nitems) break;
        const int cp = it & 7, j0 = (it >> 3) * SEG2, w = min(SEG2, wx - j0), tb = tau0 + j0;
        const float sc0 = ck * 16 + 2 * cp < Cin ? kF16XScale : 0.f, sc1 = ck * 16 + 2 * cp + 1 < Cin ? kF16XScale : 0.f;
        uint32_t* d32 = reinterpret_cast<uint32_t*>(dst) + ((2 * cp) >> 3) * (XS * 4) + (((2 * cp) & 7) >> 1);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int o = lane + 64 * k, t = tb + o;
          const bool ok = t >= 0 && t < T;
          {
#pragma clang fp contract(off)
            const float v0 = (ok ? cur[r][0][k] : 0.f) * sc0, v1 = (ok ? cur[r][1][k] : 0.f) * sc1;
            const f16x2 hi = __builtin_convertvector((f32x2){v0, v1}, f16x2);
            const f16x2 lo = __builtin_convertvector((f32x2){(v0 - (float)hi[0]) * kF16LoScale, (v1 - (float)hi[1]) * kF16LoScale}, f16x2);
            if (o < w) {
              d32[(j0 + o) * 4] = __builtin_bit_cast(uint32_t, hi);
              d32[PSZ * 4 + (j0 + o) * 4] = __builtin_bit_cast(uint32_t, lo);
            }
          }
        }
      }
      __syncthreads();
    }
    return;
  }
  float* xsb = smem + 2 * 2 * PSZ * 4 + p * SCR;           // xs[IPP][2][kXRow]
  float2* vs0 = reinterpret_cast<float2*>(xsb + IPP * 2 * kXRow);
  float2* vs1 = vs0 + kXRow;
  for (int i = lane; i < SCR; i += 64) xsb[i] = 0.f;       // idle lanes read beyond what an item writes: keep that finite
  const int T = (int)a.seg[0].Tin, Cin = a.seg[0].Cin;
  const int wx = BN + (taps - 1) * dil;                   // staged columns: the tile plus the convolution's reach
  const int tau0 = q0 - a.seg[0].pad_left;                // input time of staged column 0
  const int nseg = (wx + kSegW - 1) / kSegW, nitems = 8 * nseg;
  const float* xb = a.seg[0].x + (int64_t)b * a.seg[0].bstride;
  const int cs = (int)a.seg[0].cstride;

  float pre[IPP][2][3];
  float raw_a, raw_b;           // the chunk's Snake parameters as stored: lane l holds channel 16 chunk + (l & 15)
  auto fetch = [&](int chunk) __attribute__((always_inline)) {
    {
      const int ch = min(chunk * 16 + (lane & 15), Cin - 1);
      raw_a = sa.alpha[ch];
      raw_b = sa.beta[ch];            // Snake: the launcher passes alpha for beta (one more load beats a branch that waits for raw_a)
    }
#pragma unroll
    for (int r = 0; r < IPP; ++r) {
      const int it = p + NPROD * r;
      if (it >= nitems) break;
      const int cp = it & 7, sg = it >> 3;
      const int j0 = sg * kSegW, w = min(kSegW, wx - j0);
      const int tb = tau0 + j0;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float* xr = xb + (int64_t)min(chunk * 16 + 2 * cp + c, Cin - 1) * cs;
#pragma unroll
        for (int k = 0; k < 3; ++k) pre[r][c][k] = xr[min(max(tb - 6 + min(lane + 64 * k, w + 11), 0), T - 1)];
      }
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < IPP; ++r) {
      const int it = p + NPROD * r;
      if (it >= nitems) break;
      const int w = min(kSegW, wx - (it >> 3) * kSegW);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int i = lane + 64 * k;
        if (i < w + 12) { xsb[(r * 2) * kXRow + i] = pre[r][0][k]; xsb[(r * 2 + 1) * kXRow + i] = pre[r][1][k]; }
      }
    }
  };
  fetch(0);
  for (int ck = 0; ck < nchunk; ++ck) {
    uint4* dst = Xb + (ck & 1) * (2 * PSZ);
    stage();                                              // LDS operations of one wave execute in order: no barrier anywhere in here
    // the chunk's effective parameters, one channel per lane (aa_snake_kernel's forms, expression for expression); an item reads its
    // two channels with v_readlane -- per-item loads of alpha / beta were four dependent memory round trips per item
    float eff_a = raw_a, eff_b = raw_b;
    if (sa.logscale) {
      eff_b = expf(eff_b);        // Snake: raw_b is alpha, so this is expf(alpha) as in the activation kernel
      eff_a = expf(eff_a);
    }
    const float eff_ib = 1.0f / (eff_b + 1e-9f);
    if (ck + 1 < nchunk) fetch(ck + 1);                   // in flight while this chunk is computed
#pragma unroll 1
    for (int r = 0; r < IPP; ++r) {
      const int it = p + NPROD * r;
      if (it >= nitems) break;
      const int cp = it & 7, sg = it >> 3;
      const int j0 = sg * kSegW, w = min(kSegW, wx - j0);
      const int tb = tau0 + j0;
      const float* xs0 = xsb + (r * 2) * kXRow;
      const float* xs1 = xs0 + kXRow;
      float al[2], ib[2], sc[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        al[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eff_a), 2 * cp + c));
        ib[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eff_ib), 2 * cp + c));
        sc[c] = ck * 16 + 2 * cp + c < Cin ? kF16XScale : 0.f;
      }
      const bool edge = tb - 6 < 0 || tb + w + 6 > T;     // wave-uniform
      uint32_t* d32 = reinterpret_cast<uint32_t*>(dst) + ((2 * cp) >> 3) * (XS * 4) + (((2 * cp) & 7) >> 1);
      // pairs (v[2m], v[2m+1]) of the activated 2x signal, m = tb - 3 + pidx: NPS passes of 64 lanes, two channels, even and odd
      // sample -- 4 NPS independent dependency chains in ONE basic block
      auto pairs = [&](auto NPS_) __attribute__((always_inline)) {
        constexpr int NPS = decltype(NPS_)::value;
        float u[4 * NPS], av[4 * NPS], bv[4 * NPS];
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
          const int pidx = ps * 64 + lane;
          const int m = tb - 3 + pidx;
          int idx = pidx + 3;                             // xs index of x[m]
          if (edge) idx = min(max(min(max(m, 0), T - 1) - tb + 6, 3), kXRow - 4);
          const float* xp0 = xs0 + idx;
          const float* xp1 = xs1 + idx;
          float ue0 = 0.f, uo0 = 0.f, ue1 = 0.f, uo1 = 0.f;
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            ue0 = fmaf(sa.tu.f[2 * j + 1], xp0[2 - j], ue0);
            uo0 = fmaf(sa.tu.f[2 * j], xp0[3 - j], uo0);
            ue1 = fmaf(sa.tu.f[2 * j + 1], xp1[2 - j], ue1);
            uo1 = fmaf(sa.tu.f[2 * j], xp1[3 - j], uo1);
          }
          u[4 * ps] = ue0; u[4 * ps + 1] = uo0; u[4 * ps + 2] = ue1; u[4 * ps + 3] = uo1;
          av[4 * ps] = av[4 * ps + 1] = al[0]; av[4 * ps + 2] = av[4 * ps + 3] = al[1];
          bv[4 * ps] = bv[4 * ps + 1] = ib[0]; bv[4 * ps + 2] = bv[4 * ps + 3] = ib[1];
        }
        snake_n(u, av, bv);
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
          const int pidx = ps * 64 + lane;
          const int m = tb - 3 + pidx;
          float ve0 = u[4 * ps], vo0 = u[4 * ps + 1], ve1 = u[4 * ps + 2], vo1 = u[4 * ps + 3];
          if (edge) {
            if (m < 0) { vo0 = ve0; vo1 = ve1; }          // replicate pad of the 2x signal: v[0] on the left ...
            if (m > T - 1) { ve0 = vo0; ve1 = vo1; }      // ... v[2T-1] on the right
          }
          vs0[pidx] = make_float2(ve0, vo0);
          vs1[pidx] = make_float2(ve1, vo1);
        }
      };
      // outputs: column j0 + o, input time tb + o
      auto outs = [&](auto NPS_) __attribute__((always_inline)) {
        constexpr int NPS = decltype(NPS_)::value;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
          const int o = ps * 64 + lane;
          float y[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const float2* vs = c ? vs1 : vs0;
            float2 pv[7];
            lds_read7_b64(vs + o, pv);
            const float2 p0 = pv[0], p1 = pv[1], p2 = pv[2], p3 = pv[3], p4 = pv[4], p5 = pv[5], p6 = pv[6];
            float s = sa.td.f[0] * p0.y;
            s = fmaf(sa.td.f[1], p1.x, s);
            s = fmaf(sa.td.f[2], p1.y, s);
            s = fmaf(sa.td.f[3], p2.x, s);
            s = fmaf(sa.td.f[4], p2.y, s);
            s = fmaf(sa.td.f[5], p3.x, s);
            s = fmaf(sa.td.f[6], p3.y, s);
            s = fmaf(sa.td.f[7], p4.x, s);
            s = fmaf(sa.td.f[8], p4.y, s);
            s = fmaf(sa.td.f[9], p5.x, s);
            s = fmaf(sa.td.f[10], p5.y, s);
            s = fmaf(sa.td.f[11], p6.x, s);
            y[c] = s;
          }
          const int t = tb + o;
          const bool ok = t >= 0 && t < T;                // the convolution's zero padding
          {
            // the operand split of conv_bf16_kernel<NP = 2>::store_x: the scaled input is ROUNDED to fp32, then split (no contraction)
#pragma clang fp contract(off)
            const float v0 = (ok ? y[0] : 0.f) * sc[0], v1 = (ok ? y[1] : 0.f) * sc[1];
            const f16x2 hi = __builtin_convertvector((f32x2){v0, v1}, f16x2);
            const f16x2 lo = __builtin_convertvector((f32x2){(v0 - (float)hi[0]) * kF16LoScale, (v1 - (float)hi[1]) * kF16LoScale}, f16x2);
            if (o < w) {
              d32[(j0 + o) * 4] = __builtin_bit_cast(uint32_t, hi);
              d32[PSZ * 4 + (j0 + o) * 4] = __builtin_bit_cast(uint32_t, lo);
            }
          }
        }
      };
      if (w + 6 > 64) pairs(std::integral_constant<int, 2>{}); else pairs(std::integral_constant<int, 1>{});
      if (w > 64) outs(std::integral_constant<int, 2>{}); else outs(std::integral_constant<int, 1>{});
    }
    __syncthreads();                                      // chunk ck is staged (and chunk ck - 1 has been consumed)
  }
}

// ---- launch ---------------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int NT, int NPROD, int HALO, bool ACT>
static int launch_cs(const KArgs& ka, const SnakeArgs& sa, int B, hipStream_t st) {
  constexpr int BN = WN * NT * 32, XS = BN + HALO;
  constexpr int IPP = (8 * ((XS + kSegW - 1) / kSegW) + NPROD - 1) / NPROD;
  constexpr size_t lds = (size_t)2 * 2 * 2 * XS * 16 + (ACT ? (size_t)NPROD * ((IPP * 2 + 4) * kXRow) * 4 : 0);
  static_assert(lds <= 160 * 1024, "fused tile exceeds the LDS of a CU");
  auto kern = conv_snake_kernel<WM, WN, NT, NPROD, HALO, ACT>;
  if (lds > 64 * 1024) {
    static int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DMEL_HIP((hipError_t)rc);
  }
  const int gx = (int)((ka.Tcols + BN - 1) / BN), gy = (ka.mtiles + WM - 1) / WM;
  if (gy > 65535 || B > 65535) { set_error("conv_snake: grid too large"); return DMEL_EINVAL; }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy, (unsigned)B), dim3(64 * (WM * WN + NPROD)), lds, st, ka, sa);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

template <int WM, int WN, int NT, int NPROD, bool ACT>
static int launch_cs_h(const KArgs& ka, const SnakeArgs& sa, int B, int halo, hipStream_t st) {
  if (halo <= 16) return launch_cs<WM, WN, NT, NPROD, 16, ACT>(ka, sa, B, st);
  return launch_cs<WM, WN, NT, NPROD, 64, ACT>(ka, sa, B, st);
}

bool conv_snake_eligible(const PackedConv& pc, const ConvRun& r) {
  const PackDesc& d = pc.d;
  return d.mode == EPI_LINEAR && d.nseg == 1 && d.phases == 1 && d.seg[0].tstride == 1 && d.seg[0].toff == 0 && r.seg[0].tshift == 0 &&
         (d.seg[0].taps - 1) * d.seg[0].dil <= 64 && r.seg[0].in_len == nullptr && r.out_len == nullptr && r.seg[0].in_scale == 1.f &&
         r.seg[0].in_absmax == nullptr && r.act == ACT_NONE && r.row_scale == nullptr && r.out_tstride == 1 && r.phase_base == 0 &&
         r.fold_pitch == 0 && r.precision == DMEL_PRECISION_FP32_F16X2 && train_precision_override() != DMEL_PRECISION_BF16 &&
         r.seg[0].Tin == r.Tcols && r.seg[0].Tin < ((int64_t)1 << 30);
}

// y = conv(snake(x)) (+ bias, residual, running sum, 1 / out_div: ConvRun's LINEAR epilogue), x (B, Cin, T) BEFORE the activation
int launch_conv_snake(const PackedConv& pc, const ConvRun& r, const float* alpha, const float* beta, const float* up_taps_host,
                      const float* down_taps_host, int logscale, hipStream_t stream) {
  DMEL_CHECK_ARG(conv_snake_eligible(pc, r), "conv_snake: this convolution cannot take the fused activation (plain 'same' convolution, "
                                             "fp16-split precision, no masks)");
  DMEL_CHECK_ARG(r.seg[0].x && r.y && r.B > 0 && r.Tcols > 0 && (alpha == nullptr || (up_taps_host && down_taps_host)), "conv_snake: NULL argument / bad shape");
  const PackDesc& d = pc.d;
  const SegDesc& sd = d.seg[0];
  KArgs ka{};
  ka.nseg = 1;
  ka.steps = pc.steps;
  SegArgs& o = ka.seg[0];
  o.x = r.seg[0].x; o.bstride = r.seg[0].bstride; o.cstride = r.seg[0].cstride; o.Tin = r.seg[0].Tin;
  o.in_len = nullptr; o.in_scale = 1.f; o.in_absmax = nullptr;
  o.Cin = sd.Cin; o.nchunk = (sd.Cin + kCK - 1) / kCK; o.taps = sd.taps; o.dil = sd.dil; o.pad_left = sd.pad_left; o.tstride = 1; o.toff = 0;
  DMEL_CHECK_ARG((int64_t)sd.Cin * o.cstride < ((int64_t)1 << 30), "conv_snake: one batch item of the input exceeds 32-bit offsets");
  ka.w32h = pc.w32h.p; ka.bias = pc.bias.as<float>();
  ka.Tcols = r.Tcols; ka.mode = d.mode; ka.act = ACT_NONE; ka.C = d.C; ka.RP = pc.RP; ka.phases = 1;
  ka.out_tstride = 1; ka.phase_base = 0; ka.accumulate = r.accumulate; ka.len_div = 1; ka.out_div = r.out_div;
  ka.y = r.y; ka.y_bs = r.y_bs; ka.y_cs = r.y_cs; ka.Tout = r.Tout > 0 ? r.Tout : r.Tcols;
  ka.res = r.res; ka.res_bs = r.res_bs; ka.res_cs = r.res_cs;
  DMEL_CHECK_ARG((int64_t)d.C * ka.y_cs < ((int64_t)1 << 31) && ka.Tout < ((int64_t)1 << 30), "conv_snake: one batch item of the output exceeds 32-bit offsets");
  ka.mtiles = pc.Mpad / 32;
  SnakeArgs sa;
  sa.alpha = alpha; sa.beta = beta ? beta : alpha; sa.logscale = logscale;
  for (int i = 0; i < 12; ++i) { sa.tu.f[i] = alpha ? 2.f * up_taps_host[i] : 0.f; sa.td.f[i] = alpha ? down_taps_host[i] : 0.f; }
  const int halo = (sd.taps - 1) * sd.dil;
  const double in_elems = (double)sd.Cin * (double)r.seg[0].Tin;
  const double out_elems = (double)d.C * (double)r.Tcols * (1.0 + (r.res ? 1.0 : 0.0) + (r.accumulate ? 1.0 : 0.0));
  const double alg_bytes = 4.0 * r.B * (in_elems + out_elems) + (double)pc.Mpad * pc.steps * kCK * 4.0;
  const double alg_flops = 2.0 * r.B * (double)r.Tcols * (double)d.C * pc.k_real;
  ProfScope ps("conv_igemm", stream, alg_flops, alg_bytes, alg_flops * 3.0);
  const int mt = ka.mtiles;
  if (alpha == nullptr) {
    // plain convolution: EIGHT consumer waves in every configuration (two per SIMD hide each other's waits; with four the lean loop
    // measured 0.55-0.75x of conv_bf16_kernel, with eight 1.10-1.19x) and four producers for the conversion
    static const int cfg = [] { const char* e = getenv("DMEL_LEAN_CFG"); return e ? atoi(e) : 0; }();     // A/B: 1 = sixteen-wave workgroups
    if (mt >= 5) return launch_cs_h<8, 1, 3, 4, false>(ka, sa, r.B, halo, stream);
    if (cfg == 1) {      // 8 consumers of 32 x 64 + 8 producers, <= 128 registers, four waves per SIMD
      if (mt >= 3) return launch_cs_h<4, 2, 2, 8, false>(ka, sa, r.B, halo, stream);
      if (mt == 2) return launch_cs_h<2, 4, 2, 8, false>(ka, sa, r.B, halo, stream);
      return launch_cs_h<1, 8, 2, 8, false>(ka, sa, r.B, halo, stream);
    }
    if (mt >= 3) return launch_cs_h<4, 2, 3, 4, false>(ka, sa, r.B, halo, stream);
    if (mt == 2) return launch_cs_h<2, 4, 3, 4, false>(ka, sa, r.B, halo, stream);
    return launch_cs_h<1, 8, 3, 4, false>(ka, sa, r.B, halo, stream);
  }
  // one 32-row strip per consumer wave; fewer rows -> the consumers spread over the columns instead (BN = 96 WN), so that every
  // configuration has four (eight) matrix waves, one (two) per SIMD, next to eight (four) producers
  if (mt >= 5) return launch_cs_h<8, 1, 3, 4, true>(ka, sa, r.B, halo, stream);
  if (mt >= 3) return launch_cs_h<4, 1, 3, 8, true>(ka, sa, r.B, halo, stream);
  if (mt == 2) return launch_cs_h<2, 2, 3, 8, true>(ka, sa, r.B, halo, stream);
  return launch_cs_h<1, 3, 3, 8, true>(ka, sa, r.B, halo, stream);      // 288 columns: three full segments for the eight producers
}

}  // namespace dmel

// ---- single-op C ABI ----------------------------------------------------------------------------------------------------------------
extern "C" int dmel_conv_snake_forward(const dmel_conv* c, const float* x, const float* residual, float* y, const float* alpha,
                                       const float* beta, const float* up_filter12_host, const float* down_filter12_host, int logscale,
                                       int B, int64_t T, void* stream) {
  using namespace dmel;
  DMEL_CHECK_ARG(c && x && y && (alpha == nullptr || (up_filter12_host && down_filter12_host)), "conv_snake_forward: NULL argument");
  ConvRun r;
  r.seg[0].x = x; r.seg[0].bstride = (int64_t)c->Cin * T; r.seg[0].cstride = T; r.seg[0].Tin = T;
  r.B = B; r.Tcols = T; r.y = y; r.y_bs = (int64_t)c->Cout * T; r.y_cs = T; r.Tout = T;
  r.res = residual; r.res_bs = (int64_t)c->Cout * T; r.res_cs = T;
  r.precision = DMEL_PRECISION_FP32_F16X2;
  return launch_conv_snake(c->pc, r, alpha, beta, up_filter12_host, down_filter12_host, logscale, (hipStream_t)stream);
}
