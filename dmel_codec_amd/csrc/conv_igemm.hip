// Implicit-GEMM 1-D convolution on the fp32-input MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// One kernel serves every dense conv of the path: WaveNet dilated k3 convs with the 1x1 condition
// projection concatenated along K and the sigmoid*tanh gate in the epilogue; the 1x1 output projection with
// the residual / skip update in the epilogue; BigVGAN's k in {3,7,11}, dilation in {1,3,5} convs with the
// residual add and the 1/3 branch average in the epilogue; transposed convs as phase-major sub-convolutions;
// the ConvNeXt pointwise Linears; conv_pre / conv_post.
//
// GEMM view per batch item:  D[m, t] = sum_{ci, tap} W[m, ci, tap] * x[ci, t + tap*dil - pad]
//   M = output rows (packed, see conv.h), N = time, K = Cin * taps, walked in steps of (16 channels, 1 tap).
// Layout: activations stay (B, C, T) with T contiguous, so a K-step's B operand is 16 rows of the staged
// x tile and the tap is just a column shift inside LDS (x is read from HBM once per 16-channel chunk, not
// once per tap).  Weights are pre-tiled on the host to [m-block][step][16][BM], so a step's A tile is one
// contiguous 16*BM*4-byte run (float4, fully coalesced) and both LDS fragment reads are 32 consecutive
// dwords per half-wave: bank-conflict free for any tap shift.
// Numerics: exact fp32 fma chains in K order (MI355X_MICROARCH: f32 MFMA == fmaf chain), so results differ
// from ATen only by summation order.
#include "conv.h"
#include "conv_dev.h"
#include "ops.h"
#include <type_traits>

#include <cmath>
#include <cstdlib>

namespace dmel {


// HALO = 64: any (taps-1)*dilation <= 64.  HALO = 16: receptive fields up to 17 samples (all dilation-1 convs).
// HALO = 0: every segment is 1-tap (pointwise convs, Linear layers).  Smaller halos mean less staging work and LDS
// per chunk, which buys the short kernels a wave of occupancy.
template <int WAVES_M, int WAVES_N, int MT, int NT, int MODE, int HALO>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_igemm_kernel(KArgs a) {
  constexpr int BM = WAVES_M * MT * 32, BN = WAVES_N * NT * 32, NW = WAVES_M * WAVES_N;
  constexpr int XS = BN + HALO;                    // staged x row: BN columns + halo
  constexpr int RW = (kCK + NW - 1) / NW;          // x rows staged per wave
  constexpr int XC = (XS + 63) / 64;               // 64-column groups per x row
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                  // [2][kCK][XS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int h = lane >> 5, l31 = lane & 31;
  int tile_n, mblk, b;
  if (!conv_block_coords(a, tile_n, mblk, b)) return;   // whole workgroup exits together (padding of the 1-D grid)
  const int q0 = tile_n * BN;
  const int lb = b / a.len_div;

  floatx16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // A fragments come straight from global memory in lane order (conv.h, pack_conv): per 32-row tile and K step two
  // coalesced float4 loads per lane, prefetched one step ahead into registers.  Tiles past the end (M not a multiple
  // of BM) are clamped: the wave then recomputes the last tile and its epilogue is masked off.
  const float4* wA[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int tile = min(mblk * (BM / 32) + wave_m * MT + mi, a.mtiles - 1);
    wA[mi] = reinterpret_cast<const float4*>(a.w) + (size_t)tile * a.steps * 128 + lane;
  }
  float4 ac0[MT], ac1[MT], an0[MT], an1[MT];
  float xr[RW][XC];

  // x tile: wave w stages rows w, w+NW, ...; a lane covers columns lane, lane+64, ...  All offsets inside one
  // batch item are 32-bit (host checks C*T < 2^31).  Loads are UNCONDITIONAL on a clamped address (so they issue
  // back to back with one wait at the LDS store); out-of-range elements are zeroed by a select afterwards.
  auto load_x = [&](int sg, int chunk) {
    const float* xb = a.seg[sg].x + (int64_t)b * a.seg[sg].bstride;
    const int taps = a.seg[sg].taps, dil = a.seg[sg].dil, tstride = a.seg[sg].tstride, Cin = a.seg[sg].Cin;
    const int wx = BN + (taps - 1) * dil;
    const int tin = (int)a.seg[sg].Tin;
    const int lim = a.seg[sg].in_len ? min((int)min(a.seg[sg].in_len[lb], (int64_t)tin), tin) : tin;
    const int tau0 = q0 * tstride + a.seg[sg].toff - a.seg[sg].pad_left;
    const int cs = (int)a.seg[sg].cstride;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int r = wave + rr * NW;
      const int ci = chunk * kCK + r;
      const bool rowok = (r < kCK) && (ci < Cin);
      const float* xrow = xb + min(ci, Cin - 1) * cs;
#pragma unroll
      for (int c = 0; c < XC; ++c) {
        const int j = lane + 64 * c;
        const int tau = tau0 + j * tstride;
        const bool ok = rowok && (j < wx) && (tau >= 0) && (tau < lim);
        const float v = xrow[min(max(tau, 0), tin - 1)];
        xr[rr][c] = ok ? v : 0.f;
      }
    }
  };
  auto store_x = [&](float* dst, float scale) {
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int r = wave + rr * NW;
#pragma unroll
      for (int c = 0; c < XC; ++c) {
        const int j = lane + 64 * c;
        if ((kCK % NW == 0 || r < kCK) && (XS % 64 == 0 || j < XS)) dst[r * XS + j] = xr[rr][c] * scale;
      }
    }
  };

  int sg = 0, chunk = 0, tap = 0, xbuf = 0;
  load_x(0, 0);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) { ac0[mi] = wA[mi][0]; ac1[mi] = wA[mi][64]; }
  store_x(Xs, a.seg[0].in_scale);
  __syncthreads();

  for (int s = 0; s < a.steps; ++s) {
    int nsg = sg, nchunk = chunk, ntap = tap + 1;
    bool newx = false;
    if (ntap == a.seg[sg].taps) {
      ntap = 0;
      ++nchunk;
      newx = true;
      if (nchunk == a.seg[sg].nchunk) { nchunk = 0; ++nsg; }
    }
    const bool has_next = s + 1 < a.steps;
    if (has_next) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        an0[mi] = wA[mi][(size_t)(s + 1) * 128];
        an1[mi] = wA[mi][(size_t)(s + 1) * 128 + 64];
      }
      if (newx) load_x(nsg, nchunk);
    }
    {
      const float* xp = Xs + xbuf * (kCK * XS) + wave_n * (NT * 32) + l31 + tap * a.seg[sg].dil + h * XS;
#pragma unroll
      for (int kk = 0; kk < kCK / 2; ++kk) {
        float av[MT], bv[NT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const float4 q = (kk < 4) ? ac0[mi] : ac1[mi];
          av[mi] = (kk & 3) == 0 ? q.x : (kk & 3) == 1 ? q.y : (kk & 3) == 2 ? q.z : q.w;
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) bv[ni] = xp[2 * kk * XS + ni * 32];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
      }
    }
    if (has_next) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) { ac0[mi] = an0[mi]; ac1[mi] = an1[mi]; }
      if (newx) {   // the only synchronisation: once per 16-channel chunk, when the other x buffer is published
        store_x(Xs + (xbuf ^ 1) * (kCK * XS), a.seg[nsg].in_scale);
        __syncthreads();
        xbuf ^= 1;
      }
    }
    sg = nsg; chunk = nchunk; tap = ntap;
  }

  conv_epilogue<MT, NT, MODE>(a, acc, mblk * BM + wave_m * (MT * 32), q0 + wave_n * (NT * 32) + l31, b, lb, h);
}

// -----------------------------------------------------------------------------------------------------------
// Register-direct variant: no LDS, no barriers.  Every wave streams BOTH operands of its (32 MT) x (32 NT) tile from
// global memory / L1 into double-buffered registers: A as in the kernel above, B one dword per lane and MFMA
// (lanes 0-31: 32 consecutive time steps of channel k, lanes 32-63: of channel k+1 -- two 128-byte segments per
// load).  A tap is a shifted re-read that hits L1.  Per K step a (1,1) wave issues 10 loads, 8 selects and ~15
// integer instructions for its 8 MFMAs, against ~340 instructions in the LDS-staged kernel, whose x-tile staging and
// barrier dominate when a wave owns few MFMAs per step: small-N WaveNet GEMMs, 32- and 64-channel BigVGAN stages.
// Operand traffic at full MFMA rate is 16-32 B/clk/CU, within the 64 B/clk of L1.
template <int WAVES_M, int WAVES_N, int MT, int NT, int MODE>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_direct_kernel(KArgs a) {
  constexpr int BM = WAVES_M * MT * 32, BN = WAVES_N * NT * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int h = lane >> 5, l31 = lane & 31;
  int tile_n, mblk, b;
  if (!conv_block_coords(a, tile_n, mblk, b)) return;   // whole workgroup exits together (padding of the 1-D grid)
  const int q0 = tile_n * BN;
  const int lb = b / a.len_div;

  floatx16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const float4* wA[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int tile = min(mblk * (BM / 32) + wave_m * MT + mi, a.mtiles - 1);
    wA[mi] = reinterpret_cast<const float4*>(a.w) + (size_t)tile * a.steps * 128 + lane;
  }
  const int qcol = q0 + wave_n * (NT * 32) + l31;   // this lane's column of n-tile 0

  float4 ac0[MT], ac1[MT], an0[MT], an1[MT];
  float bc[NT][8], bn[NT][8];
  int sbase = 0;   // global K-step index of the segment's first step

  for (int sg = 0; sg < a.nseg; ++sg) {
    const SegArgs& sa = a.seg[sg];
    const float* xb = sa.x + (int64_t)b * sa.bstride;
    const int taps = sa.taps, dil = sa.dil, cin = sa.Cin, cs = (int)sa.cstride, tin = (int)sa.Tin;
    const int lim = sa.in_len ? min((int)min(sa.in_len[lb], (int64_t)tin), tin) : tin;
    const float scale = sa.in_scale;
    const int nst = sa.nchunk * taps;
    int tcol[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) tcol[ni] = (qcol + ni * 32) * sa.tstride + sa.toff - sa.pad_left;
    int rowoff[8];

    // loads of local step (chunk, tap) into (A0, A1, B)
#define DMEL_DIRECT_LOAD(A0, A1, B, chunk, tap, gstep)                                                  \
    {                                                                                                   \
      _Pragma("unroll") for (int mi = 0; mi < MT; ++mi) {                                               \
        A0[mi] = wA[mi][(size_t)(gstep) * 128];                                                         \
        A1[mi] = wA[mi][(size_t)(gstep) * 128 + 64];                                                    \
      }                                                                                                 \
      if ((tap) == 0) {                                                                                 \
        _Pragma("unroll") for (int kk = 0; kk < 8; ++kk)                                                \
          rowoff[kk] = min((chunk) * kCK + 2 * kk + h, cin - 1) * cs;                                   \
      }                                                                                                 \
      _Pragma("unroll") for (int ni = 0; ni < NT; ++ni) {                                               \
        const int tau = tcol[ni] + (tap) * dil;                                                         \
        const bool ok = (tau >= 0) && (tau < lim);                                                      \
        const int tc = min(max(tau, 0), tin - 1);                                                       \
        _Pragma("unroll") for (int kk = 0; kk < 8; ++kk) {                                              \
          const float v = xb[rowoff[kk] + tc];                                                          \
          B[ni][kk] = ok ? v * scale : 0.f;                                                             \
        }                                                                                               \
      }                                                                                                 \
    }

    int chunk = 0, tap = 0;
    DMEL_DIRECT_LOAD(ac0, ac1, bc, 0, 0, sbase)
    for (int ls = 0; ls < nst; ++ls) {
      int ntap = tap + 1, nchunk = chunk;
      if (ntap == taps) { ntap = 0; ++nchunk; }
      const bool has_next = ls + 1 < nst;
      if (has_next) DMEL_DIRECT_LOAD(an0, an1, bn, nchunk, ntap, sbase + ls + 1)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float av[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const float4 q = (kk < 4) ? ac0[mi] : ac1[mi];
          av[mi] = (kk & 3) == 0 ? q.x : (kk & 3) == 1 ? q.y : (kk & 3) == 2 ? q.z : q.w;
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bc[ni][kk], acc[mi][ni], 0, 0, 0);
      }
      if (has_next) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) { ac0[mi] = an0[mi]; ac1[mi] = an1[mi]; }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) bc[ni][kk] = bn[ni][kk];
      }
      chunk = nchunk; tap = ntap;
    }
#undef DMEL_DIRECT_LOAD
    sbase += nst;
  }
  conv_epilogue<MT, NT, MODE>(a, acc, mblk * BM + wave_m * (MT * 32), qcol, b, lb, h);
}

// Fills the logical grid into the kernel arguments and returns the launch grid.  The XCD-chunked 1-D form is used
// when the weights of the launch do not fit one XCD's 4 MiB L2 next to the activations (large-M GEMMs: the WaveNet
// decoder); small weight sets are L2-resident on every XCD anyway and keep the plain 3-D grid.
static int conv_grid(KArgs& k, int gx, int gy, int gz, dim3& grid) {
  static int mode = [] { const char* e = getenv("DMEL_CONV_XCD"); return e ? atoi(e) : -1; }();   // -1 auto, 0 off, 1 on
  k.gx = gx; k.gy = gy; k.gz = gz;
  const double weight_bytes = (double)k.mtiles * 32.0 * k.steps * kCK * 4.0;
  const bool want = mode == 1 || (mode == -1 && weight_bytes > 2.5e6 && gy >= 8);
  const int64_t total = (int64_t)gx * gy * gz;
  if (want && total < ((int64_t)1 << 30)) {
    k.xcd_chunk = (int)((total + 7) / 8);
    grid = dim3((unsigned)(k.xcd_chunk * 8));
    return DMEL_OK;
  }
  k.xcd_chunk = 0;
  if (gy > 65535 || gz > 65535) {
    set_error("conv_igemm: grid too large (mblocks %d, batch %d)", gy, gz);
    return DMEL_EINVAL;
  }
  grid = dim3((unsigned)gx, (unsigned)gy, (unsigned)gz);
  return DMEL_OK;
}

template <int WM, int WN, int MT, int NT, int MODE> static int launch_d(const KArgs& ka, int B, int mblocks, hipStream_t st) {
  constexpr int BN = WN * NT * 32;
  KArgs k2 = ka;
  dim3 grid;
  DMEL_TRY(conv_grid(k2, (int)((ka.Tcols + BN - 1) / BN), mblocks, B, grid));
  hipLaunchKernelGGL((conv_direct_kernel<WM, WN, MT, NT, MODE>), grid, dim3(64 * WM * WN), 0, st, k2);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

template <int WM, int WN, int MT, int NT, int MODE, int HALO>
static int launch_h(const KArgs& ka, int B, int mblocks, hipStream_t st) {
  constexpr int BN = WN * NT * 32;
  const size_t lds = (size_t)(2 * kCK * (BN + HALO)) * sizeof(float);
  KArgs k2 = ka;
  dim3 grid;
  DMEL_TRY(conv_grid(k2, (int)((ka.Tcols + BN - 1) / BN), mblocks, B, grid));
  hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MT, NT, MODE, HALO>), grid, dim3(64 * WM * WN), lds, st, k2);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

template <int WM, int WN, int MT, int NT, int MODE> static int launch_t(const KArgs& ka, int B, int mblocks, hipStream_t st) {
  int halo = 0;
  for (int s = 0; s < ka.nseg; ++s) halo = std::max(halo, (ka.seg[s].taps - 1) * ka.seg[s].dil);
  if (halo == 0) return launch_h<WM, WN, MT, NT, MODE, 0>(ka, B, mblocks, st);
  // dilation-1 convs (every second conv of an AMP block, k2 transposed-conv phases): a 16-column halo is enough
  if (halo <= 16) return launch_h<WM, WN, MT, NT, MODE, 16>(ka, B, mblocks, st);
  return launch_h<WM, WN, MT, NT, MODE, 64>(ka, B, mblocks, st);
}

// -----------------------------------------------------------------------------------------------------------
// bf16 matrix-core kernels: v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 tensors in HBM on both sides.
//
//   NP = 3  "split fp32" (DMEL_PRECISION_FP32, the default): every fp32 operand is cut into three bf16 pieces by
//           truncation, a = a1 + a2 + a3 EXACTLY (3 x 8 significant bits = the 24 of fp32; a1 = a & 0xffff0000,
//           a2 = (a - a1) & 0xffff0000, a3 = a - a1 - a2, every subtraction exact), and a product is formed from the six
//           partial products a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1, each exact in the fp32 accumulator's input.  The
//           dropped terms a2b3 + a3b2 + a3b3 are below 2^-21 |ab| (worst case; ~2^-23 typical), i.e. at the level of the
//           fp32 rounding of the product itself, and far below the accumulation rounding of a K ~ 10^3 reduction.  Six
//           32-cycle bf16 MFMAs per 32x32x16 block replace eight 64-cycle fp32 MFMAs: 2.67x less matrix-core time for the
//           same fp32-grade result (tests: same error against the fp64 oracle as the fp32 MFMA kernel).
//   NP = 2  "fp16 split" (DMEL_PRECISION_FP32_F16X2; the vocoder's default): every fp32 operand as TWO fp16 pieces, a = a_hi + 2^-11 a_lo
//           with a_hi = fp16(a) and a_lo = fp16((a - a_hi) * 2^11), both round-to-nearest: |a - (a_hi + 2^-11 a_lo)| <= 2^-22 |a| (22 significant bits; fp32 has 24), four times the unit
//           roundoff of fp32.  THREE partial products per block (v_mfma_f32_32x32x16_f16): a_hi b_hi into `acc`, a_hi b_lo + a_lo b_hi
//           into a second accumulator `acl` that is folded in once, acc + 2^-11 acl, before the epilogue; the dropped a_lo b_lo is
//           below 2^-22 |ab|.  Same measured error against fp64 as NP = 3 and as an fp32 fma chain, half the matrix-core time of NP = 3.
//           The second piece is kept SCALED so that it stays a normal fp16 number wherever the first one is; activations are staged
//           x 2^-6 and the weight image carries 2^6 (exact), which centres the fp16 range on audio-network magnitudes: |x| < 4.19e6,
//           absolute error 2^-30 below |x| = 2^-8.  Not used for gradients (too many binades) nor for the encoder (ids are defined
//           by NP = 3).
//   NP = 1  "bf16 operands" (DMEL_PRECISION_BF16, opt-in): operands rounded to bf16 (RNE), one MFMA per block.  This is the
//           arithmetic the reference's LM configs ask for when they run the codec under dtype: bfloat16
//           (config/lm/lm_config.yaml:1,83); it is outside the 1e-4 bar and never the default.
//
// x is staged KG*8 channels at a time and kept in LDS as [piece][8-channel group][column][8 x bf16]: the B fragment of any
// tap is one conflict-free ds_read_b128 per piece, the staging write one ds_write_b128 per (piece, group, column) item.
// The weight pieces are pre-split on the host and stored in A-fragment order (one coalesced dwordx4 per lane, piece and
// step, straight from global/L2).  The x loads for chunk c+1 are issued at the first step of chunk c and land in LDS at its
// last step, so their latency hides behind a whole chunk of MFMAs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#ifndef DMEL_PD
#define DMEL_PD 2   // weight prefetch distance in K steps
#endif
#ifndef DMEL_EXP
#define DMEL_EXP 0   // what-if switches for profiling builds (wrong results): 2 no x staging, 4 no weight loads, 8 no epilogue
#endif
#ifndef DMEL_PD2
#define DMEL_PD2 2  // ... of the fp16-split kernel (half the MFMA time per step)
#endif
#ifndef DMEL_XTOP
// 1: issue the next chunk's x loads at the TOP of a chunk's first step, in front of that step's weight prefetch (two steps of MFMAs between
// issue and use instead of one).  Round-3 experiment, A/B'd interleaved on one device against 0 (profiles/r03_conv_experiments.txt): no
// shape gained (-4 .. +2 %), with or without a weight prefetch distance of 3 -- the x loads are not what the waves wait for.  Stays 0.
#define DMEL_XTOP 0
#endif
#ifndef DMEL_BPF
// 1: the B fragments of step s + 1 are read from LDS BEFORE the MFMAs of step s (two fragment register sets), whenever the next step
// belongs to the same staged chunk.  A lone wave on its SIMD (the decoder WaveNet at 32 x 92 frames: 288 workgroups for 256 CUs) otherwise
// meets the LDS latency at the top of every step with nothing else to issue.
#define DMEL_BPF 0
#endif
#ifndef DMEL_KG2
#define DMEL_KG2 2  // 8-channel groups staged per barrier by the fp16-split kernel when the convolution has taps
#endif

__device__ __forceinline__ uint32_t pack_hi16(float lo, float hi) {      // {bf16 bits of lo, bf16 bits of hi} by truncation
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}

// (amdgpu_waves_per_eu(2, 2) -- 166 VGPRs, accumulators out of the AGPRs -- was measured: 2-12 % slower on every bench shape, and the
// allocator still parks one weight set on the B-fragment registers; the default register budget stays.)
template <int WAVES_M, int WAVES_N, int MT, int NT, int MODE, int HALO, int NP, int KG, int PS = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, NP == 2 ? 2 : 1) void conv_bf16_kernel(KArgs a) {
  static_assert(PS == 0 || NP == 2, "pre-split inputs exist for the fp16 split only");
  constexpr int BM = WAVES_M * MT * 32, BN = WAVES_N * NT * 32;
  constexpr int XS = BN + HALO;
  constexpr int SUB = KG / 2;                            // 16-channel K steps (per tap) per staged chunk
  // Staging map: an item is (8-channel group kg, column j) -- eight channels of one staged column, i.e. one lane's B operand.  The group is
  // WAVE-UNIFORM (a wave owns one group, or a segment of one, or a few whole groups), so the eight row addresses are scalar registers and
  // a load is global_load_dword v, v_column_offset, s[row_base]: no vector address arithmetic per element.
  constexpr int NWV = WAVES_M * WAVES_N;
  static_assert(NWV % KG == 0 || KG % NWV == 0, "waves and staged channel groups must divide one another");
  constexpr int SEG = NWV >= KG ? NWV / KG : 1;          // waves sharing one group's row of columns
  constexpr int RPW = NWV >= KG ? 1 : KG / NWV;          // groups per wave
  constexpr int SL = (XS + SEG - 1) / SEG;               // columns per wave and group
  constexpr int NPASS = (SL + 63) / 64;
  constexpr int NIT = RPW * NPASS;                       // items per thread per chunk
  constexpr int PSZ = KG * XS;                           // uint4 per piece
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifndef DMEL_CONV_LDS_PAD
#define DMEL_CONV_LDS_PAD 0      // debug: bytes of unused LDS in front of and behind the kernel's own
#endif
  constexpr int kLdsPad = ((size_t)2 * NP * KG * XS * 16 + 2 * DMEL_CONV_LDS_PAD <= 65536) ? DMEL_CONV_LDS_PAD : 0;
  uint4* Xb = reinterpret_cast<uint4*>(smem) + kLdsPad / 16;            // [2][NP][KG][XS] x 16 bytes

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int h = lane >> 5, l31 = lane & 31;
  int tile_n, mblk, b;
  if (!conv_block_coords(a, tile_n, mblk, b)) return;
  const int q0 = tile_n * BN;
  const int lb = b / a.len_div;
  // (Round 2, fp16-split kernel: waves in odd SIMD slots sleeping 256-2048 cycles before the K loop -- to put the two workgroups of a CU out of
  // phase -- changed nothing either: 0 +- 3 % on every shape.)
  // (Delaying the workgroups that land in odd wave slots -- so that a CU's two workgroups alternate between their MFMA loop and their
  // HBM-bound epilogue instead of doing both in phase -- was measured with 20 k / 50 k / 100 k cycle delays: 3-40 % slower on every
  // shape when applied to every such workgroup, no gain beyond the +-10 % run-to-run noise when applied to the first round only.
  // The k = 3 layers' time is still close to MFMA time + epilogue traffic time, i.e. the two do not overlap.)

  floatx16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  // NP = 2: the cross terms a_hi b_lo + a_lo b_hi, 2^11 times their true size
  floatx16 acl[NP == 2 ? MT : 1][NP == 2 ? NT : 1];
  if constexpr (NP == 2) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acl[mi][ni][r] = 0.f;
  }

  // Weight fragments: wave-uniform (scalar) base per row tile + a 16-byte lane offset, so a step's loads are
  // global_load_dwordx4 v, v_lane, s[base] offset:piece*1024 with one scalar add per step and no vector address math.
  const char* wT[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int tile = __builtin_amdgcn_readfirstlane(min(mblk * (BM / 32) + wave_m * MT + mi, a.mtiles - 1));
    wT[mi] = reinterpret_cast<const char*>(NP == 3 ? a.w48 : NP == 2 ? a.w32h : a.w16) + (size_t)tile * a.steps * (NP * 1024);
  }
  const uint32_t lane16 = lane * 16;
  auto load_w = [&](uint4 (&dst)[MT][NP], int step) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const char* sp = wT[mi] + (size_t)step * (NP * 1024);
#pragma unroll
      for (int p = 0; p < NP; ++p) dst[mi][p] = *reinterpret_cast<const uint4*>(sp + p * 1024 + lane16);
    }
  };
  float xr[NIT][8];       // PS: the same registers hold the two 16-byte units (hi, lo) of an item

  // valid input length per segment, read once (a global load inside the K loop would drain the weight prefetch behind it)
  auto seg_limit = [&](int sg) {
    const int tin = (int)a.seg[sg].Tin;
    return a.seg[sg].in_len ? (int)min(a.seg[sg].in_len[lb], (int64_t)tin) : tin;
  };
  const int lim0 = seg_limit(0), lim1 = a.nseg > 1 ? seg_limit(1) : 0;
  // load_x only issues loads (clamped addresses, no predication, no use of the values): validity is applied by store_x,
  // a K step of MFMAs later, so no s_waitcnt lands between the loads and the math.  Addresses are a scalar base (batch
  // item) plus one unsigned 32-bit byte offset per lane: row offsets advance by additions, clamped to the last channel.
  // NP = 2 over a gradient tensor: staged x 2^(13 - e), e = exponent of the tensor's max |value| (the largest gradient lands in [2^13, 2^14),
  // everything within 2^-27 of it keeps 2^-22 relative precision), instead of the fixed 2^-6 of activations; undone on the accumulators
  float f16_in = kF16XScale, f16_out = 1.f;
  if (NP == 2 && a.seg[0].in_absmax) {
    const int e = min(max((int)(*a.seg[0].in_absmax >> 23) - 127, -100), 100);
    f16_in = __uint_as_float((uint32_t)(127 + 13 - e) << 23);
    f16_out = __uint_as_float((uint32_t)(127 - 13 + e) << 23) * (1.f / kF16WScale);
  }
  bool xok[NIT];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int st_kg0 = (wave_u / SEG) * RPW, st_j0 = (wave_u % SEG) * SL;
  auto load_x = [&](int sg, int chunk) {
    if constexpr (PS != 0) {
      // pre-split input: an item is two 16-byte loads (hi and lo unit of (8-channel group, column)) and no arithmetic
      const int taps = a.seg[sg].taps, dil = a.seg[sg].dil;
      const int wx = BN + (taps - 1) * dil;
      const int tin = (int)a.seg[sg].Tin, g8 = a.seg[sg].xp_g8;
      const int lim = sg == 0 ? lim0 : lim1;
      const int tau0 = q0 + a.seg[sg].toff - a.seg[sg].pad_left;
      const uint4* xpb = a.seg[sg].xp + (int64_t)b * g8 * tin;
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int kgi = chunk * KG + st_kg0 + rr;                              // scalar: the item's 8-channel group
        const uint4* rowp = xpb + (int64_t)min(kgi, g8 - 1) * tin;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int it = rr * NPASS + ps;
          const int j = min(st_j0 + ps * 64 + lane, XS - 1);
          const int tau = tau0 + j;
          xok[it] = (j < wx) && (tau >= 0) && (tau < lim) && (kgi < g8);
          const uint4* ptr = rowp + min(max(tau, 0), tin - 1);
          const uint4 hi = *ptr, lo = *(ptr + a.seg[sg].xp_plane);
          xr[it][0] = __uint_as_float(hi.x); xr[it][1] = __uint_as_float(hi.y); xr[it][2] = __uint_as_float(hi.z); xr[it][3] = __uint_as_float(hi.w);
          xr[it][4] = __uint_as_float(lo.x); xr[it][5] = __uint_as_float(lo.y); xr[it][6] = __uint_as_float(lo.z); xr[it][7] = __uint_as_float(lo.w);
        }
      }
      return;
    }
    const char* xb = reinterpret_cast<const char*>(a.seg[sg].x + (int64_t)b * a.seg[sg].bstride);
    const int taps = a.seg[sg].taps, dil = a.seg[sg].dil, tstride = a.seg[sg].tstride, Cin = a.seg[sg].Cin;
    const int wx = BN + (taps - 1) * dil;
    const int tin = (int)a.seg[sg].Tin;
    const int lim = sg == 0 ? lim0 : lim1;
    const int tau0 = q0 * tstride + a.seg[sg].toff - a.seg[sg].pad_left;
    const uint32_t cs4 = (uint32_t)a.seg[sg].cstride * 4u, last = (uint32_t)(Cin - 1) * cs4;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      // rows past the last channel are clamped here and multiplied by a scalar zero in store_x
      uint32_t row = (uint32_t)(chunk * (KG * 8) + (st_kg0 + rr) * 8) * cs4;      // scalar
      const char* rowp[8];
#pragma unroll
      for (int e = 0; e < 8; ++e, row += cs4) rowp[e] = xb + min(row, last);
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int it = rr * NPASS + ps;
        const int j = min(st_j0 + ps * 64 + lane, XS - 1);
        const int tau = tau0 + j * tstride;
        xok[it] = (j < wx) && (tau >= 0) && (tau < lim);
        const uint32_t col = (uint32_t)min(max(tau, 0), tin - 1) * 4u;
#pragma unroll
        for (int e = 0; e < 8; ++e) xr[it][e] = *reinterpret_cast<const float*>(rowp[e] + col);
      }
    }
  };
  auto store_x = [&](uint4* dst, int sg, int chunk) {
#pragma clang fp contract(off)
    if constexpr (PS != 0) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int rr = it / NPASS, ps = it % NPASS;
        const int jl = ps * 64 + lane;
        const int j = st_j0 + jl;
        if ((SL % 64 != 0 && jl >= SL) || (XS % SL != 0 && j >= XS)) continue;
        const int i = (st_kg0 + rr) * XS + j;
        const bool ok = xok[it];
        dst[i] = make_uint4(ok ? __float_as_uint(xr[it][0]) : 0u, ok ? __float_as_uint(xr[it][1]) : 0u, ok ? __float_as_uint(xr[it][2]) : 0u,
                            ok ? __float_as_uint(xr[it][3]) : 0u);
        dst[PSZ + i] = make_uint4(ok ? __float_as_uint(xr[it][4]) : 0u, ok ? __float_as_uint(xr[it][5]) : 0u, ok ? __float_as_uint(xr[it][6]) : 0u,
                                  ok ? __float_as_uint(xr[it][7]) : 0u);
      }
      return;
    }
    // No contraction in here: the scaled input is ROUNDED to fp32 and then split.  Fused into the first subtraction of the split (fma), the
    // pieces would sum to the unrounded product, and the whole-stack WaveNet kernel, which rounds, would differ in the last bit.
    const float scale = NP == 2 ? a.seg[sg].in_scale * f16_in : a.seg[sg].in_scale;
    const int Cin = a.seg[sg].Cin;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int rr = it / NPASS, ps = it % NPASS;
      const int jl = ps * 64 + lane;                         // column inside this wave's segment
      const int j = st_j0 + jl;
      if ((SL % 64 != 0 && jl >= SL) || (XS % SL != 0 && j >= XS)) continue;
      const int i = (st_kg0 + rr) * XS + j;
      const int c0 = chunk * (KG * 8) + (st_kg0 + rr) * 8;        // scalar: the group's first channel
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (xok[it] ? xr[it][e] : 0.f) * (c0 + e < Cin ? scale : 0.f);
      if constexpr (NP == 1) {
        bf16x8 p;
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = (__bf16)v[e];
        dst[i] = __builtin_bit_cast(uint4, p);
      } else if constexpr (NP == 2) {
        f16x8 ph, pl;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          // v_cvt_pk_f16_f32: round to nearest even, subnormals kept; the subtraction below is exact
          const f16x2 hi = __builtin_convertvector((f32x2){v[e], v[e + 1]}, f16x2);
          const f16x2 lo = __builtin_convertvector((f32x2){(v[e] - (float)hi[0]) * kF16LoScale, (v[e + 1] - (float)hi[1]) * kF16LoScale}, f16x2);
          ph[e] = hi[0]; ph[e + 1] = hi[1];
          pl[e] = lo[0]; pl[e + 1] = lo[1];
        }
        dst[i] = __builtin_bit_cast(uint4, ph);
        dst[PSZ + i] = __builtin_bit_cast(uint4, pl);
      } else {
        uint32_t p1[4], p2[4], p3[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const float r0 = v[e] - __uint_as_float(__float_as_uint(v[e]) & 0xffff0000u);
          const float r1 = v[e + 1] - __uint_as_float(__float_as_uint(v[e + 1]) & 0xffff0000u);
          const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u);
          const float s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
          p1[e >> 1] = pack_hi16(v[e], v[e + 1]);
          p2[e >> 1] = pack_hi16(r0, r1);
          p3[e >> 1] = pack_hi16(s0, s1);
        }
        dst[i] = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        dst[PSZ + i] = make_uint4(p2[0], p2[1], p2[2], p2[3]);
        dst[2 * PSZ + i] = make_uint4(p3[0], p3[1], p3[2], p3[3]);
      }
    }
  };
  // scalars of the segment being reduced, kept in SGPRs: read from the kernel arguments inside the loop they cost a
  // scalar-cache round trip (and an s_waitcnt lgkmcnt(0) that also drains the LDS reads) per K step
  int cur_taps = a.seg[0].taps, cur_nchunk = a.seg[0].nchunk, cur_dil = a.seg[0].dil;
  // chunk (sg, ck) -> the chunk staged after it; false when it is the last one
  auto next_chunk = [&](int sg, int ck, int& nsg, int& nck) -> bool {
    nsg = sg; nck = ck + 1;
    if (nck * SUB >= cur_nchunk) { nck = 0; ++nsg; }
    return nsg < a.nseg;
  };

  int sg = 0, c16 = 0, tap = 0, xbuf = 0;
  int psg = 0, pck = 0;          // chunk whose x is on its way to LDS
  bool pending = false;
  // Weight fragments are prefetched PD K steps ahead into PD + 1 register sets; the K loop is unrolled so the sets rotate
  // without copies.  One step of MFMAs (<= 0.5 us) does not cover an L2 hit under load (~1 us): PD = 2.
  // Weight loads first, x loads last in the prologue: store_x then waits for the youngest load, so nothing is outstanding
  // when the loop is entered (a prologue load still pending at loop entry forces a static s_waitcnt vmcnt(0) into the loop
  // body, which then drains the NEXT steps' weight prefetch in every iteration).
  constexpr int PD = NP == 2 ? DMEL_PD2 : DMEL_PD;
  uint4 wa[PD + 1][MT][NP];
#pragma unroll
  for (int d = 0; d < PD; ++d) load_w(wa[d], min(d, a.steps - 1));
  load_x(0, 0);
  store_x(Xb, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  __syncthreads();
  int cstep = 0;                 // K step inside the staged chunk

  // Wait placement.  The compiler's s_waitcnt insertion is static: where paths with different numbers of loads in flight
  // merge (steps that also issue the next chunk's x loads vs. steps that do not), it emits the wait that is safe for the
  // path with the FEWEST younger loads, which on the other paths drains loads issued a few instructions earlier (measured:
  // a full vmcnt(0) in front of the MFMAs of every chunk's first step).  So the order inside a step is fixed such that one
  // count is right on all paths:
  //   top:  issue the weight loads of step s+PD;  read the x fragments of step s from LDS;  MFMAs of step s (their
  //         weights were waited for at the end of step s-1);
  //   end:  EXPLICIT vmcnt(#weight loads issued at the top): the weights of step s+1 and any x loads issued at the end of
  //         step s-1 have landed (both had >= one step of MFMAs);  then the staging work, off the barrier's critical path:
  //         chunk step 0 issues the next chunk's x loads, chunk step 1 converts them and writes the idle LDS buffer (released
  //         by the barrier that opened this chunk), so the barrier closing the chunk has only MFMAs in front of it.
  // (Register double-buffering of the x fragments was tried and dropped: the ds_read latency hides behind the other wave.)
  constexpr int kWLoads = MT * NP;                                              // weight loads per step and wave
  // at the end of step s only the weights of step s + 1 have to be home: the loads of steps s + 2 .. s + PD stay in flight.  (Until round 3
  // the wait was vmcnt(kWLoads) whatever PD: a prefetch distance above 2 then fetched earlier but was drained just the same -- the
  // "PD = 3" experiment of profiles/r03_conv_experiments.txt measured nothing.)
  constexpr int kWKeep = (PD - 1) * kWLoads < 63 ? (PD - 1) * kWLoads : 63;
  constexpr int kWaitW = (kWKeep & 15) | (7 << 4) | (15 << 8) | ((kWKeep >> 4) << 14);     // s_waitcnt vmcnt((PD - 1) kWLoads)
  constexpr int kXLoads = NIT * (PS != 0 ? 2 : 8);                                          // x loads per thread and chunk
  constexpr int kWX = kWKeep + kXLoads < 63 ? kWKeep + kXLoads : 63;
  [[maybe_unused]] constexpr int kWaitWX = (kWX & 15) | (7 << 4) | (15 << 8) | ((kWX >> 4) << 14);           // ... that also leaves a chunk of x loads in flight
#if DMEL_BPF
  bf16x8 bfrag[2][NT][NP];
  bool have_frag = false;
#endif
  auto k_step = [&](auto R, int s) {
    constexpr int r = decltype(R)::value;
    uint4 (&use)[MT][NP] = wa[r % (PD + 1)];
    int nsg = sg, nc16 = c16, ntap = tap + 1;
    bool newx = false;
    if (ntap == cur_taps) {
      ntap = 0;
      ++nc16;
      if (nc16 == cur_nchunk) { nc16 = 0; ++nsg; newx = true; }
      else newx = (nc16 % SUB) == 0;
    }
    const bool has_next = s + 1 < a.steps;
#if DMEL_XTOP
    // x loads of the NEXT chunk at the TOP of a chunk's first step, in front of this step's weight prefetch.  vmcnt retires in order: the wait
    // for the weights of step s + 1 at the end of this step may leave these loads (and the weights of step s + 2) in flight, and only the
    // wait at the end of step s + 1 -- for weights issued AFTER them -- forces them home.  Issued at the END of the step (as before) they
    // sat behind one step of MFMAs only: ~300 cycles against an L2 / Infinity-Cache latency of 500-900, the rest a stall in every chunk.
    bool x_younger = false;
    if (cstep == 0) {
      pending = next_chunk(sg, c16 / SUB, psg, pck);
      if (pending && !(DMEL_EXP & 2)) { load_x(psg, pck); x_younger = !newx; }
      if (DMEL_EXP & 2) pending = false;
    }
#endif
    // unconditional (the last PD steps re-fetch the final step's fragments into a set nobody reads again): a branch here
    // would put a wait-free path into the CFG and with it a conservative vmcnt(0) in front of the MFMAs
    if (!(DMEL_EXP & 4)) load_w(wa[(r + PD) % (PD + 1)], min(s + PD, a.steps - 1));
    {
#if DMEL_BPF
      bf16x8 (&bcur)[NT][NP] = bfrag[r % 2];
      bf16x8 (&bnxt)[NT][NP] = bfrag[(r + 1) % 2];
      if (!have_frag) {
        const uint4* xp = Xb + xbuf * (NP * PSZ) + ((c16 % SUB) * 2 + h) * XS + wave_n * (NT * 32) + l31 + tap * cur_dil;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
          for (int p = 0; p < NP; ++p) bcur[ni][p] = __builtin_bit_cast(bf16x8, xp[p * PSZ + ni * 32]);
      }
      have_frag = has_next && !newx;
      if (have_frag) {      // next step reads the same staged chunk: its fragments travel while this step multiplies
        const uint4* xq = Xb + xbuf * (NP * PSZ) + ((nc16 % SUB) * 2 + h) * XS + wave_n * (NT * 32) + l31 + ntap * cur_dil;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
          for (int p = 0; p < NP; ++p) bnxt[ni][p] = __builtin_bit_cast(bf16x8, xq[p * PSZ + ni * 32]);
      }
#else
      const uint4* xp = Xb + xbuf * (NP * PSZ) + ((c16 % SUB) * 2 + h) * XS + wave_n * (NT * 32) + l31 + tap * cur_dil;
      bf16x8 bcur[NT][NP];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int p = 0; p < NP; ++p) bcur[ni][p] = __builtin_bit_cast(bf16x8, xp[p * PSZ + ni * 32]);
#endif
      if constexpr (NP == 2) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) {
            const f16x8 ah = __builtin_bit_cast(f16x8, use[mi][0]), al = __builtin_bit_cast(f16x8, use[mi][1]);
            const f16x8 bh = __builtin_bit_cast(f16x8, bcur[ni][0]), bl = __builtin_bit_cast(f16x8, bcur[ni][1]);
            acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acl[mi][ni], 0, 0, 0);
            acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acl[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mi][ni], 0, 0, 0);
          }
      } else {
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest partial products first
#pragma unroll
        for (int t = 0; t < (NP == 3 ? 6 : 1); ++t)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, use[mi][NP == 3 ? PA[t] : 0]),
                                                                    bcur[ni][NP == 3 ? PB[t] : 0], acc[mi][ni], 0, 0, 0);
      }
    }
#if DMEL_XTOP
    if (x_younger) __builtin_amdgcn_s_waitcnt(kWaitWX);      // the weights of step s + 1 are older than the x loads: those may stay in flight
    else __builtin_amdgcn_s_waitcnt(kWaitW);
#else
    __builtin_amdgcn_s_waitcnt(kWaitW);
    if (cstep == 0) {
      pending = next_chunk(sg, c16 / SUB, psg, pck);
      if (pending && !(DMEL_EXP & 2)) load_x(psg, pck);
      if (DMEL_EXP & 2) pending = false;
    }
#endif
    if (pending && (((DMEL_EXP & 1) ? false : cstep == 1) || newx)) {       // second step of the chunk, or its only one
      store_x(Xb + (xbuf ^ 1) * (NP * PSZ), psg, pck);
      pending = false;
    }
    ++cstep;
    if (has_next && newx) {
      __syncthreads();
      xbuf ^= 1;
      cstep = 0;
      if (nsg != sg) { cur_taps = a.seg[nsg].taps; cur_nchunk = a.seg[nsg].nchunk; cur_dil = a.seg[nsg].dil; }
    }
    sg = nsg; c16 = nc16; tap = ntap;
  };
  // unrolled by the PD + 1 weight sets so that they rotate without register copies
  // (DMEL_BPF: and by the two fragment sets -- the unroll is the least common multiple, so that both rotations are compile-time indices)
  constexpr int UNR = DMEL_BPF ? ((PD + 1) % 2 ? 2 * (PD + 1) : PD + 1) : PD + 1;
  static_assert(UNR <= 8, "k_step calls below cover an unroll of up to eight");
  for (int s = 0; s < a.steps; s += UNR) {
    k_step(std::integral_constant<int, 0>{}, s);
    if (s + 1 < a.steps) k_step(std::integral_constant<int, 1>{}, s + 1);
    if (UNR > 2 && s + 2 < a.steps) k_step(std::integral_constant<int, 2 % UNR>{}, s + 2);
    if (UNR > 3 && s + 3 < a.steps) k_step(std::integral_constant<int, 3 % UNR>{}, s + 3);
    if (UNR > 4 && s + 4 < a.steps) k_step(std::integral_constant<int, 4 % UNR>{}, s + 4);
    if (UNR > 5 && s + 5 < a.steps) k_step(std::integral_constant<int, 5 % UNR>{}, s + 5);
    if (UNR > 6 && s + 6 < a.steps) k_step(std::integral_constant<int, 6 % UNR>{}, s + 6);
    if (UNR > 7 && s + 7 < a.steps) k_step(std::integral_constant<int, 7 % UNR>{}, s + 7);
  }
  if constexpr (NP == 2) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = fmaf(acl[mi][ni][r], 1.f / kF16LoScale, acc[mi][ni][r]) * f16_out;
  }
  if ((DMEL_EXP & 8) && acc[0][0][0] != 12345.f) return;
  conv_epilogue<MT, NT, MODE, NP == 2 ? 8 : 4>(a, acc, mblk * BM + wave_m * (MT * 32), q0 + wave_n * (NT * 32) + l31, b, lb, h);
}

template <int WM, int WN, int MT, int NT, int MODE, int HALO, int NP, int KG, int PS = 0>
static int launch_b16k(const KArgs& ka, int B, int mblocks, hipStream_t st) {
  constexpr int BN = WN * NT * 32;
#ifndef DMEL_CONV_LDS_PAD
#define DMEL_CONV_LDS_PAD 0
#endif
  constexpr size_t lds0 = (size_t)2 * NP * KG * (BN + HALO) * 16;
  constexpr size_t lds = lds0 + (lds0 + 2 * DMEL_CONV_LDS_PAD <= 65536 ? 2 * DMEL_CONV_LDS_PAD : 0);
  static_assert(lds <= 64 * 1024, "bf16 conv tile exceeds the default dynamic LDS limit");
  KArgs k2 = ka;
  dim3 grid;
  DMEL_TRY(conv_grid(k2, (int)((ka.Tcols + BN - 1) / BN), mblocks, B, grid));
  hipLaunchKernelGGL((conv_bf16_kernel<WM, WN, MT, NT, MODE, HALO, NP, KG, PS>), grid, dim3(64 * WM * WN), lds, st, k2);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

template <int WM, int WN, int MT, int NT, int MODE, int NP> static int launch_b16(const KArgs& ka, int B, int mblocks, hipStream_t st) {
  int halo = 0;
  for (int s = 0; s < ka.nseg; ++s) halo = std::max(halo, (ka.seg[s].taps - 1) * ka.seg[s].dil);
  // 1x1 convolutions stage 32 channels per barrier (two K steps); everything else 16 (taps K steps)
  constexpr int KG0 = (WN * NT * 32 >= 256 && NP == 3) ? 2 : 4;       // keep the widest tile inside 64 KiB of LDS
  if (halo == 0) return launch_b16k<WM, WN, MT, NT, MODE, 0, NP, KG0>(ka, B, mblocks, st);
  // dilation-1 convs (every second conv of an AMP block, k2 transposed-conv phases): a 16-column halo is enough
  constexpr int KGT = (NP == 2 && 2 * NP * DMEL_KG2 * (WN * NT * 32 + 64) * 16 <= 65536) ? DMEL_KG2 : 2;
  if (halo <= 16) return launch_b16k<WM, WN, MT, NT, MODE, 16, NP, KGT>(ka, B, mblocks, st);
  return launch_b16k<WM, WN, MT, NT, MODE, 64, NP, KGT>(ka, B, mblocks, st);
}

#ifndef DMEL_TILE5_NARROW
#define DMEL_TILE5_NARROW 1      // tile 5 = 128 x 32 (four waves of 32 x 32): the few-column launches of pick_tile_bf16; 0: the old 64 x 128 (2,1,1,4), never chosen
#endif
template <int MODE, int NP> static int launch_mode_bf16(const KArgs& ka, int tile, int B, hipStream_t st) {
  const int bm[8] = {128, 128, 64, 32, 128, DMEL_TILE5_NARROW ? 128 : 64, 128, 256};
  // bf16 tiles: 0: 128x128 (2,2,2,2), 1: 128x96 (4,1,1,3), 2: 64x128 (2,2,1,2), 3: 32x256 (1,4,1,2), 4: 128x128 (4,1,1,4),
  // 5: 128x32 (4,1,1,1) [64x128 (2,1,1,4) with DMEL_TILE5_NARROW=0], 6: 128x64 (4,1,1,2), 7: 256x96 (8,1,1,3: eight waves)
  const int mblocks = (ka.mtiles * 32 + bm[tile] - 1) / bm[tile];
  switch (tile) {
    case 0: return launch_b16<2, 2, 2, 2, MODE, NP>(ka, B, mblocks, st);
    case 1: return launch_b16<4, 1, 1, 3, MODE, NP>(ka, B, mblocks, st);
    case 2: return launch_b16<2, 2, 1, 2, MODE, NP>(ka, B, mblocks, st);
    case 3: return launch_b16<1, 4, 1, 2, MODE, NP>(ka, B, mblocks, st);
    case 4: return launch_b16<4, 1, 1, 4, MODE, NP>(ka, B, mblocks, st);
    case 6: return launch_b16<4, 1, 1, 2, MODE, NP>(ka, B, mblocks, st);
    case 7: return launch_b16<8, 1, 1, 3, MODE, NP>(ka, B, mblocks, st);
#if DMEL_TILE5_NARROW
    default: return launch_b16<4, 1, 1, 1, MODE, NP>(ka, B, mblocks, st);      // 128 x 32 (four waves of 32 x 32)
#else
    default: return launch_b16<2, 1, 1, 4, MODE, NP>(ka, B, mblocks, st);
#endif
  }
}

// Measured on MI355X (tools/bench_conv.py, DMEL_CONV_TILE_BF16 sweep): long rows of >= 128 output channels run best with
// one 32-row strip per wave (4: weights fetched once per block), short ones with the 96-column tile (less padding at
// T = 92 / 736); 64-row problems on 64x128, 32-row problems on 32x256.
// NP = 2 (two accumulator sets): the 96-column tile keeps 2-3 waves per SIMD where the 128-column one has 208 registers, and wins at every
// length (bv2, 128 rows x 5888 columns: 176 / 233 / 250 TF/s against 117 / 182 / 206).
// Within NP = 2 (MI355X, tools/bench_conv.py --precision 3): 256-row problems with long rows gain 5-8 % from the eight-wave 256 x 96 tile (x is
// converted once per 256 rows); 128-row problems with >= 40 K steps gain 6-17 % from 128 x 64 (124 registers, four waves per SIMD).
static int pick_tile_bf16(int mtiles, int64_t T, int np, int steps, int B) {
  const char* e = getenv("DMEL_CONV_TILE_BF16");        // per call: tools/ab_wavenet.py switches tiles inside one process
  const int forced = (e && e[0]) ? atoi(e) : -1;
  if (forced >= 0 && forced < 8) return forced;
  int t;
  if (np == 2 && mtiles >= 8 && T > 96) t = 7;
  else if (np == 2 && mtiles >= 4 && mtiles < 8 && steps >= 40) t = 6;
  else if (mtiles >= 4) t = (T > 2048 && np != 2) ? 4 : 1;
  else if (mtiles >= 2) t = 2;
  else t = 3;
  // FEW COLUMNS (streaming decode of one stream: 128-512 frames per push; round 3): when the choice above makes fewer workgroups than half
  // the CUs -- 18 for the decoder WaveNet's gate convolution at a 64-token chunk -- the launch is a handful of long K loops on an idle
  // chip.  The 128 x 64 tile (four waves) makes 2-3x as many.  Every tile accumulates K in the same order: same bits, so a streamed chunk
  // still equals the whole-sequence decode (tests).  Measured (tools/bench_stream.py, batch 1, pipelined): 64-token chunks 480 -> 589
  // audio-s/s, 128-token 877 -> 1025 (profiles/r03_stream.txt).
  if (np == 2 && mtiles >= 4) {
    static const int bm[8] = {128, 128, 64, 32, 128, DMEL_TILE5_NARROW ? 128 : 64, 128, 256};
    static const int bn[8] = {128, 96, 128, 256, 128, DMEL_TILE5_NARROW ? 32 : 128, 64, 96};
    auto wgs = [&](int tt) { return (int64_t)((mtiles * 32 + bm[tt] - 1) / bm[tt]) * ((T + bn[tt] - 1) / bn[tt]) * B; };
    static const bool off = [] { const char* f = getenv("DMEL_CONV_SMALLN"); return f && f[0] == '0'; }();
    if (!off && wgs(t) < 128) {
      t = 6;
      // ... and the 128 x 32 tile where even that leaves most CUs idle (a 64-token chunk: 44 -> 88 workgroups; forced on every launch it
      // measured 580 -> 634 audio-s/s at 64-token chunks, 299 -> 332 at 32)
      if (DMEL_TILE5_NARROW && wgs(6) < 128) t = 5;
    }
  }
  return t;
}

// Pre-split inputs (SegRun::xp; the decoder WaveNet's gate and residual / skip convolutions): the fp16-split kernel with a staging pass
// that copies 16-byte units.  Instantiated for the tiles those layers use: 128 x 96 (short rows) and the eight-wave 256 x 96.
template <int MODE> static int launch_presplit(const KArgs& ka, int B, int64_t Tcols, hipStream_t st) {
  int halo = 0;
  for (int s = 0; s < ka.nseg; ++s) halo = std::max(halo, (ka.seg[s].taps - 1) * ka.seg[s].dil);
  if (halo > 16) { set_error("conv (pre-split input): receptive field %d exceeds the 16-column halo built for this path", halo); return DMEL_EUNSUPPORTED; }
  const bool wide = ka.mtiles >= 8 && Tcols > 96;
  if (wide) {
    const int mblocks = (ka.mtiles * 32 + 255) / 256;
    if (halo == 0) return launch_b16k<8, 1, 1, 3, MODE, 0, 2, 4, 1>(ka, B, mblocks, st);
    return launch_b16k<8, 1, 1, 3, MODE, 16, 2, DMEL_KG2, 1>(ka, B, mblocks, st);
  }
  const int mblocks = (ka.mtiles * 32 + 127) / 128;
  if (halo == 0) return launch_b16k<4, 1, 1, 3, MODE, 0, 2, 4, 1>(ka, B, mblocks, st);
  return launch_b16k<4, 1, 1, 3, MODE, 16, 2, DMEL_KG2, 1>(ka, B, mblocks, st);
}

template <int NP> static int launch_bf16_any(const KArgs& ka, EpiMode mode, int B, int64_t Tcols, hipStream_t st) {
  const int t16 = pick_tile_bf16(ka.mtiles, Tcols, NP, ka.steps, B);
  switch (mode) {
    case EPI_LINEAR: return launch_mode_bf16<EPI_LINEAR, NP>(ka, t16, B, st);
    case EPI_GATE: return launch_mode_bf16<EPI_GATE, NP>(ka, t16, B, st);
    default: return launch_mode_bf16<EPI_RESSKIP, NP>(ka, t16, B, st);
  }
}

// ---- tile selection -------------------------------------------------------------------------------------------
// The kernel is MFMA-bound, so a launch costs (rounds of workgroups over the 256 CUs) x (tile area) / (how well a wave
// of that shape keeps the pipe fed).  Padding waste and the last, partly filled round are what the choice trades.
struct TileCfg {
  int wm, wn, mt, nt;
  float eff;
};
static const TileCfg kTiles[] = {
    {2, 2, 2, 2, 0.93f},  // 128 x 128, 4 waves (212 registers: 2 waves per SIMD)
    {4, 1, 1, 3, 1.00f},  // 128 x  96 (164 registers: 3 waves per SIMD), measured fastest per MFMA
    {2, 2, 1, 2, 0.92f},  //  64 x 128
    {2, 1, 1, 3, 0.85f},  //  64 x  96, 2 waves
    {1, 4, 1, 2, 0.90f},  //  32 x 256
    {1, 4, 1, 1, 0.80f},  //  32 x 128
    {1, 3, 1, 1, 0.80f},  //  32 x  96, 3 waves
    // register-direct variants (conv_direct_kernel), ids 7..: not picked automatically yet
    {1, 4, 1, 1, 0.00f},  //  7: 32 x 128, 4 waves
    {1, 3, 1, 1, 0.00f},  //  8: 32 x  96, 3 waves
    {1, 4, 1, 2, 0.00f},  //  9: 32 x 256
    {2, 2, 1, 2, 0.00f},  // 10: 64 x 128
    {4, 1, 1, 3, 0.00f},  // 11: 128 x 96
    {2, 2, 2, 2, 0.00f},  // 12: 128 x 128
    {2, 2, 1, 1, 0.00f},  // 13: 64 x 64
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

static int pick_tile(int mtiles, int64_t T, int B) {
  static int forced = [] {
    const char* e = getenv("DMEL_CONV_TILE");
    return e ? atoi(e) : -1;
  }();
  if (forced >= 0 && forced < kNumTiles) return forced;
  // One column tile per batch item (1 s clips through the WaveNets, conv_pre): too few workgroups for x-tile sharing
  // through LDS to pay; the register-direct kernel measured 1.06-1.4x faster there (tools/ab_tiles.sh).
  if (T <= 96 && mtiles >= 4) return 8;
  int best = 0;
  double best_cost = 1e300;
  for (int i = 0; i < kNumTiles; ++i) {
    if (kTiles[i].eff <= 0.f) continue;
    const int bm = kTiles[i].wm * kTiles[i].mt * 32, bn = kTiles[i].wn * kTiles[i].nt * 32;
    const double wgs = (double)((mtiles * 32 + bm - 1) / bm) * (double)((T + bn - 1) / bn) * B;
    const double rounds = std::ceil(wgs / 256.0);
    // above ~4 rounds the partly filled last round stops mattering
    const double cost = (wgs > 1024 ? wgs / 256.0 : rounds) * bm * bn / kTiles[i].eff;
    if (cost < best_cost) { best_cost = cost; best = i; }
  }
  return best;
}

template <int MODE> static int launch_mode(const KArgs& ka, int tile, int B, hipStream_t st) {
  const TileCfg& t = kTiles[tile];
  const int mblocks = (ka.mtiles * 32 + t.wm * t.mt * 32 - 1) / (t.wm * t.mt * 32);
  switch (tile) {
    case 0: return launch_t<2, 2, 2, 2, MODE>(ka, B, mblocks, st);
    case 1: return launch_t<4, 1, 1, 3, MODE>(ka, B, mblocks, st);
    case 2: return launch_t<2, 2, 1, 2, MODE>(ka, B, mblocks, st);
    case 3: return launch_t<2, 1, 1, 3, MODE>(ka, B, mblocks, st);
    case 4: return launch_t<1, 4, 1, 2, MODE>(ka, B, mblocks, st);
    case 5: return launch_t<1, 4, 1, 1, MODE>(ka, B, mblocks, st);
    case 6: return launch_t<1, 3, 1, 1, MODE>(ka, B, mblocks, st);
    case 7: return launch_d<1, 4, 1, 1, MODE>(ka, B, mblocks, st);
    case 8: return launch_d<1, 3, 1, 1, MODE>(ka, B, mblocks, st);
    case 9: return launch_d<1, 4, 1, 2, MODE>(ka, B, mblocks, st);
    case 10: return launch_d<2, 2, 1, 2, MODE>(ka, B, mblocks, st);
    case 11: return launch_d<4, 1, 1, 3, MODE>(ka, B, mblocks, st);
    case 12: return launch_d<2, 2, 2, 2, MODE>(ka, B, mblocks, st);
    default: return launch_d<2, 2, 1, 1, MODE>(ka, B, mblocks, st);
  }
}

// x (B, C, T) fp32 -> pre-split planes [piece][b][C / 8][T] x 16 bytes (SegRun::xp): the operand split of store_x, once per element
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, uint4* __restrict__ planes, int64_t plane_units,
                                                           const int64_t* __restrict__ len, int len_div, int C, int T) {
  const int b = blockIdx.z, g = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float* xr = x + ((int64_t)b * C + g * 8) * T + t;
  const bool live = !len || t < len[b / len_div];
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = live ? xr[(int64_t)e * T] : 0.f;
  f16x8 ph, pl;
  {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const float s0 = v[e] * kF16XScale, s1 = v[e + 1] * kF16XScale;
      const f16x2 hi = __builtin_convertvector((f32x2){s0, s1}, f16x2);
      const f16x2 lo = __builtin_convertvector((f32x2){(s0 - (float)hi[0]) * kF16LoScale, (s1 - (float)hi[1]) * kF16LoScale}, f16x2);
      ph[e] = hi[0]; ph[e + 1] = hi[1];
      pl[e] = lo[0]; pl[e + 1] = lo[1];
    }
  }
  const int64_t i = ((int64_t)b * (C / 8) + g) * T + t;
  planes[i] = __builtin_bit_cast(uint4, ph);
  planes[plane_units + i] = __builtin_bit_cast(uint4, pl);
}
int launch_split_planes(const float* x, void* planes, int64_t plane_units, const int64_t* len, int len_div, int B, int C, int64_t T,
                        hipStream_t stream) {
  DMEL_CHECK_ARG(x && planes && B > 0 && B <= 65535 && C > 0 && C % 8 == 0 && C / 8 <= 65535 && T > 0 && T < ((int64_t)1 << 30), "split_planes: bad argument");
  ProfScope ps("small", stream, 0.0, 8.0 * B * C * (double)T);
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)(C / 8), (unsigned)B), dim3(256), 0, stream, x,
                     reinterpret_cast<uint4*>(planes), plane_units, len, len_div > 0 ? len_div : 1, C, (int)T);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

// Debug range check of the fp16 split's documented domain (include/dmel_hip.h: activations are staged x 2^-6, so |x| must stay below
// 2^6 * 65504 = 4.19e6; beyond that the first piece overflows to inf).  DMEL_DEBUG_F16_RANGE=1: reduce max |x| of every contiguous
// input of such a launch, synchronise, and fail the launch LOUDLY if it is outside the domain.  Off by default (it serialises).
static int check_f16_range(const PackedConv& pc, const ConvRun& r, hipStream_t stream) {
  static const int range_check = [] { const char* e = getenv("DMEL_DEBUG_F16_RANGE"); return e ? atoi(e) : 0; }();
  if (!range_check) return DMEL_OK;
  const PackDesc& d = pc.d;
  for (int s = 0; s < d.nseg; ++s) {
    const SegRun& sr = r.seg[s];
    if (sr.in_absmax || sr.cstride != sr.Tin || sr.bstride != (int64_t)d.seg[s].Cin * sr.cstride) continue;     // scaled by its own maximum / a view
    const uint32_t* slot = nullptr;
    DMEL_TRY(launch_absmax(sr.x, (int64_t)r.B * sr.bstride, stream, &slot));
    uint32_t bits = 0;
    DMEL_HIP(hipMemcpyAsync(&bits, slot, sizeof(bits), hipMemcpyDeviceToHost, stream));
    DMEL_HIP(hipStreamSynchronize(stream));
    float mx;
    std::memcpy(&mx, &bits, sizeof(mx));
    if (!(mx * sr.in_scale < kF16WScale * 65504.f)) {
      set_error("conv (fp16 split): max |x| = %g of input segment %d is outside the documented domain |x| < %g "
                "(DMEL_PRECISION_FP32_F16X2; use DMEL_PRECISION_FP32_BF16X3 for such tensors)", (double)mx * sr.in_scale, s,
                (double)(kF16WScale * 65504.f));
      return DMEL_EINVAL;
    }
  }
  return DMEL_OK;
}

int launch_conv(const PackedConv& pc, const ConvRun& r, hipStream_t stream) {
  {  // producer / consumer kernel for the launches it was measured to win on (conv_pc.hip); DMEL_CONV_PC=0 is the A/B switch, read per call
    const char* e = getenv("DMEL_CONV_PC");
    if (!(e && e[0] == '0') && conv_pc_eligible(pc, r, e && e[0] == '2')) {      // 2: also tiny launches (bit-identity tests on small shapes)
      DMEL_TRY(check_f16_range(pc, r, stream));
      return launch_conv_pc(pc, r, stream);
    }
  }
  KArgs ka{};
  const PackDesc& d = pc.d;
  ka.nseg = d.nseg;
  ka.steps = pc.steps;
  int max_halo = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const SegDesc& sd = d.seg[s];
    SegArgs& o = ka.seg[s];
    DMEL_CHECK_ARG(r.seg[s].x != nullptr || r.seg[s].xp != nullptr, "conv: input pointer of segment %d is NULL", s);
    DMEL_CHECK_ARG(r.seg[s].xp == nullptr || (r.precision == DMEL_PRECISION_FP32_F16X2 && sd.Cin % 8 == 0 && sd.tstride == 1 &&
                                              r.seg[s].in_scale == 1.f && r.seg[s].tshift == 0 && r.seg[s].in_absmax == nullptr && r.fold_pitch == 0),
                   "conv: a pre-split input needs the fp16-split precision, channels %% 8 == 0, unit stride and scale");
    DMEL_CHECK_ARG((r.seg[s].xp != nullptr) == (r.seg[0].xp != nullptr), "conv: either every input segment is pre-split or none is");
    // taps on a strided view (polyphase branches of a stride-2 convolution: the discriminator) are staged correctly by the bf16
    // matrix-core kernels only; the native fp32-MFMA kernels were never built for it
    DMEL_CHECK_ARG(sd.taps == 1 || sd.tstride == 1 || (r.precision != DMEL_PRECISION_FP32_MFMA && !getenv("DMEL_CONV_FP32_MFMA")),
                   "conv: taps>1 with strided input is unsupported by the native fp32-MFMA kernel");
    o.x = r.seg[s].x; o.bstride = r.seg[s].bstride; o.cstride = r.seg[s].cstride; o.Tin = r.seg[s].Tin;
    o.in_len = r.seg[s].in_len; o.in_scale = r.seg[s].in_scale;
    o.in_absmax = (s == 0 && d.nseg == 1 && r.precision == DMEL_PRECISION_FP32_F16X2) ? r.seg[s].in_absmax : nullptr;
    o.xp = reinterpret_cast<const uint4*>(r.seg[s].xp); o.xp_plane = r.seg[s].xp_plane; o.xp_g8 = sd.Cin / 8;
    o.Cin = sd.Cin; o.nchunk = (sd.Cin + kCK - 1) / kCK; o.taps = sd.taps; o.dil = sd.dil;
    o.pad_left = sd.pad_left; o.tstride = sd.tstride; o.toff = sd.toff + (int)(r.seg[s].tshift * sd.tstride);
    max_halo = std::max(max_halo, (sd.taps - 1) * sd.dil);
    // the staging code addresses one batch item with unsigned 32-bit BYTE offsets
    DMEL_CHECK_ARG((int64_t)sd.Cin * o.cstride < ((int64_t)1 << 30) && o.Tin * sd.tstride < ((int64_t)1 << 30),
                   "conv: one batch item of the input exceeds 32-bit byte offsets (4 GiB)");
  }
  if (max_halo > 64) {
    set_error("conv_igemm: receptive field (taps-1)*dilation = %d exceeds the 64-sample LDS halo", max_halo);
    return DMEL_EUNSUPPORTED;
  }
  DMEL_CHECK_ARG(r.y != nullptr && r.B > 0 && r.Tcols > 0, "conv: bad output/batch/length");
  DMEL_CHECK_ARG(d.mode != EPI_RESSKIP || r.skip != nullptr, "conv: skip buffer missing");
  ka.w = pc.w.as<float>(); ka.w16 = pc.w16.p; ka.w48 = pc.w48.p; ka.w32h = pc.w32h.p; ka.bias = pc.bias.as<float>();
  ka.Tcols = r.Tcols; ka.mode = d.mode; ka.act = r.act; ka.C = d.C; ka.RP = pc.RP; ka.phases = d.phases;
  ka.out_tstride = r.out_tstride; ka.phase_base = r.phase_base; ka.accumulate = r.accumulate;
  ka.len_div = r.len_div > 0 ? r.len_div : 1; ka.skip_first = r.skip_first; ka.out_div = r.out_div;
  ka.y = r.y; ka.y_bs = r.y_bs; ka.y_cs = r.y_cs; ka.Tout = r.Tout > 0 ? r.Tout : r.Tcols * r.out_tstride;
  ka.res = r.res; ka.res_bs = r.res_bs; ka.res_cs = r.res_cs; ka.row_scale = r.row_scale;
  ka.out_len = r.out_len; ka.skip = r.skip;
  ka.fold_pitch = r.fold_pitch; ka.fold_valid = r.fold_valid;
  DMEL_CHECK_ARG(r.yp == nullptr || (d.mode != EPI_LINEAR && d.C % 8 == 0 && r.out_tstride == 1), "conv: pre-split output needs a paired mode and C %% 8 == 0");
  ka.yp = reinterpret_cast<uint4*>(r.yp); ka.yp_plane = r.yp_plane; ka.yp_g8 = d.C / 8; ka.yp_only = r.yp_only;
  DMEL_CHECK_ARG(r.fold_pitch == 0 || (r.fold_pitch >= r.fold_valid && r.fold_valid > 0 && r.B == 1 && r.out_tstride == 1),
                 "conv: folded-batch launches are single-item, unit-stride and need 0 < fold_valid <= fold_pitch");
  DMEL_CHECK_ARG((int64_t)d.C * ka.y_cs < ((int64_t)1 << 31) && ka.Tout < ((int64_t)1 << 30) && ka.Tcols < ((int64_t)1 << 30),
                 "conv: one batch item of the output exceeds 32-bit offsets");
  ka.mtiles = pc.Mpad / 32;
  const double rows_real = (d.mode == EPI_LINEAR ? (double)d.C * d.phases : 2.0 * d.C);
  // algorithmic HBM bytes of the launch: every input row once, every output element once (+ the residual / running-sum / skip rows the
  // epilogue reads), the split weight image once
  double in_elems = 0.0;
  for (int s = 0; s < d.nseg; ++s) in_elems += (double)d.seg[s].Cin * (double)std::min<int64_t>(r.seg[s].Tin, r.Tcols * d.seg[s].tstride + max_halo);
  double out_elems;
  if (d.mode == EPI_LINEAR) out_elems = (double)d.C * d.phases * (double)r.Tcols * (1.0 + (r.res ? 1.0 : 0.0) + (r.accumulate ? 1.0 : 0.0));
  else if (d.mode == EPI_GATE) out_elems = (double)d.C * (double)r.Tcols;
  else out_elems = (double)d.C * (double)r.Tcols * (r.skip_first ? 3.0 : 4.0);
  double alg_bytes = 4.0 * r.B * (in_elems + out_elems);      // + the weight image read once, added below when the arithmetic is known
  static const int native_fp32 = [] { const char* e = getenv("DMEL_CONV_FP32_MFMA"); return e ? atoi(e) : 0; }();
  const bool one_piece = r.precision == DMEL_PRECISION_BF16 || train_precision_override() == DMEL_PRECISION_BF16;
  const bool native = !one_piece && (native_fp32 || r.precision == DMEL_PRECISION_FP32_MFMA);
  const double products = one_piece ? 1.0 : native ? 16.0 /* fp32 MFMA: 1/16 of the bf16 rate */ : r.precision == DMEL_PRECISION_FP32_F16X2 ? 3.0 : 6.0;
  alg_bytes += (double)pc.Mpad * pc.steps * kCK * (one_piece ? 2.0 : native ? 4.0 : r.precision == DMEL_PRECISION_FP32_F16X2 ? 4.0 : 6.0);
  const double alg_flops = 2.0 * r.B * (double)r.Tcols * rows_real * pc.k_real;
  ProfScope ps("conv_igemm", stream, alg_flops, alg_bytes, alg_flops * products);
  if (r.precision == DMEL_PRECISION_BF16 || train_precision_override() == DMEL_PRECISION_BF16)
    return launch_bf16_any<1>(ka, d.mode, r.B, r.Tcols, stream);
  // fp32: the split kernel is the default everywhere (after the wait-placement fixes it also wins on the 32-row, K < 128
  // layers of the last vocoder stage: 58 vs 72 us); the native fp32-MFMA kernels serve DMEL_PRECISION_FP32_MFMA
  if (r.seg[0].xp != nullptr) {
    DMEL_CHECK_ARG(!native_fp32 && train_precision_override() != DMEL_PRECISION_BF16 && d.mode != EPI_LINEAR,
                   "conv: pre-split inputs are built for the paired modes of the fp16-split kernel");
    return d.mode == EPI_GATE ? launch_presplit<EPI_GATE>(ka, r.B, r.Tcols, stream) : launch_presplit<EPI_RESSKIP>(ka, r.B, r.Tcols, stream);
  }
  if (r.precision == DMEL_PRECISION_FP32_F16X2 && !native_fp32) {
    DMEL_TRY(check_f16_range(pc, r, stream));
    return launch_bf16_any<2>(ka, d.mode, r.B, r.Tcols, stream);
  }
  if ((r.precision == DMEL_PRECISION_FP32 || r.precision == DMEL_PRECISION_FP32_BF16X3 || r.precision == DMEL_PRECISION_FP32_F16X2) && !native_fp32)
    return launch_bf16_any<3>(ka, d.mode, r.B, r.Tcols, stream);
  const int tile = pick_tile(ka.mtiles, r.Tcols, r.B);
  switch (d.mode) {
    case EPI_LINEAR: return launch_mode<EPI_LINEAR>(ka, tile, r.B, stream);
    case EPI_GATE: return launch_mode<EPI_GATE>(ka, tile, r.B, stream);
    default: return launch_mode<EPI_RESSKIP>(ka, tile, r.B, stream);
  }
}

}  // namespace dmel

using namespace dmel;

// ---- single-op C ABI (tests / module mirrors) ---------------------------------------------------

extern "C" int dmel_conv_set_precision(dmel_conv* c, int precision) {
  DMEL_CHECK_ARG(c && valid_precision(precision), "conv_set_precision: not a DMEL_PRECISION_* value");
  c->precision = precision;
  return DMEL_OK;
}

extern "C" int dmel_conv_create(dmel_conv** out, const float* w_host, const float* bias_host, int Cout, int Cin, int k,
                                int dilation) {
  DMEL_CHECK_ARG(out && w_host, "NULL argument");
  DMEL_CHECK_ARG(Cout > 0 && Cin > 0 && k > 0 && (k % 2) == 1 && dilation > 0, "conv: bad shape (odd k required)");
  auto* c = new dmel_conv();
  c->Cout = Cout; c->Cin = Cin; c->k = k; c->dil = dilation;
  PackDesc d;
  d.mode = EPI_LINEAR; d.nseg = 1; d.C = Cout; d.phases = 1;
  d.seg[0].Cin = Cin; d.seg[0].taps = k; d.seg[0].dil = dilation; d.seg[0].pad_left = dilation * (k - 1) / 2;
  int rc = pack_conv(c->pc, d,
                     [&](int, int row, int ci, int tap) { return w_host[((size_t)row * Cin + ci) * k + tap]; },
                     [&](int row) { return bias_host ? bias_host[row] : 0.f; });
  if (rc != DMEL_OK) { delete c; return rc; }
  c->w_host.assign(w_host, w_host + (size_t)Cout * Cin * k);      // kept for the lazily packed backward-data weights
  *out = c;
  return DMEL_OK;
}

extern "C" void dmel_conv_destroy(dmel_conv* c) { delete c; }

extern "C" int dmel_conv_forward(const dmel_conv* c, const float* x, float* y, int B, int64_t T, void* stream) {
  DMEL_CHECK_ARG(c && x && y, "NULL argument");
  ConvRun r;
  r.seg[0].x = x; r.seg[0].bstride = (int64_t)c->Cin * T; r.seg[0].cstride = T; r.seg[0].Tin = T;
  r.B = B; r.Tcols = T; r.y = y; r.y_bs = (int64_t)c->Cout * T; r.y_cs = T; r.Tout = T;
  r.precision = c->precision;
  return launch_conv(c->pc, r, (hipStream_t)stream);
}
