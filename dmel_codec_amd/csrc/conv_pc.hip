// Producer / consumer form of the fp16-split convolution (gfx950): conv_bf16_kernel<NP = 2> with the two jobs of its waves taken apart.
//
//   * WM x WN CONSUMER waves (eight in every configuration built: two per SIMD) run nothing but the K loop -- weight fragments straight
//     from L2 into registers two steps ahead, B fragments by ds_read_b128, three MFMAs per 32 x 32 x 16 block, one barrier per staged chunk;
//   * NPROD PRODUCER waves stage x for the next chunk: loads one chunk ahead, scale, two-piece fp16 split, LDS.
//
// Why: rocprofv3 --pmc on conv_bf16_kernel (profiles/r03_pmc_conv.txt) shows its waves issuing ~60 scalar and ~35-60 vector
// instructions per K step next to 6-9 MFMAs -- the bookkeeping of a flat loop that is generic over segments, taps and chunk sizes,
// plus the staging arithmetic -- 16-20 % of wave time in scalar issue alone, on waves that are in-order.  Here the consumer's step is
// a pointer bump, six LDS reads and nine MFMAs, and the staging instructions belong to other waves.  Measured (tools/bench_fused.py,
// interleaved, one device): 1.07-1.20x conv_bf16_kernel on the 256-row vocoder stage, 0.8-1.0x on the 128 / 64 / 32-row stages
// (there the 128 x 64 tile at four waves per SIMD stays ahead), and the decoder WaveNet's 1120-row GEMMs (tools/bench_wavenet.py).
// launch_conv_pc is therefore asked for by the callers that gain: BigVGAN stages of >= 160 channels and the conditioned WaveNet's gate /
// residual convolutions.  Same operand split, K order, MFMA order and epilogue as conv_bf16_kernel<NP = 2>: BIT-IDENTICAL outputs.
#include "conv_dev.h"
#include "ops.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#ifndef DMEL_PC_ASMW
#define DMEL_PC_ASMW 0       // 1: weight stream of the consumers from inline assembly with hand-placed waits.  Removes the compiler's
                            // vmcnt(0) in front of one step in three (checked in the ISA), bit-identical -- and measured within +-3 % of the
                            // compiler-visible loads (profiles/r03_conv_experiments.txt): the weight latency is not the limiter.  Stays 0.
#endif
#ifndef DMEL_PC_SCHED
#define DMEL_PC_SCHED 0      // 1: all six fragment reads issued before the first MFMA of a step (measured 3-8 % SLOWER than the compiler's interleave)
#endif

namespace dmel {

typedef _Float16 pc_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pc_f16x2 __attribute__((ext_vector_type(2)));
typedef float pc_f32x2 __attribute__((ext_vector_type(2)));

template <int WM, int WN, int NT, int NPROD, int HALO, int MODE>
__global__ __launch_bounds__(64 * (WM * WN + NPROD)) void conv_pc_kernel(KArgs a) {
  constexpr int NC = WM * WN;
  constexpr int BN = WN * NT * 32;
  constexpr int XS = BN + HALO;
  constexpr int PSZ = 2 * XS;                             // uint4 per piece: two 8-channel groups per staged chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint4* Xb = reinterpret_cast<uint4*>(smem);             // [2 buffers][2 pieces][2 groups][XS] x 16 bytes

  const int lane = threadIdx.x & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tile_n = blockIdx.x, mblk = blockIdx.y, b = blockIdx.z;
  const int q0 = tile_n * BN;
  const int steps = a.steps;
  const int nch0 = a.seg[0].nchunk, nch1 = a.nseg > 1 ? a.seg[1].nchunk : 0;

  if (wave_u < NC) {
    // ------------------------------------------------------------------------------------------------ consumer
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
    // The consumer's only vector-memory traffic inside the K loop is this weight stream, and it is issued from inline assembly so that the
    // compiler does not see it: hipcc's own s_waitcnt insertion is static, and at the head of the unrolled loop it merged the paths into
    // a vmcnt(0) in front of one step in three -- a full L2 round trip for weights that are not needed until the NEXT step (the ISA of
    // the builtin-load version: profiles/r03_pmc_conv.txt).  Ordering is by the explicit s_waitcnt vmcnt(2) at the end of every step
    // (vmcnt retires in order: all but the two loads of step s + 2 have landed) followed by a scheduling barrier, so that no MFMA of the
    // next step is hoisted above the wait (cdna_hip_programming.md, section 5.7 and rule 18); the loop's tail prefetches are drained by
    // a vmcnt(0) before the registers are handed to the epilogue.
    auto load_w = [&](uint4 (&dst)[2], int step) __attribute__((always_inline)) {
      const char* sp = wT + (size_t)step * 2048 + lane16;
#if DMEL_PC_ASMW
      asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:1024" : "=&v"(dst[0]), "=&v"(dst[1]) : "v"(sp) : "memory");
#else
      dst[0] = *reinterpret_cast<const uint4*>(sp);
      dst[1] = *reinterpret_cast<const uint4*>(sp + 1024);
#endif
    };
    constexpr int PD = 2;
    uint4 wa[PD + 1][2];
    load_w(wa[0], 0);
    load_w(wa[1], min(1, steps - 1));
#if DMEL_PC_ASMW
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // simplest correct start: both sets landed before the loop (once per workgroup)
    __builtin_amdgcn_sched_barrier(0);
#endif
    __syncthreads();                                      // chunk 0 is staged
    int tap = 0, xbuf = 0, ckl = 0, taps = a.seg[0].taps, dil = a.seg[0].dil, nch = nch0;
    constexpr int kWaitW = (2 & 15) | (7 << 4) | (15 << 8);      // s_waitcnt vmcnt(2): the weights of the next step have landed
    auto k_step = [&](auto R, int s) __attribute__((always_inline)) {
      constexpr int r = decltype(R)::value;
      uint4 (&use)[2] = wa[r % (PD + 1)];
      load_w(wa[(r + PD) % (PD + 1)], min(s + PD, steps - 1));   // unconditional: a branch here costs a vmcnt(0) (conv_bf16_kernel)
      const uint4* xp = Xb + xbuf * (2 * PSZ) + h * XS + wave_n * (NT * 32) + l31 + tap * dil;
      pc_f16x8 bh[NT], bl[NT];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        bh[ni] = __builtin_bit_cast(pc_f16x8, xp[ni * 32]);
        bl[ni] = __builtin_bit_cast(pc_f16x8, xp[PSZ + ni * 32]);
      }
      const pc_f16x8 ah = __builtin_bit_cast(pc_f16x8, use[0]), al = __builtin_bit_cast(pc_f16x8, use[1]);
#if DMEL_PC_SCHED
      // all six fragment reads (and the weight prefetch) are ISSUED before the first MFMA: left alone the compiler reads one fragment
      // into one register quad, waits, multiplies, and re-uses the quad for the next read -- five exposed LDS round trips per step
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        acl[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ni], acl[ni], 0, 0, 0);
        acl[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ni], acl[ni], 0, 0, 0);
        acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ni], acc[0][ni], 0, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(kWaitW);
#if DMEL_PC_ASMW
      __builtin_amdgcn_sched_barrier(0);
#endif
      if (++tap == taps) {
        tap = 0;
        ++ckl;
        if (s + 1 < steps) {
          if (ckl == nch) {                               // the next chunk opens the second segment
            ckl = 0; taps = a.seg[1].taps; dil = a.seg[1].dil; nch = nch1;
          }
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
#if DMEL_PC_ASMW
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail's redundant prefetches: nothing may still be landing in registers the epilogue re-uses
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][ni][r] = fmaf(acl[ni][r], 1.f / kF16LoScale, acc[0][ni][r]);
    conv_epilogue<1, NT, MODE, 8, MODE == EPI_LINEAR>(a, acc, (mblk * WM + wave_m) * 32, q0 + wave_n * (NT * 32) + l31, b, b / a.len_div, h);
    return;
  }

  // -------------------------------------------------------------------------------------------------- producer
  // A chunk = 16 channels x wx columns of one segment = 8 channel pairs x segments of up to 128 columns ("items", two passes of 64 lanes);
  // producer p takes items p, p + NPROD, ...  The loads of chunk c + 1 are in flight while chunk c is converted.
  const int p = wave_u - NC;
  constexpr int SEG = 128, NS = (XS + SEG - 1) / SEG, IP = (8 * NS + NPROD - 1) / NPROD;
  const int nchunks = nch0 + nch1;
  float pre[IP][2][2];
  auto fetch = [&](int ck) __attribute__((always_inline)) {
    const int sg = ck >= nch0 ? 1 : 0, chunk = ck - (sg ? nch0 : 0);
    const int T = (int)a.seg[sg].Tin, Cin = a.seg[sg].Cin, cs = (int)a.seg[sg].cstride;
    const int wx = BN + (a.seg[sg].taps - 1) * a.seg[sg].dil;
    const int tau0 = q0 - a.seg[sg].pad_left;
    const int nitems = 8 * ((wx + SEG - 1) / SEG);
    const float* xb = a.seg[sg].x + (int64_t)b * a.seg[sg].bstride;
#pragma unroll
    for (int r = 0; r < IP; ++r) {
      const int it = p + NPROD * r;
      if (it >= nitems) break;
      const int cp = it & 7, tb = tau0 + (it >> 3) * SEG;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float* xr = xb + (int64_t)min(chunk * 16 + 2 * cp + c, Cin - 1) * cs;
#pragma unroll
        for (int k = 0; k < 2; ++k) pre[r][c][k] = xr[min(max(tb + lane + 64 * k, 0), T - 1)];
      }
    }
  };
  fetch(0);
  for (int ck = 0; ck < nchunks; ++ck) {
    uint4* dst = Xb + (ck & 1) * (2 * PSZ);
    float cur[IP][2][2];
#pragma unroll
    for (int r = 0; r < IP; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 2; ++k) cur[r][c][k] = pre[r][c][k];
    if (ck + 1 < nchunks) fetch(ck + 1);
    const int sg = ck >= nch0 ? 1 : 0, chunk = ck - (sg ? nch0 : 0);
    const int T = (int)a.seg[sg].Tin, Cin = a.seg[sg].Cin;
    const int wx = BN + (a.seg[sg].taps - 1) * a.seg[sg].dil;
    const int tau0 = q0 - a.seg[sg].pad_left;
    const int nitems = 8 * ((wx + SEG - 1) / SEG);
    const float scale = a.seg[sg].in_scale * kF16XScale;
#pragma unroll
    for (int r = 0; r < IP; ++r) {
      const int it = p + NPROD * r;
      if (it >= nitems) break;
      const int cp = it & 7, j0 = (it >> 3) * SEG, w = min(SEG, wx - j0), tb = tau0 + j0;
      const float sc0 = chunk * 16 + 2 * cp < Cin ? scale : 0.f, sc1 = chunk * 16 + 2 * cp + 1 < Cin ? scale : 0.f;
      uint32_t* d32 = reinterpret_cast<uint32_t*>(dst) + ((2 * cp) >> 3) * (XS * 4) + (((2 * cp) & 7) >> 1);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int o = lane + 64 * k, t = tb + o;
        const bool ok = t >= 0 && t < T;
        {
          // the operand split of conv_bf16_kernel<NP = 2>::store_x: the scaled input is ROUNDED to fp32, then split (no contraction)
#pragma clang fp contract(off)
          const float v0 = (ok ? cur[r][0][k] : 0.f) * sc0, v1 = (ok ? cur[r][1][k] : 0.f) * sc1;
          const pc_f16x2 hi = __builtin_convertvector((pc_f32x2){v0, v1}, pc_f16x2);
          const pc_f16x2 lo = __builtin_convertvector((pc_f32x2){(v0 - (float)hi[0]) * kF16LoScale, (v1 - (float)hi[1]) * kF16LoScale}, pc_f16x2);
          if (o < w) {
            d32[(j0 + o) * 4] = __builtin_bit_cast(uint32_t, hi);
            d32[PSZ * 4 + (j0 + o) * 4] = __builtin_bit_cast(uint32_t, lo);
          }
        }
      }
    }
    __syncthreads();                                      // chunk ck is staged (and chunk ck - 1 has been consumed)
  }
}

template <int WM, int WN, int NT, int NPROD, int HALO, int MODE>
static int launch_pc(const KArgs& ka, int B, hipStream_t st) {
  constexpr int BN = WN * NT * 32, XS = BN + HALO;
  constexpr size_t lds = (size_t)2 * 2 * 2 * XS * 16;
  static_assert(lds <= 64 * 1024, "producer / consumer tile exceeds the default dynamic LDS limit");
  const int gx = (int)((ka.Tcols + BN - 1) / BN), gy = (ka.mtiles + WM - 1) / WM;
  if (gy > 65535 || B > 65535) { set_error("conv_pc: grid too large"); return DMEL_EINVAL; }
  hipLaunchKernelGGL((conv_pc_kernel<WM, WN, NT, NPROD, HALO, MODE>), dim3((unsigned)gx, (unsigned)gy, (unsigned)B), dim3(64 * (WM * WN + NPROD)), lds, st, ka);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

bool conv_pc_eligible(const PackedConv& pc, const ConvRun& r, bool any_size) {
  const PackDesc& d = pc.d;
  if (d.phases != 1 || r.out_tstride != 1 || r.phase_base != 0 || r.fold_pitch != 0 || r.row_scale || r.yp) return false;
  if (r.precision != DMEL_PRECISION_FP32_F16X2 || train_precision_override() == DMEL_PRECISION_BF16 || getenv("DMEL_CONV_FP32_MFMA")) return false;
  if (d.mode == EPI_LINEAR && (r.act != ACT_NONE || r.out_len)) return false;
  int halo = 0;
  for (int s = 0; s < d.nseg; ++s) {
    const SegDesc& sd = d.seg[s];
    if (sd.tstride != 1 || sd.toff != 0 || r.seg[s].tshift != 0 || r.seg[s].in_len || r.seg[s].in_absmax || r.seg[s].xp || !r.seg[s].x) return false;
    if (r.seg[s].Tin >= ((int64_t)1 << 30) || (int64_t)sd.Cin * r.seg[s].cstride >= ((int64_t)1 << 30)) return false;
    halo = std::max(halo, (sd.taps - 1) * sd.dil);
  }
  if (halo > 64 || (halo > 16 && d.mode != EPI_LINEAR)) return false;       // the paired modes are built with the 16-column halo only
  if (halo == 0) return false;      // pointwise convolutions: a staged chunk is ONE K step here (a barrier per step); conv_bf16_kernel stages 32
                                    // channels per barrier for them and stays ahead (wn_dec_1x1: 43 vs 46 us)
  // few columns (one stream of the streaming decoder): a 256 x 96 tile per workgroup leaves most CUs without one; conv_bf16_kernel's
  // small-N tile choice (pick_tile_bf16) serves those launches
  const int64_t wgs = (int64_t)((pc.Mpad + 255) / 256) * ((r.Tcols + 95) / 96) * r.B;
  if (wgs < 128 && !any_size) return false;
  return pc.Mpad / 32 >= 5;                                                   // eight strips of 32 rows per workgroup: the layout that gains
}

int launch_conv_pc(const PackedConv& pc, const ConvRun& r, hipStream_t stream) {
  DMEL_CHECK_ARG(conv_pc_eligible(pc, r, true), "conv_pc: this convolution is outside what the producer / consumer kernel is built for");
  DMEL_CHECK_ARG(r.y && r.B > 0 && r.Tcols > 0, "conv_pc: bad output / shape");
  const PackDesc& d = pc.d;
  DMEL_CHECK_ARG(d.mode != EPI_RESSKIP || r.skip != nullptr, "conv_pc: skip buffer missing");
  KArgs ka{};
  ka.nseg = d.nseg;
  ka.steps = pc.steps;
  int halo = 0;
  double in_elems = 0.0;
  for (int s = 0; s < d.nseg; ++s) {
    const SegDesc& sd = d.seg[s];
    SegArgs& o = ka.seg[s];
    o.x = r.seg[s].x; o.bstride = r.seg[s].bstride; o.cstride = r.seg[s].cstride; o.Tin = r.seg[s].Tin;
    o.in_len = nullptr; o.in_scale = r.seg[s].in_scale; o.in_absmax = nullptr;
    o.Cin = sd.Cin; o.nchunk = (sd.Cin + kCK - 1) / kCK; o.taps = sd.taps; o.dil = sd.dil; o.pad_left = sd.pad_left; o.tstride = 1; o.toff = 0;
    halo = std::max(halo, (sd.taps - 1) * sd.dil);
    in_elems += (double)sd.Cin * (double)std::min<int64_t>(r.seg[s].Tin, r.Tcols + halo);
  }
  ka.w32h = pc.w32h.p; ka.bias = pc.bias.as<float>();
  ka.Tcols = r.Tcols; ka.mode = d.mode; ka.act = r.act; ka.C = d.C; ka.RP = pc.RP; ka.phases = 1;
  ka.out_tstride = 1; ka.phase_base = 0; ka.accumulate = r.accumulate; ka.len_div = r.len_div > 0 ? r.len_div : 1;
  ka.skip_first = r.skip_first; ka.out_div = r.out_div;
  ka.y = r.y; ka.y_bs = r.y_bs; ka.y_cs = r.y_cs; ka.Tout = r.Tout > 0 ? r.Tout : r.Tcols;
  ka.res = r.res; ka.res_bs = r.res_bs; ka.res_cs = r.res_cs; ka.skip = r.skip;
  DMEL_CHECK_ARG((int64_t)d.C * ka.y_cs < ((int64_t)1 << 31) && ka.Tout < ((int64_t)1 << 30), "conv_pc: one batch item of the output exceeds 32-bit offsets");
  ka.mtiles = pc.Mpad / 32;
  const double rows_real = d.mode == EPI_LINEAR ? (double)d.C : 2.0 * d.C;
  double out_elems;
  if (d.mode == EPI_LINEAR) out_elems = (double)d.C * (double)r.Tcols * (1.0 + (r.res ? 1.0 : 0.0) + (r.accumulate ? 1.0 : 0.0));
  else if (d.mode == EPI_GATE) out_elems = (double)d.C * (double)r.Tcols;
  else out_elems = (double)d.C * (double)r.Tcols * (r.skip_first ? 3.0 : 4.0);
  const double alg_bytes = 4.0 * r.B * (in_elems + out_elems) + (double)pc.Mpad * pc.steps * kCK * 4.0;
  const double alg_flops = 2.0 * r.B * (double)r.Tcols * rows_real * pc.k_real;
  ProfScope ps("conv_igemm", stream, alg_flops, alg_bytes, alg_flops * 3.0);
  switch (d.mode) {
    case EPI_GATE: return launch_pc<8, 1, 3, 4, 16, EPI_GATE>(ka, r.B, stream);
    case EPI_RESSKIP: return launch_pc<8, 1, 3, 4, 16, EPI_RESSKIP>(ka, r.B, stream);
    default:
      if (halo <= 16) return launch_pc<8, 1, 3, 4, 16, EPI_LINEAR>(ka, r.B, stream);
      return launch_pc<8, 1, 3, 4, 64, EPI_LINEAR>(ka, r.B, stream);
  }
}

}  // namespace dmel
