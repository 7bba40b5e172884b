// Whole-WaveNet kernel for narrow, unconditioned stacks on short items (the dMel ENCODER: 10 -> 70 channels, 20 gated dilated blocks,
// 93 frames per (utterance, mel group) item): models/modules/wavenet.py:204-225 in ONE launch.
//
// The layered path issues 42 implicit-GEMM launches of ~24 us for the encoder of the bench step -- tiny GEMMs (160 x 210 x 93 per item)
// whose activations go to HBM and come back between any two of them: launch- and latency-bound at 30 TF/s.  Here one workgroup owns
// one item for the whole stack:
//   * x lives in LDS for all blocks: an fp32 master copy (the residual update must stay exact) and its three-way exact bf16 split in
//     B-fragment order [piece][8-channel group][column + 8 halo][8] -- the layout conv_bf16_kernel stages per chunk (conv_igemm.hip) --
//     so a tap is a column offset and the "same" zero padding is the zeroed halo;
//   * fifteen waves: (32-row tile of the paired (gate|filter, residual|skip) row layout) x (32-column block of the item), one 32 x 32
//     accumulator each -- a lone wave per tile spent more time in the sigmoid / tanh / divide epilogues than in its MFMAs (0.49 ms per
//     forward); spread over three waves per tile the epilogues of one wave hide behind the MFMAs of the others;
//   * per block: 15 K steps of the dilated k = 3 conv -> sigmoid * tanh -> z (split, LDS) -> 5 K steps of the 1x1 projection ->
//     x = (x + r) / sqrt2 (master + split), skip += s in REGISTERS for all 20 blocks;
//   * weights stream from L2 straight into registers in A-fragment order (the handle's existing split-bf16 images, untouched), prefetched
//     three K steps ahead across the gate -> projection -> next block boundaries;
//   * input projection (+ SiLU) in front, skip_projection behind, masks fused at both ends.
// Arithmetic is that of the layered path, term for term (same six partial products per 32 x 32 x 16 block in the same order, K walked
// chunk-major / tap-minor, same epilogue expressions), so the two paths produce IDENTICAL bits -- the token ids cannot move.
// Eligibility (else the layered path runs): residual channels <= 80 (five row tiles), no condition input, no output projection,
// T <= 96, dilations <= 8.  LDS: 128 KiB per workgroup (one per CU; a 1 s batch of 32 x 8 groups is exactly one round of 256).
#include "ops.h"

namespace dmel {

typedef float fx16 __attribute__((ext_vector_type(16)));
typedef __bf16 fbf16x8 __attribute__((ext_vector_type(8)));

constexpr int kFT = 96;                 // columns per item (three 32-column MFMA blocks)
constexpr int kFH = 8;                  // halo columns on each side (max dilation)
constexpr int kFXS = kFT + 2 * kFH;     // staged row length of x
constexpr int kFPD = 3;                 // weight prefetch distance in K steps (kFPD + 1 register sets)

struct FusedArgs {
  const float* x;            // (N, Cin, T)
  float* y;                  // (N, C, T)
  const int64_t* in_len;     // nullable, (N / len_div)
  const int64_t* out_len;    // nullable
  int len_div, N, T, Cin, C, L, cycle, has_in;
  float skip_scale;
  const void* in_w;          // split-bf16 image of input_projection (LINEAR rows), nullable
  const float* in_b;
  const void* skip_w;        // skip_projection (LINEAR rows)
  const float* skip_b;
  const void* const* gate_w; // [L] images of the gated convs (paired rows), device array
  const float* const* gate_b;
  const void* const* rs_w;   // [L] images of the residual / skip projections
  const float* const* rs_b;
};

__device__ __forceinline__ uint32_t f_pack_hi16(float lo, float hi) {
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
// exact three-way split of four consecutive channels -> one 8-byte LDS store per piece
__device__ __forceinline__ void split4_store(const float (&v)[4], char* p0, char* p1, char* p2) {
  float r[4], s[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    r[e] = v[e] - __uint_as_float(__float_as_uint(v[e]) & 0xffff0000u);
    s[e] = r[e] - __uint_as_float(__float_as_uint(r[e]) & 0xffff0000u);
  }
  *reinterpret_cast<uint2*>(p0) = make_uint2(f_pack_hi16(v[0], v[1]), f_pack_hi16(v[2], v[3]));
  *reinterpret_cast<uint2*>(p1) = make_uint2(f_pack_hi16(r[0], r[1]), f_pack_hi16(r[2], r[3]));
  *reinterpret_cast<uint2*>(p2) = make_uint2(f_pack_hi16(s[0], s[1]), f_pack_hi16(s[2], s[3]));
}
// eight consecutive channels of one column -> one 16-byte unit per piece (the staging form of conv_bf16_kernel::store_x)
__device__ __forceinline__ void split8_store(const float (&v)[8], uint4* d0, uint4* d1, uint4* d2) {
  uint32_t p1[4], p2[4], p3[4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const float r0 = v[e] - __uint_as_float(__float_as_uint(v[e]) & 0xffff0000u);
    const float r1 = v[e + 1] - __uint_as_float(__float_as_uint(v[e + 1]) & 0xffff0000u);
    const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u);
    const float s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    p1[e >> 1] = f_pack_hi16(v[e], v[e + 1]);
    p2[e >> 1] = f_pack_hi16(r0, r1);
    p3[e >> 1] = f_pack_hi16(s0, s1);
  }
  *d0 = make_uint4(p1[0], p1[1], p1[2], p1[3]);
  *d1 = make_uint4(p2[0], p2[1], p2[2], p2[3]);
  *d2 = make_uint4(p3[0], p3[1], p3[2], p3[3]);
}

// NCH = 16-channel chunks of the residual width (C <= 16 NCH); NG = 2 NCH eight-channel groups
template <int NCH>
__global__ __launch_bounds__(192 * NCH) void wavenet_fused_kernel(FusedArgs a) {
  constexpr int NG = 2 * NCH, CP = 16 * NCH, NW = 3 * NCH, NTHR = 64 * NW;
  constexpr int GS = 3 * NCH, RS = NCH, LS = GS + RS;             // K steps of the gated conv, of the projection, per block
  static_assert(LS % (kFPD + 1) == 0, "the weight register sets must rotate consistently across blocks");
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  uint4* Xp = reinterpret_cast<uint4*>(lds_raw);                  // [3][NG][kFXS]   x, split, with halo
  uint4* Zp = Xp + 3 * NG * kFXS;                                 // [3][NG][kFT]    gate output z / projection inputs, split
  float* Xf = reinterpret_cast<float*>(Zp + 3 * NG * kFT);        // [CP][kFT]       x, fp32 master
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int wave = wv / 3, nb = wv - 3 * wave;          // row tile, column block of this wave
  const int q = nb * 32 + l31;                          // this lane's column
  const int n = blockIdx.x, T = a.T, C = a.C;
  const int lb = n / a.len_div;
  const int ilim = a.in_len ? (int)min(a.in_len[lb], (int64_t)T) : T;
  const int olim = a.out_len ? (int)min(a.out_len[lb], (int64_t)T) : T;
  const uint32_t lane16 = lane * 16;

  // ---- zero the split buffers (halo columns, channel padding and columns >= T must read as zeros forever)
  for (int i = tid; i < 3 * NG * (kFXS + kFT); i += NTHR) Xp[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < CP * kFT; i += NTHR) Xf[i] = 0.f;
  __syncthreads();

  fx16 acc;
  auto zero_acc = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  };
  // one K step: A fragments (three pieces) in `w`, B fragments of 8-channel groups (2 c16 + h) at column offset `col0` of `src`
  auto mma_step = [&](const uint4 (&w)[3], const uint4* src, int row_len, int c16, int col0) {
    const uint4* bp = src + (2 * c16 + h) * row_len + col0 + q;
    fbf16x8 b[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(fbf16x8, bp[p * NG * row_len]);
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest partial products first (as conv_bf16_kernel)
#pragma unroll
    for (int t = 0; t < 6; ++t)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fbf16x8, w[PA[t]]), b[PB[t]], acc, 0, 0, 0);
  };
  auto load_w = [&](uint4 (&dst)[3], const void* image, int steps, int step) {
    const char* sp = reinterpret_cast<const char*>(image) + ((size_t)wave * steps + step) * 3072 + lane16;
#pragma unroll
    for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const uint4*>(sp + p * 1024);
  };
  // byte address of the 4-channel slot (c0 .. c0 + 3, c0 % 4 == 0) of column j in piece plane p of a split buffer
  auto slot = [&](uint4* buf, int row_len, int p, int c0, int j) -> char* {
    return reinterpret_cast<char*>(buf + (p * NG + (c0 >> 3)) * row_len + j) + (c0 & 7) * 2;
  };

  // ---- input: x0 = silu(input_projection(x * mask)) or x * mask            (wavenet.py:205-207)
  const float* xin = a.x + (int64_t)n * a.Cin * T;
  if (a.has_in) {
    // stage the raw input (<= 16 channels: one chunk) into Zp groups 0..1, one (group, column) item per thread
    for (int i = tid; i < 2 * kFT; i += NTHR) {
      const int g = i / kFT, q = i - g * kFT;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 8 * g + e;
        v[e] = (c < a.Cin && q < ilim) ? xin[(int64_t)c * T + q] : 0.f;
      }
      split8_store(v, Zp + (0 * NG + g) * kFT + q, Zp + (1 * NG + g) * kFT + q, Zp + (2 * NG + g) * kFT + q);
    }
    __syncthreads();
    const int rows = (C + 31) / 32;                      // LINEAR packing: identity rows, ceil(C / 32) tiles, one K step
    if (wave < rows) {
      uint4 w[3];
      load_w(w, a.in_w, 1, 0);
      zero_acc();
      mma_step(w, Zp, kFT, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; r += 4) {                  // rows m0 .. m0 + 3: four consecutive channels per lane
        const int m0 = wave * 32 + 8 * (r >> 2) + 4 * h;
        if (m0 < C) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u = acc[r + e] + a.in_b[m0 + e];
            v[e] = (q < T && m0 + e < C) ? u / (1.f + expf(-u)) : 0.f;
            Xf[(m0 + e) * kFT + q] = v[e];
          }
          split4_store(v, slot(Xp, kFXS, 0, m0, q + kFH), slot(Xp, kFXS, 1, m0, q + kFH), slot(Xp, kFXS, 2, m0, q + kFH));
        }
      }
    }
  } else {
    for (int i = tid; i < NG * kFT; i += NTHR) {
      const int g = i / kFT, q = i - g * kFT;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 8 * g + e;
        v[e] = (c < C && q < ilim) ? xin[(int64_t)c * T + q] : 0.f;
        Xf[c * kFT + q] = v[e];
      }
      split8_store(v, Xp + (0 * NG + g) * kFXS + q + kFH, Xp + (1 * NG + g) * kFXS + q + kFH, Xp + (2 * NG + g) * kFXS + q + kFH);
    }
  }
  __syncthreads();
  if (a.has_in) {
    // Zp held the raw input: clear the two groups again so that the gate output of channel-padded rows stays zero
    for (int i = tid; i < 3 * 2 * kFT; i += NTHR) {
      const int p = i / (2 * kFT), rem = i - p * (2 * kFT);
      Zp[(p * NG + rem / kFT) * kFT + rem % kFT] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- the blocks.  Weight stream: global K-step index k = block * LS + s; register set k % (kFPD + 1); prefetched kFPD steps ahead.
  float skip[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) skip[e] = 0.f;
  uint4 wa[kFPD + 1][3];
  auto fetch = [&](uint4 (&dst)[3], int block, int s) {      // step s of block `block` (s < LS); past the last block: harmless re-read
    const int bl = min(block, a.L - 1);
    if (s < GS) load_w(dst, a.gate_w[bl], GS, s);
    else load_w(dst, a.rs_w[bl], RS, s - GS);
  };
#pragma unroll
  for (int d = 0; d < kFPD; ++d) fetch(wa[d], 0, d);

  for (int blk = 0; blk < a.L; ++blk) {
    const int dil = a.cycle ? 1 << (blk % a.cycle) : 1;
    const float* gb = a.gate_b[blk];
    const float* rb = a.rs_b[blk];
    // gated dilated conv: chunk-major, tap-minor (pack_conv's step order)
    zero_acc();
#pragma unroll
    for (int s = 0; s < GS; ++s) {
      const int ns = s + kFPD;
      fetch(wa[ns % (kFPD + 1)], ns < LS ? blk : blk + 1, ns < LS ? ns : ns - LS);
      mma_step(wa[s % (kFPD + 1)], Xp, kFXS, s / 3, kFH + (s % 3 - 1) * dil);
    }
    // z = sigmoid(gate) * tanh(filter)                                        (wavenet.py:129-130)
#pragma unroll
    for (int r = 0; r < 16; r += 8) {                     // r = 0..3 -> channels c0..c0+3 of group A, r = 8..11 -> group B
      const int rho0 = 8 * (r >> 2) + 4 * h;              // first of four consecutive packed rows (gate); partner rows + 8 (filter)
      const int c0 = wave * 16 + ((rho0 >> 3) >> 1) * 8 + (rho0 & 7);
      if (c0 < C) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v0 = acc[r + e] + gb[wave * 32 + rho0 + e];
          const float v1 = acc[(r + e + 4) & 15] + gb[wave * 32 + rho0 + e + 8];
          v[e] = (q < T && c0 + e < C) ? (1.f / (1.f + expf(-v0))) * tanhf(v1) : 0.f;
        }
        split4_store(v, slot(Zp, kFT, 0, c0, q), slot(Zp, kFT, 1, c0, q), slot(Zp, kFT, 2, c0, q));
      }
    }
    __syncthreads();
    // residual / skip projection (1x1)                                         (wavenet.py:131-134)
    zero_acc();
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      const int ks = GS + s, ns = ks + kFPD;
      fetch(wa[ns % (kFPD + 1)], ns < LS ? blk : blk + 1, ns < LS ? ns : ns - LS);
      mma_step(wa[ks % (kFPD + 1)], Zp, kFT, s, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; r += 8) {
      const int rho0 = 8 * (r >> 2) + 4 * h;
      const int c0 = wave * 16 + ((rho0 >> 3) >> 1) * 8 + (rho0 & 7);
      if (c0 < C) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v0 = acc[r + e] + rb[wave * 32 + rho0 + e];
          const float v1 = acc[(r + e + 4) & 15] + rb[wave * 32 + rho0 + e + 8];
          const bool ok = q < T && c0 + e < C;
          float* xm = Xf + (c0 + e) * kFT + q;
          v[e] = ok ? (*xm + v0) / 1.41421356237309504880f : 0.f;
          *xm = v[e];
          const int si = (r >> 1) + e;
          skip[si] = ok ? (blk == 0 ? v1 : skip[si] + v1) : 0.f;
        }
        split4_store(v, slot(Xp, kFXS, 0, c0, q + kFH), slot(Xp, kFXS, 1, c0, q + kFH), slot(Xp, kFXS, 2, c0, q + kFH));
      }
    }
    __syncthreads();
  }

  // ---- skip_projection(sum of skips / sqrt(L))                                (wavenet.py:218-219), masked store
#pragma unroll
  for (int r = 0; r < 16; r += 8) {
    const int rho0 = 8 * (r >> 2) + 4 * h;
    const int c0 = wave * 16 + ((rho0 >> 3) >> 1) * 8 + (rho0 & 7);
    if (c0 < C) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (q < T && c0 + e < C) ? skip[(r >> 1) + e] * a.skip_scale : 0.f;
      split4_store(v, slot(Zp, kFT, 0, c0, q), slot(Zp, kFT, 1, c0, q), slot(Zp, kFT, 2, c0, q));
    }
  }
  __syncthreads();
  const int rows = (C + 31) / 32;
  if (wave < rows) {
    zero_acc();
    uint4 w[2][3];
    load_w(w[0], a.skip_w, RS, 0);
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      if (s + 1 < RS) load_w(w[(s + 1) & 1], a.skip_w, RS, s + 1);
      mma_step(w[s & 1], Zp, kFT, s, 0);
    }
    float* yb = a.y + (int64_t)n * C * T;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co >= C) continue;
      const float bias = a.skip_b[co];
      if (q < T) yb[(int64_t)co * T + q] = q < olim ? acc[r] + bias : 0.f;
    }
  }
}

template <int NCH> static int launch_fused_t(const FusedArgs& a, hipStream_t st) {
  constexpr int NG = 2 * NCH, CP = 16 * NCH;
  constexpr size_t lds = (size_t)3 * NG * (kFXS + kFT) * 16 + (size_t)CP * kFT * 4;
  static bool raised = false;
  if (!raised && lds > 64 * 1024) {
    DMEL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wavenet_fused_kernel<NCH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    raised = true;
  }
  hipLaunchKernelGGL(wavenet_fused_kernel<NCH>, dim3((unsigned)a.N), dim3(192 * NCH), lds, st, a);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

int launch_wavenet_fused(const WaveNetFused& f, const float* x, float* y, const int64_t* in_len, const int64_t* out_len, int len_div, int N,
                         int64_t T, hipStream_t st) {
  FusedArgs a{};
  a.x = x; a.y = y; a.in_len = in_len; a.out_len = out_len; a.len_div = len_div > 0 ? len_div : 1;
  a.N = N; a.T = (int)T; a.Cin = f.Cin; a.C = f.C; a.L = f.L; a.cycle = f.cycle; a.has_in = f.has_in;
  a.skip_scale = f.skip_scale;
  a.in_w = f.in_w; a.in_b = f.in_b; a.skip_w = f.skip_w; a.skip_b = f.skip_b;
  a.gate_w = reinterpret_cast<const void* const*>(f.table.p);
  a.gate_b = reinterpret_cast<const float* const*>(reinterpret_cast<const char*>(f.table.p) + (size_t)f.L * sizeof(void*));
  a.rs_w = reinterpret_cast<const void* const*>(reinterpret_cast<const char*>(f.table.p) + (size_t)2 * f.L * sizeof(void*));
  a.rs_b = reinterpret_cast<const float* const*>(reinterpret_cast<const char*>(f.table.p) + (size_t)3 * f.L * sizeof(void*));
  const int nch = (f.C + 15) / 16;
  double flops = 2.0 * N * (double)T * ((double)f.L * (2.0 * f.C * 3 * f.C + 2.0 * f.C * f.C) + (double)f.C * f.C + (f.has_in ? (double)f.C * f.Cin : 0.0));
  const double wbytes = ((double)f.L * (2.0 * ((f.C + 15) / 16 * 16)) * (4.0 * ((f.C + 15) / 16 * 16)) + 2.0 * 96 * 96) * 6.0;
  ProfScope ps("conv_igemm", st, flops, 4.0 * N * (double)T * (f.Cin + f.C) + wbytes, 6.0 * flops);
  switch (nch) {
    case 5: return launch_fused_t<5>(a, st);
    case 4: return launch_fused_t<4>(a, st);
    case 3: return launch_fused_t<3>(a, st);
    default: set_error("wavenet_fused: %d channels not instantiated", f.C); return DMEL_EUNSUPPORTED;
  }
}

}  // namespace dmel
