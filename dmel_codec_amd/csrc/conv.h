// Internal API of the implicit-GEMM 1-D convolution kernel (conv_igemm.hip).
#pragma once
#include "common.h"

namespace dmel {

constexpr int kCK = 16;  // input channels per K step

enum EpiMode : int {
  EPI_LINEAR = 0,   // y[co] = f(acc + bias) ...           identity row map (optionally phase-major rows)
  EPI_GATE = 1,     // z[c]  = sigmoid(acc_gate) * tanh(acc_filter)        paired row map
  EPI_RESSKIP = 2,  // x[c]  = (x[c] + acc_res) / sqrt2 ; skip[c] (+)= acc_skip   paired row map
};
enum Act : int { ACT_NONE = 0, ACT_SILU = 1, ACT_TANH = 2, ACT_CLAMP1 = 3, ACT_GELU = 4 };

// One K segment = one input tensor contributing Cin*taps reduction rows.
struct SegDesc {
  int Cin = 0, taps = 1, dil = 1, pad_left = 0, tstride = 1, toff = 0;
};

// Host description of how source weight rows map to packed rows.
struct PackDesc {
  EpiMode mode = EPI_LINEAR;
  int nseg = 1;
  SegDesc seg[2];
  int C = 0;               // paired modes: channels C (source rows 2C); LINEAR: rows per phase (Cout)
  int phases = 1;          // LINEAR only: phase-major row blocks (transposed conv)
};

struct PackedConv {
  PackDesc d;
  int Mpad = 0, steps = 0, RP = 0;   // Mpad: packed rows, multiple of 32
  double k_real = 0;       // sum over segments of Cin*taps (algorithmic reduction length)
  DevBuf w, bias;
  DevBuf w16;              // the same weights rounded to bf16 (RNE), in 32x32x16 MFMA A-fragment order (DMEL_PRECISION_BF16)
  DevBuf w48;              // the same weights as three exact bf16 pieces (truncation split), same order (split-fp32 kernel)
  DevBuf w32h;             // the same weights x 2^6 as two fp16 pieces (hi = fp16(w), lo = fp16((w - hi) 2^11)), same order (fp16-split kernel)
};

// Device-side re-pack (training: the weights change every optimiser step and live on the device).  A segment's source is an
// affine view of a device tensor: value(src_row, ci, tap) = w[src_row*rs + ci*cs + (rev ? taps-1-tap : tap)*ts]; the bias of
// a source row is b0[row] (+ b1[row]), or 0.  launch_repack rewrites pc.w / pc.w16 / pc.w48 / pc.bias in place from such
// views, producing bit-identical images to pack_conv on the same values (tests hold it to that).
struct RepackSeg {
  const float* w = nullptr;
  int64_t rs = 0, cs = 0, ts = 0;
  int rev = 0;
  // phase-major packed rows (transposed convs): source row sr = ph * pC + c addresses w[c * rs + ph * ps + ...] when pC > 0
  int64_t ps = 0;
  int pC = 0;
};
struct RepackSrc {
  RepackSeg seg[2];
  const float* b0 = nullptr;
  const float* b1 = nullptr;
  int bias_mod = 0;          // > 0: the bias of source row sr is b0[sr % bias_mod] (one bias per output channel, shared by the phases)
};
int launch_repack(PackedConv& pc, const RepackSrc& src, hipStream_t stream);
// While a RepackBatch of this thread is alive, launch_repack only records its job; flush() sends all of them as ONE launch (a refresh
// of a module is 2-120 convolutions: one launch instead of one each).  Nested batches join the outermost one.
struct RepackBatch {
  explicit RepackBatch(hipStream_t s);
  ~RepackBatch();
  int flush();
  hipStream_t stream;
  bool owner;
};

// get_w(seg, src_row, ci, tap) returns the source weight; get_b(src_row) the bias (0 if none).
// Source row numbering: LINEAR: phase*C + co ; paired: kind*C + c.
template <class FW, class FB> int pack_conv(PackedConv& pc, const PackDesc& d, FW get_w, FB get_b);

struct SegRun {
  const float* x = nullptr;
  int64_t bstride = 0, cstride = 0, Tin = 0;
  const int64_t* in_len = nullptr;
  float in_scale = 1.f;
  int64_t tshift = 0;      // output column q reads input column (q + tshift) * tstride + ...: a launch over a sub-range of a longer row
  // fp16-split launches over a GRADIENT tensor (single segment): bit pattern of its max |value| in device memory (launch_absmax on the same
  // stream); the kernel stages x * 2^(13 - exponent) instead of x * 2^-6 and undoes it on the accumulators.  nullptr: activations.
  const uint32_t* in_absmax = nullptr;
  // Pre-split operand (fp16-split launches only): the input ALREADY as two planes of fp16 pieces in the kernel's B-fragment order,
  // [piece][b][channel / 8][t] x (8 x fp16 = 16 bytes), values x 2^-6, second piece x 2^11 -- what the staging code would have produced.
  // Written by the producing kernel's epilogue (ConvRun::yp) or by launch_split_planes.  The staging pass then copies 16-byte units
  // instead of converting 8 floats per unit, once per row block of the weights.  xp != nullptr: x is ignored; channels % 8 == 0.
  const void* xp = nullptr;
  int64_t xp_plane = 0;      // 16-byte units between the two planes (= B * channels / 8 * T)
};

struct ConvRun {
  SegRun seg[2];
  int B = 0;
  int64_t Tcols = 0;       // output columns per batch item (before out_tstride)
  // outputs
  float* y = nullptr;
  int64_t y_bs = 0, y_cs = 0, Tout = 0;
  int out_tstride = 1, phase_base = 0;
  Act act = ACT_NONE;
  const float* res = nullptr;     // added after activation/row scale, indexed like y
  int64_t res_bs = 0, res_cs = 0;
  const float* row_scale = nullptr;
  int accumulate = 0;             // v += y_old
  float out_div = 1.f;            // v /= out_div
  const int64_t* out_len = nullptr;
  int len_div = 1;                // lengths index = b / len_div (both in_len and out_len)
  float* skip = nullptr;          // RESSKIP
  int skip_first = 0;
  int precision = 0;              // DMEL_PRECISION_* (include/dmel_hip.h)
  // Folded batch ("B = 1" launches over a (C, N * fold_pitch) buffer that holds N short items side by side, fold_valid real columns
  // each followed by fold_pitch - fold_valid ZERO columns that play the role of the convolution's zero padding): outputs in the gap
  // columns are written as zeros so the invariant survives the layer.  0 = off.
  int fold_pitch = 0, fold_valid = 0;
  // paired modes (GATE / RESSKIP): write the result (also: yp_only = instead of y) as pre-split planes [piece][b][C / 8][Tcols], for the
  // convolution that consumes it (SegRun::xp).  C % 8 == 0.
  void* yp = nullptr;
  int64_t yp_plane = 0;
  int yp_only = 0;
};

int launch_conv(const PackedConv& pc, const ConvRun& r, hipStream_t stream);
// x (B, C, T) fp32 -> the two pre-split planes (SegRun::xp layout), columns at or behind len[b / len_div] written as zero; C % 8 == 0
int launch_split_planes(const float* x, void* planes, int64_t plane_units, const int64_t* len, int len_div, int B, int C, int64_t T,
                        hipStream_t stream);

// ---- host packing (template, header-only) ---------------------------------------------------
template <class FW, class FB> int pack_conv(PackedConv& pc, const PackDesc& d, FW get_w, FB get_b) {
  pc.d = d;
  const bool paired = d.mode != EPI_LINEAR;
  int M;
  if (paired) {
    M = (int)align_up(d.C, 16) * 2;
    pc.RP = M;
  } else {
    pc.RP = d.phases > 1 ? (int)align_up(d.C, 32) : d.C;
    M = pc.RP * d.phases;
  }
  pc.Mpad = (int)align_up(M, 32);
  pc.steps = 0;
  pc.k_real = 0;
  for (int s = 0; s < d.nseg; ++s) {
    pc.steps += (int)((d.seg[s].Cin + kCK - 1) / kCK) * d.seg[s].taps;
    pc.k_real += (double)d.seg[s].Cin * d.seg[s].taps;
  }
  auto src_row = [&](int m) -> int {  // packed row -> source row or -1
    if (paired) {
      int q32 = m >> 5, rho = m & 31, grp = rho >> 3, i = rho & 7;
      int c = q32 * 16 + (grp >> 1) * 8 + i;
      if (c >= d.C) return -1;
      return (grp & 1) * d.C + c;
    }
    if (m >= pc.RP * d.phases) return -1;
    int ph = m / pc.RP, co = m % pc.RP;
    if (co >= d.C) return -1;
    return ph * d.C + co;
  };
  // Weight image in MFMA A-fragment order: [32-row tile][K step][half][lane 0..63][4].  Lane l = 32*h + r holds,
  // for half hf and j = 0..3, W[row 32*tile + r][k = 2*(4*hf + j) + h] of the step's 16 reduction rows, so a wave
  // fetches a step's fragments with two fully coalesced 1 KiB dwordx4 loads and no LDS round trip.
  std::vector<float> w((size_t)pc.Mpad * pc.steps * kCK, 0.f), b(pc.Mpad, 0.f);
  for (int m = 0; m < pc.Mpad; ++m) {
    int sr = src_row(m);
    if (sr < 0) continue;
    b[m] = get_b(sr);
    const int tile = m >> 5, r = m & 31;
    int step = 0;
    for (int s = 0; s < d.nseg; ++s) {
      const SegDesc& sd = d.seg[s];
      int nchunk = (sd.Cin + kCK - 1) / kCK;
      for (int ch = 0; ch < nchunk; ++ch)
        for (int tp = 0; tp < sd.taps; ++tp, ++step)
          for (int k = 0; k < kCK; ++k) {
            int ci = ch * kCK + k;
            if (ci >= sd.Cin) continue;
            const int h = k & 1, kk = k >> 1, hf = kk >> 2, j = kk & 3, lane = 32 * h + r;
            w[((((size_t)tile * pc.steps + step) * 2 + hf) * 64 + lane) * 4 + j] = get_w(s, sr, ci, tp);
          }
    }
  }
  // bf16 images for the 32x32x16 MFMA: [32-row tile][K step][piece][lane 0..63][8]; lane l = 32*h + r holds W[row r][k = 8*h + j].
  // w16: one piece, round-to-nearest-even.  w48: three pieces by truncation, w = p1 + p2 + p3 exactly (see conv_igemm.hip).
  {
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto from_bits = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    auto to_bf16 = [&](float f) -> uint16_t {
      uint32_t u = bits(f);
      if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
      return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);                    // round to nearest even
    };
    std::vector<uint16_t> w16((size_t)pc.Mpad * pc.steps * kCK, 0), w48((size_t)pc.Mpad * pc.steps * kCK * 3, 0);
    std::vector<uint16_t> w32h((size_t)pc.Mpad * pc.steps * kCK * 2, 0);
    for (int m = 0; m < pc.Mpad; ++m) {
      int sr = src_row(m);
      if (sr < 0) continue;
      const int tile = m >> 5, r = m & 31;
      int step = 0;
      for (int s2 = 0; s2 < d.nseg; ++s2) {
        const SegDesc& sd = d.seg[s2];
        int nchunk = (sd.Cin + kCK - 1) / kCK;
        for (int ch = 0; ch < nchunk; ++ch)
          for (int tp = 0; tp < sd.taps; ++tp, ++step)
            for (int k = 0; k < kCK; ++k) {
              int ci = ch * kCK + k;
              if (ci >= sd.Cin) continue;
              const int h = k >> 3, j = k & 7, lane = 32 * h + r;
              const float v = get_w(s2, sr, ci, tp);
              w16[(((size_t)tile * pc.steps + step) * 64 + lane) * 8 + j] = to_bf16(v);
              const float p1 = from_bits(bits(v) & 0xffff0000u), r1 = v - p1;
              const float p2 = from_bits(bits(r1) & 0xffff0000u), r2 = r1 - p2;
              const size_t base = ((size_t)tile * pc.steps + step) * 3;
              w48[((base + 0) * 64 + lane) * 8 + j] = (uint16_t)(bits(p1) >> 16);
              w48[((base + 1) * 64 + lane) * 8 + j] = (uint16_t)(bits(p2) >> 16);
              w48[((base + 2) * 64 + lane) * 8 + j] = (uint16_t)(bits(r2) >> 16);
              const float vs = v * kF16WScale;
              const uint16_t hi = f32_to_f16_bits(vs);
              const size_t bh = ((size_t)tile * pc.steps + step) * 2;
              w32h[((bh + 0) * 64 + lane) * 8 + j] = hi;
              w32h[((bh + 1) * 64 + lane) * 8 + j] = f32_to_f16_bits((vs - f16_bits_to_f32(hi)) * kF16LoScale);
            }
      }
    }
    DMEL_TRY(pc.w16.upload(w16.data(), w16.size() * sizeof(uint16_t)));
    DMEL_TRY(pc.w48.upload(w48.data(), w48.size() * sizeof(uint16_t)));
    DMEL_TRY(pc.w32h.upload(w32h.data(), w32h.size() * sizeof(uint16_t)));
  }
  DMEL_TRY(pc.w.upload(w.data(), w.size() * sizeof(float)));
  DMEL_TRY(pc.bias.upload(b.data(), b.size() * sizeof(float)));
  return DMEL_OK;
}

}  // namespace dmel

// Single-op handle of the C ABI (dmel_conv_*): forward image, and -- packed on first use -- the transposed, tap-reversed
// image that turns backward-data into the same convolution kernel.
struct dmel_conv {
  dmel::PackedConv pc;
  int Cout = 0, Cin = 0, k = 0, dil = 0;
  int precision = 0;
  std::vector<float> w_host;
  dmel::PackedConv pc_dgrad;
  bool dgrad_ready = false;
};
