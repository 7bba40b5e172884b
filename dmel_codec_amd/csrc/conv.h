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
};

// get_w(seg, src_row, ci, tap) returns the source weight; get_b(src_row) the bias (0 if none).
// Source row numbering: LINEAR: phase*C + co ; paired: kind*C + c.
template <class FW, class FB> int pack_conv(PackedConv& pc, const PackDesc& d, FW get_w, FB get_b);

struct SegRun {
  const float* x = nullptr;
  int64_t bstride = 0, cstride = 0, Tin = 0;
  const int64_t* in_len = nullptr;
  float in_scale = 1.f;
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
};

int launch_conv(const PackedConv& pc, const ConvRun& r, hipStream_t stream);

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
  DMEL_TRY(pc.w.upload(w.data(), w.size() * sizeof(float)));
  DMEL_TRY(pc.bias.upload(b.data(), b.size() * sizeof(float)));
  return DMEL_OK;
}

}  // namespace dmel
