// Device-side pieces shared by the convolution kernels (conv_igemm.hip, conv_snake.hip): kernel arguments, the epilogue, the
// workgroup -> tile map.  Internal to the library.
#pragma once
#include "conv.h"

namespace dmel {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct SegArgs {
  const float* x;
  int64_t bstride, cstride, Tin;
  const int64_t* in_len;
  float in_scale;
  int Cin, nchunk, taps, dil, pad_left, tstride, toff;
  const uint32_t* in_absmax;    // fp16 split over a gradient tensor (segment 0 only): see conv.h SegRun
  // Pre-split input (conv.h SegRun::xp): the operand already as two fp16 planes in B-fragment order, [piece][b][channel / 8][t] x 16 bytes
  const uint4* xp;
  int64_t xp_plane;             // uint4 between the two pieces
  int xp_g8;                    // 8-channel groups per batch item
};

struct KArgs {
  SegArgs seg[2];
  int nseg, steps, mtiles;
  int gx, gy, gz, xcd_chunk;   // logical grid (column tiles, m blocks, batch); xcd_chunk > 0: 1-D XCD-chunked launch
  const float* w;
  const void* w16;
  const void* w48;
  const void* w32h;
  const float* bias;
  int64_t Tcols;
  int mode, act, C, RP, phases, out_tstride, phase_base, accumulate, len_div, skip_first;
  float out_div;
  float* y;
  int64_t y_bs, y_cs, Tout;
  const float* res;
  int64_t res_bs, res_cs;
  const float* row_scale;
  const int64_t* out_len;
  float* skip;
  int fold_pitch, fold_valid;
  // paired modes: also (yp_only: instead) write the output as pre-split fp16 planes for the convolution that reads it next
  uint4* yp;
  int64_t yp_plane;
  int yp_g8, yp_only;
};

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case ACT_SILU: return v / (1.f + expf(-v));
    case ACT_TANH: return tanhf(v);
    case ACT_CLAMP1: return fminf(fmaxf(v, -1.f), 1.f);
    case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    default: return v;
  }
}

// folded batch (ConvRun::fold_pitch): is output column q one of the zero columns between two items?
__device__ __forceinline__ bool in_gap(const KArgs& a, int q) {
  return a.fold_pitch > 0 && (q % a.fold_pitch) >= a.fold_valid;
}

// ------------------------------------------------------------------ epilogue (shared by both kernels)
// Lane (h, l31) holds, for each 32x32 tile, column l31 and rows (r&3) + 8*(r>>2) + 4*h, r = 0..15.
// Row-only quantities (bias, row offsets, channel map) are computed once per row, outside the column loop.
// LEAN_ONLY: the caller guarantees the lean LINEAR case (no activation, row scale, phases or output stride): the general loop is not compiled
template <int MT, int NT, int MODE, int RB = 4, bool LEAN_ONLY = false>      // RB: rows whose residual / running-sum loads are in flight together (lean LINEAR path)
__device__ __forceinline__ void conv_epilogue(const KArgs& a, floatx16 (&acc)[MT][NT], int mrow0, int colbase, int b, int lb,
                                              int h) {
  const int tcols = (int)a.Tcols;
  if (MODE == EPI_LINEAR) {
    const int olim = a.out_len ? (int)min(a.out_len[lb], (int64_t)0x7fffffff) : 0x7fffffff;
    float* yb = a.y + (int64_t)b * a.y_bs;
    const float* rb = a.res ? a.res + (int64_t)b * a.res_bs : nullptr;
    const int ycs = (int)a.y_cs, rcs = (int)a.res_cs, tout = (int)a.Tout;
    // The common case (plain conv, optionally + residual: 5 of 6 vocoder convs, every WaveNet projection) gets a loop
    // with no per-element flag tests: the general loop below spends more time in uniform branches than in stores.
    const bool lean = LEAN_ONLY || (a.act == ACT_NONE && !a.row_scale && a.phases == 1 && a.out_tstride == 1 && a.phase_base == 0);
    if (lean) {
      const int tlim = min(tcols, tout);
      bool colok[NT], live[NT];
      int cq[NT];               // column offsets clamped into the row, so residual loads need no predicate
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        colok[ni] = colbase + ni * 32 < tlim;
        live[ni] = colbase + ni * 32 < olim && !in_gap(a, colbase + ni * 32);
        cq[ni] = min(colbase + ni * 32, tlim - 1);
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int mtile = mrow0 + mi * 32;
        if (mtile >= a.mtiles * 32) continue;
        // RB rows at a time: all their residual / running-sum loads are issued before the first use, so the wave meets the HBM latency
        // 16 / RB times per tile.  The fp16-split kernel asks for 8 (its second accumulator set is dead here: -0.15 ms per bench step); for the
        // others 8 rows cost 32 registers and a wave of occupancy (100 -> 132 VGPRs on the 128 x 96 tile), so they keep 4.
#pragma unroll
        for (int g = 0; g < 16 / RB; ++g) {
          float bias[RB], rv[RB][NT];
          int co[RB];
          float ov[RB][NT];          // previous output, for the running sum of the AMP branches (accumulate)
#pragma unroll
          for (int k = 0; k < RB; ++k) {
            const int r = g * RB + k;                                  // accumulator register -> row (r & 3) + 8 (r >> 2) + 4 h
            co[k] = mtile + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int cc = min(co[k], a.C - 1);
            bias[k] = a.bias[cc];
            if (rb) {
#pragma unroll
              for (int ni = 0; ni < NT; ++ni) rv[k][ni] = rb[cc * rcs + cq[ni]];
            }
            if (a.accumulate) {
#pragma unroll
              for (int ni = 0; ni < NT; ++ni) ov[k][ni] = yb[cc * ycs + cq[ni]];
            }
          }
#pragma unroll
          for (int k = 0; k < RB; ++k) {
            if (co[k] >= a.C) continue;
            float* yrow = yb + co[k] * ycs + colbase;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
              float v = acc[mi][ni][g * RB + k] + bias[k];
              if (rb) v += rv[k][ni];
              if (a.accumulate) v += ov[k][ni];
              if (a.out_div != 1.f) v = v / a.out_div;
              if (colok[ni]) yrow[ni * 32] = live[ni] ? v : 0.f;
            }
          }
        }
      }
      return;
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int mtile = mrow0 + mi * 32;
      if (mtile >= a.mtiles * 32) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mtile + (r & 3) + 8 * (r >> 2) + 4 * h;
        int ph = 0, co = m;
        if (a.phases > 1) { ph = m / a.RP; co = m - ph * a.RP; }
        if (co >= a.C || ph >= a.phases) continue;
        const float bias = a.bias[m];
        const float rs = a.row_scale ? a.row_scale[co] : 1.f;
        const int tph = a.phase_base + ph;
        float* yrow = yb + co * ycs;
        const float* rrow = rb ? rb + co * rcs : nullptr;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          const int q = colbase + ni * 32;
          const int t = q * a.out_tstride + tph;
          if (q >= tcols || t >= tout) continue;
          float v = act_apply(acc[mi][ni][r] + bias, a.act) * rs;
          if (rrow) v += rrow[t];
          if (a.accumulate) v += yrow[t];
          if (a.out_div != 1.f) v = v / a.out_div;
          if (t >= olim || in_gap(a, q)) v = 0.f;
          yrow[t] = v;
        }
      }
    }
  } else {
    float* yb = a.y + (int64_t)b * a.y_bs;
    float* sb = (MODE == EPI_RESSKIP) ? a.skip + (int64_t)b * a.y_bs : nullptr;
    const int ycs = (int)a.y_cs;
    bool gap[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) gap[ni] = in_gap(a, colbase + ni * 32);
    typedef _Float16 ep_f16x2 __attribute__((ext_vector_type(2)));
    typedef float ep_f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int mtile = mrow0 + mi * 32;
      if (mtile >= a.mtiles * 32) continue;
      const int q32 = mtile >> 5;
      // A lane holds, of each 8-channel group of the tile, FOUR consecutive channels (4 h .. 4 h + 3: accumulator registers r0 .. r0 + 3,
      // partner rows in r0 + 4 .. r0 + 7), i.e. one 8-byte half of the group's 16-byte unit in the pre-split plane layout.
#pragma unroll
      for (int gs = 0; gs < 2; ++gs) {
        const int r0 = gs * 8;
        const int c0 = q32 * 16 + gs * 8 + 4 * h;
        if (c0 >= a.C) continue;
        float b0[4], b1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rho = j + 8 * (r0 >> 2) + 4 * h;
          b0[j] = a.bias[mtile + rho];
          b1[j] = a.bias[mtile + rho + 8];
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          const int q = colbase + ni * 32;
          if (q >= tcols) continue;
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = 0.f;
            if (c0 + j >= a.C) continue;
            const float v0 = acc[mi][ni][r0 + j] + b0[j];
            const float v1 = acc[mi][ni][r0 + j + 4] + b1[j];
            float* yrow = yb + (c0 + j) * ycs;
            if (MODE == EPI_GATE) {
              o[j] = gap[ni] ? 0.f : (1.f / (1.f + expf(-v0))) * tanhf(v1);
              if (!a.yp_only) yrow[q] = o[j];
            } else {
              float* srow = sb + (c0 + j) * ycs;
              o[j] = gap[ni] ? 0.f : (yrow[q] + v0) / 1.41421356237309504880f;
              yrow[q] = o[j];
              srow[q] = gap[ni] ? 0.f : (a.skip_first ? v1 : srow[q] + v1);
            }
          }
          if (a.yp) {
            // the operand split of conv_bf16_kernel<NP = 2>::store_x, done once here instead of once per row block of the reader
#pragma clang fp contract(off)
            const float s0 = o[0] * kF16XScale, s1 = o[1] * kF16XScale, s2 = o[2] * kF16XScale, s3 = o[3] * kF16XScale;
            const ep_f16x2 h01 = __builtin_convertvector((ep_f32x2){s0, s1}, ep_f16x2), h23 = __builtin_convertvector((ep_f32x2){s2, s3}, ep_f16x2);
            const ep_f16x2 l01 = __builtin_convertvector((ep_f32x2){(s0 - (float)h01[0]) * kF16LoScale, (s1 - (float)h01[1]) * kF16LoScale}, ep_f16x2);
            const ep_f16x2 l23 = __builtin_convertvector((ep_f32x2){(s2 - (float)h23[0]) * kF16LoScale, (s3 - (float)h23[1]) * kF16LoScale}, ep_f16x2);
            uint2* dst = reinterpret_cast<uint2*>(a.yp + ((int64_t)b * a.yp_g8 + (c0 >> 3)) * tcols + q) + h;
            *dst = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
            *(dst + 2 * a.yp_plane) = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
          }
        }
      }
    }
  }
}

// Workgroup -> (column tile, m block, batch item).  Plain 3-D launch, or (xcd_chunk > 0) a 1-D launch in which the
// hardware's round-robin of consecutive workgroup ids over the 8 XCDs is undone: the m-block-major work list is cut
// into 8 contiguous chunks and XCD label j = id % 8 walks chunk j, so one XCD's L2 holds a contiguous 1/8 of the
// weight rows (and re-reads each x tile for its few m blocks) instead of every XCD streaming all weights through the
// Infinity Cache.  Purely a locality choice: any placement gives the same result.
__device__ __forceinline__ bool conv_block_coords(const KArgs& a, int& tile_n, int& mblk, int& b) {
  if (a.xcd_chunk <= 0) {
    tile_n = blockIdx.x; mblk = blockIdx.y; b = blockIdx.z;
    return true;
  }
  const int id = blockIdx.x;
  const int w = (id & 7) * a.xcd_chunk + (id >> 3);
  const int per_m = a.gx * a.gz;
  if ((id >> 3) >= a.xcd_chunk || w >= per_m * a.gy) return false;
  mblk = w / per_m;
  const int rest = w - mblk * per_m;
  b = rest / a.gx;
  tile_n = rest - b * a.gx;
  return true;
}

}  // namespace dmel
