// Backward of the dense 1-D convolution (first piece of SURVEY.md section 8(f) rank 1: the training path).
//
// The reference has no hand-written backward: `loss.backward()` runs ATen's conv1d double-backward kernels through autograd
// (codec_lit_modules.py:236,315 -> torch/nn/functional conv1d).  For y = conv1d(x, W, b, dilation d, "same" padding p):
//   dx[ci, t]      = sum_{co, k} W[co, ci, K-1-k] * dy[co, t + k*d - p]         backward-data: the SAME convolution with the
//                                                                               weights transposed and tap-reversed
//   dW[co, ci, k]  = sum_{b, t} dy[b, co, t] * x[b, ci, t + k*d - p]            backward-weight: a GEMM over (b, t)
//   db[co]         = sum_{b, t} dy[b, co, t]
// backward-data therefore reuses conv_igemm.hip unchanged (fp32-grade, the forward kernel's tiles and rates).
// backward-weight is its own kernel below: per tap a (Cout x Cin) GEMM whose reduction runs over batch x time, on the fp32
// MFMA (exact fp32 products; gradients are summed over ~10^5 samples, so accumulation order is the only noise), split over
// K across workgroups with fp32 atomics.
#include "ops.h"

#include <cstdlib>
#include <type_traits>

namespace dmel {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kWgTile = 64;     // workgroup tile: 64 output channels x 64 input channels of one tap
constexpr int kWgK = 64;        // time samples staged per step
constexpr int kWgPitch = 68;    // LDS row pitch in floats: rows 16-byte aligned for 128-bit fragment reads, 4 banks apart

struct WgArgs {
  const float* dy;
  const float* x;
  float* dw;
  int Cout, Cin, taps, dil, pad, B, T;
  int chunks_per_item, slices, chunks_per_slice;
  // strided form (k2s2 conv / transposed conv): x[b, ci, t * xstride + xoff] of rows of length Tx; the result of the single
  // launch tap goes to dw[(co * Cin + ci) * taps_out + tap_out]
  int xstride, xoff, Tx, taps_out, tap_out;
  // short rows (T <= 32: the deep discriminator layers have 12-24 frames per image row): ipc = 64 / T batch items share one staged
  // 64-sample step, lane -> (item lane / T, sample lane % T); 1 = one item per step
  int ipc;
  // fp16-split instantiation: bit pattern of max |dy| of the launch (device memory, written by absmax_kernel on the same stream)
  const uint32_t* dy_absmax;
  // three-tap kernel: raw accumulator tiles of every workgroup ([workgroup][tap][wave][16][64] floats), summed over the K slices by
  // wgrad_reduce_kernel -- no atomics (thousands of device-scope float atomics on a few 10^5 addresses cost more than the GEMM)
  float* partial;
};

// grid (ceil(Cin/64), ceil(Cout/64), taps * slices); 256 threads = 2 x 2 waves of 32 x 32
// PACK = 1: several short batch items per staged step (per-lane item index).  PACK = 2: the reduction runs over the FLATTENED sample index
// n = b * T + t in steps of 64, whatever T is (a 92-frame row wastes 30 % of its second 64-sample step otherwise): lane -> (n / T, n % T).
// PACK = 0: the item is wave-uniform.
template <int PACK>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgArgs a) {
  __shared__ __attribute__((aligned(16))) float dys[kWgTile * kWgPitch];
  __shared__ __attribute__((aligned(16))) float xs[kWgTile * kWgPitch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ci0 = blockIdx.x * kWgTile, co0 = blockIdx.y * kWgTile;
  const int tap = blockIdx.z % a.taps, slice = blockIdx.z / a.taps;
  const int shift = tap * a.dil - a.pad;
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int total = PACK == 1 ? (a.B + a.ipc - 1) / a.ipc : PACK == 2 ? (a.B * a.T + kWgK - 1) / kWgK : a.B * a.chunks_per_item;
  const int c_begin = slice * a.chunks_per_slice, c_end = min(total, c_begin + a.chunks_per_slice);
  const int r31 = lane & 31, h = lane >> 5;
  const int lane_item = PACK == 1 ? lane / a.T : 0, lane_t = PACK == 1 ? lane - lane_item * a.T : lane;
  // global loads of step c into registers (clamped addresses, validity kept as bit masks and applied when the values are staged):
  // they are issued before the MFMA loop of step c - 1, so their latency hides behind it
  float vd[16], vx[16];
  unsigned md = 0, mx = 0;
  auto fetch = [&](int c) {
    int b, t0;
    bool lane_ok = true;
    if (PACK == 1) {                     // c indexes groups of ipc items; every lane has its own item
      b = c * a.ipc + lane_item;
      t0 = 0;
      lane_ok = lane_item < a.ipc && b < a.B;
      b = min(b, a.B - 1);
    } else if (PACK == 2) {              // c indexes 64 flattened samples
      const int n = c * kWgK + lane;
      b = n / a.T;
      t0 = n - b * a.T - lane_t;         // so that t below is n % T
      lane_ok = b < a.B;
      b = min(b, a.B - 1);
    } else {
      b = c / a.chunks_per_item;
      t0 = (c - b * a.chunks_per_item) * kWgK;
    }
    const float* dyb = a.dy + (int64_t)b * a.Cout * a.T;
    const float* xb = a.x + (int64_t)b * a.Cin * a.Tx;
    const int t = t0 + lane_t, tx = t * a.xstride + a.xoff + shift;
    const bool tok = lane_ok && t < a.T, txok = tok && tx >= 0 && tx < a.Tx;
    const int tc = min(t, a.T - 1), txc = min(max(tx, 0), a.Tx - 1);
    md = 0; mx = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {       // one 256-byte row segment per wave and pass
      const int row = wave + 4 * i;
      const int co = co0 + row, ci = ci0 + row;
      vd[i] = dyb[(int64_t)min(co, a.Cout - 1) * a.T + tc];
      vx[i] = xb[(int64_t)min(ci, a.Cin - 1) * a.Tx + txc];
      md |= (unsigned)(tok && co < a.Cout) << i;
      mx |= (unsigned)(txok && ci < a.Cin) << i;
    }
  };
  if (c_begin < c_end) fetch(c_begin);
  for (int c = c_begin; c < c_end; ++c) {
    __syncthreads();      // the previous step's fragment reads are done
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = wave + 4 * i;
      dys[row * kWgPitch + lane] = (md >> i) & 1 ? vd[i] : 0.f;
      xs[row * kWgPitch + lane] = (mx >> i) & 1 ? vx[i] : 0.f;
    }
    __syncthreads();
    if (c + 1 < c_end) fetch(c + 1);
    // MFMA j of lane half h multiplies staged sample 32 h + j (any pairing of samples works as long as both operands agree),
    // so every lane reads 32 contiguous floats of its row
    const float4* ap = reinterpret_cast<const float4*>(dys + (wm * 32 + r31) * kWgPitch + 32 * h);
    const float4* bp = reinterpret_cast<const float4*>(xs + (wn * 32 + r31) * kWgPitch + 32 * h);
    float4 av[8], bv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { av[q] = ap[q]; bv[q] = bp[q]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bv[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bv[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bv[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bv[q].w, acc, 0, 0, 0);
    }
  }
  // D[row = co][col = ci]: acc[r] <-> row (r & 3) + 8 * (r >> 2) + 4 * h, col = lane & 31
  const int ci = ci0 + wn * 32 + r31;
  if (ci < a.Cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co < a.Cout) atomicAdd(a.dw + ((int64_t)co * a.Cin + ci) * a.taps_out + a.tap_out + tap, acc[r]);
    }
  }
}

// ---- split-fp32 variant ------------------------------------------------------------------------------------------------
// The same GEMM on the bf16 matrix core with every fp32 operand written as the exact sum of three bf16 pieces (conv_igemm.hip's
// arithmetic: six partial products per 32 x 32 x 16 block, dropped terms < 2^-21 relative, fp32 accumulate).  Both operands are
// time-contiguous, i.e. already K-major: a thread loads eight consecutive samples of one row, splits them and stores three
// 16-byte units -- exactly one lane's MFMA operand -- into the piece planes of the LDS tile.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t wg_pack_hi16(float lo, float hi) {
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wg_f16x2 __attribute__((ext_vector_type(2)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ void wg_split_store(const float (&v)[8], uint4* plane0, int plane_stride, int idx) {
  if constexpr (NP == 2) {          // two fp16 pieces, the second one scaled by 2^11 (conv_igemm.hip, "fp16 split"); v is already scaled
    wg_f16x8 ph, pl;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const wg_f16x2 hi = __builtin_convertvector((wg_f32x2){v[e], v[e + 1]}, wg_f16x2);
      const wg_f16x2 lo = __builtin_convertvector((wg_f32x2){(v[e] - (float)hi[0]) * kF16LoScale, (v[e + 1] - (float)hi[1]) * kF16LoScale}, wg_f16x2);
      ph[e] = hi[0]; ph[e + 1] = hi[1];
      pl[e] = lo[0]; pl[e + 1] = lo[1];
    }
    plane0[idx] = __builtin_bit_cast(uint4, ph);
    plane0[plane_stride + idx] = __builtin_bit_cast(uint4, pl);
    return;
  }
  if constexpr (NP == 1) {
    wg_bf16x8 p;
#pragma unroll
    for (int e = 0; e < 8; ++e) p[e] = (__bf16)v[e];
    plane0[idx] = __builtin_bit_cast(uint4, p);
    return;
  }
  uint32_t p1[4], p2[4], p3[4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const float r0 = v[e] - __uint_as_float(__float_as_uint(v[e]) & 0xffff0000u);
    const float r1 = v[e + 1] - __uint_as_float(__float_as_uint(v[e + 1]) & 0xffff0000u);
    const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u);
    const float s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    p1[e >> 1] = wg_pack_hi16(v[e], v[e + 1]);
    p2[e >> 1] = wg_pack_hi16(r0, r1);
    p3[e >> 1] = wg_pack_hi16(s0, s1);
  }
  plane0[idx] = make_uint4(p1[0], p1[1], p1[2], p1[3]);
  plane0[plane_stride + idx] = make_uint4(p2[0], p2[1], p2[2], p2[3]);
  plane0[2 * plane_stride + idx] = make_uint4(p3[0], p3[1], p3[2], p3[3]);
}

// XS = a.xstride (1 or 2); workgroup tile (64 MT) x (64 NT) output x input channels, 2 x 2 waves of (32 MT) x (32 NT); KS samples per
// staged step.  <1, 1, 64>: the fp32 kernel's tile (default).  <2, 2, 32>: 128 x 128, twice the flops per byte moved from L2 (opt-in).
// grid (ceil(Cin / (64 NT)), ceil(Cout / (64 MT)), taps * slices); a.chunks_per_item counts KS-sample steps.
// NP = 3: split fp32 (exact products); NP = 1: operands rounded to bf16 (round to nearest even), one MFMA per block -- the bf16 training mode
// FLAT (round 3): the reduction runs over the flattened sample index n = b * T + t in steps of KS, as conv_wgrad_kernel<2> does, for rows of
// 33..255 samples (the 92-frame WaveNet GEMMs ran on the fp32 MFMA at its peak, 158 TF/s: the matrix pipe itself was the limit).  Every
// step takes the clamped scalar loads of the boundary path, with the item index carried per element (a group of 8 samples crosses at
// most one item boundary because T > 8).
template <int XS, int MT, int NT, int KS, int NP = 3, bool FLAT = false>
__global__ __launch_bounds__(256) void conv_wgrad_split_kernel(WgArgs a) {
  constexpr int UR = KS / 8 + 1;                                 // uint4 units per LDS row incl. one pad unit: conflict-free 128-bit reads
  constexpr int RM = 64 * MT, RN = 64 * NT;                      // staged rows of dy / x
  constexpr int kPlaneD = RM * UR, kPlaneX = RN * UR;            // uint4 units per piece plane
  constexpr int G = KS / 8, RP = 256 / G;                        // 8-sample groups per row, rows per staging pass
  constexpr int PD = RM / RP, PX = RN / RP;                      // staging passes
  static_assert(RM % RP == 0 && RN % RP == 0 && NP * (kPlaneD + kPlaneX) * 16 <= 65536, "tile does not fit");
  __shared__ uint4 lds[NP * (kPlaneD + kPlaneX)];                 // [operand][piece][row][unit]
  uint4* const dys = lds;
  uint4* const xs = lds + NP * kPlaneD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ci0 = blockIdx.x * RN, co0 = blockIdx.y * RM;
  const int tap = blockIdx.z % a.taps, slice = blockIdx.z / a.taps;
  const int shift = tap * a.dil - a.pad;
  floatx16 acc[MT][NT];
  floatx16 acl[NP == 2 ? MT : 1][NP == 2 ? NT : 1];            // NP = 2: the cross terms, 2^11 times their size
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[mi][ni][r] = 0.f;
        if constexpr (NP == 2) acl[mi][ni][r] = 0.f;
      }
  // NP = 2: gradients span too many binades for fp16 as they are: dy is staged x 2^(13 - e), e the exponent of the launch's max |dy|
  // (absmax_kernel), so that the largest gradient lands in [2^13, 2^14) and everything within 2^-28 of it keeps 2^-22 relative precision;
  // x is staged x 2^-6 like the forward kernel.  The product of the two scales is undone on the accumulators.
  float dy_mul = 1.f, x_mul = 1.f, out_mul = 1.f;
  if constexpr (NP == 2) {
    const uint32_t mb = *a.dy_absmax;
    const int e = min(max((int)(mb >> 23) - 127, -100), 100);
    dy_mul = __uint_as_float((uint32_t)(127 + 13 - e) << 23);
    x_mul = kF16XScale;
    out_mul = __uint_as_float((uint32_t)(127 - 13 + e) << 23) * (1.f / kF16XScale);
  }
  const int total = FLAT ? (a.B * a.T + KS - 1) / KS : a.B * a.chunks_per_item;
  const int c_begin = slice * a.chunks_per_slice, c_end = min(total, c_begin + a.chunks_per_slice);
  const int r31 = lane & 31, hh = lane >> 5;
  const int srow = tid / G, sg = tid % G;                        // staging: row within a pass, group of 8 samples

  // two register sets: the global loads of step c + 2 are issued while step c is multiplied, so a load has two MFMA phases and a
  // staging phase to arrive
  float rd[2][PD][8], rx[2][PX][8];
  unsigned md[2][PD], mx[2][PX];
  auto fetch = [&](auto set, int c) {
    constexpr int S = decltype(set)::value;
    if constexpr (FLAT) {
      const int n0 = c * KS + 8 * sg;
      const int b0 = n0 / a.T, t0f = n0 - b0 * a.T;
      int64_t offd[8], offx[8];              // element offsets without the row term
      unsigned okd = 0, okx = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int te = t0f + e, be = b0;
        if (te >= a.T) { te -= a.T; be += 1; }
        const bool ok = be < a.B;
        const int bc = min(be, a.B - 1);
        const int tx = te * XS + a.xoff + shift;
        offd[e] = (int64_t)bc * a.Cout * a.T + te;
        offx[e] = (int64_t)bc * a.Cin * a.Tx + min(max(tx, 0), a.Tx - 1);
        okd |= (unsigned)ok << e;
        okx |= (unsigned)(ok && tx >= 0 && tx < a.Tx) << e;
      }
#pragma unroll
      for (int ps = 0; ps < PD; ++ps) {
        const int co = co0 + srow + RP * ps;
        const float* p = a.dy + (int64_t)min(co, a.Cout - 1) * a.T;
#pragma unroll
        for (int e = 0; e < 8; ++e) rd[S][ps][e] = p[offd[e]];
        md[S][ps] = co < a.Cout ? okd : 0u;
      }
#pragma unroll
      for (int ps = 0; ps < PX; ++ps) {
        const int ci = ci0 + srow + RP * ps;
        const float* q = a.x + (int64_t)min(ci, a.Cin - 1) * a.Tx;
#pragma unroll
        for (int e = 0; e < 8; ++e) rx[S][ps][e] = q[offx[e]];
        mx[S][ps] = ci < a.Cin ? okx : 0u;
      }
      return;
    }
    const int b = c / a.chunks_per_item;
    const int t0 = (c - b * a.chunks_per_item) * KS;
    const int t = t0 + 8 * sg;
    const int txs = t * XS + a.xoff + shift;
    const float* dyb = a.dy + (int64_t)b * a.Cout * a.T;
    const float* xb = a.x + (int64_t)b * a.Cin * a.Tx;
    // wave-uniform choice (a real branch, so the number of loads in flight stays static on either side): a step whose samples
    // and their displaced x window lie inside the rows uses 128-bit loads, a boundary step clamped scalar loads with validity bits
    const int xw0 = t0 * XS + a.xoff + shift;
    const bool interior = t0 + KS <= a.T && xw0 >= 0 && xw0 + KS * XS <= a.Tx;
    if (interior) {
#pragma unroll
      for (int ps = 0; ps < PD; ++ps) {
        const int co = co0 + srow + RP * ps;
        const float* p = dyb + (int64_t)min(co, a.Cout - 1) * a.T;
        const f4u u0 = *reinterpret_cast<const f4u*>(p + t), u1 = *reinterpret_cast<const f4u*>(p + t + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { rd[S][ps][e] = u0[e]; rd[S][ps][4 + e] = u1[e]; }
        md[S][ps] = co < a.Cout ? 0xffu : 0u;
      }
#pragma unroll
      for (int ps = 0; ps < PX; ++ps) {
        const int ci = ci0 + srow + RP * ps;
        const float* q = xb + (int64_t)min(ci, a.Cin - 1) * a.Tx;
        if (XS == 1) {
          const f4u w0 = *reinterpret_cast<const f4u*>(q + txs), w1 = *reinterpret_cast<const f4u*>(q + txs + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { rx[S][ps][e] = w0[e]; rx[S][ps][4 + e] = w1[e]; }
        } else {
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const f4u w = *reinterpret_cast<const f4u*>(q + txs + 4 * e4);
            rx[S][ps][2 * e4] = w[0]; rx[S][ps][2 * e4 + 1] = w[2];
          }
        }
        mx[S][ps] = ci < a.Cin ? 0xffu : 0u;
      }
    } else {
#pragma unroll
      for (int ps = 0; ps < PD; ++ps) {
        const int co = co0 + srow + RP * ps;
        const float* p = dyb + (int64_t)min(co, a.Cout - 1) * a.T;
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          rd[S][ps][e] = p[min(t + e, a.T - 1)];
          m |= (unsigned)(t + e < a.T) << e;
        }
        md[S][ps] = co < a.Cout ? m : 0u;
      }
#pragma unroll
      for (int ps = 0; ps < PX; ++ps) {
        const int ci = ci0 + srow + RP * ps;
        const float* q = xb + (int64_t)min(ci, a.Cin - 1) * a.Tx;
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int tx = txs + e * XS;
          rx[S][ps][e] = q[min(max(tx, 0), a.Tx - 1)];
          m |= (unsigned)(t + e < a.T && tx >= 0 && tx < a.Tx) << e;
        }
        mx[S][ps] = ci < a.Cin ? m : 0u;
      }
    }
  };
  auto step = [&](auto set, int c) {
    constexpr int S = decltype(set)::value;
    __syncthreads();      // the previous step's fragment reads are done
#pragma unroll
    for (int ps = 0; ps < PD; ++ps) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (md[S][ps] >> e) & 1 ? (NP == 2 ? rd[S][ps][e] * dy_mul : rd[S][ps][e]) : 0.f;
      wg_split_store<NP>(v, dys, kPlaneD, (srow + RP * ps) * UR + sg);
    }
#pragma unroll
    for (int ps = 0; ps < PX; ++ps) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (mx[S][ps] >> e) & 1 ? (NP == 2 ? rx[S][ps][e] * x_mul : rx[S][ps][e]) : 0.f;
      wg_split_store<NP>(v, xs, kPlaneX, (srow + RP * ps) * UR + sg);
    }
    __syncthreads();
    fetch(set, min(c + 2, c_end - 1));      // unconditional (the tail re-reads the last step): keeps the load count per step static
    const uint4* ap = dys + (wm * 32 * MT + r31) * UR + hh;
    const uint4* bp = xs + (wn * 32 * NT + r31) * UR + hh;
#pragma unroll
    for (int kk = 0; kk < KS / 16; ++kk) {
      wg_bf16x8 af[MT][NP], bf[NT][NP];
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) af[mi][pc] = __builtin_bit_cast(wg_bf16x8, ap[pc * kPlaneD + mi * 32 * UR + 2 * kk]);
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) bf[ni][pc] = __builtin_bit_cast(wg_bf16x8, bp[pc * kPlaneX + ni * 32 * UR + 2 * kk]);
      }
      if constexpr (NP == 2) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) {
            const wg_f16x8 ah = __builtin_bit_cast(wg_f16x8, af[mi][0]), al = __builtin_bit_cast(wg_f16x8, af[mi][1]);
            const wg_f16x8 bh = __builtin_bit_cast(wg_f16x8, bf[ni][0]), bl = __builtin_bit_cast(wg_f16x8, bf[ni][1]);
            acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acl[mi][ni], 0, 0, 0);
            acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acl[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mi][ni], 0, 0, 0);
          }
      } else {
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // smallest partial products first
#pragma unroll
        for (int u = 0; u < (NP == 3 ? 6 : 1); ++u)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][NP == 3 ? PA[u] : 0], bf[ni][NP == 3 ? PB[u] : 0], acc[mi][ni], 0, 0, 0);
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (c_begin >= c_end) return;
  fetch(S0{}, c_begin);
  fetch(S1{}, min(c_begin + 1, c_end - 1));
  int c = c_begin;
  for (; c + 1 < c_end; c += 2) {     // pairs, so that neither step sits behind a condition
    step(S0{}, c);
    step(S1{}, c + 1);
  }
  if (c < c_end) step(S0{}, c);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int ci = ci0 + (wn * NT + ni) * 32 + r31;
      if (ci < a.Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + (wm * MT + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          float v = acc[mi][ni][r];
          if constexpr (NP == 2) v = fmaf(acl[mi][ni][r], 1.f / kF16LoScale, v) * out_mul;
          if (co < a.Cout) atomicAdd(a.dw + ((int64_t)co * a.Cin + ci) * a.taps_out + a.tap_out + tap, v);
        }
      }
    }
}

// ---- three taps per workgroup (fp16 split) -------------------------------------------------------------------------------------
// The kernels above give every (tap, tile, K slice) its own workgroup, so a 64 x 64 dy tile is fetched from L2 and converted once per TAP
// and a step moves 32 KB for 0.5 MFLOP: the discriminator's 27-tap layers are bound by that traffic, not by the matrix core.  Here a workgroup
// owns THREE adjacent taps of a unit-dilation convolution: the dy tile is staged once, the x window (8 XS + 2 samples per thread instead of
// 3 x 8) once, the three shifted x tiles are cut from it in registers -- 2.8x fewer bytes and 1.5x fewer conversions per flop.  Operands are the
// fp16 split of conv_igemm.hip (two pieces, three products, two accumulator sets per tap); dy is scaled by the launch's max (absmax_kernel).
// grid (ceil(Cin / 64), ceil(Cout / 64), taps / 3 * slices); 73.7 KB of dynamic LDS: [dy | x tap 0 | x tap 1 | x tap 2] x [hi, lo] planes.
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
template <int XS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16g_kernel(WgArgs a) {
  constexpr int TG = 3, KS = 64, UR = KS / 8 + 1, G = KS / 8, RP = 256 / G, PS = 64 / RP;
  constexpr int W = 8 * XS + (TG - 1);                           // x samples a thread needs for its 8 output samples and 3 taps
  constexpr int kPlane = 64 * UR;
  extern __shared__ uint4 wg_lds[];
  uint4* const dys = wg_lds;
  uint4* const xs = wg_lds + 2 * kPlane;                         // tap g at xs + g * 2 * kPlane
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const int groups = a.taps / TG;
  const int tap0 = (blockIdx.z % groups) * TG, slice = blockIdx.z / groups;
  const int shift = tap0 - a.pad;                                 // dilation 1
  floatx16 acc[TG], acl[TG];
#pragma unroll
  for (int g = 0; g < TG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[g][r] = 0.f; acl[g][r] = 0.f; }
  const uint32_t mb = *a.dy_absmax;
  const int ex = min(max((int)(mb >> 23) - 127, -100), 100);
  const float dy_mul = __uint_as_float((uint32_t)(127 + 13 - ex) << 23), x_mul = kF16XScale;
  const float out_mul = __uint_as_float((uint32_t)(127 - 13 + ex) << 23) * (1.f / kF16XScale);
  const int total = a.B * a.chunks_per_item;
  const int c_begin = slice * a.chunks_per_slice, c_end = min(total, c_begin + a.chunks_per_slice);
  // (an empty K slice still writes its zero tiles: the reduction reads every slice)
  const int r31 = lane & 31, hh = lane >> 5;
  const int srow = tid / G, sg = tid % G;

  // one register set: a step's values are consumed by the staging at its top and the next step's loads are issued right behind it, so they
  // have the 36 MFMAs of the step to arrive
  float rd[PS][8], rx[PS][W];
  unsigned md[PS], mx[PS];
  auto fetch = [&](int c) {
    const int b = c / a.chunks_per_item;
    const int t0 = (c - b * a.chunks_per_item) * KS;
    const int t = t0 + 8 * sg;
    const int txs = t * XS + a.xoff + shift;
    const float* dyb = a.dy + (int64_t)b * a.Cout * a.T;
    const float* xb = a.x + (int64_t)b * a.Cin * a.Tx;
    const int xw0 = t0 * XS + a.xoff + shift;
    const bool interior = t0 + KS <= a.T && xw0 >= 0 && xw0 + KS * XS + TG - 1 <= a.Tx;       // wave-uniform; covers the W floats every thread loads
    if (interior) {
#pragma unroll
      for (int ps = 0; ps < PS; ++ps) {
        const int co = co0 + srow + RP * ps, ci = ci0 + srow + RP * ps;
        const float* p = dyb + (int64_t)min(co, a.Cout - 1) * a.T + t;
        const float* q = xb + (int64_t)min(ci, a.Cin - 1) * a.Tx + txs;
        const f4u u0 = *reinterpret_cast<const f4u*>(p), u1 = *reinterpret_cast<const f4u*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { rd[ps][e] = u0[e]; rd[ps][4 + e] = u1[e]; }
        md[ps] = co < a.Cout ? 0xffu : 0u;
#pragma unroll
        for (int e4 = 0; e4 < (W - 2) / 4; ++e4) {
          const f4u w4 = *reinterpret_cast<const f4u*>(q + 4 * e4);
#pragma unroll
          for (int e = 0; e < 4; ++e) rx[ps][4 * e4 + e] = w4[e];
        }
        const f2u w2 = *reinterpret_cast<const f2u*>(q + W - 2);
        rx[ps][W - 2] = w2[0]; rx[ps][W - 1] = w2[1];
        mx[ps] = ci < a.Cin ? (1u << W) - 1u : 0u;
      }
    } else {
#pragma unroll
      for (int ps = 0; ps < PS; ++ps) {
        const int co = co0 + srow + RP * ps, ci = ci0 + srow + RP * ps;
        const float* p = dyb + (int64_t)min(co, a.Cout - 1) * a.T;
        const float* q = xb + (int64_t)min(ci, a.Cin - 1) * a.Tx;
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          rd[ps][e] = p[min(t + e, a.T - 1)];
          m |= (unsigned)(t + e < a.T) << e;
        }
        md[ps] = co < a.Cout ? m : 0u;
        m = 0;
#pragma unroll
        for (int e = 0; e < W; ++e) {
          const int tx = txs + e;
          rx[ps][e] = q[min(max(tx, 0), a.Tx - 1)];
          m |= (unsigned)(tx >= 0 && tx < a.Tx) << e;
        }
        mx[ps] = ci < a.Cin ? m : 0u;
      }
    }
  };
  if (c_begin < c_end) fetch(c_begin);
  for (int c = c_begin; c < c_end; ++c) {
    __syncthreads();      // the previous step's fragment reads are done
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
      const int idx = (srow + RP * ps) * UR + sg;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (md[ps] >> e) & 1 ? rd[ps][e] * dy_mul : 0.f;
      wg_split_store<2>(v, dys, kPlane, idx);
      // an output sample past the end of dy's row has a zero dy factor, so x needs no extra mask for it
#pragma unroll
      for (int g = 0; g < TG; ++g) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (mx[ps] >> (e * XS + g)) & 1 ? rx[ps][e * XS + g] * x_mul : 0.f;
        wg_split_store<2>(v, xs + g * 2 * kPlane, kPlane, idx);
      }
    }
    __syncthreads();
    fetch(min(c + 1, c_end - 1));      // unconditional (the last step re-reads itself): the load count per step stays static
    const uint4* ap = dys + (wm * 32 + r31) * UR + hh;
    const uint4* bp = xs + (wn * 32 + r31) * UR + hh;
#pragma unroll
    for (int kk = 0; kk < KS / 16; ++kk) {
      const wg_f16x8 ah = __builtin_bit_cast(wg_f16x8, ap[2 * kk]), al = __builtin_bit_cast(wg_f16x8, ap[kPlane + 2 * kk]);
#pragma unroll
      for (int g = 0; g < TG; ++g) {
        const wg_f16x8 bh = __builtin_bit_cast(wg_f16x8, bp[g * 2 * kPlane + 2 * kk]);
        const wg_f16x8 bl = __builtin_bit_cast(wg_f16x8, bp[g * 2 * kPlane + kPlane + 2 * kk]);
        acl[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acl[g], 0, 0, 0);
        acl[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acl[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[g], 0, 0, 0);
      }
    }
  }
  const int64_t wg = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float* dst = a.partial + ((wg * TG) * 4 + wave) * 1024 + lane;
#pragma unroll
  for (int g = 0; g < TG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(int64_t)g * 4096 + r * 64] = fmaf(acl[g][r], 1.f / kF16LoScale, acc[g][r]) * out_mul;
}
constexpr size_t kWgF16gLds = (size_t)4 * 2 * 64 * (64 / 8 + 1) * 16;

// dw[co, ci, tap_out + tap] += sum over the K slices of the three-tap kernel's partial tiles.  One thread per (co, ci, tap) of the launch.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Cout, int Cin, int taps,
                                                           int taps_out, int tap_out, int tm, int tn, int slices) {
  const int64_t n = (int64_t)Cout * Cin * taps;
  const int groups = taps / 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int tap = (int)((i / Cin) % taps);
    const int co = (int)(i / ((int64_t)Cin * taps));
    const int y = co >> 6, x = ci >> 6, wave = ((co >> 5) & 1) * 2 + ((ci >> 5) & 1);
    const int rr = co & 31, hh = (rr >> 2) & 1, r = (rr & 3) + 4 * (rr >> 3), lane = (ci & 31) + 32 * hh;
    const int grp = tap / 3, g = tap - 3 * grp;
    const int64_t tile_stride = (int64_t)groups * tm * tn * 3 * 4096;            // one K slice
    const float* p = partial + ((((int64_t)grp * tm + y) * tn + x) * 3 + g) * 4096 + wave * 1024 + r * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four loads in flight per thread: the slices are 10^5-10^6 floats apart
    int sl = 0;
    for (; sl + 4 <= slices; sl += 4) {
      s0 += p[(int64_t)sl * tile_stride];
      s1 += p[(int64_t)(sl + 1) * tile_stride];
      s2 += p[(int64_t)(sl + 2) * tile_stride];
      s3 += p[(int64_t)(sl + 3) * tile_stride];
    }
    for (; sl < slices; ++sl) s0 += p[(int64_t)sl * tile_stride];
    dw[((int64_t)co * Cin + ci) * taps_out + tap_out + tap] += (s0 + s1) + (s2 + s3);
  }
}

// max |v| over n floats as a bit pattern (non-negative floats order like unsigned integers); *out zeroed by the launcher
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ v, int64_t n, uint32_t* out) {
  uint32_t m = 0;
  const int64_t head = min(n, (int64_t)((4 - ((reinterpret_cast<uintptr_t>(v) >> 2) & 3)) & 3));      // floats up to 16-byte alignment
  const int64_t n4 = (n - head) >> 2, tail0 = head + 4 * n4;
  const uint4* v4 = reinterpret_cast<const uint4*>(v + head);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 q = v4[i];
    m = max(max(m, q.x & 0x7fffffffu), max(max(q.y & 0x7fffffffu, q.z & 0x7fffffffu), q.w & 0x7fffffffu));
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x < head) m = max(m, __float_as_uint(v[threadIdx.x]) & 0x7fffffffu);
    if (threadIdx.x >= 64 && threadIdx.x - 64 < n - tail0) m = max(m, __float_as_uint(v[tail0 + threadIdx.x - 64]) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  __shared__ uint32_t part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {           // one atomic per workgroup: thousands of them on one word serialise for longer than the reduction takes
    m = max(max(part[0], part[1]), max(part[2], part[3]));
    if (m != 0) atomicMax(out, min(m, 0x7f7fffffu));      // inf / NaN gradients: scale as for the largest finite value
  }
}

// DMEL_WGRAD_FP32_MFMA=1 keeps every weight gradient on the native fp32 MFMA kernel (A/B switch)
static bool wgrad_native_only() {
  static const bool v = [] { const char* e = getenv("DMEL_WGRAD_FP32_MFMA"); return e && e[0] == '1'; }();
  return v;
}
// DMEL_WGRAD_TILE128=1 moves the split kernel to its 128 x 128 x 32 instantiation where both channel counts allow (A/B switch).
// Measured slower than 64 x 64 x 64 on every shape (42-81 vs 67-92 TF/s): 256 VGPRs and 60 KB of LDS leave one or two workgroups
// per CU, and a 32-sample step has two barriers per 48 MFMAs -- the kernel is bound by latency between barriers, not by L2 bytes.
static bool wgrad_big_tile() {
  static const bool v = [] { const char* e = getenv("DMEL_WGRAD_TILE128"); return e && e[0] == '1'; }();
  return v;
}
int launch_absmax(const float* v, int64_t n, hipStream_t st, const uint32_t** out) {
  // a ring of device words: launches of one thread are ordered on their stream; the ring keeps earlier launches' words alive while later
  // ones are queued
  DevBuf& ring = *thread_scratch(0, st);
  static thread_local unsigned next = 0;      // one counter for all rings of the thread: a ring is only ever entered at its own slots
  constexpr unsigned kSlots = 1024;
  if (!ring.p) {
    DMEL_HIP(hipMalloc(&ring.p, kSlots * sizeof(uint32_t)));
    ring.bytes = kSlots * sizeof(uint32_t);
  }
  uint32_t* slot = ring.as<uint32_t>() + (next++ % kSlots);
  DMEL_HIP(hipMemsetAsync(slot, 0, sizeof(uint32_t), st));
  const unsigned blocks = (unsigned)std::min<int64_t>((n / 4 + 255) / 256 + 1, 512);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, st, v, n, slot);
  DMEL_HIP(hipGetLastError());
  *out = slot;
  return DMEL_OK;
}

// picks the kernel, then the K slicing for its tile and step (a: everything but chunks_per_item / ipc / slices / chunks_per_slice)
static int launch_wgrad_any(WgArgs a, hipStream_t st) {
  a.ipc = a.T <= 32 ? kWgK / a.T : 1;
  // short rows are mostly boundary steps (scalar loads in the split kernel): they stay on the fp32-MFMA kernel
  const bool split = a.ipc == 1 && !wgrad_native_only() && a.xstride <= 2 && a.T >= 256;
  // the three-tap kernel also serves the bf16 training mode (its fp16 split is more accurate than bf16 operands and, without atomics, faster)
  const bool can_group = split && a.dil == 1 && a.taps % 3 == 0;
  // rows of 33..255 samples whose last 64-sample step would be mostly padding (T = 92: 30 % of the MFMA work) reduce over the flattened
  // index; in fp32 training mode on the fp16-split kernel (FLAT instantiation), which needs the launch's max |dy| like the long rows do
  static const bool flat_off = [] { const char* e = getenv("DMEL_WGRAD_NOFLAT"); return e != nullptr; }();
  // measured (round 3, tools/bench_train_step.py): weight-gradient time per step 25.8 -> 30.5 ms with the FLAT fp16-split instantiation --
  // its staging (scalar loads with per-element item indices, masks, scale, split, pack: ~500 vector instructions per 64-sample step and
  // wave) costs what the 32 fp32 MFMAs it replaces cost (2048 cycles), plus one absmax launch per gradient tensor.  Opt-in, for A/B.
  static const bool flat16_off = [] { const char* e = getenv("DMEL_WGRAD_FLAT16"); return !(e && e[0] == '1'); }();
  static const bool f16_off0 = [] { const char* e = getenv("DMEL_WGRAD_F16X2"); return e && e[0] == '0'; }();
  const bool flat_shape = !split && a.ipc == 1 && a.T < 256 && (int64_t)a.B * a.T < ((int64_t)1 << 30) &&
                          ((a.T + kWgK - 1) / kWgK) * kWgK * 10 > a.T * 11 && !flat_off;
  const bool flat16 = flat_shape && a.T >= 33 && a.xstride <= 2 && !wgrad_native_only() && !flat16_off && !f16_off0 &&
                      train_precision_override() != DMEL_PRECISION_BF16;
  if ((split && (train_precision_override() != DMEL_PRECISION_BF16 || can_group)) || flat16) {
    if (!a.dy_absmax) DMEL_TRY(launch_absmax(a.dy, (int64_t)a.B * a.Cout * a.T, st, &a.dy_absmax));
  } else {
    a.dy_absmax = nullptr;
  }
  // bf16 training mode (TrainPrecisionScope): long rows take the one-piece instantiation of the split kernel; short rows (the 92-frame
  // WaveNet GEMMs, packed image rows) stay on the exact fp32-MFMA kernel, which is at least as accurate
  static const bool group_off0 = [] { const char* e = getenv("DMEL_WGRAD_TAPGROUP"); return e && e[0] == '0'; }();
  static const bool f16_off = [] { const char* e = getenv("DMEL_WGRAD_F16X2"); return e && e[0] == '0'; }();
  const bool bf16 = split && train_precision_override() == DMEL_PRECISION_BF16 && !(can_group && !group_off0 && !f16_off && a.dy_absmax);
  // fp32 mode: long rows take the fp16-split instantiation (three products, cheaper operand conversion; DMEL_WGRAD_F16X2=0: the six-product one)
  const bool f16 = split && !bf16 && !f16_off && a.dy_absmax != nullptr;
  static const bool group_off = [] { const char* e = getenv("DMEL_WGRAD_TAPGROUP"); return e && e[0] == '0'; }();
  const bool grouped = f16 && !group_off && a.dil == 1 && a.taps % 3 == 0;
  if (grouped) {
    a.chunks_per_item = (a.T + 63) / 64;
    const int tm = (a.Cout + 63) / 64, tn = (a.Cin + 63) / 64, groups = a.taps / 3;
    const int total = a.B * a.chunks_per_item;
    // ~1024 workgroups (two per CU resident): 256 -> 33.9 ms, 512 -> 30.4, 1024 -> 29.9, 2048 -> 29.8 ms of weight-gradient time per training step
    const char* we = getenv("DMEL_WGRAD_WANT");
    const int wantg = we ? atoi(we) : 1024;
    int slices = std::max(1, std::min((wantg + tm * tn * groups - 1) / (tm * tn * groups), (total + 7) / 8));
    slices = std::min(slices, 65535 / groups);
    a.slices = slices;
    a.chunks_per_slice = (total + slices - 1) / slices;
    const dim3 grid((unsigned)tn, (unsigned)tm, (unsigned)(groups * slices));
    static const bool attr_ok = [] {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_f16g_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgF16gLds) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_f16g_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgF16gLds) == hipSuccess;
    }();
    if (!attr_ok) { set_error("conv_wgrad: could not raise the dynamic LDS limit to %zu bytes", kWgF16gLds); return DMEL_EUNSUPPORTED; }
    // partial tiles of all workgroups: library-owned scratch of this (thread, stream), reused by every launch on it (launches and their
    // reductions are ordered on the stream)
    DevBuf& scratch = *thread_scratch(1, st);
    const size_t need = (size_t)tn * tm * groups * slices * 3 * 4096 * sizeof(float);
    if (scratch.bytes < need) {
      DMEL_HIP(hipStreamSynchronize(st));
      scratch.release();
      DMEL_HIP(hipMalloc(&scratch.p, need));
      scratch.bytes = need;
    }
    a.partial = scratch.as<float>();
    if (a.xstride == 1) hipLaunchKernelGGL(conv_wgrad_f16g_kernel<1>, grid, dim3(256), kWgF16gLds, st, a);
    else hipLaunchKernelGGL(conv_wgrad_f16g_kernel<2>, grid, dim3(256), kWgF16gLds, st, a);
    DMEL_HIP(hipGetLastError());
    const int64_t nout = (int64_t)a.Cout * a.Cin * a.taps;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<int64_t>((nout + 255) / 256, 4096)), dim3(256), 0, st, a.partial, a.dw, a.Cout,
                       a.Cin, a.taps, a.taps_out, a.tap_out, tm, tn, slices);
    DMEL_HIP(hipGetLastError());
    return DMEL_OK;
  }
  const bool big = split && !bf16 && !f16 && a.Cout >= 128 && a.Cin >= 128 && wgrad_big_tile();
  const int tile = big ? 128 : kWgTile, ks = big ? 32 : kWgK;
  a.chunks_per_item = (a.T + ks - 1) / ks;
  const int tm = (a.Cout + tile - 1) / tile, tn = (a.Cin + tile - 1) / tile;
  const int tiles = tm * tn * a.taps;
  const bool flat = flat_shape;
  const int total = a.ipc > 1 ? (a.B + a.ipc - 1) / a.ipc : flat ? (a.B * a.T + kWgK - 1) / kWgK : a.B * a.chunks_per_item;
  // enough K slices to fill the chip (~8 workgroups of the small tile / ~4 of the large one per CU), each at least 8 staged steps long
  const char* we2 = getenv("DMEL_WGRAD_WANT");
  const int want = we2 ? atoi(we2) : (big ? 1024 : 2048);
  int slices = std::max(1, std::min((want + tiles - 1) / tiles, (total + 7) / 8));
  slices = std::min(slices, 65535 / a.taps);
  a.slices = slices;
  a.chunks_per_slice = (total + slices - 1) / slices;
  const dim3 grid((unsigned)tn, (unsigned)tm, (unsigned)(a.taps * slices));
  if (a.ipc > 1) hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), 0, st, a);
  else if (flat16 && a.xstride == 1) hipLaunchKernelGGL((conv_wgrad_split_kernel<1, 1, 1, 64, 2, true>), grid, dim3(256), 0, st, a);
  else if (flat16) hipLaunchKernelGGL((conv_wgrad_split_kernel<2, 1, 1, 64, 2, true>), grid, dim3(256), 0, st, a);
  else if (flat) hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, dim3(256), 0, st, a);
  else if (!split) hipLaunchKernelGGL(conv_wgrad_kernel<0>, grid, dim3(256), 0, st, a);
  else if (big && a.xstride == 1) hipLaunchKernelGGL((conv_wgrad_split_kernel<1, 2, 2, 32>), grid, dim3(256), 0, st, a);
  else if (big) hipLaunchKernelGGL((conv_wgrad_split_kernel<2, 2, 2, 32>), grid, dim3(256), 0, st, a);
  else if (f16 && a.xstride == 1) hipLaunchKernelGGL((conv_wgrad_split_kernel<1, 1, 1, 64, 2>), grid, dim3(256), 0, st, a);
  else if (f16) hipLaunchKernelGGL((conv_wgrad_split_kernel<2, 1, 1, 64, 2>), grid, dim3(256), 0, st, a);
  else if (bf16 && a.xstride == 1) hipLaunchKernelGGL((conv_wgrad_split_kernel<1, 1, 1, 64, 1>), grid, dim3(256), 0, st, a);
  else if (bf16) hipLaunchKernelGGL((conv_wgrad_split_kernel<2, 1, 1, 64, 1>), grid, dim3(256), 0, st, a);
  else if (a.xstride == 1) hipLaunchKernelGGL((conv_wgrad_split_kernel<1, 1, 1, 64>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_wgrad_split_kernel<2, 1, 1, 64>), grid, dim3(256), 0, st, a);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

// db[co] = sum over (b, t).  grid (Cout, splits): a workgroup reduces every splits-th 2048-sample piece of its channel's (b, t) plane
// and adds its partial sum to db (zeroed by the launcher) -- a single workgroup per channel left a 64-channel layer of the
// discriminator (32 x 10496 samples per channel) on 64 CUs for 350 us.
constexpr int kBgPiece = 2048;
__global__ __launch_bounds__(256) void conv_bgrad_kernel(const float* __restrict__ dy, float* __restrict__ db, int Cout, int B, int T) {
  __shared__ float part[4];
  const int co = blockIdx.x;
  const int pieces = (T + kBgPiece - 1) / kBgPiece, total = B * pieces;
  float s = 0.f;
  if (T < 256) {      // short rows (the 92-frame WaveNet GEMMs): threads walk the flattened (item, frame) index, or most of them would idle
    const int64_t n = (int64_t)B * T;
    for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.y * 256) {
      const int b = (int)(i / T);
      s += dy[((int64_t)b * Cout + co) * T + (i - (int64_t)b * T)];
    }
  } else {
    for (int i = blockIdx.y; i < total; i += gridDim.y) {
      const int b = i / pieces, t0 = (i - b * pieces) * kBgPiece, t1 = min(T, t0 + kBgPiece);
      const float* row = dy + ((int64_t)b * Cout + co) * T;
      for (int t = t0 + threadIdx.x; t < t1; t += 256) s += row[t];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = part[0] + part[1] + part[2] + part[3];
    if (gridDim.y == 1) db[co] = v;
    else atomicAdd(db + co, v);
  }
}
static int launch_bgrad(const float* dy, float* db, int Cout, int B, int T, hipStream_t st) {
  const int total = B * ((T + kBgPiece - 1) / kBgPiece);
  const int splits = std::max(1, std::min(total, 2048 / std::max(1, Cout)));
  if (splits > 1) DMEL_TRY(zero_unless_cleared(db, (size_t)Cout * sizeof(float), st));
  hipLaunchKernelGGL(conv_bgrad_kernel, dim3((unsigned)Cout, (unsigned)splits), dim3(256), 0, st, dy, db, Cout, B, T);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}

int launch_conv_wgrad(const float* x, const float* dy, float* dw, float* db, int Cout, int Cin, int taps, int dil, int B, int64_t T,
                      hipStream_t st) {
  DMEL_CHECK_ARG(x && dy && dw, "conv_wgrad: NULL argument");
  DMEL_CHECK_ARG(B > 0 && T > 0 && T < ((int64_t)1 << 30) && Cout > 0 && Cin > 0 && taps > 0 && dil > 0, "conv_wgrad: bad shape");
  WgArgs a;
  a.dy = dy; a.x = x; a.dw = dw;
  a.Cout = Cout; a.Cin = Cin; a.taps = taps; a.dil = dil; a.pad = dil * (taps - 1) / 2; a.B = B; a.T = (int)T;
  a.xstride = 1; a.xoff = 0; a.Tx = (int)T; a.taps_out = taps; a.tap_out = 0;
  a.dy_absmax = nullptr; a.partial = nullptr;
  DMEL_TRY(zero_unless_cleared(dw, (size_t)Cout * Cin * taps * sizeof(float), st));
  {
    ProfScope ps("conv_wgrad", st, 2.0 * B * (double)T * Cout * Cin * taps, 0.0);
    DMEL_TRY(launch_wgrad_any(a, st));
  }
  if (db) DMEL_TRY(launch_bgrad(dy, db, Cout, B, (int)T, st));
  return DMEL_OK;
}

// One tap of a stride-s convolution / transposed convolution: dw[(r * Ccols + c) * taps_out + tap_out] (+)= sum_{b,t} rows[b, r, t] *
// cols[b, c, t * xstride + xoff]; rows (B, Crows, T), cols (B, Ccols, Tx).  The caller zeroes dw once before the taps.
int launch_conv_wgrad_strided(const float* rows, const float* cols, float* dw, int Crows, int Ccols, int xstride, int xoff, int64_t T,
                              int64_t Tx, int taps_out, int tap_out, int B, hipStream_t st, int taps, const uint32_t* rows_absmax) {
  DMEL_CHECK_ARG(rows && cols && dw, "conv_wgrad_strided: NULL argument");
  DMEL_CHECK_ARG(B > 0 && T > 0 && Tx > 0 && Tx < ((int64_t)1 << 30) && Crows > 0 && Ccols > 0 && xstride > 0 && taps_out > 0 &&
                     tap_out >= 0 && taps > 0 && tap_out + taps <= taps_out, "conv_wgrad_strided: bad shape");
  WgArgs a;
  a.dy = rows; a.x = cols; a.dw = dw;
  a.Cout = Crows; a.Cin = Ccols; a.taps = taps; a.dil = 1; a.pad = 0; a.B = B; a.T = (int)T;
  a.xstride = xstride; a.xoff = xoff; a.Tx = (int)Tx; a.taps_out = taps_out; a.tap_out = tap_out;
  a.dy_absmax = rows_absmax; a.partial = nullptr;
  {
    ProfScope ps("conv_wgrad", st, 2.0 * B * (double)T * Crows * Ccols * taps, 0.0);
    DMEL_TRY(launch_wgrad_any(a, st));
  }
  return DMEL_OK;
}

int launch_conv_bgrad(const float* dy, float* db, int Cout, int B, int64_t T, hipStream_t st) {
  DMEL_CHECK_ARG(dy && db && Cout > 0 && B > 0 && T > 0 && T < ((int64_t)1 << 31), "conv_bgrad: bad argument");
  return launch_bgrad(dy, db, Cout, B, (int)T, st);
}

}  // namespace dmel

using namespace dmel;

extern "C" int dmel_conv_backward_data(dmel_conv* c, const float* dy, float* dx, int B, int64_t T, void* stream) {
  DMEL_CHECK_ARG(c && dy && dx, "conv_backward_data: NULL argument");
  DMEL_CHECK_ARG(B > 0 && T > 0, "conv_backward_data: bad shape");
  if (!c->dgrad_ready) {
    DMEL_CHECK_ARG(!c->w_host.empty(), "conv_backward_data: handle holds no host weights");
    PackDesc d;
    d.mode = EPI_LINEAR; d.nseg = 1; d.C = c->Cin; d.phases = 1;
    d.seg[0].Cin = c->Cout; d.seg[0].taps = c->k; d.seg[0].dil = c->dil; d.seg[0].pad_left = c->dil * (c->k - 1) / 2;
    const int Cin = c->Cin, K = c->k;
    const float* w = c->w_host.data();
    DMEL_TRY(pack_conv(c->pc_dgrad, d,
                       [&](int, int row /*ci*/, int cc /*co*/, int tap) { return w[((size_t)cc * Cin + row) * K + (K - 1 - tap)]; },
                       [&](int) { return 0.f; }));
    c->dgrad_ready = true;
  }
  ConvRun r;
  r.seg[0].x = dy; r.seg[0].bstride = (int64_t)c->Cout * T; r.seg[0].cstride = T; r.seg[0].Tin = T;
  r.B = B; r.Tcols = T; r.y = dx; r.y_bs = (int64_t)c->Cin * T; r.y_cs = T; r.Tout = T;
  r.precision = exact_precision(c->precision);     // gradients stay on the exact split
  return launch_conv(c->pc_dgrad, r, (hipStream_t)stream);
}

extern "C" int dmel_conv_backward_weight(const dmel_conv* c, const float* x, const float* dy, float* dw, float* db, int B,
                                         int64_t T, void* stream) {
  DMEL_CHECK_ARG(c && x && dy && dw, "conv_backward_weight: NULL argument");
  return launch_conv_wgrad(x, dy, dw, db, c->Cout, c->Cin, c->k, c->dil, B, T, (hipStream_t)stream);
}
