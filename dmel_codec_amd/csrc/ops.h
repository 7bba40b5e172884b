// Internal launchers shared between the kernel files and the module orchestration (modules.cpp).
#pragma once
#include "conv.h"

namespace dmel {

struct FsqConst {
  int n_levels;
  int levels[4], half_width[4], basis[4];
  float half_l[4], offset[4], shift[4];
  int prebound;
  int strict;      // encode only: project_in and the bound(s) evaluated in float64 (dmel_quantizer_set_strict)
};

int make_fsq_const(FsqConst& k, const int* levels, int n, int prebound);
int launch_fsq_encode(const float* z, const float* w_in, const float* b_in, int32_t* ids, float* prequant,
                      const FsqConst& k, int B, int G, int C, int64_t T4, hipStream_t s);
int launch_fsq_decode(const int32_t* ids, const float* w_out, const float* b_out, float* z, const FsqConst& k, int B,
                      int G, int C, int64_t T4, hipStream_t s);
// FSQ straight-through backward: dx (B*G, C, T4) from dout (same layout) and the saved FSQ input x; parameter gradients in the
// handle's packed layouts (w_in (G,D,C), b_in (G,D), w_out (G,C,D), b_out (G,C)); scratch >= 2*B*G*T4*D floats
int launch_fsq_backward(const float* x, const float* dout, const float* w_in, const float* b_in, const float* w_out, float* dx,
                        float* dw_in, float* db_in, float* dw_out, float* db_out, float* scratch, const FsqConst& k, int B, int G, int C,
                        int64_t T4, hipStream_t s);
// h0 (nullable): also store the pre-norm depthwise-conv output (N, C, T) -- the training path keeps it for the LayerNorm backward
int launch_dwconv_ln(const float* x, float* y, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                     int N, int C, int64_t T, hipStream_t s, float* h0 = nullptr);
// act: 2 = tanh, 3 = clamp(-1, 1), else none (values of enum Act)
int launch_conv_post(const float* x, float* y, const float* w_dev, float bias, int act, int B, int C, int K, int64_t T,
                     hipStream_t s);
int launch_masked_copy(const float* x, float* y, const int64_t* len, int div, int N, int C, int64_t T, hipStream_t s);
// Folded batch of short items (see ConvRun::fold_pitch): xf (C, pitch) holds item n's T columns at [n*P, n*P + T), zeros elsewhere.
// fold: xf[c][n*P + t] = x[n][c][t] * (t < len[n / div]) (len nullable), every other column of the `pitch`-wide rows zeroed.
// unfold: y[n][c][t] = xf[c][n*P + t] * (t < len[n / div]).
int launch_fold(const float* x, float* xf, const int64_t* len, int div, int N, int C, int64_t T, int P, int64_t pitch, hipStream_t s);
int launch_unfold(const float* xf, float* y, const int64_t* len, int div, int N, int C, int64_t T, int P, int64_t pitch, hipStream_t s);
int launch_resample(const float* x, float* y, const float* bank_dev, int B, int64_t L, int64_t Lout, int down, int up, int width,
                    hipStream_t s);
// Whole-WaveNet kernel for narrow unconditioned stacks on short items (wavenet_fused.hip): device pointers into the handle's existing
// split-bf16 weight images and biases; `table` = device array [gate_w[L] | gate_b[L] | rs_w[L] | rs_b[L]] of pointers.
struct WaveNetFused {
  bool ok = false;
  int Cin = 0, C = 0, L = 0, cycle = 0, has_in = 0;
  float skip_scale = 1.f;
  const void* in_w = nullptr;
  const float* in_b = nullptr;
  const void* skip_w = nullptr;
  const float* skip_b = nullptr;
  DevBuf table;
};
int launch_wavenet_fused(const WaveNetFused& f, const float* x, float* y, const int64_t* in_len, const int64_t* out_len, int len_div, int N,
                         int64_t T, hipStream_t st);
int launch_aa_snake(const float* x, float* y, const float* alpha, const float* beta, const float* up_taps_host,
                    const float* down_taps_host, int logscale, int B, int C, int64_t T, hipStream_t s);


// Anti-aliased Snake fused into the convolution that reads it (conv_snake.hip): y = conv(snake(x)) with ConvRun's LINEAR epilogue
// (bias, residual, running sum, 1 / out_div); r.seg[0].x is the tensor BEFORE the activation.  Bit-identical to launch_aa_snake +
// launch_conv at DMEL_PRECISION_FP32_F16X2.  conv_snake_eligible: plain "same" convolution, fp16-split precision, no masks / strides.
bool conv_snake_eligible(const PackedConv& pc, const ConvRun& r);
int launch_conv_snake(const PackedConv& pc, const ConvRun& r, const float* alpha, const float* beta, const float* up_taps_host,
                      const float* down_taps_host, int logscale, hipStream_t stream);

// Producer / consumer form of the fp16-split convolution (conv_pc.hip): eight consumer waves that only run the MFMA loop + four producer
// waves that stage x.  launch_conv hands eligible launches (>= 5 row tiles, fp16-split precision, no masks / strides / phases) to it
// unless DMEL_CONV_PC=0.  Bit-identical to conv_bf16_kernel<NP = 2>.
bool conv_pc_eligible(const PackedConv& pc, const ConvRun& r, bool any_size = false);      // any_size: also launches of < 128 workgroups (tests: DMEL_CONV_PC=2)
int launch_conv_pc(const PackedConv& pc, const ConvRun& r, hipStream_t stream);

// ---- training path (train_ops.hip, conv_bwd.hip) --------------------------------------------------------------
int launch_gate_fwd(const float* pre, float* z, int N, int C, int64_t T, hipStream_t s);
int launch_gate_bwd(const float* dz, const float* pre, float* dpre, int N, int C, int64_t T, hipStream_t s);
int launch_resskip_fwd(const float* x, const float* o, float* xn, float* skipsum, int first, int N, int C, int64_t T, hipStream_t s);
int launch_resskip_bwd(const float* gx, const float* gs, float* go, float* gxs, int N, int C, int64_t T, hipStream_t s);
int launch_silu_fwd(const float* u, float* y, int64_t total, hipStream_t s);
int launch_silu_bwd(const float* g, const float* u, float* du, int64_t total, hipStream_t s);
int launch_scale(const float* x, float* y, float k, int64_t total, hipStream_t s);
// ConvNeXt block pieces (firefly.py:383-402): exact-erf GELU, layer scale + residual, and the backward of the front half
int launch_gelu_fwd(const float* u, float* g, int64_t total, hipStream_t s);
int launch_gelu_bwd(const float* dg, const float* u, float* du, int64_t total, hipStream_t s);
int launch_layerscale_res_fwd(const float* x, const float* v, const float* gamma, float* y, int N, int C, int64_t T, hipStream_t s);
// dv = dy * gamma[c]; dgamma[c] = sum_{n,t} dy * v   (dgamma overwritten)
int launch_layerscale_bwd(const float* dy, const float* v, const float* gamma, float* dv, float* dgamma, int N, int C, int64_t T,
                          hipStream_t s);
// LayerNorm over channels (eps 1e-6, biased variance) backward: dh0 from dh1 and the saved pre-norm h0; dln_w / dln_b accumulated
// with atomics into zero-initialised buffers
int launch_ln_bwd(const float* dh1, const float* h0, const float* ln_w, float* dh0, float* dln_w, float* dln_b, int N, int C, int64_t T,
                  hipStream_t s);
// depthwise k=7 conv backward: dx = dres + conv_transpose(dh0) (dres nullable), ddw (C, 7), ddb (C) overwritten
int launch_dwconv_bwd(const float* dh0, const float* x, const float* dw_w, const float* dres, float* dx, float* ddw, float* ddb, int N, int C,
                      int64_t T, hipStream_t s);
// dw (Cout, Cin, taps) = sum_{b,t} dy[b,co,t] * x[b,ci,t + k*dil - pad], pad = dil*(taps-1)/2; db (Cout) nullable.  Overwrites.
int launch_conv_wgrad(const float* x, const float* dy, float* dw, float* db, int Cout, int Cin, int taps, int dil, int B, int64_t T,
                      hipStream_t s);
// `taps` adjacent taps of a strided (transposed) conv:
//   dw[(r*Ccols + c)*taps_out + tap_out + k] += sum_{b,t} rows[b,r,t] * cols[b,c,t*xstride + xoff + k],  k < taps
// rows_absmax: the bit pattern of max |rows| if the caller already has it (launch_absmax), else nullptr (reduced inside when needed)
int launch_conv_wgrad_strided(const float* rows, const float* cols, float* dw, int Crows, int Ccols, int xstride, int xoff, int64_t T,
                              int64_t Tx, int taps_out, int tap_out, int B, hipStream_t s, int taps = 1, const uint32_t* rows_absmax = nullptr);
// max |v| over n floats, as a bit pattern, into a device word owned by the library (a ring of words per host thread): the scale of the
// fp16-split kernels that read gradient tensors.  Runs on `s`; *slot stays valid for the next ~1000 calls of this thread.
int launch_absmax(const float* v, int64_t n, hipStream_t s, const uint32_t** slot);
int launch_conv_bgrad(const float* dy, float* db, int Cout, int B, int64_t T, hipStream_t s);

}  // namespace dmel
