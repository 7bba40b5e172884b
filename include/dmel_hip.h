/*
 * dmel_hip.h -- C ABI of libdmel_hip.so, the MI355X (gfx950) native dMel codec hot path.
 *
 * Drop-in boundary: the reference's plugin surface is its Python module API
 * (Hydra `_target_` strings); its one native ABI is `fwd_cuda` of the anti-alias activation
 * (models/modules/bigvgan/alias_free_activation/cuda/anti_alias_activation_cuda.cu:212,
 *  anti_alias_activation.cpp:21-23), a C++ (torch::Tensor) ABI.  This header is what a binding of the
 * reference's path binds instead: plain pointers, sizes and an opaque stream -- no torch types.
 *
 * Conventions
 *   - every `const float*`/`float*` tensor argument is DEVICE memory unless the name ends in `_host`;
 *   - the caller owns every buffer, including the workspace (size from the matching *_workspace_bytes);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all launches are asynchronous on
 *     it and the library never synchronises inside a forward call;
 *   - return value: 0 = ok, <0 = invalid argument / unsupported configuration, >0 = hipError_t;
 *     dmel_last_error() returns a thread-local description of the last failure;
 *   - layouts are the reference's: activations (B, C, T) fp32 with T contiguous, audio (B, L) fp32,
 *     token ids (B, G*R, T4) int32.
 */
#ifndef DMEL_HIP_H
#define DMEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMEL_OK 0
#define DMEL_EINVAL (-1)
#define DMEL_EUNSUPPORTED (-2)
#define DMEL_EMISSING (-3)

/* Arithmetic of the convolutions of a handle (dmel_*_set_precision).  Tensors are fp32 in HBM in every mode.
 *  FP32 (default, the parity path): fp32-grade products, fp32 accumulation; results within 1e-4 of the reference run in
 *      fp32.  The library is free to form the products either on the fp32 MFMA or from an exact three-way bf16 split of
 *      both operands on the bf16 matrix cores (six partial products, dropped terms < 2^-21 relative; DESIGN.md "split
 *      fp32") -- both meet the same error bound against an fp64 evaluation and tests hold them to it.
 *  FP32_MFMA: force the native v_mfma_f32_32x32x2_f32 kernel (every product one exact fp32 fma).
 *  FP32_F16X2: fp32-grade products from a TWO-way fp16 split of both operands (a = a_hi + 2^-11 a_lo, a_hi = fp16(a) round to nearest,
 *      a_lo = fp16((a - a_hi) 2^11): 22 significant bits, |a - (a_hi + 2^-11 a_lo)| <= 2^-22 |a|, against the 24 of fp32) and THREE partial products on the fp16 matrix
 *      cores (a_hi b_hi into one fp32 accumulator, a_hi b_lo + a_lo b_hi into a second one that is folded in with 2^-11 at the end; the
 *      dropped a_lo b_lo is < 2^-22 |ab|): the "3xTF32" construction with fp16 pieces (same 22 bits), whose errors are random in sign and
 *      vanish under the rounding of a K ~ 10^3 fp32 accumulation -- half the matrix-core time of the six-product
 *      bf16 split for the same error against an fp64 evaluation (tests hold both to the same bound).  Inputs are staged as x 2^-6 and
 *      the weight image carries 2^6, so the fp16 range covers |x| < 4.19e6 (larger magnitudes overflow to inf -- loudly) and values
 *      below 2^-8 keep an ABSOLUTE error of 2^-30: use it for activations, not for back-propagated gradients.
 *      DOMAIN, stated once more because inference launches do not rescale their inputs (training launches over gradients do):
 *          |x| < 4.19e6 (= 2^6 * 65504), relative error 2^-22 for |x| >= 2^-8 = 3.9e-3, absolute error <= 2^-30 = 9.3e-10 below that
 *          (0.5 % of a 1e-7 input, 7e-5 of a 1e-5 one: tests/test_gpu_parity.py::test_f16_split_conv_over_the_magnitude_range).
 *      Audio-network activations (|x| ~ 1e-3 .. 1e2) sit inside it.  DMEL_DEBUG_F16_RANGE=1 makes every such launch reduce max |x| of
 *      its input, synchronise and FAIL if the bound is exceeded (a debugging aid: it serialises the stream).  The vocoder handle
 *      selects it by default (DMEL_PRECISION_FP32 there means "fp32-grade, library's choice"); WaveNet handles keep the six-product
 *      split unless asked, because the encoder's token ids are defined by it.  Training entry points choose per launch: the six-product split wherever a gradient
 *      tensor is an operand as it is, the fp16 split where the operand is an activation (discriminator / decoder forward) or a gradient
 *      tensor the library first scales by its own maximum (the discriminator's backward-data and long-row weight gradients: a reduction on
 *      the same stream finds max |dy|, the kernel stages dy x 2^(13 - exponent)).
 *  FP32_BF16X3: force the six-product bf16 split where FP32 would pick the fp16 one.
 *  BF16: opt-in throughput mode, the library-side equivalent of running the reference's codec under dtype: bfloat16
 *      (config/lm/lm_config.yaml:1,83; models/lm_lit_modules.py:52-55): weights and staged activations rounded to bf16
 *      (nearest-even), fp32 accumulation; activations, biases, residuals and the quantizer stay fp32.  More accurate than
 *      the reference's bf16 autocast, but NOT within the 1e-4 bar. */
#define DMEL_PRECISION_FP32 0
#define DMEL_PRECISION_BF16 1
#define DMEL_PRECISION_FP32_MFMA 2
#define DMEL_PRECISION_FP32_F16X2 3
#define DMEL_PRECISION_FP32_BF16X3 4

const char* dmel_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int dmel_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * STFT -> magnitude -> mel -> log            replaces utils/spectrogram.py:41-81 (LinearSpectrogram.forward)
 * The plan caches the hann window, FFT twiddles and the sparse Slaney mel basis on the device, as the
 * reference caches mel basis + window per (config, device) (utils/spectrogram.py:43-56).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dmel_stft_plan dmel_stft_plan;

/* f_max <= 0 means sample_rate/2 (librosa default when fmax=None).  window_host: win_length floats or
 * NULL for the periodic hann window torch.hann_window(win_length) (spectrogram.py:53).
 * Supported: n_fft in {512, 1024, 2048}, win_length <= n_fft, (n_fft - hop_length) even, n_mels <= 128. */
int dmel_stft_plan_create(dmel_stft_plan** plan, int sample_rate, int n_fft, int win_length, int hop_length,
                          int n_mels, double f_min, double f_max, const float* window_host);
void dmel_stft_plan_destroy(dmel_stft_plan* plan);
/* Host-only helper (no device needed): the dense (n_mels, n_fft/2+1) Slaney mel basis, librosa 0.10.2
 * filters.mel(htk=False, norm="slaney") semantics (call site utils/spectrogram.py:45-52).  f_max <= 0: sr/2. */
int dmel_mel_basis_host(int sample_rate, int n_fft, int n_mels, double f_min, double f_max, float* basis_host);
/* Copy the dense (n_mels, n_fft/2+1) mel basis the plan was built from to host memory. */
int dmel_stft_plan_mel_basis(const dmel_stft_plan* plan, float* basis_host);
/* Number of frames for an L-sample clip: 1 + (L + 2*pad - n_fft) / hop with pad = (n_fft-hop)/2. */
int64_t dmel_stft_num_frames(const dmel_stft_plan* plan, int64_t L);
/* audio (B, L) with row stride audio_row_stride (elements).  out (B, n_mels, T).
 * lengths: NULL, or B int64 sample counts: frames t >= lengths[b]/hop are written as 0 (the `mels * mask`
 * of codec_lit_modules.py:492-506 fused in). */
int dmel_stft_logmel_f32(const dmel_stft_plan* plan, const float* audio, int64_t audio_row_stride,
                         const int64_t* lengths, float* out, int B, int64_t L, void* stream);

/* Several streams on one GPU (round 3; dmel_codec_amd.pipeline.CodecLanes sets it): on != 0 makes every later STFT launch of the process ask for
 * 152 KB of LDS, i.e. one workgroup per CU and no room beside it for a workgroup of the convolution kernels.  Launched from different
 * streams, the two kinds of workgroup otherwise end up on one CU, and 3-11 % of the STFT launches then return one wrong frame (measured,
 * mechanism not understood: csrc/stft_logmel.hip, profiles/r03_stft_concurrency.txt).  No reference counterpart (utils/spectrogram.py:58-79 is
 * one torch.stft call on the current stream).  Costs nothing at the codec's sizes (<= 256 workgroups), 2.5x at 256 x 60 s. */
int dmel_stft_set_exclusive_cu(int on);
/* The same launch with the linear magnitudes sqrt(re^2 + im^2 + 1e-9) (utils/spectrogram.py:76) as a second, optional output:
 * linear (B, T, n_fft/2 + 1), frame-major so that every frame is one coalesced row; logmel_out may be NULL (then the mel stage is
 * skipped).  Consumer: the multi-resolution STFT loss BASELINE.json's north_star names (absent from the reference). */
int dmel_stft_f32(const dmel_stft_plan* plan, const float* audio, int64_t audio_row_stride, const int64_t* lengths,
                  float* logmel_out /*nullable*/, float* linear_out /*nullable*/, int B, int64_t L, void* stream);

/* Backward of the linear magnitudes of dmel_stft_f32 -- what turns the multi-resolution STFT loss BASELINE.json's north_star names into a
 * LOSS (the reference has neither; its STFT framing is utils/spectrogram.py:58-76).  grad_linear (B, T, n_fft/2 + 1) = dL/d|S|, frame-major
 * like linear_out; daudio (B, L) is overwritten with dL/daudio.  The handle holds the windowed DFT matrix and its transpose in MFMA tile
 * order: both transforms run as GEMMs on the library's convolution kernel (split-fp32 arithmetic), followed by an overlap-add that folds
 * the reflect padding back.  n_fft any multiple of 16 up to 4096 (the forward kernel: 512 / 1024 / 2048). */
typedef struct dmel_stft_grad dmel_stft_grad;
int dmel_stft_grad_create(dmel_stft_grad** out, int n_fft, int win_length, int hop_length, const float* window_host /*nullable: periodic hann*/);
void dmel_stft_grad_destroy(dmel_stft_grad* h);
size_t dmel_stft_grad_workspace_bytes(const dmel_stft_grad* h, int B, int64_t L);
int dmel_stft_magnitude_backward_f32(const dmel_stft_grad* h, const float* audio, int64_t audio_row_stride, const float* grad_linear,
                                     float* daudio, int64_t daudio_row_stride, int B, int64_t L, void* workspace, size_t workspace_bytes,
                                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sample-rate conversion in front of the STFT    replaces torchaudio.functional.resample as called by
 * LogMelSpectrogram.forward(x, sample_rate=...) (utils/spectrogram.py:122-123): polyphase windowed-sinc filter bank,
 * y[b, n * new + p] = sum_k bank[p][k] * xpad[b, n * orig + k], orig_freq / new_freq already divided by their gcd, xpad = x with `width`
 * zeros in front, kw = 2 * width + orig_freq taps per phase.  filter_bank_dev: (new_freq, kw) fp32 on the device (the caller builds it:
 * it depends on torchaudio's rolloff / window choices, which are host-side policy); x (B, L), y (B, Lout), Lout <= ceil(new * L / orig).
 * ---------------------------------------------------------------------------------------------- */
int dmel_resample_f32(const float* x, float* y, const float* filter_bank_dev, int B, int64_t L, int64_t Lout, int orig_freq,
                      int new_freq, int width, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data front end on the GPU: what LhotseTTSDataset.__getitem__ + collate_fn do to the decoded clips of
 * a batch (dataset/lhotse_tts_dataset.py:29-32, :46-65): every clip peak-normalised, `librosa.util.normalize(audio) * 0.95` =
 * x / max|x| * peak (a clip whose peak is below the smallest normal float is left unscaled, as librosa does), right-padded with zeros
 * to Lmax and stacked: audios (B, 1, Lmax) f32, audio_lengths (1, B) int32.  clips_dev: B device pointers (device array) to the mono
 * clips where the decoder / resampler left them; lengths_dev: their sample counts (device, int64); order_dev (nullable, device int32):
 * output row b takes clip order[b] (the reference sorts a batch longest-first on the host, :20); peaks_scratch: B words.
 * ---------------------------------------------------------------------------------------------- */
int dmel_collate_peak_f32(const float* const* clips_dev, const int64_t* lengths_dev, const int32_t* order_dev /*nullable*/, float* audios,
                          int32_t* audio_lengths, uint32_t* peaks_scratch, int B, int64_t Lmax, float peak, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Anti-aliased Snake / SnakeBeta           replaces fwd_cuda (anti_alias_activation_cuda.cu:212-246) and the
 * torch Activation1d (alias_free_activation/torch/act.py:25-30): x2 up (12-tap kaiser-sinc, replicate pad)
 * -> x + 1/(b+1e-9) sin^2(a x) -> x2 down (12 taps, replicate pad 5/6).  alpha/beta: (C) as stored in the
 * state dict; logscale != 0 applies exp() as the reference does.  beta == NULL means Snake (beta := alpha).
 * up_filter12_host / down_filter12_host: the 12 taps of UpSample1d.filter and DownSample1d.lowpass.filter (host pointers; they
 * become launch constants) -- the two arguments of fwd_cuda(input, up_filter, down_filter, alpha, beta)
 * (anti_alias_activation.cpp:19-23); the reference registers the same kaiser-sinc taps for both.
 * ---------------------------------------------------------------------------------------------- */
int dmel_aa_snake_f32(const float* x, float* y, const float* alpha, const float* beta, const float* up_filter12_host,
                      const float* down_filter12_host, int logscale, int B, int C, int64_t T, void* stream);
/* Backward of dmel_aa_snake_f32 (the reference's fused kernel has none: alias_free_activation/cuda/activation1d.py:29-32;
 * SURVEY.md section 8(f) rank 1, C-ABI row `aa_snake(+_bwd)`): dx (B, C, T), dalpha (C), dbeta (C; NULL exactly when beta is NULL,
 * i.e. Snake, whose single parameter then receives both contributions).  Gradients are with respect to the STORED parameters
 * (the log-scale ones when logscale != 0).  dalpha / dbeta are overwritten.  Summation order is not fixed (atomics). */
int dmel_aa_snake_backward_f32(const float* x, const float* dy, float* dx, const float* alpha, const float* beta /*nullable*/,
                               float* dalpha, float* dbeta /*nullable*/, const float* up_filter12_host,
                               const float* down_filter12_host, int logscale, int B, int C, int64_t T, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Module handles.  Weights are handed over as HOST fp32 arrays under the reference's state-dict key names
 * (key names are part of the contract); the library folds weight norm (weight_g / weight_v),
 * re-tiles them for the MFMA kernels and uploads them.  *_finalize fails with DMEL_EMISSING when a required
 * key was never set.
 * ---------------------------------------------------------------------------------------------- */

/* WaveNet      replaces models/modules/wavenet.py:138-225 (forward :204-225, block :116-135) */
typedef struct dmel_wavenet dmel_wavenet;
int dmel_wavenet_create(dmel_wavenet** m, int input_channels /*0 = same as residual*/, int output_channels /*0 = none*/,
                        int residual_channels, int residual_layers, int dilation_cycle, int condition_channels /*0 = none*/);
void dmel_wavenet_destroy(dmel_wavenet* m);
int dmel_wavenet_set_precision(dmel_wavenet* m, int precision);   /* DMEL_PRECISION_*; may be changed between forwards */
int dmel_wavenet_set_tensor(dmel_wavenet* m, const char* key, const float* data_host, const int64_t* shape, int ndim);
int dmel_wavenet_finalize(dmel_wavenet* m);
size_t dmel_wavenet_workspace_bytes(const dmel_wavenet* m, int N, int64_t T);
/* x (N, Cin, T), condition (N, Ccond, T) or NULL, y (N, Cout, T).
 * in_lengths / out_lengths: NULL or N int64 frame counts; x is read as x * (t < in_lengths[n]) and y is
 * written as y * (t < out_lengths[n])  (the mask multiplies of codec_lit_modules.py:471-477,505-506).
 * group_repeat: lengths index = n / group_repeat (expand_mask, codec_lit_modules.py:156-157). */
int dmel_wavenet_forward(const dmel_wavenet* m, const float* x, const float* condition, float* y, int N, int64_t T,
                         const int64_t* in_lengths, const int64_t* out_lengths, int group_repeat,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Training path of the WaveNet (first module of SURVEY.md section 8(f) rank 1; C-ABI row `wavenet_block(+_bwd)`): what
 * `manual_backward(loss)` differentiates through autograd in the reference (codec_lit_modules.py:236,315 over
 * wavenet.py:116-135,204-225).  enable_training must precede finalize (the transposed weight images are packed there).
 * forward_train = forward without masks, keeping in `workspace` what backward needs; backward must be given the SAME
 * workspace, untouched, plus x / condition again.  Parameter gradients are written (overwritten) into one flat fp32 device
 * buffer of dmel_wavenet_grad_floats() elements; dmel_wavenet_grad_slot maps a state-dict key to its (offset, numel) there
 * (layout of each slot = the parameter's own layout).  dx / dcondition may be NULL when not needed.  Gradients are
 * fp32-grade regardless of the handle's precision. */
int dmel_wavenet_enable_training(dmel_wavenet* m, int on);
/* bf16 training mode (BASELINE config 3 "DDP bf16"; the reference switches dtype / Lightning precision, codec_lit_modules.py:52-56,
 * config/lm/lm_config.yaml:1,83): DMEL_PRECISION_BF16 makes every convolution of forward_train / backward run with operands rounded to
 * bf16 and fp32 accumulation -- forward, backward-data, and the weight gradients of rows of >= 256 samples (shorter rows keep the exact
 * fp32 kernel) -- while parameters, activations in HBM, gradients and the optimiser stay fp32 (autocast semantics).  The same setter
 * exists for the other three trainable handles.  Default DMEL_PRECISION_FP32 (the parity path). */
int dmel_wavenet_set_train_precision(dmel_wavenet* m, int precision);
/* Re-pack every weight image of a finalized handle from DEVICE tensors (after an optimiser step): keys / device_tensors
 * name the state-dict tensors (weights (Cout, Cin, k) and biases, contiguous fp32) as they currently live on the device.
 * Runs on `stream`, no host copy, no allocation; produces bit-identical images to set_tensor + finalize on the same values. */
int dmel_wavenet_refresh(dmel_wavenet* m, int n, const char* const* keys, const float* const* device_tensors, void* stream);
size_t dmel_wavenet_train_workspace_bytes(const dmel_wavenet* m, int N, int64_t T);
int64_t dmel_wavenet_grad_floats(const dmel_wavenet* m);
int dmel_wavenet_grad_slot(const dmel_wavenet* m, const char* key, int64_t* offset, int64_t* numel);
int dmel_wavenet_forward_train(const dmel_wavenet* m, const float* x, const float* condition, float* y, int N, int64_t T,
                               void* workspace, size_t workspace_bytes, void* stream);
int dmel_wavenet_backward(const dmel_wavenet* m, const float* x, const float* condition, const float* dy, float* dx /*nullable*/,
                          float* dcondition /*nullable*/, float* grads, int N, int64_t T, void* workspace, size_t workspace_bytes,
                          void* stream);
/* The exchange step of data-parallel training (the reference gets it from Lightning's DDP wrapper, config/codec/dMel_example.yaml:14:
 * gradient buckets are all-reduced while backward is still running).  dmel_wavenet_backward_hooked is dmel_wavenet_backward that calls
 * `on_ready(user, offset, numel)` on the calling host thread right after it has ENQUEUED the last kernel that writes
 * grads[offset, offset + numel): first the tail (skip_projection + output_projection), then residual_layers L-1 ... 0 (one contiguous
 * region per block: the bucket SURVEY section 8(e) asks for), then input_projection.  The regions are disjoint and cover the whole
 * buffer.  A caller that records an event on `stream` inside the callback and lets a communication stream wait for it overlaps the
 * all-reduce of block k with the backward of blocks k-1 ... 0.  on_ready == NULL: plain dmel_wavenet_backward. */
typedef void (*dmel_grad_ready_fn)(void* user, int64_t offset, int64_t numel);
int dmel_wavenet_backward_hooked(const dmel_wavenet* m, const float* x, const float* condition, const float* dy, float* dx /*nullable*/,
                                 float* dcondition /*nullable*/, float* grads, int N, int64_t T, void* workspace,
                                 size_t workspace_bytes, void* stream, dmel_grad_ready_fn on_ready, void* user);

/* Incremental (streaming) forward with state carry -- SURVEY.md section 8(f) rank 2: the LM emits tokens, the codec decodes while it
 * does (the reference decodes once at the end, models/lm_lit_modules.py:467-471).  Every block is a "same"-padded convolution, so block
 * l's output at time t needs block l-1's output up to t + dilation_l: instead of re-running a halo of old frames through the stack for
 * every chunk, the caller keeps the OUTPUT HISTORY of every block and each step only computes the new columns of every block.
 *   hist (L + 1, N, C, cap): hist[0] = the (masked) input x_0 (after input_projection, when the model has one: not supported here),
 *                            hist[l] = output of block l; time axis in window coordinates [0, cap)
 *   skip (N, C, cap): running sum of the blocks' skip outputs;  cond (N, Ccond, cap) or NULL;  y (N, Cout, cap)
 *   prev[l], next[l], l = 0..L: columns [0, prev[l]) of level l are already valid; the call makes [prev[l], next[l]) valid.
 *     next[0] = how far x_0 / cond have been written by the caller; mid-stream next[l] <= next[l-1] - dilation_l (checked);
 *     at the end of the sequence next[l] = next[0] = total length for every l (zero padding past the end, like the whole-sequence call).
 *   y gets columns [prev[L], next[L]).  scratch: N * 2 C * cap floats.
 * Numerically identical to dmel_wavenet_forward column by column (same kernels, same reduction order). */
int dmel_wavenet_stream_step(const dmel_wavenet* m, float* hist, float* skip, const float* cond /*nullable*/, float* y, float* scratch,
                             int N, int64_t cap, const int64_t* prev, const int64_t* next, void* stream);

/* ConvNeXtBlock (models/modules/firefly.py:337-402; C-ABI row `convnext_block`), standalone: y = x + gamma * pwconv2(gelu(pwconv1(
 * LayerNorm_C(dwconv7(x))))), x / y (N, dim, T).  set_tensor keys: dwconv.weight (dim,1,7), dwconv.bias, norm.weight, norm.bias,
 * pwconv1.weight (4 dim, dim), pwconv1.bias, pwconv2.weight (dim, 4 dim), pwconv2.bias, gamma.  The training entry points follow the
 * WaveNet ones (enable_training before finalize; forward_train keeps its intermediates in the workspace that backward is given again;
 * parameter gradients in one flat buffer addressed through grad_slot; dx is always produced). */
typedef struct dmel_convnext dmel_convnext;
int dmel_convnext_create(dmel_convnext** m, int dim);
void dmel_convnext_destroy(dmel_convnext* m);
int dmel_convnext_set_tensor(dmel_convnext* m, const char* key, const float* data_host, const int64_t* shape, int ndim);
int dmel_convnext_enable_training(dmel_convnext* m, int on);
int dmel_convnext_set_train_precision(dmel_convnext* h, int precision);   /* see dmel_wavenet_set_train_precision */
int dmel_convnext_finalize(dmel_convnext* m);
size_t dmel_convnext_workspace_bytes(const dmel_convnext* m, int N, int64_t T);
int dmel_convnext_forward(const dmel_convnext* m, const float* x, float* y, int N, int64_t T, void* workspace, size_t workspace_bytes,
                          void* stream);
size_t dmel_convnext_train_workspace_bytes(const dmel_convnext* m, int N, int64_t T);
int64_t dmel_convnext_grad_floats(const dmel_convnext* m);
int dmel_convnext_grad_slot(const dmel_convnext* m, const char* key, int64_t* offset, int64_t* numel);
int dmel_convnext_forward_train(const dmel_convnext* m, const float* x, float* y, int N, int64_t T, void* workspace,
                                size_t workspace_bytes, void* stream);
int dmel_convnext_backward(const dmel_convnext* m, const float* x, const float* dy, float* dx, float* grads, int N, int64_t T,
                           void* workspace, size_t workspace_bytes, void* stream);

/* DownsampleFiniteScalarQuantize (is_dmel=True, n_codebooks=1)   replaces models/modules/dowmsample_fsq.py:124-147
 * and vector_quantize_pytorch GroupedResidualFSQ.forward / get_output_from_indices. */
typedef struct dmel_quantizer dmel_quantizer;
int dmel_quantizer_create(dmel_quantizer** q, int input_dim /*= groups * dim_per_group*/, int n_groups,
                          const int* levels, int n_levels, const int* downsample_factor, int n_factors,
                          int fsq_prebound);
void dmel_quantizer_destroy(dmel_quantizer* q);
/* Strict encode ("hard parts"): project_in (Linear C -> n_levels) and the tanh bound(s) of
 * dmel_quantizer_encode are evaluated in float64 and rounded to fp32 once, so the value that is rounded to an id no longer depends
 * on a summation order or a tanhf implementation.  Off by default (the fp32 path follows the reference's arithmetic). */
int dmel_quantizer_set_strict(dmel_quantizer* q, int on);
int dmel_quantizer_set_tensor(dmel_quantizer* q, const char* key, const float* data_host, const int64_t* shape, int ndim);
int dmel_quantizer_finalize(dmel_quantizer* q);
size_t dmel_quantizer_workspace_bytes(const dmel_quantizer* q, int B, int64_t T);
/* z (B*G, C, T) -> ids (B, G, T4) int32, T4 = T / prod(factors).  prequant (optional, may be NULL):
 * (G, B, T4, n_levels) fp32, the bounded value that is rounded (for the near-tie analysis of the tests). */
int dmel_quantizer_encode(const dmel_quantizer* q, const float* z, int32_t* ids, float* prequant, int B, int64_t T,
                          void* workspace, size_t workspace_bytes, void* stream);
/* The same, additionally returning the down-sampled features the FSQ sees -- latents (B*G, C, T4), nullable -- so that a checker can
 * feed another FSQ implementation exactly the same input (strict-mode parity test). */
int dmel_quantizer_encode_ex(const dmel_quantizer* q, const float* z, int32_t* ids, float* prequant /*nullable*/,
                             float* latents /*nullable*/, int B, int64_t T, void* workspace, size_t workspace_bytes, void* stream);
/* ids (B, G, T4) -> z (B, G*C, T4*prod(factors)) */
int dmel_quantizer_decode(const dmel_quantizer* q, const int32_t* ids, float* z, int B, int64_t T4,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Training path of the quantiser: DownsampleFiniteScalarQuantize.forward (dowmsample_fsq.py:86-122) with the straight-through FSQ of
 * vector_quantize_pytorch, and its backward.  z (B*G, Cg, T) -> zq (B*G, Cg, T) (= (B, G*Cg, T); zero-padded from 2^nf * T4 back to T
 * with left = diff / 2), ids (B, G, T4) and latents (B*G, Cg, T4) (both nullable).  Same conventions as the WaveNet training entry
 * points (enable_training before finalize; the workspace of forward_train is handed to backward unchanged; one flat gradient buffer
 * addressed through grad_slot with the state-dict keys).  dz is always produced. */
int dmel_quantizer_enable_training(dmel_quantizer* q, int on);
int dmel_quantizer_set_train_precision(dmel_quantizer* h, int precision);   /* see dmel_wavenet_set_train_precision */
/* as dmel_wavenet_refresh: re-pack every weight image / parameter buffer from device tensors named by their state-dict keys */
int dmel_quantizer_refresh(dmel_quantizer* q, int n, const char* const* keys, const float* const* device_tensors, void* stream);
size_t dmel_quantizer_train_workspace_bytes(const dmel_quantizer* q, int B, int64_t T);
int64_t dmel_quantizer_grad_floats(const dmel_quantizer* q);
int dmel_quantizer_grad_slot(const dmel_quantizer* q, const char* key, int64_t* offset, int64_t* numel);
int dmel_quantizer_forward_train(const dmel_quantizer* q, const float* z, float* zq, int32_t* ids /*nullable*/,
                                 float* latents /*nullable*/, int B, int64_t T, void* workspace, size_t workspace_bytes, void* stream);
int dmel_quantizer_backward(const dmel_quantizer* q, const float* z, const float* dzq, float* dz, float* grads, int B, int64_t T,
                            void* workspace, size_t workspace_bytes, void* stream);

/* z[b,c,t] = z[b,c,t] * (t < lengths[b]) + (w[c] * value + bias[c])      codec_lit_modules.py:520-526
 * (quality_projection = nn.Linear(1, C) applied to the constant 2.0).  w, bias: device (C). */
int dmel_mask_add_quality_f32(float* z, const int64_t* lengths, const float* w, const float* bias, float value,
                              int B, int C, int64_t T, void* stream);

/* BigVGAN generator      replaces models/modules/bigvgan/bigvgan.py:367-393 (+ AMPBlock1 :132-141, AMPBlock2 :232-237) */
typedef struct dmel_bigvgan dmel_bigvgan;
typedef struct dmel_bigvgan_config {
  int num_mels;
  int upsample_initial_channel;
  int num_upsamples;
  int upsample_rates[8];
  int upsample_kernel_sizes[8];
  int num_kernels;
  int resblock_kernel_sizes[8];
  int resblock_dilations[8][3];
  int snake_logscale;      /* h.snake_logscale */
  int activation_snake;    /* 1: "snake", 0: "snakebeta" */
  int use_tanh_at_final;
  int use_bias_at_final;
  int resblock_type;       /* 1 (or 0): AMPBlock1 (bigvgan.py:31-147), 2: AMPBlock2 (bigvgan.py:150-241: per dilation act -> conv -> + x, keys convs.{l}) */
} dmel_bigvgan_config;
int dmel_bigvgan_create(dmel_bigvgan** m, const dmel_bigvgan_config* cfg);
void dmel_bigvgan_destroy(dmel_bigvgan* m);
int dmel_bigvgan_set_tensor(dmel_bigvgan* m, const char* key, const float* data_host, const int64_t* shape, int ndim);
int dmel_bigvgan_finalize(dmel_bigvgan* m);
size_t dmel_bigvgan_workspace_bytes(const dmel_bigvgan* m, int B, int64_t T);
/* Number of streams the AMP blocks of a stage are spread over: 1 = everything on the caller's stream, 3 (default) =
 * caller's stream + two library-owned side streams forked/joined with events inside every stage (the VALU-bound
 * activations of one block then overlap the MFMA-bound convolutions of another).  Results are identical.  A handle's
 * forward is not re-entrant from two host threads at once (it owns the fork/join events). */
int dmel_bigvgan_set_streams(dmel_bigvgan* m, int n_streams);
int dmel_bigvgan_set_precision(dmel_bigvgan* m, int precision);   /* DMEL_PRECISION_*; conv_post (C -> 1) always runs in fp32 */
/* mel (B, num_mels, T) -> audio (B, 1, T * prod(upsample_rates)) */
int dmel_bigvgan_forward(const dmel_bigvgan* m, const float* mel, float* audio, int B, int64_t T,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Discriminator (models/modules/discriminator.py:6-35): six weight-normed Conv2d (3 x 9 | 3 x 3, stride (1, 1|2)) + SiLU over the mel
 * image.  set_tensor keys as in the reference's state dict: blocks.{0,2,..,10}.bias, blocks.{i}.parametrizations.weight.original0|1
 * (weight norm is folded at finalize).  x (B, H = n_mels, W = frames) -> logits (B, H, dmel_discriminator_out_frames(W)).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dmel_discriminator dmel_discriminator;
int dmel_discriminator_create(dmel_discriminator** d);
void dmel_discriminator_destroy(dmel_discriminator* d);
int dmel_discriminator_set_tensor(dmel_discriminator* d, const char* key, const float* data_host, const int64_t* shape, int ndim);
int dmel_discriminator_finalize(dmel_discriminator* d);
int64_t dmel_discriminator_out_frames(const dmel_discriminator* d, int64_t W);
size_t dmel_discriminator_workspace_bytes(const dmel_discriminator* d, int B, int H, int64_t W);
int dmel_discriminator_forward(const dmel_discriminator* d, const float* x, float* y, int B, int H, int64_t W, void* workspace,
                               size_t workspace_bytes, void* stream);
/* Training path (same conventions as the WaveNet's): forward_train keeps every layer's input and pre-activation in the workspace,
 * backward returns dx (nullable) and the gradients of bias / weight-norm g (original0) / v (original1) of every layer in the flat
 * buffer (the chain through torch._weight_norm is applied here).  enable_training requires the weight-normed form of the weights. */
int dmel_discriminator_enable_training(dmel_discriminator* d, int on);
int dmel_discriminator_set_train_precision(dmel_discriminator* h, int precision);   /* see dmel_wavenet_set_train_precision */
/* as dmel_wavenet_refresh (keys: blocks.{i}.bias, ...original0, ...original1); the weight-norm fold runs on the device */
int dmel_discriminator_refresh(dmel_discriminator* d, int n, const char* const* keys, const float* const* device_tensors, void* stream);
size_t dmel_discriminator_train_workspace_bytes(const dmel_discriminator* d, int B, int H, int64_t W);
int64_t dmel_discriminator_grad_floats(const dmel_discriminator* d);
int dmel_discriminator_grad_slot(const dmel_discriminator* d, const char* key, int64_t* offset, int64_t* numel);
int dmel_discriminator_forward_train(const dmel_discriminator* d, const float* x, float* y, int B, int H, int64_t W, void* workspace,
                                     size_t workspace_bytes, void* stream);
int dmel_discriminator_backward(const dmel_discriminator* d, const float* dy, float* dx /*nullable*/, float* grads, int B, int H, int64_t W,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Single-op entry point for the implicit-GEMM conv kernel (tests, module mirrors).
 * y = conv1d(x, w, bias, dilation, padding = dilation*(k-1)/2)   w_host: (Cout, Cin, k) as nn.Conv1d stores it.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dmel_conv dmel_conv;
int dmel_conv_create(dmel_conv** c, const float* w_host, const float* bias_host /*nullable*/, int Cout, int Cin, int k,
                     int dilation);
void dmel_conv_destroy(dmel_conv* c);
int dmel_conv_set_precision(dmel_conv* c, int precision);
int dmel_conv_forward(const dmel_conv* c, const float* x, float* y, int B, int64_t T, void* stream);
/* Activation1d fused into the convolution that reads it: y = conv1d(Activation1d(x)) + bias (+ residual), the act -> conv pair of
 * AMPBlock1 / AMPBlock2.forward (bigvgan/bigvgan.py:132-141, :232-237; alias_free_activation/torch/act.py:25-30) as ONE kernel --
 * producer waves compute the anti-aliased Snake of the staged x tile in LDS while consumer waves run the MFMA loop, so the activated
 * tensor never exists in HBM.  x (B, Cin, T) is the tensor BEFORE the activation; alpha / beta (Cin; beta NULL = Snake), filters and
 * logscale as in dmel_aa_snake_f32; residual (B, Cout, T) nullable.  Always computes in DMEL_PRECISION_FP32_F16X2 and is
 * BIT-IDENTICAL to dmel_aa_snake_f32 followed by dmel_conv_forward at that precision.  (k - 1) * dilation <= 64. */
int dmel_conv_snake_forward(const dmel_conv* c, const float* x, const float* residual /*nullable*/, float* y, const float* alpha,
                            const float* beta /*nullable*/, const float* up_filter12_host, const float* down_filter12_host,
                            int logscale, int B, int64_t T, void* stream);
/* Backward of the same convolution -- what autograd runs for the reference (`loss.backward()`, codec_lit_modules.py:236,315 ->
 * ATen conv1d backward); first piece of the training path (SURVEY.md section 8(f) rank 1, C-ABI row `conv1d_dilated(+_bwd)`).
 *   backward_data:   dx (B, Cin, T)  = conv1d(dy, W transposed and tap-reversed)   -- the forward kernel on a second weight image
 *   backward_weight: dw (Cout, Cin, k) = sum_{b,t} dy[b,co,t] * x[b,ci,t + k*d - p], db (Cout) = sum_{b,t} dy   (db nullable)
 * dw / db are overwritten.  The K split of backward_weight uses fp32 atomics: results are reproducible to rounding, not
 * bitwise.  Gradients are always fp32-grade (a BF16 handle falls back to FP32 here). */
int dmel_conv_backward_data(dmel_conv* c, const float* dy, float* dx, int B, int64_t T, void* stream);
int dmel_conv_backward_weight(const dmel_conv* c, const float* x, const float* dy, float* dw, float* db /*nullable*/, int B,
                              int64_t T, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Standalone transposed convolution and output convolution      (convT1d, conv_post)
 * ConvTranspose1d(Cin, Cout, k = 2 * stride, stride, padding = stride / 2): every up-sampler of BigVGAN
 * (models/modules/bigvgan/bigvgan.py:320-334, applied at :371-374), run as `stride` phase sub-convolutions on the implicit-GEMM
 * kernel.  w_host (Cin, Cout, k) and bias_host (Cout, nullable) are HOST arrays in torch's ConvTranspose1d layout, weight norm
 * already folded.  x (B, Cin, T) -> y (B, Cout, T * stride), device, fp32.  Other k / stride / padding: DMEL_EUNSUPPORTED.
 * dmel_conv_post_f32: the C -> 1 convolution that ends the vocoder (bigvgan.py:386-391): y[b, 0, t] = act(bias + sum_{c,k} w[c, k]
 * x[b, c, t + k - K/2]), act 0 none / 2 tanh / 3 clamp to [-1, 1]; w_dev (C, K) DEVICE floats, K odd, C * K * 4 <= 48 KB.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dmel_conv_transpose dmel_conv_transpose;
int dmel_conv_transpose1d_create(dmel_conv_transpose** out, const float* w_host, const float* bias_host, int Cin, int Cout, int k, int stride);
void dmel_conv_transpose1d_destroy(dmel_conv_transpose* h);
int dmel_conv_transpose1d_set_precision(dmel_conv_transpose* h, int precision);   /* DMEL_PRECISION_* */
int dmel_conv_transpose1d_forward(const dmel_conv_transpose* h, const float* x, float* y, int B, int64_t T, void* stream);
int dmel_conv_post_f32(const float* x, const float* w_dev, float bias, int act, float* y, int B, int C, int K, int64_t T, void* stream);

/* Timing hook used by bench.py: when enabled, every launch of the named kernel family on `stream` is
 * bracketed by hipEvents; dmel_prof_read returns the launch count and total milliseconds since the last reset
 * (synchronises the events it reads).  family: "conv_igemm", "aa_snake", "stft_logmel", "small". */
int dmel_prof_enable(int on);
int dmel_prof_reset(void);
int dmel_prof_read(const char* family, int64_t* launches, double* total_ms, double* total_flops, double* total_bytes);
/* the same plus the matrix-core flops the launches ISSUED for those algorithmic flops (6x under the bf16 split, 3x under the fp16 split,
 * 1x with bf16 operands; 0 for families without a matrix-core kernel): total_flops / (total_issue_flops / dense peak) is the family's
 * matrix-core ceiling */
int dmel_prof_read_ex(const char* family, int64_t* launches, double* total_ms, double* total_flops, double* total_bytes,
                      double* total_issue_flops);

#ifdef __cplusplus
}
#endif
#endif /* DMEL_HIP_H */
