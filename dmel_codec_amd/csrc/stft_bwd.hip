// Backward of the STFT magnitude (what makes the multi-resolution STFT loss a LOSS), for gfx950.
//
// Forward (utils/spectrogram.py:58-76, dmel_stft_f32): ypad = reflect_pad(y, pad, pad), pad = (n_fft - hop) / 2;
//   X[k, t] = sum_n ypad[t hop + n] w[n] e^{-2 pi i k n / N},  k = 0 .. N/2;   mag = sqrt(re^2 + im^2 + 1e-9).
// Given g = dL/dmag (B, T, N/2 + 1):
//   dre = g re / mag,  dim = g im / mag;
//   dframe[n, t] = w[n] sum_k (dre[k, t] cos(2 pi k n / N) - dim[k, t] sin(2 pi k n / N))      (one-sided: no doubling of the inner bins)
//   dypad[s] = sum_t dframe[s - t hop, t]  (overlap-add);   dy = dypad with the reflect padding folded back onto the samples it mirrors.
// The forward kernel keeps only magnitudes, so the backward recomputes re / im.  Both transforms are dense (N + 2) x N matrices applied
// to every frame -- GEMMs, and they run on the library's own implicit-GEMM convolution kernel (split-fp32 arithmetic):
//   1. X (B, N + 2, T) = conv1d(ypad viewed as N "channels" at element stride 1 and frame stride hop, F),  F = [w cos ; -w sin]
//      (the strided view costs nothing: SegDesc::tstride = hop, SegRun::cstride = 1);
//   2. G (B, N + 2, T): dre / dim from g and X                                                     (stft_grad_spectrum_kernel);
//   3. dframe (B, N, T) = conv1d(G, F^T) (pointwise);
//   4. overlap-add + reflect fold -> dy (B, L)                                                      (stft_overlap_add_kernel, a gather: no atomics).
// Tested against torch.autograd through the torch.stft restatement of the forward in float64 (tests/test_gpu_train.py).
#include "ops.h"

#include <cmath>
#include <vector>

namespace dmel {

__global__ __launch_bounds__(256) void stft_reflect_pad_kernel(const float* __restrict__ y, int64_t ystride, float* __restrict__ yp, int L, int pad,
                                                               int Lp) {
  const int b = blockIdx.y;
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= Lp) return;
  int j = s - pad;
  if (j < 0) j = -j;                                  // reflect without repeating the edge sample: ypad[pad - 1 - i] = y[1 + i]
  if (j >= L) j = 2 * (L - 1) - j;                    // ypad[pad + L + i] = y[L - 2 - i]
  yp[(int64_t)b * Lp + s] = y[(int64_t)b * ystride + j];
}

// X: (B, 2 K, T) rows [re(0..K-1) | im(0..K-1)], g: (B, T, K) frame-major; G overwrites X
__global__ __launch_bounds__(256) void stft_grad_spectrum_kernel(float* __restrict__ X, const float* __restrict__ g, int K, int T) {
  const int b = blockIdx.z, k = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float* re = X + ((int64_t)b * 2 * K + k) * T + t;
  float* im = re + (int64_t)K * T;
  const float r = *re, i = *im;
  const float mag = sqrtf(r * r + i * i + 1e-9f);
  const float s = g[((int64_t)b * T + t) * K + k] / mag;
  *re = s * r;
  *im = s * i;
}

// dy[b, j] = sum over the positions s of ypad that mirror sample j (itself, and its reflections) of sum_t F[s - t hop, t]
__global__ __launch_bounds__(256) void stft_overlap_add_kernel(const float* __restrict__ F, float* __restrict__ dy, int64_t dystride, int N, int T,
                                                               int hop, int L, int pad) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= L) return;
  const float* Fb = F + (int64_t)b * N * T;
  auto at = [&](int s) {                               // dypad[s]
    // frames t with 0 <= s - t hop < N
    const int t_hi = min(T - 1, s / hop);
    const int t_lo = max(0, (s - N + hop) / hop);      // ceil((s - N + 1) / hop) for s - N + 1 > 0
    float acc = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
      const int n = s - t * hop;
      if (n >= 0 && n < N) acc += Fb[(int64_t)n * T + t];
    }
    return acc;
  };
  float v = at(pad + j);
  if (j >= 1 && j <= pad) v += at(pad - j);                       // left mirror: ypad[pad - j] = y[j]
  if (j <= L - 2 && j >= L - 1 - pad) v += at(pad + 2 * (L - 1) - j);   // right mirror: ypad[pad + L + i] = y[L - 2 - i]
  dy[(int64_t)b * dystride + j] = v;
}

}  // namespace dmel

struct dmel_stft_grad {
  int n_fft = 0, win = 0, hop = 0, pad = 0;
  dmel::PackedConv fwd, bwd;      // F: (N + 2) x N on the strided frame view;  F^T: N x (N + 2), pointwise
};

extern "C" int dmel_stft_grad_create(dmel_stft_grad** out, int n_fft, int win_length, int hop_length, const float* window_host) {
  using namespace dmel;
  DMEL_CHECK_ARG(out, "stft_grad_create: NULL argument");
  DMEL_CHECK_ARG(n_fft >= 16 && n_fft <= 4096 && (n_fft % 16) == 0 && win_length > 0 && win_length <= n_fft && hop_length > 0 &&
                     hop_length <= n_fft && ((n_fft - hop_length) % 2) == 0,
                 "stft_grad_create: n_fft %d (multiple of 16, <= 4096), win_length %d, hop_length %d ((n_fft - hop) even) out of range", n_fft,
                 win_length, hop_length);
  auto* h = new dmel_stft_grad();
  h->n_fft = n_fft; h->win = win_length; h->hop = hop_length; h->pad = (n_fft - hop_length) / 2;
  const int N = n_fft, K = N / 2 + 1;
  std::vector<double> w(N, 0.0);
  const int woff = (N - win_length) / 2;               // torch.stft centres a short window in n_fft
  for (int n = 0; n < win_length; ++n)
    w[woff + n] = window_host ? (double)window_host[n] : 0.5 - 0.5 * std::cos(2.0 * M_PI * (double)n / (double)win_length);
  // F[row, n]: rows 0..K-1 = w[n] cos(2 pi k n / N), rows K..2K-1 = -w[n] sin(2 pi k n / N); angles reduced exactly (k n mod N) in float64
  std::vector<float> Fm((size_t)2 * K * N);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) {
      const double th = 2.0 * M_PI * (double)(((int64_t)k * n) % N) / (double)N;
      Fm[(size_t)k * N + n] = (float)(w[n] * std::cos(th));
      Fm[(size_t)(K + k) * N + n] = (float)(-w[n] * std::sin(th));
    }
  PackDesc d;
  d.mode = EPI_LINEAR; d.nseg = 1; d.C = 2 * K; d.phases = 1;
  d.seg[0].Cin = N; d.seg[0].taps = 1; d.seg[0].dil = 1; d.seg[0].pad_left = 0; d.seg[0].tstride = hop_length; d.seg[0].toff = 0;
  int rc = pack_conv(h->fwd, d, [&](int, int row, int ci, int) { return Fm[(size_t)row * N + ci]; }, [](int) { return 0.f; });
  if (rc == DMEL_OK) {
    PackDesc e;
    e.mode = EPI_LINEAR; e.nseg = 1; e.C = N; e.phases = 1;
    e.seg[0].Cin = 2 * K; e.seg[0].taps = 1; e.seg[0].dil = 1; e.seg[0].pad_left = 0;
    rc = pack_conv(h->bwd, e, [&](int, int row, int ci, int) { return Fm[(size_t)ci * N + row]; }, [](int) { return 0.f; });
  }
  if (rc != DMEL_OK) { delete h; return rc; }
  *out = h;
  return DMEL_OK;
}
extern "C" void dmel_stft_grad_destroy(dmel_stft_grad* h) { delete h; }

static size_t stft_grad_plan(const dmel_stft_grad* h, int B, int64_t L, void* ws, float** yp, float** X, float** F) {
  using namespace dmel;
  const int64_t T = L / h->hop, Lp = L + 2 * h->pad;
  const int K = h->n_fft / 2 + 1;
  Arena a(ws, (size_t)-1);
  float* p0 = a.take<float>((size_t)B * Lp + 64);
  float* p1 = a.take<float>((size_t)B * 2 * K * T);
  float* p2 = a.take<float>((size_t)B * h->n_fft * T);
  if (yp) { *yp = p0; *X = p1; *F = p2; }
  return align_up(a.off, 256);
}
extern "C" size_t dmel_stft_grad_workspace_bytes(const dmel_stft_grad* h, int B, int64_t L) {
  if (!h || B <= 0 || L <= 0) return 0;
  return stft_grad_plan(h, B, L, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int dmel_stft_magnitude_backward_f32(const dmel_stft_grad* h, const float* audio, int64_t audio_row_stride, const float* grad_linear,
                                                float* daudio, int64_t daudio_row_stride, int B, int64_t L, void* workspace, size_t workspace_bytes,
                                                void* stream) {
  using namespace dmel;
  DMEL_CHECK_ARG(h && audio && grad_linear && daudio && workspace, "stft_magnitude_backward: NULL argument");
  DMEL_CHECK_ARG(B > 0 && B <= 65535 && L > h->pad && L >= h->hop && (L + h->n_fft) * (int64_t)h->hop < ((int64_t)1 << 30),
                 "stft_magnitude_backward: bad shape (L must exceed the reflect pad %d and L * hop must stay below 2^30)", h->pad);
  DMEL_CHECK_ARG(audio_row_stride >= L && daudio_row_stride >= L, "stft_magnitude_backward: row stride shorter than the row");
  float *yp, *X, *F;
  const size_t need = stft_grad_plan(h, B, L, workspace, &yp, &X, &F);
  DMEL_CHECK_ARG(workspace_bytes >= need, "stft_magnitude_backward: workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const int N = h->n_fft, K = N / 2 + 1, hop = h->hop, pad = h->pad;
  const int Lp = (int)L + 2 * pad;
  const int T = (int)(L / hop);                           // = 1 + (Lp - N) / hop for every L >= hop
  hipLaunchKernelGGL(stft_reflect_pad_kernel, dim3((unsigned)((Lp + 255) / 256), (unsigned)B), dim3(256), 0, st, audio, audio_row_stride, yp, (int)L,
                     pad, Lp);
  DMEL_HIP(hipGetLastError());
  {  // 1. X = F applied to the frames (a strided view of ypad)
    ConvRun r;
    r.seg[0].x = yp; r.seg[0].bstride = Lp; r.seg[0].cstride = 1; r.seg[0].Tin = (int64_t)(T - 1) * hop + 1;
    r.B = B; r.Tcols = T; r.y = X; r.y_bs = (int64_t)2 * K * T; r.y_cs = T; r.Tout = T;
    r.precision = DMEL_PRECISION_FP32;
    DMEL_TRY(launch_conv(h->fwd, r, st));
  }
  hipLaunchKernelGGL(stft_grad_spectrum_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)K, (unsigned)B), dim3(256), 0, st, X, grad_linear, K, T);
  DMEL_HIP(hipGetLastError());
  {  // 3. dframe = F^T G
    ConvRun r;
    r.seg[0].x = X; r.seg[0].bstride = (int64_t)2 * K * T; r.seg[0].cstride = T; r.seg[0].Tin = T;
    r.B = B; r.Tcols = T; r.y = F; r.y_bs = (int64_t)N * T; r.y_cs = T; r.Tout = T;
    r.precision = DMEL_PRECISION_FP32;
    DMEL_TRY(launch_conv(h->bwd, r, st));
  }
  hipLaunchKernelGGL(stft_overlap_add_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)B), dim3(256), 0, st, F, daudio, daudio_row_stride, N, T, hop,
                     (int)L, pad);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}
