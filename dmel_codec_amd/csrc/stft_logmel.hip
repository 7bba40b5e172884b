// Fused reflect-pad -> hann -> 1024-point real FFT -> |.| -> sparse Slaney mel -> log(clamp) for gfx950.
//
// Replaces the six ATen launches of utils/spectrogram.py:58-79 (pad, stft, pow/sum/sqrt, matmul, clamp, log)
// and never materialises the (B, 513, T) complex spectrum.
//
// Mapping: one workgroup = 32 consecutive frames of one clip; each of its 4 waves transforms 8 frames, one
// frame at a time.  A frame's 1024 real samples are packed as 512 complex points z[n] = x[2n] + i x[2n+1];
// the 512-point complex FFT is three in-register radix-8 passes (64 lanes x 8 points) with two LDS
// transposes in between (padded so every ds_read/ds_write is bank-conflict free), then the split-radix
// style real-FFT recombination, magnitudes into LDS, one lane per mel band walking its triangle (the mel
// basis is 99 % zeros: each FFT bin feeds at most two bands), log, and a (n_mels x 32) LDS tile so that
// the (B, n_mels, T) store is made of 128-byte row segments.
//
// Algorithmic HBM bytes per frame: hop*4 read + n_mels*4 written.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <mutex>

namespace dmel {

// A/B switches (tools/build_variant.sh): waves per workgroup, and up to which P the window / pass-1 twiddles stay in registers
#ifndef DMEL_STFT_WAVES
#define DMEL_STFT_WAVES 4
#endif
#ifndef DMEL_STFT_REGTAB_MAXP
#define DMEL_STFT_REGTAB_MAXP 8
#endif
#ifndef DMEL_STFT_LDS_PAD
#define DMEL_STFT_LDS_PAD 0      // debug: bytes of unused LDS in front of and behind the kernel's own
#endif
#ifndef DMEL_STFT_FULL_TILE
#define DMEL_STFT_FULL_TILE 0      // debug: 1 = LDS sized for 128 bands whatever the plan (the round-2 allocation)
#endif
#ifndef DMEL_STFT_WPE
#define DMEL_STFT_WPE 0
#endif
#ifndef DMEL_STFT_FRAMES
#define DMEL_STFT_FRAMES 32
#endif
constexpr int kFramesPerWG = DMEL_STFT_FRAMES;
constexpr int kWaves = DMEL_STFT_WAVES;
constexpr int kThreads = 64 * kWaves;
constexpr int kFramesPerWave = kFramesPerWG / kWaves;
constexpr int kMaxMels = 128;

struct cf {
  float x, y;
};
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf mul_negi(cf a) { return {a.y, -a.x}; }  // a * (-i)

// 4-point forward DFT, natural order in and out.
__device__ __forceinline__ void dft4(cf c0, cf c1, cf c2, cf c3, cf& y0, cf& y1, cf& y2, cf& y3) {
  cf d0 = cadd(c0, c2), d1 = csub(c0, c2), d2 = cadd(c1, c3), d3 = mul_negi(csub(c1, c3));
  y0 = cadd(d0, d2);
  y1 = cadd(d1, d3);
  y2 = csub(d0, d2);
  y3 = csub(d1, d3);
}

// 8-point forward DFT (decimation in frequency), natural order in and out.
__device__ __forceinline__ void dft8(cf (&a)[8]) {
  const float h = 0.70710678118654752440f;
  cf b0 = cadd(a[0], a[4]), b4 = csub(a[0], a[4]);
  cf b1 = cadd(a[1], a[5]), t5 = csub(a[1], a[5]);
  cf b2 = cadd(a[2], a[6]), t6 = csub(a[2], a[6]);
  cf b3 = cadd(a[3], a[7]), t7 = csub(a[3], a[7]);
  cf b5 = {(t5.x + t5.y) * h, (t5.y - t5.x) * h};    // * (1 - i)/sqrt2
  cf b6 = mul_negi(t6);                              // * -i
  cf b7 = {(t7.y - t7.x) * h, -(t7.x + t7.y) * h};   // * (-1 - i)/sqrt2
  dft4(b0, b1, b2, b3, a[0], a[2], a[4], a[6]);
  dft4(b4, b5, b6, b7, a[1], a[3], a[5], a[7]);
}

// 16-point forward DFT (decimation in time over two 8-point DFTs), natural order in and out.
__device__ __forceinline__ void dft16(cf (&a)[16]) {
  cf e[8], o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { e[i] = a[2 * i]; o[i] = a[2 * i + 1]; }
  dft8(e);
  dft8(o);
  const cf w[8] = {{1.f, 0.f},
                   {0.92387953251128675613f, -0.38268343236508977173f},
                   {0.70710678118654752440f, -0.70710678118654752440f},
                   {0.38268343236508977173f, -0.92387953251128675613f},
                   {0.f, -1.f},
                   {-0.38268343236508977173f, -0.92387953251128675613f},
                   {-0.70710678118654752440f, -0.70710678118654752440f},
                   {-0.92387953251128675613f, -0.38268343236508977173f}};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const cf t = k == 0 ? o[0] : cmul(w[k], o[k]);
    a[k] = cadd(e[k], t);
    a[k + 8] = csub(e[k], t);
  }
}

template <int P> __device__ __forceinline__ void dft_p(cf (&a)[P]);
template <> __device__ __forceinline__ void dft_p<4>(cf (&a)[4]) { dft4(a[0], a[1], a[2], a[3], a[0], a[1], a[2], a[3]); }
template <> __device__ __forceinline__ void dft_p<8>(cf (&a)[8]) { dft8(a); }
template <> __device__ __forceinline__ void dft_p<16>(cf (&a)[16]) { dft16(a); }

struct StftTables {
  const cf* winz;       // [H]     (w[2n], w[2n+1]),  H = n_fft / 2
  const cf* tw1;        // [P][64] W_H^(k1*lane),     P = H / 64
  const cf* tw2;        // [8][8]  W64^(j1*m2)
  const cf* twr;        // [H/2+1] W_N^k
  const int* mel_start; // [n_mels] first FFT bin of the band
  const int* mel_cnt;   // [n_mels] number of bins
  const int* mel_ptr;   // [n_mels] offset into mel_w
  const float* mel_w;   // packed non-zero weights
  // balanced mel stage: every band's bin range is cut into chunks of kMelChunk bins (weights zero-padded); chunk i belongs to lane
  // i % 64, pass i / 64
  const int* ch_k0;     // [64 * passes] first FFT bin of the chunk (clamped so that k0 + kMelChunk - 1 stays inside the magnitude row)
  const float* ch_w;    // [64 * passes][kMelChunk] weights, zeros past the band's end and for unused chunks
  const int* band_pbeg; // [n_mels] first chunk of the band
  const int* band_pcnt; // [n_mels] number of chunks of the band
};
constexpr int kMelChunk = 8;
constexpr int kMaxMelPasses = 6;   // <= 384 chunks per frame: 2050 non-zero weights / 8 + one partly filled chunk per band (128 bands)

__device__ __forceinline__ int64_t reflect_index(int64_t s, int64_t L) {
  if (s < 0) s = -s;
  if (s >= L) s = 2 * (L - 1) - s;
  return s;
}

// LDS operations of one wave execute in order, so the phases of a frame (which only exchange data between the lanes of
// ONE wave) need a wave-local fence, not a workgroup barrier: the four waves of a workgroup run their frames
// independently and meet once, before the coalesced tile store.
#ifndef DMEL_STFT_SYNC
#define DMEL_STFT_SYNC 0      // debug: 1 = explicit s_waitcnt lgkmcnt(0) in wave_sync, 2 = workgroup-scope fences
#endif
__device__ __forceinline__ void wave_sync() {
#if DMEL_STFT_SYNC == 2
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#if DMEL_STFT_SYNC == 1
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0), vmcnt / expcnt untouched
#endif
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

constexpr int kMaxMelW = 2304;   // non-zero mel weights (each FFT bin feeds at most two bands: <= 2*1025)

// n_fft = 128 P (P = 4, 8, 16).  The N/2 = 64 P complex points of a frame are n = 64 n1 + n2 (n1 < P, n2 = lane):
// pass 1 is a P-point DFT per lane (+ twiddle W_{N/2}^{k1 n2}); every k1 row then needs a 64-point DFT over n2, done as
// 8 x 8 with two LDS transposes (passes 2 and 3: P*8 independent 8-point DFTs each, i.e. P/8 per lane; for P = 4 half
// the lanes idle in those passes).  Output index k = k1 + P (j1 + 8 j2).
template <int P>
#if DMEL_STFT_WPE
__attribute__((amdgpu_waves_per_eu(DMEL_STFT_WPE, DMEL_STFT_WPE)))
#endif
__global__ __launch_bounds__(kThreads) void stft_logmel_kernel(StftTables tb, const float* __restrict__ audio,
                                                          int64_t row_stride, const int64_t* __restrict__ lengths,
                                                          float* __restrict__ out, float* __restrict__ linear, int64_t L, int64_t T,
                                                          int hop, int pad, int n_mels, int mel_passes) {
  constexpr int H = 64 * P, N = 128 * P, U = (P + 7) / 8, EX = P * 72, NR = P / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) float smem_stft[];
#ifndef DMEL_STFT_LDS_PAD
#define DMEL_STFT_LDS_PAD 0      // debug: bytes of unused LDS in front of and behind the kernel's own
#endif
  float* const smem_base = smem_stft + DMEL_STFT_LDS_PAD / 4;
  cf* buf_all = reinterpret_cast<cf*>(smem_base);                 // [kWaves][EX]   exchange buffer, reused by every pass
  float* mag_all = smem_base + 2 * kWaves * EX;                   // [kWaves][H+8]
  float* chw = mag_all + kWaves * (H + 8);                        // [mel_passes * 64][kMelChunk] chunk weights (LDS sized per plan)
  int* chk0 = reinterpret_cast<int*>(chw + mel_passes * 64 * kMelChunk);   // [mel_passes * 64] first bin of every chunk
  float (*tile)[kFramesPerWG + 1] = reinterpret_cast<float (*)[kFramesPerWG + 1]>(chk0 + mel_passes * 64);   // [n_mels][33]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int b = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * kFramesPerWG;
  const float* x = audio + (int64_t)b * row_stride;

  for (int i = tid; i < mel_passes * 64 * kMelChunk; i += kThreads) chw[i] = tb.ch_w[i];
  for (int i = tid; i < mel_passes * 64; i += kThreads) chk0[i] = tb.ch_k0[i];
  if (tid < kWaves * 7) mag_all[(tid / 7) * (H + 8) + H + 1 + tid % 7] = 0.f;   // pad bins: read (times a zero weight) by the last chunks

  // per-lane constants kept in registers across the wave's frames (window and pass-1 twiddles only while they fit)
  constexpr bool kRegTables = P <= DMEL_STFT_REGTAB_MAXP;
  cf wz[kRegTables ? P : 1], w1[kRegTables ? P : 1], w2[8], wr[NR];
  if constexpr (kRegTables) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      wz[j] = tb.winz[64 * j + lane];
      w1[j] = tb.tw1[64 * j + lane];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) w2[j] = tb.tw2[8 * j + (lane & 7)];
#pragma unroll
  for (int j = 0; j < NR; ++j) wr[j] = tb.twr[min(lane + 64 * j, H / 2)];
  // mel stage, balanced: this lane's (up to three) chunks of <= chunk_len bins, and the (up to two) bands whose chunk sums it adds up.
  // One lane per BAND walked up to ~80 bins on the widest bands while 63 lanes idled (the wave pays for its slowest lane): 90
  // dependent iterations per frame; chunked, the wave does 3 x chunk_len + (chunks of the widest band) ~ 35.
  int bbeg[2], bcnt[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int m = lane + 64 * q;
    const bool has = m < n_mels;
    bbeg[q] = has ? tb.band_pbeg[m] : 0;
    bcnt[q] = has ? tb.band_pcnt[m] : 0;
  }
  const int k1l = lane >> 3, l7 = lane & 7;
  cf* A = buf_all + wave * EX;
  float* mg = mag_all + wave * (H + 8);
  __syncthreads();   // mel tables visible

  for (int fi = 0; fi < kFramesPerWave; ++fi) {
    const int f = wave * kFramesPerWave + fi;
    const int64_t t = t0 + f;
    if (t >= T) break;                 // wave-uniform: the rest of this wave's frames are past the clip
    cf r[P];
    // ---- load + reflect pad + window; lane = n2, reg j = n1, point n = 64 j + lane
    const int fstart = (int)t * hop - pad;                       // host checks L < 2^30: 32-bit sample offsets
    const bool interior = (fstart >= 0) && (fstart + N <= (int)L);
    const float* xf = x + fstart + 2 * lane;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const cf w = kRegTables ? wz[kRegTables ? j : 0] : tb.winz[64 * j + lane];
      if (interior) {
        r[j] = {xf[128 * j] * w.x, xf[128 * j + 1] * w.y};
      } else {
        const int64_t s0 = reflect_index(fstart + 2 * lane + 128 * j, L), s1 = reflect_index(fstart + 2 * lane + 128 * j + 1, L);
        r[j] = {x[s0] * w.x, x[s1] * w.y};
      }
    }
    // ---- pass 1: DFT over n1, twiddle W_H^(k1 n2)
    dft_p<P>(r);
#pragma unroll
    for (int j = 1; j < P; ++j) r[j] = cmul(r[j], kRegTables ? w1[kRegTables ? j : 0] : tb.tw1[64 * j + lane]);
#pragma unroll
    for (int j = 0; j < P; ++j) A[j * 72 + lane] = r[j];
    wave_sync();
    // ---- pass 2: lane = (k1, m2), k1 = lane/8 + 8u; reg m1 <- A[k1][8 m1 + m2]; DFT over m1; twiddle W64^(j1 m2)
    cf rr[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k1 = min(k1l + 8 * u, P - 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) rr[u][j] = A[k1 * 72 + 8 * j + l7];
    }
    wave_sync();   // all reads done before the buffer is overwritten
#pragma unroll
    for (int u = 0; u < U; ++u) {
      dft8(rr[u]);
#pragma unroll
      for (int j = 1; j < 8; ++j) rr[u][j] = cmul(rr[u][j], w2[j]);
      // transpose inside each 8-lane group: writer (k1, m2) reg j1 -> reader (k1, j1) reg m2
      if (P >= 8 || k1l + 8 * u < P) {
#pragma unroll
        for (int j = 0; j < 8; ++j) A[(k1l + 8 * u) * 72 + j + 9 * l7] = rr[u][j];
      }
    }
    wave_sync();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k1 = min(k1l + 8 * u, P - 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) rr[u][j] = A[k1 * 72 + l7 + 9 * j];
    }
    wave_sync();
    // ---- pass 3: DFT over m2 -> Z[k1 + P (j1 + 8 j2)]; store in natural order, one pad slot per 8
#pragma unroll
    for (int u = 0; u < U; ++u) {
      dft8(rr[u]);
      if (P >= 8 || k1l + 8 * u < P) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = (k1l + 8 * u) + P * (l7 + 8 * j);
          A[k + (k >> 3)] = rr[u][j];
        }
      }
    }
    wave_sync();
    // ---- real-FFT recombination + magnitude; lane handles k = lane + 64 j and its mirror H - k
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int k = lane + 64 * j;
      if (j < NR - 1 || lane == 0) {
        const int km = (H - k) & (H - 1);
        cf zk = A[k + (k >> 3)], zm = A[km + (km >> 3)];
        cf e = {0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};
        cf o = {0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x)};
        cf wo = cmul(wr[j], o);
        cf xk = cadd(e, wo), xm = csub(e, wo);
        // v_sqrt_f32 (1 ulp; the argument is >= 1e-9, never denormal) instead of the 12-instruction IEEE sequence
        mg[k] = __builtin_amdgcn_sqrtf(xk.x * xk.x + xk.y * xk.y + 1e-9f);
        mg[H - k] = __builtin_amdgcn_sqrtf(xm.x * xm.x + xm.y * xm.y + 1e-9f);
      }
    }
    wave_sync();
    // ---- optional linear-magnitude output, frame-major (B, T, H + 1): one coalesced row per frame (multi-resolution STFT loss)
    if (linear) {
      float* lo = linear + ((int64_t)b * T + t) * (H + 1);
      for (int k = lane; k <= H; k += 64) lo[k] = mg[k];
    }
    // ---- mel: chunk partial sums (ascending-bin fma chains, weights from LDS) into the free exchange buffer, then per band the
    //      ascending sum of its chunks
    if (out) {
      float* part = reinterpret_cast<float*>(A);
      for (int q = 0; q < mel_passes; ++q) {                // wave-uniform trip count; the eight products are straight-line code
        const int c = lane + 64 * q;
        const float* w = chw + c * kMelChunk;
        const float* g = mg + chk0[c];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < kMelChunk; ++i) acc = fmaf(w[i], g[i], acc);
        part[c] = acc;
      }
      wave_sync();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (lane + 64 * q < n_mels) {
          const float* pp = part + bbeg[q];
          float acc = 0.f;
          for (int i = 0; i < bcnt[q]; ++i) acc += pp[i];
          tile[lane + 64 * q][f] = logf(fmaxf(acc, 1e-5f));
        }
      }
    }
    wave_sync();   // mg and A are rewritten by the next frame
  }
  __syncthreads();
  if (!out) return;
  const int64_t n_valid = lengths ? lengths[b] / hop : T;
  float* o = out + (int64_t)b * n_mels * T;
  for (int idx = tid; idx < n_mels * kFramesPerWG; idx += kThreads) {
    const int m = idx / kFramesPerWG, f = idx % kFramesPerWG;
    const int64_t t = t0 + f;
    if (t < T) o[(int64_t)m * T + t] = (t < n_valid) ? tile[m][f] : 0.f;
  }
}

// the output tile is sized by the plan's band count (80 bands at n_fft = 1024: 48.6 KB, three workgroups per CU instead of two)
template <int P> static size_t stft_lds_bytes(int mel_passes = kMaxMelPasses, int n_mels = kMaxMels) {
  return (size_t)(2 * kWaves * P * 72 + kWaves * (64 * P + 8) + mel_passes * 64 * (kMelChunk + 1) + n_mels * (kFramesPerWG + 1)) *
             sizeof(float) + 2 * DMEL_STFT_LDS_PAD;
}

}  // namespace dmel

using namespace dmel;

struct dmel_stft_plan {
  int sample_rate, n_fft, win_length, hop, n_mels, pad;
  double f_min, f_max;
  std::vector<float> basis;  // dense (n_mels, 513), host copy
  DevBuf winz, tw1, tw2, twr, mel_start, mel_cnt, mel_ptr, mel_w, ch_k0, ch_w, band_pbeg, band_pcnt;
  int n_melw = 0, mel_passes = 1;
};

namespace {

// Slaney mel scale (librosa filters.mel, htk=False): linear below 1 kHz, log above.
const double kFsp = 200.0 / 3.0, kMinLogHz = 1000.0, kMinLogMel = kMinLogHz / kFsp;
double hz_to_mel(double f) {
  static const double logstep = std::log(6.4) / 27.0;
  return f >= kMinLogHz ? kMinLogMel + std::log(f / kMinLogHz) / logstep : f / kFsp;
}
double mel_to_hz(double m) {
  static const double logstep = std::log(6.4) / 27.0;
  return m >= kMinLogMel ? kMinLogHz * std::exp(logstep * (m - kMinLogMel)) : kFsp * m;
}

void build_mel_basis(int sr, int n_fft, int n_mels, double fmin, double fmax, std::vector<float>& w) {
  const int nb = n_fft / 2 + 1;
  w.assign((size_t)n_mels * nb, 0.f);
  std::vector<double> mel_f(n_mels + 2);
  const double m_lo = hz_to_mel(fmin), m_hi = hz_to_mel(fmax);
  for (int i = 0; i < n_mels + 2; ++i) {
    // numpy.linspace: start + i*step, last point exact
    double step = (m_hi - m_lo) / (n_mels + 1);
    double m = (i == n_mels + 1) ? m_hi : m_lo + i * step;
    mel_f[i] = mel_to_hz(m);
  }
  for (int i = 0; i < n_mels; ++i) {
    const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
    const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
    for (int k = 0; k < nb; ++k) {
      const double fk = (double)k * sr / n_fft;
      const double lower = -(mel_f[i] - fk) / fd0, upper = (mel_f[i + 2] - fk) / fd1;
      const double v = std::fmax(0.0, std::fmin(lower, upper));
      const float v32 = (float)v;                       // librosa stores float32, then scales
      w[(size_t)i * nb + k] = (float)((double)v32 * enorm);
    }
  }
}

}  // namespace

extern "C" int dmel_stft_plan_create(dmel_stft_plan** out, int sample_rate, int n_fft, int win_length, int hop_length,
                                     int n_mels, double f_min, double f_max, const float* window_host) {
  DMEL_CHECK_ARG(out != nullptr, "plan out pointer is NULL");
  *out = nullptr;
  if (n_fft != 512 && n_fft != 1024 && n_fft != 2048) {
    set_error("stft_logmel: n_fft=%d unsupported (the gfx950 kernel is built for n_fft in {512, 1024, 2048})", n_fft);
    return DMEL_EUNSUPPORTED;
  }
  const int kHalf = n_fft / 2, P = kHalf / 64;
  DMEL_CHECK_ARG(win_length > 0 && win_length <= n_fft, "win_length %d out of range", win_length);
  DMEL_CHECK_ARG(hop_length > 0 && hop_length <= n_fft && (n_fft - hop_length) % 2 == 0,
                 "hop_length %d unsupported (need 0 < hop <= n_fft, n_fft-hop even)", hop_length);
  DMEL_CHECK_ARG(n_mels > 0 && n_mels <= kMaxMels, "n_mels %d out of range (1..128)", n_mels);
  DMEL_CHECK_ARG(sample_rate > 0, "bad sample_rate");
  if (f_max <= 0) f_max = sample_rate / 2.0;
  auto* p = new dmel_stft_plan();
  p->sample_rate = sample_rate; p->n_fft = n_fft; p->win_length = win_length; p->hop = hop_length;
  p->n_mels = n_mels; p->pad = (n_fft - hop_length) / 2; p->f_min = f_min; p->f_max = f_max;

  // window (zero-padded to n_fft, centred, as torch.stft does for win_length < n_fft)
  std::vector<float> win(n_fft, 0.f);
  const int woff = (n_fft - win_length) / 2;
  for (int n = 0; n < win_length; ++n) {
    win[woff + n] = window_host ? window_host[n]
                                : (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * (double)n / (double)win_length));
  }
  std::vector<cf> winz(kHalf), tw1((size_t)P * 64), tw2(64), twr(kHalf / 2 + 1);
  for (int n = 0; n < kHalf; ++n) winz[n] = {win[2 * n], win[2 * n + 1]};
  for (int k = 0; k < P; ++k)
    for (int l = 0; l < 64; ++l) {
      double a = -2.0 * M_PI * (double)(k * l) / (double)kHalf;
      tw1[k * 64 + l] = {(float)std::cos(a), (float)std::sin(a)};
    }
  for (int j = 0; j < 8; ++j)
    for (int m = 0; m < 8; ++m) {
      double a = -2.0 * M_PI * (double)(j * m) / 64.0;
      tw2[j * 8 + m] = {(float)std::cos(a), (float)std::sin(a)};
    }
  for (int k = 0; k <= kHalf / 2; ++k) {
    double a = -2.0 * M_PI * (double)k / (double)n_fft;
    twr[k] = {(float)std::cos(a), (float)std::sin(a)};
  }
  build_mel_basis(sample_rate, n_fft, n_mels, f_min, f_max, p->basis);
  const int nb = n_fft / 2 + 1;
  std::vector<int> st(n_mels), cnt(n_mels), ptr(n_mels);
  std::vector<float> packed;
  for (int m = 0; m < n_mels; ++m) {
    int lo = nb, hi = -1;
    for (int k = 0; k < nb; ++k)
      if (p->basis[(size_t)m * nb + k] != 0.f) { lo = std::min(lo, k); hi = std::max(hi, k); }
    st[m] = hi < 0 ? 0 : lo;
    cnt[m] = hi < 0 ? 0 : hi - lo + 1;
    ptr[m] = (int)packed.size();
    for (int k = 0; k < cnt[m]; ++k) packed.push_back(p->basis[(size_t)m * nb + st[m] + k]);
  }
  {  // the n_fft = 2048 instantiation needs 78 KB of dynamic LDS: raise the per-kernel limit (default 64 KB)
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_logmel_kernel<16>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)stft_lds_bytes<16>());
    if (e1 != hipSuccess && P == 16) {
      set_error("stft_logmel: cannot reserve %zu bytes of LDS: %s", stft_lds_bytes<16>(), hipGetErrorString(e1));
      delete p;
      return (int)e1;
    }
    (void)hipGetLastError();
  }
  if (packed.empty()) packed.push_back(0.f);
  if ((int)packed.size() > kMaxMelW) {
    set_error("stft_logmel: %zu non-zero mel weights exceed the LDS table (%d)", packed.size(), kMaxMelW);
    delete p;
    return DMEL_EUNSUPPORTED;
  }
  p->n_melw = (int)packed.size();
  // chunk table of the balanced mel stage: every band cut into kMelChunk-bin chunks, weights zero-padded
  std::vector<int> ck0, bpb(n_mels), bpc(n_mels);
  std::vector<float> cw;
  for (int m = 0; m < n_mels; ++m) {
    bpb[m] = (int)ck0.size();
    for (int o = 0; o < cnt[m]; o += kMelChunk) {
      ck0.push_back(st[m] + o);
      for (int i = 0; i < kMelChunk; ++i) cw.push_back(o + i < cnt[m] ? packed[(size_t)ptr[m] + o + i] : 0.f);
    }
    bpc[m] = (int)ck0.size() - bpb[m];
  }
  p->mel_passes = std::max(1, (int)((ck0.size() + 63) / 64));
  if (p->mel_passes > kMaxMelPasses) {
    set_error("stft_logmel: %zu mel chunks exceed the %d the kernel holds per frame", ck0.size(), 64 * kMaxMelPasses);
    delete p;
    return DMEL_EUNSUPPORTED;
  }
  ck0.resize((size_t)64 * p->mel_passes, 0);
  cw.resize((size_t)64 * p->mel_passes * kMelChunk, 0.f);
  int rc = DMEL_OK;
  if ((rc = p->ch_k0.upload(ck0.data(), ck0.size() * sizeof(int))) || (rc = p->ch_w.upload(cw.data(), cw.size() * sizeof(float))) ||
      (rc = p->band_pbeg.upload(bpb.data(), bpb.size() * sizeof(int))) || (rc = p->band_pcnt.upload(bpc.data(), bpc.size() * sizeof(int)))) {
    delete p;
    return rc;
  }
  if ((rc = p->winz.upload(winz.data(), winz.size() * sizeof(cf))) || (rc = p->tw1.upload(tw1.data(), tw1.size() * sizeof(cf))) ||
      (rc = p->tw2.upload(tw2.data(), tw2.size() * sizeof(cf))) || (rc = p->twr.upload(twr.data(), twr.size() * sizeof(cf))) ||
      (rc = p->mel_start.upload(st.data(), st.size() * sizeof(int))) || (rc = p->mel_cnt.upload(cnt.data(), cnt.size() * sizeof(int))) ||
      (rc = p->mel_ptr.upload(ptr.data(), ptr.size() * sizeof(int))) || (rc = p->mel_w.upload(packed.data(), packed.size() * sizeof(float)))) {
    delete p;
    return rc;
  }
  *out = p;
  return DMEL_OK;
}

extern "C" int dmel_mel_basis_host(int sample_rate, int n_fft, int n_mels, double f_min, double f_max, float* basis_host) {
  DMEL_CHECK_ARG(basis_host && sample_rate > 0 && n_fft > 0 && n_mels > 0, "mel_basis_host: bad argument");
  if (f_max <= 0) f_max = sample_rate / 2.0;
  std::vector<float> w;
  build_mel_basis(sample_rate, n_fft, n_mels, f_min, f_max, w);
  std::memcpy(basis_host, w.data(), w.size() * sizeof(float));
  return DMEL_OK;
}

extern "C" void dmel_stft_plan_destroy(dmel_stft_plan* p) { delete p; }

extern "C" int dmel_stft_plan_mel_basis(const dmel_stft_plan* p, float* basis_host) {
  DMEL_CHECK_ARG(p && basis_host, "NULL argument");
  std::memcpy(basis_host, p->basis.data(), p->basis.size() * sizeof(float));
  return DMEL_OK;
}

extern "C" int64_t dmel_stft_num_frames(const dmel_stft_plan* p, int64_t L) {
  if (!p) return -1;
  int64_t padded = L + 2 * p->pad;
  if (padded < p->n_fft) return 0;
  return 1 + (padded - p->n_fft) / p->hop;
}

// ---- one workgroup per CU, and no other LDS user beside it (round 3) -------------------------------------------------------------
// Measured (tools/dbg_lanes5.py / dbg_lanes7.py, profiles/r03_stft_concurrency.txt): when workgroups of this kernel share a CU with
// workgroups of the convolution kernels -- which only happens when the two are launched from different streams -- 3-11 % of the launches
// come back with ONE frame wrong (a subset of its bands, or all of them; values near the right ones).  What was ruled out: LDS overflow on
// either side (8 KB guard regions; the hardware isolates LDS between workgroups: tools/probe/ldsoob.hip), residue in registers or LDS
// (tools/probe/poison.hip), LDS ordering inside the wave (explicit lgkmcnt(0), workgroup-scope fences), stale cache lines (the wrong
// values are not what the address held before), the LDS size of this round.  rocBLAS GEMMs, the activation kernel, torch element-wise
// kernels and this kernel itself as the neighbour do not trigger it.  The mechanism is not understood; the cure is to keep the two apart:
// with dmel_stft_set_exclusive_cu(1) a launch asks for kExclusiveLds bytes of LDS, so that a CU holding one of its workgroups has no room
// for a workgroup of any kernel that uses more than 8 KB of LDS (every convolution kernel does), and the other way round.  One workgroup
// per CU instead of three: nothing for the 96-workgroup launch of the codec, 2.5x slower at 256 x 60 s -- hence a switch, set by
// dmel_codec_amd.pipeline.CodecLanes (several batches in flight on their own streams) and off otherwise.
static std::atomic<int> g_exclusive_cu{0};
constexpr size_t kExclusiveLds = 152 * 1024;
extern "C" int dmel_stft_set_exclusive_cu(int on) {
  g_exclusive_cu.store(on ? 1 : 0);
  return DMEL_OK;
}
template <int P> static size_t stft_launch_lds(const dmel_stft_plan* p) {
  const size_t need = stft_lds_bytes<P>(p->mel_passes, DMEL_STFT_FULL_TILE ? kMaxMels : p->n_mels);
  if (!g_exclusive_cu.load()) return need;
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_logmel_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)kExclusiveLds) == hipSuccess;
  return ok ? std::max(need, kExclusiveLds) : need;
}

extern "C" int dmel_stft_logmel_f32(const dmel_stft_plan* p, const float* audio, int64_t row_stride, const int64_t* lengths,
                                    float* out, int B, int64_t L, void* stream) {
  DMEL_CHECK_ARG(out, "NULL argument");
  return dmel_stft_f32(p, audio, row_stride, lengths, out, nullptr, B, L, stream);
}

extern "C" int dmel_stft_f32(const dmel_stft_plan* p, const float* audio, int64_t row_stride, const int64_t* lengths, float* out,
                             float* linear, int B, int64_t L, void* stream) {
  DMEL_CHECK_ARG(p && audio && (out || linear), "NULL argument");
  DMEL_CHECK_ARG(B > 0 && B <= 65535, "batch %d out of range", B);
  DMEL_CHECK_ARG(L > p->pad, "clip length %lld must exceed the reflect pad %d", (long long)L, p->pad);
  DMEL_CHECK_ARG(row_stride >= L, "row stride smaller than L");
  DMEL_CHECK_ARG(L < ((int64_t)1 << 30), "clip longer than 2^30 samples");
  const int64_t T = dmel_stft_num_frames(p, L);
  DMEL_CHECK_ARG(T > 0, "clip too short for one frame");
  StftTables tb{p->winz.as<cf>(), p->tw1.as<cf>(), p->tw2.as<cf>(), p->twr.as<cf>(), p->mel_start.as<int>(),
                p->mel_cnt.as<int>(), p->mel_ptr.as<int>(), p->mel_w.as<float>(), p->ch_k0.as<int>(), p->ch_w.as<float>(),
                p->band_pbeg.as<int>(), p->band_pcnt.as<int>()};
  dim3 grid((unsigned)((T + kFramesPerWG - 1) / kFramesPerWG), (unsigned)B);
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope ps("stft_logmel", s, 0.0, (double)B * (4.0 * (double)L + 4.0 * p->n_mels * (double)T));
    switch (p->n_fft) {
      case 512:
        hipLaunchKernelGGL(stft_logmel_kernel<4>, grid, dim3(kThreads), stft_launch_lds<4>(p), s, tb, audio, row_stride, lengths, out,
                           linear, L, T, p->hop, p->pad, p->n_mels, p->mel_passes);
        break;
      case 1024:
        hipLaunchKernelGGL(stft_logmel_kernel<8>, grid, dim3(kThreads), stft_launch_lds<8>(p), s, tb, audio, row_stride, lengths, out,
                           linear, L, T, p->hop, p->pad, p->n_mels, p->mel_passes);
        break;
      default:
        hipLaunchKernelGGL(stft_logmel_kernel<16>, grid, dim3(kThreads), stft_launch_lds<16>(p), s, tb, audio, row_stride, lengths,
                           out, linear, L, T, p->hop, p->pad, p->n_mels, p->mel_passes);
    }
  }
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}
