// Drives the host side of every handle of the C ABI (include/dmel_hip.h) under AddressSanitizer on the fake HIP runtime: create ->
// set_tensor for every state-dict key -> finalize (weight-norm folding, re-tiling into MFMA fragment order, three weight images,
// transposed / training images, mel and chunk tables) -> workspace planning -> forward / training entry points (argument checks, tile
// selection, launch assembly; the kernels themselves do not run).  Exits non-zero on any error return; ASan aborts on any bad access.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/dmel_hip.h"

static std::mt19937 rng(7);
static int failures = 0;
#define CK(call)                                                                                   \
  do {                                                                                             \
    int rc__ = (call);                                                                             \
    if (rc__ != 0) { std::printf("FAIL %s -> %d: %s\n", #call, rc__, dmel_last_error()); ++failures; } \
  } while (0)

template <class H, class F> static void set(H* h, F fn, const std::string& key, std::vector<int64_t> shape, float scale = 0.05f) {
  size_t n = 1;
  for (auto d : shape) n *= (size_t)d;
  std::vector<float> v(n);
  std::normal_distribution<float> nd(0.f, scale);
  for (auto& x : v) x = nd(rng);
  CK(fn(h, key.c_str(), v.data(), shape.data(), (int)shape.size()));
}
static std::vector<float> buf(size_t n) { return std::vector<float>(n + 64, 0.25f); }

static void wavenet(int Cin, int Cout, int C, int L, int cycle, int Cc, int N, int T, bool train) {
  dmel_wavenet* m = nullptr;
  CK(dmel_wavenet_create(&m, Cin, Cout, C, L, cycle, Cc));
  if (train) CK(dmel_wavenet_enable_training(m, 1));
  auto S = [&](const std::string& k, std::vector<int64_t> s) { set(m, dmel_wavenet_set_tensor, k, s); };
  if (Cin != C) { S("input_projection.conv.weight", {C, Cin, 1}); S("input_projection.conv.bias", {C}); }
  for (int i = 0; i < L; ++i) {
    const std::string p = "residual_layers." + std::to_string(i) + ".";
    S(p + "conv_layer.conv.weight", {2 * C, C, 3}); S(p + "conv_layer.conv.bias", {2 * C});
    S(p + "output_projection.conv.weight", {2 * C, C, 1}); S(p + "output_projection.conv.bias", {2 * C});
    if (Cc) { S(p + "condition_projection.conv.weight", {2 * C, Cc, 1}); S(p + "condition_projection.conv.bias", {2 * C}); }
  }
  S("skip_projection.conv.weight", {C, C, 1}); S("skip_projection.conv.bias", {C});
  if (Cout != C) { S("output_projection.conv.weight", {Cout, C, 1}); S("output_projection.conv.bias", {Cout}); }
  CK(dmel_wavenet_finalize(m));
  auto x = buf((size_t)N * Cin * T), c = buf((size_t)N * (Cc ? Cc : 1) * T), y = buf((size_t)N * Cout * T);
  std::vector<int64_t> lens(N, T - 1);
  std::vector<char> ws(dmel_wavenet_workspace_bytes(m, N, T) + 256);
  CK(dmel_wavenet_forward(m, x.data(), Cc ? c.data() : nullptr, y.data(), N, T, lens.data(), lens.data(), 1, ws.data(), ws.size(), nullptr));
  for (int prec : {DMEL_PRECISION_BF16, DMEL_PRECISION_FP32_MFMA, DMEL_PRECISION_FP32}) {
    CK(dmel_wavenet_set_precision(m, prec));
    CK(dmel_wavenet_forward(m, x.data(), Cc ? c.data() : nullptr, y.data(), N, T, nullptr, nullptr, 1, ws.data(), ws.size(), nullptr));
  }
  if (train) {
    std::vector<char> tw(dmel_wavenet_train_workspace_bytes(m, N, T) + 256);
    auto dy = buf((size_t)N * Cout * T), dx = buf((size_t)N * Cin * T), dc = buf((size_t)N * (Cc ? Cc : 1) * T);
    auto g = buf((size_t)dmel_wavenet_grad_floats(m));
    CK(dmel_wavenet_forward_train(m, x.data(), Cc ? c.data() : nullptr, y.data(), N, T, tw.data(), tw.size(), nullptr));
    static int64_t covered;
    covered = 0;
    CK(dmel_wavenet_backward_hooked(m, x.data(), Cc ? c.data() : nullptr, dy.data(), dx.data(), Cc ? dc.data() : nullptr, g.data(), N, T,
                                    tw.data(), tw.size(), nullptr, [](void*, int64_t, int64_t n) { covered += n; }, nullptr));
    if (covered != dmel_wavenet_grad_floats(m)) { std::printf("FAIL hook regions cover %lld of %lld\n", (long long)covered, (long long)dmel_wavenet_grad_floats(m)); ++failures; }
    CK(dmel_wavenet_set_train_precision(m, DMEL_PRECISION_BF16));
    CK(dmel_wavenet_forward_train(m, x.data(), Cc ? c.data() : nullptr, y.data(), N, T, tw.data(), tw.size(), nullptr));
  }
  if (Cin == C) {   // incremental forward
    const int64_t cap = 256;
    auto hist = buf((size_t)(L + 1) * N * C * cap), skip = buf((size_t)N * C * cap), cond = buf((size_t)N * (Cc ? Cc : 1) * cap),
         yy = buf((size_t)N * Cout * cap), sc = buf((size_t)2 * N * C * cap);
    std::vector<int64_t> prev(L + 1, 0), next(L + 1);
    next[0] = 200;
    for (int l = 1; l <= L; ++l) next[l] = next[l - 1] - (cycle ? 1 << ((l - 1) % cycle) : 1);
    CK(dmel_wavenet_stream_step(m, hist.data(), skip.data(), Cc ? cond.data() : nullptr, yy.data(), sc.data(), N, cap, prev.data(), next.data(), nullptr));
    prev = next;
    for (int l = 0; l <= L; ++l) next[l] = 230;
    CK(dmel_wavenet_stream_step(m, hist.data(), skip.data(), Cc ? cond.data() : nullptr, yy.data(), sc.data(), N, cap, prev.data(), next.data(), nullptr));
  }
  dmel_wavenet_destroy(m);
}

static void convnext_keys(const std::string& p, int C, const std::function<void(const std::string&, std::vector<int64_t>)>& S) {
  S(p + "dwconv.weight", {C, 1, 7}); S(p + "dwconv.bias", {C}); S(p + "norm.weight", {C}); S(p + "norm.bias", {C});
  S(p + "pwconv1.weight", {4 * C, C}); S(p + "pwconv1.bias", {4 * C}); S(p + "pwconv2.weight", {C, 4 * C}); S(p + "pwconv2.bias", {C});
  S(p + "gamma", {C});
}

static void quantizer(int G, int C, std::vector<int> levels, int B, int T) {
  dmel_quantizer* q = nullptr;
  int f[2] = {2, 2};
  CK(dmel_quantizer_create(&q, G * C, G, levels.data(), (int)levels.size(), f, 2, 1));
  CK(dmel_quantizer_enable_training(q, 1));
  std::function<void(const std::string&, std::vector<int64_t>)> S = [&](const std::string& k, std::vector<int64_t> s) { set(q, dmel_quantizer_set_tensor, k, s); };
  const int D = (int)levels.size();
  for (int i = 0; i < 2; ++i) {
    S("downsample." + std::to_string(i) + ".0.weight", {C, C, 2}); S("downsample." + std::to_string(i) + ".0.bias", {C});
    convnext_keys("downsample." + std::to_string(i) + ".1.", C, S);
    S("upsample." + std::to_string(i) + ".0.weight", {C, C, 2}); S("upsample." + std::to_string(i) + ".0.bias", {C});
    convnext_keys("upsample." + std::to_string(i) + ".1.", C, S);
  }
  for (int g = 0; g < G; ++g) {
    const std::string p = "residual_fsq.rvqs." + std::to_string(g) + ".";
    S(p + "project_in.weight", {D, C}); S(p + "project_in.bias", {D}); S(p + "project_out.weight", {C, D}); S(p + "project_out.bias", {C});
  }
  CK(dmel_quantizer_finalize(q));
  const int T4 = T / 4;
  auto z = buf((size_t)B * G * C * T), zq = buf((size_t)B * G * C * T), pre = buf((size_t)G * B * T4 * D), lat = buf((size_t)B * G * C * T4);
  std::vector<int32_t> ids((size_t)B * G * T4 + 16, 3);
  std::vector<char> ws(dmel_quantizer_workspace_bytes(q, B, T) + 256);
  CK(dmel_quantizer_set_strict(q, 1));
  CK(dmel_quantizer_encode_ex(q, z.data(), ids.data(), pre.data(), lat.data(), B, T, ws.data(), ws.size(), nullptr));
  CK(dmel_quantizer_decode(q, ids.data(), zq.data(), B, T4, ws.data(), ws.size(), nullptr));
  std::vector<char> tw(dmel_quantizer_train_workspace_bytes(q, B, T) + 256);
  auto g = buf((size_t)dmel_quantizer_grad_floats(q));
  CK(dmel_quantizer_forward_train(q, z.data(), zq.data(), ids.data(), lat.data(), B, T, tw.data(), tw.size(), nullptr));
  CK(dmel_quantizer_backward(q, z.data(), zq.data(), z.data(), g.data(), B, T, tw.data(), tw.size(), nullptr));
  dmel_quantizer_destroy(q);
}

static void discriminator(int B, int H, int W) {
  dmel_discriminator* d = nullptr;
  CK(dmel_discriminator_create(&d));
  CK(dmel_discriminator_enable_training(d, 1));
  const int cin[6] = {1, 64, 128, 256, 512, 1024}, cout[6] = {64, 128, 256, 512, 1024, 1}, kw[6] = {9, 9, 9, 9, 3, 3};
  for (int i = 0; i < 6; ++i) {
    const std::string p = "blocks." + std::to_string(2 * i) + ".";
    set(d, dmel_discriminator_set_tensor, p + "bias", {cout[i]});
    set(d, dmel_discriminator_set_tensor, p + "parametrizations.weight.original0", {cout[i], 1, 1, 1}, 1.f);
    set(d, dmel_discriminator_set_tensor, p + "parametrizations.weight.original1", {cout[i], cin[i], 3, kw[i]});
  }
  CK(dmel_discriminator_finalize(d));
  const int64_t Wo = dmel_discriminator_out_frames(d, W);
  auto x = buf((size_t)B * H * W), y = buf((size_t)B * H * Wo), dx = buf((size_t)B * H * W);
  std::vector<char> ws(dmel_discriminator_workspace_bytes(d, B, H, W) + 256), tw(dmel_discriminator_train_workspace_bytes(d, B, H, W) + 256);
  CK(dmel_discriminator_forward(d, x.data(), y.data(), B, H, W, ws.data(), ws.size(), nullptr));
  auto g = buf((size_t)dmel_discriminator_grad_floats(d));
  CK(dmel_discriminator_forward_train(d, x.data(), y.data(), B, H, W, tw.data(), tw.size(), nullptr));
  CK(dmel_discriminator_backward(d, y.data(), dx.data(), g.data(), B, H, W, tw.data(), tw.size(), nullptr));
  dmel_discriminator_destroy(d);
}

static void bigvgan(int n_up, const int* rates, const int* ks, int C0, int mels, int resblock, bool weight_norm, int B, int T) {
  dmel_bigvgan_config c;
  std::memset(&c, 0, sizeof(c));
  c.num_mels = mels; c.upsample_initial_channel = C0; c.num_upsamples = n_up; c.num_kernels = 3;
  const int rk[3] = {3, 7, 11}, rd[3] = {1, 3, 5};
  for (int i = 0; i < n_up; ++i) { c.upsample_rates[i] = rates[i]; c.upsample_kernel_sizes[i] = ks[i]; }
  for (int j = 0; j < 3; ++j) { c.resblock_kernel_sizes[j] = rk[j]; for (int l = 0; l < 3; ++l) c.resblock_dilations[j][l] = rd[l]; }
  c.snake_logscale = 1; c.use_tanh_at_final = 1; c.use_bias_at_final = 1; c.resblock_type = resblock;
  dmel_bigvgan* m = nullptr;
  CK(dmel_bigvgan_create(&m, &c));
  auto W = [&](const std::string& p, std::vector<int64_t> shape, bool bias) {
    if (weight_norm) {
      set(m, dmel_bigvgan_set_tensor, p + "weight_g", {shape[0], 1, 1}, 1.f);
      set(m, dmel_bigvgan_set_tensor, p + "weight_v", shape);
    } else {
      set(m, dmel_bigvgan_set_tensor, p + "weight", shape);
    }
    if (bias) set(m, dmel_bigvgan_set_tensor, p + "bias", {p.find("ups.") == 0 ? shape[1] : shape[0]});
  };
  auto A = [&](const std::string& p, int ch) {
    set(m, dmel_bigvgan_set_tensor, p + "act.alpha", {ch}); set(m, dmel_bigvgan_set_tensor, p + "act.beta", {ch});
    set(m, dmel_bigvgan_set_tensor, p + "upsample.filter", {1, 1, 12}); set(m, dmel_bigvgan_set_tensor, p + "downsample.lowpass.filter", {1, 1, 12});
  };
  // all up filters / all down filters must agree: overwrite them with constants afterwards
  W("conv_pre.", {C0, mels, 7}, true);
  int ch = C0;
  for (int i = 0; i < n_up; ++i) {
    W("ups." + std::to_string(i) + ".0.", {C0 >> i, C0 >> (i + 1), ks[i]}, true);
    ch = C0 >> (i + 1);
    for (int j = 0; j < 3; ++j) {
      const std::string bp = "resblocks." + std::to_string(i * 3 + j) + ".";
      for (int l = 0; l < 3; ++l) {
        if (resblock == 2) W(bp + "convs." + std::to_string(l) + ".", {ch, ch, rk[j]}, true);
        else { W(bp + "convs1." + std::to_string(l) + ".", {ch, ch, rk[j]}, true); W(bp + "convs2." + std::to_string(l) + ".", {ch, ch, rk[j]}, true); }
      }
      for (int a = 0; a < (resblock == 2 ? 3 : 6); ++a) {
        set(m, dmel_bigvgan_set_tensor, bp + "activations." + std::to_string(a) + ".act.alpha", {ch});
        set(m, dmel_bigvgan_set_tensor, bp + "activations." + std::to_string(a) + ".act.beta", {ch});
      }
    }
  }
  set(m, dmel_bigvgan_set_tensor, "activation_post.act.alpha", {ch}); set(m, dmel_bigvgan_set_tensor, "activation_post.act.beta", {ch});
  W("conv_post.", {1, ch, 7}, true);
  (void)A;
  CK(dmel_bigvgan_finalize(m));
  int up = 1;
  for (int i = 0; i < n_up; ++i) up *= rates[i];
  auto mel = buf((size_t)B * mels * T), y = buf((size_t)B * T * up);
  std::vector<char> ws(dmel_bigvgan_workspace_bytes(m, B, T) + 256);
  for (int streams : {1, 3}) {
    CK(dmel_bigvgan_set_streams(m, streams));
    CK(dmel_bigvgan_forward(m, mel.data(), y.data(), B, T, ws.data(), ws.size(), nullptr));
  }
  CK(dmel_bigvgan_set_precision(m, DMEL_PRECISION_BF16));
  CK(dmel_bigvgan_forward(m, mel.data(), y.data(), B, T, ws.data(), ws.size(), nullptr));
  dmel_bigvgan_destroy(m);
}

int main() {
  // single convolutions: ragged channel counts, every tap count / dilation of the vocoder, forward + both backward entry points
  const int shapes[][6] = {{256, 256, 7, 3, 736, 2}, {128, 128, 11, 5, 3000, 1}, {1120, 560, 3, 2, 92, 2}, {70, 10, 1, 1, 93, 5}, {33, 17, 3, 1, 50, 1},
                           {32, 32, 3, 1, 23552, 1}, {100, 560, 1, 1, 92, 3}, {128, 64, 9, 1, 4592, 2}, {96, 80, 3, 1, 700, 2}};
  for (auto& s : shapes) {
    const int Cout = s[0], Cin = s[1], k = s[2], dil = s[3], T = s[4], B = s[5];
    std::vector<float> w((size_t)Cout * Cin * k, 0.01f), b(Cout, 0.1f);
    dmel_conv* c = nullptr;
    CK(dmel_conv_create(&c, w.data(), b.data(), Cout, Cin, k, dil));
    auto x = buf((size_t)B * Cin * T), y = buf((size_t)B * Cout * T), dw = buf((size_t)Cout * Cin * k), db = buf(Cout);
    for (int prec : {DMEL_PRECISION_FP32, DMEL_PRECISION_BF16, DMEL_PRECISION_FP32_MFMA, DMEL_PRECISION_FP32_F16X2, DMEL_PRECISION_FP32_BF16X3}) {
      CK(dmel_conv_set_precision(c, prec));
      CK(dmel_conv_forward(c, x.data(), y.data(), B, T, nullptr));
    }
    CK(dmel_conv_backward_data(c, y.data(), x.data(), B, T, nullptr));
    CK(dmel_conv_backward_weight(c, x.data(), y.data(), dw.data(), db.data(), B, T, nullptr));
    dmel_conv_destroy(c);
  }
  wavenet(10, 70, 70, 20, 4, 0, 16, 93, true);          // encoder (whole-stack kernel table)
  wavenet(48, 48, 48, 3, 4, 0, 4, 17, false);
  wavenet(560, 80, 560, 4, 4, 560, 3, 92, true);        // decoder (conditioned, output projection, XCD-chunked grids, streaming)
  wavenet(64, 64, 64, 5, 0, 64, 2, 40, false);
  quantizer(8, 70, {7, 5, 5}, 3, 93);
  quantizer(2, 70, {8, 6}, 2, 47);
  {
    dmel_convnext* cx = nullptr;
    CK(dmel_convnext_create(&cx, 70));
    CK(dmel_convnext_enable_training(cx, 1));
    std::function<void(const std::string&, std::vector<int64_t>)> S = [&](const std::string& k, std::vector<int64_t> s) { set(cx, dmel_convnext_set_tensor, k, s); };
    convnext_keys("", 70, S);
    CK(dmel_convnext_finalize(cx));
    const int N = 6, T = 46;
    auto x = buf((size_t)N * 70 * T), y = buf((size_t)N * 70 * T), g = buf((size_t)dmel_convnext_grad_floats(cx));
    std::vector<char> ws(dmel_convnext_workspace_bytes(cx, N, T) + 256), tw(dmel_convnext_train_workspace_bytes(cx, N, T) + 256);
    CK(dmel_convnext_forward(cx, x.data(), y.data(), N, T, ws.data(), ws.size(), nullptr));
    CK(dmel_convnext_forward_train(cx, x.data(), y.data(), N, T, tw.data(), tw.size(), nullptr));
    CK(dmel_convnext_backward(cx, x.data(), y.data(), x.data(), g.data(), N, T, tw.data(), tw.size(), nullptr));
    dmel_convnext_destroy(cx);
  }
  discriminator(2, 80, 93);
  discriminator(1, 16, 60);
  { const int r[4] = {8, 8, 2, 2}, k[4] = {16, 16, 4, 4}; bigvgan(4, r, k, 512, 80, 1, true, 2, 12); }
  { const int r[2] = {4, 2}, k[2] = {8, 4}; bigvgan(2, r, k, 32, 20, 2, false, 2, 9); }
  { const int r[6] = {4, 4, 2, 2, 2, 2}, k[6] = {8, 8, 4, 4, 4, 4}; bigvgan(6, r, k, 96, 100, 1, true, 1, 3); }
  {   // standalone transposed convolution / output convolution
    for (int stride : {2, 8}) {
      const int Ci = 40, Co = 24, k = 2 * stride, B = 2, T = 37;
      std::vector<float> w((size_t)Ci * Co * k, 0.02f), b(Co, 0.1f);
      dmel_conv_transpose* h = nullptr;
      CK(dmel_conv_transpose1d_create(&h, w.data(), b.data(), Ci, Co, k, stride));
      auto x = buf((size_t)B * Ci * T), y = buf((size_t)B * Co * T * stride);
      for (int prec : {DMEL_PRECISION_FP32, DMEL_PRECISION_FP32_F16X2, DMEL_PRECISION_BF16}) {
        CK(dmel_conv_transpose1d_set_precision(h, prec));
        CK(dmel_conv_transpose1d_forward(h, x.data(), y.data(), B, T, nullptr));
      }
      dmel_conv_transpose1d_destroy(h);
    }
    dmel_conv_transpose* bad = nullptr;
    if (dmel_conv_transpose1d_create(&bad, buf(64).data(), nullptr, 4, 4, 3, 2) == 0) { std::printf("FAIL conv_transpose1d accepted k != 2 stride\n"); ++failures; }
    auto xp = buf(3 * 32 * 500), wp = buf(32 * 7), yp = buf(3 * 500);
    CK(dmel_conv_post_f32(xp.data(), wp.data(), 0.3f, 2, yp.data(), 3, 32, 7, 500, nullptr));
  }
  for (int nfft : {512, 1024, 2048}) {
    dmel_stft_plan* p = nullptr;
    CK(dmel_stft_plan_create(&p, nfft == 2048 ? 44100 : 24000, nfft, nfft, nfft / 4, nfft == 2048 ? 128 : 80, 0.0, 0.0, nullptr));
    const int B = 2; const int64_t L = 24000;
    auto a = buf((size_t)B * L);
    const int64_t T = dmel_stft_num_frames(p, L);
    auto mel = buf((size_t)B * 128 * T), lin = buf((size_t)B * T * (nfft / 2 + 1));
    CK(dmel_stft_logmel_f32(p, a.data(), L, nullptr, mel.data(), B, L, nullptr));
    CK(dmel_stft_f32(p, a.data(), L, nullptr, nullptr, lin.data(), B, L, nullptr));
    std::vector<float> basis((size_t)128 * (nfft / 2 + 1));
    CK(dmel_stft_plan_mel_basis(p, basis.data()));
    dmel_stft_plan_destroy(p);
  }
  {
    auto x = buf(3 * 5000), y = buf(3 * 7500), bank = buf(3 * 16), al = buf(8), be = buf(8), taps = buf(12);
    CK(dmel_resample_f32(x.data(), y.data(), bank.data(), 3, 5000, 7500, 2, 3, 7, nullptr));
    auto xs = buf(2 * 8 * 3000), ys = buf(2 * 8 * 3000), da = buf(8), db = buf(8);
    CK(dmel_aa_snake_f32(xs.data(), ys.data(), al.data(), be.data(), taps.data(), taps.data(), 1, 2, 8, 3000, nullptr));
    CK(dmel_aa_snake_backward_f32(xs.data(), ys.data(), xs.data(), al.data(), be.data(), da.data(), db.data(), taps.data(), taps.data(), 1, 2, 8, 3000, nullptr));
  }
  std::printf(failures ? "host ASan run: %d FAILURES\n" : "host ASan run: all entry points returned success, no sanitizer report\n", failures);
  return failures ? 1 : 0;
}
