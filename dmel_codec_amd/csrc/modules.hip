// Module-level orchestration: WaveNet, DownsampleFiniteScalarQuantize, BigVGAN.
// Host code only: consumes reference state-dict tensors, folds weight norm, re-tiles weights for the MFMA conv
// kernel and issues the kernel sequence of one forward on the caller's stream (no allocation, no sync).
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>

#include "ops.h"

using namespace dmel;

namespace {

struct HostTensor {
  std::vector<float> v;
  std::vector<int64_t> shape;
  int64_t numel() const { return (int64_t)v.size(); }
};

struct TensorStore {
  std::map<std::string, HostTensor> t;
  int set(const char* key, const float* data, const int64_t* shape, int ndim) {
    DMEL_CHECK_ARG(key && data && shape && ndim >= 0 && ndim <= 4, "set_tensor: bad argument");
    HostTensor h;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { h.shape.push_back(shape[i]); n *= shape[i]; }
    h.v.assign(data, data + n);
    t[key] = std::move(h);
    return DMEL_OK;
  }
  const HostTensor* find(const std::string& k) const {
    auto it = t.find(k);
    return it == t.end() ? nullptr : &it->second;
  }
  // exact-shape lookup; sets the error and returns nullptr when missing or mis-shaped
  const HostTensor* need(const std::string& k, std::initializer_list<int64_t> shape) const {
    const HostTensor* h = find(k);
    if (!h) { set_error("missing state-dict tensor '%s'", k.c_str()); return nullptr; }
    if (h->shape != std::vector<int64_t>(shape)) {
      std::string got, want;
      for (auto s : h->shape) got += std::to_string(s) + ",";
      for (auto s : shape) want += std::to_string(s) + ",";
      set_error("tensor '%s' has shape (%s) but (%s) is required", k.c_str(), got.c_str(), want.c_str());
      return nullptr;
    }
    return h;
  }
  // conv weight with optional old-style weight norm (weight_g, weight_v; norm over all dims but 0), torch._weight_norm
  bool conv_weight(const std::string& prefix, std::initializer_list<int64_t> shape, std::vector<float>& out) const {
    if (find(prefix + "weight")) {
      const HostTensor* w = need(prefix + "weight", shape);
      if (!w) return false;
      out = w->v;
      return true;
    }
    if (find(prefix + "parametrizations.weight.original1")) {      // torch.nn.utils.parametrizations.weight_norm (dim 0)
      const HostTensor* v1 = need(prefix + "parametrizations.weight.original1", shape);
      const HostTensor* g1 = find(prefix + "parametrizations.weight.original0");
      const int64_t n1 = *shape.begin();
      if (!v1 || !g1 || g1->numel() != n1) { set_error("missing or mis-shaped '%sparametrizations.weight.original0'", prefix.c_str()); return false; }
      const int64_t inner1 = v1->numel() / n1;
      out.resize(v1->v.size());
      for (int64_t i = 0; i < n1; ++i) {
        double ss = 0;
        for (int64_t j = 0; j < inner1; ++j) ss += (double)v1->v[i * inner1 + j] * v1->v[i * inner1 + j];
        const float scale = g1->v[i] / (float)std::sqrt(ss);
        for (int64_t j = 0; j < inner1; ++j) out[i * inner1 + j] = v1->v[i * inner1 + j] * scale;
      }
      return true;
    }
    const HostTensor* v = need(prefix + "weight_v", shape);
    if (!v) { set_error("missing state-dict tensor '%sweight' (or weight_g/weight_v)", prefix.c_str()); return false; }
    const HostTensor* g = find(prefix + "weight_g");
    const int64_t n0 = *shape.begin();
    if (!g || g->numel() != n0) { set_error("missing or mis-shaped '%sweight_g'", prefix.c_str()); return false; }
    const int64_t inner = v->numel() / n0;
    out.resize(v->v.size());
    for (int64_t i = 0; i < n0; ++i) {
      double ss = 0;
      for (int64_t j = 0; j < inner; ++j) ss += (double)v->v[i * inner + j] * v->v[i * inner + j];
      const float scale = g->v[i] / (float)std::sqrt(ss);
      for (int64_t j = 0; j < inner; ++j) out[i * inner + j] = v->v[i * inner + j] * scale;
    }
    return true;
  }
};

int upload_vec(DevBuf& d, const std::vector<float>& v) { return d.upload(v.data(), v.size() * sizeof(float)); }

ConvRun run_1seg(const float* x, int Cin, int64_t Tin, float* y, int Cout, int64_t Tout, int B) {
  ConvRun r;
  r.seg[0].x = x; r.seg[0].bstride = (int64_t)Cin * Tin; r.seg[0].cstride = Tin; r.seg[0].Tin = Tin;
  r.B = B; r.Tcols = Tout; r.y = y; r.y_bs = (int64_t)Cout * Tout; r.y_cs = Tout; r.Tout = Tout;
  return r;
}

}  // namespace

// =====================================================================================================
// WaveNet                                               models/modules/wavenet.py:138-225
// =====================================================================================================
struct dmel_wavenet {
  int Cin, Cout, C, L, cycle, Ccond;
  TensorStore ts;
  bool ready = false;
  bool has_in = false, has_out = false;
  PackedConv in_proj, skip_proj, out_proj;
  std::vector<PackedConv> gate, resskip;
  int precision = 0;
  int train_precision = 0;     // DMEL_PRECISION_BF16: bf16 training mode (dmel_wavenet_set_train_precision)
  // training path (enable_training before finalize): unfused forward images that expose the gate pre-activations, and the
  // transposed images that turn every backward-data into a forward convolution
  bool train = false, train_ready = false;
  struct TrainLayer { PackedConv pre_lin, out_lin, pre_dx, pre_dc, out_dz; int dil = 1; };
  std::vector<TrainLayer> tl;
  PackedConv in_dx, skip_dx, out_dx;
  // how every weight image derives from the state-dict tensors, for the device-side re-pack after an optimiser step
  struct Recipe {
    PackedConv* pc;
    struct Seg { std::string key; int64_t rs, cs, ts; int rev; } seg[2];
    std::string b0, b1;
  };
  std::vector<Recipe> recipes;
  void recipe(PackedConv* pc, Recipe::Seg s0, Recipe::Seg s1 = {"", 0, 0, 0, 0}, std::string b0 = "", std::string b1 = "") {
    recipes.push_back({pc, {s0, s1}, b0, b1});
  }
  struct GradSlot { std::string key; int64_t offset, numel; };
  std::vector<GradSlot> slots;
  int64_t grad_floats = 0;
  WaveNetFused fused;          // whole-stack kernel for narrow unconditioned WaveNets on short items (the dMel encoder)
};


extern "C" int dmel_wavenet_create(dmel_wavenet** out, int input_channels, int output_channels, int residual_channels,
                                   int residual_layers, int dilation_cycle, int condition_channels) {
  DMEL_CHECK_ARG(out, "NULL out");
  DMEL_CHECK_ARG(residual_channels > 0 && residual_layers > 0 && dilation_cycle >= 0 && condition_channels >= 0,
                 "wavenet: bad configuration");
  auto* m = new dmel_wavenet();
  m->C = residual_channels;
  m->Cin = input_channels > 0 ? input_channels : residual_channels;
  m->Cout = output_channels > 0 ? output_channels : residual_channels;
  m->L = residual_layers; m->cycle = dilation_cycle; m->Ccond = condition_channels;
  m->has_in = m->Cin != m->C;     // wavenet.py:152-156
  m->has_out = m->Cout != m->C;   // wavenet.py:181-186
  *out = m;
  return DMEL_OK;
}
extern "C" void dmel_wavenet_destroy(dmel_wavenet* m) { delete m; }
extern "C" int dmel_wavenet_set_precision(dmel_wavenet* m, int precision) {
  DMEL_CHECK_ARG(m && valid_precision(precision), "wavenet_set_precision: not a DMEL_PRECISION_* value");
  m->precision = precision;
  return DMEL_OK;
}
extern "C" int dmel_wavenet_set_tensor(dmel_wavenet* m, const char* key, const float* data, const int64_t* shape, int ndim) {
  DMEL_CHECK_ARG(m, "NULL handle");
  m->ready = false;
  return m->ts.set(key, data, shape, ndim);
}

static int pack_pointwise(PackedConv& pc, const TensorStore& ts, const std::string& prefix, int Cout, int Cin) {
  const HostTensor* w = ts.need(prefix + "weight", {Cout, Cin, 1});
  const HostTensor* b = ts.need(prefix + "bias", {Cout});
  if (!w || !b) return DMEL_EMISSING;
  PackDesc d;
  d.mode = EPI_LINEAR; d.nseg = 1; d.C = Cout; d.seg[0].Cin = Cin;
  return pack_conv(pc, d, [&](int, int row, int ci, int) { return w->v[(size_t)row * Cin + ci]; },
                   [&](int row) { return b->v[row]; });
}

// transposed 1x1: y (rows = Cin of the forward conv) = W^T applied to a (Cout-channel) gradient; no bias
static int pack_pointwise_T(PackedConv& pc, const HostTensor* w, int Cout, int Cin) {
  PackDesc d;
  d.mode = EPI_LINEAR; d.nseg = 1; d.C = Cin; d.seg[0].Cin = Cout;
  return pack_conv(pc, d, [&](int, int row, int cc, int) { return w->v[(size_t)cc * Cin + row]; }, [&](int) { return 0.f; });
}

static int wavenet_pack_training(dmel_wavenet* m) {
  const int C = m->C, Cc = m->Ccond;
  m->tl.clear();
  m->tl.resize(m->L);
  m->slots.clear();
  int64_t off = 0;
  auto slot = [&](const std::string& key, int64_t numel) { m->slots.push_back({key, off, numel}); off += numel; };
  if (m->has_in) {
    const HostTensor* w = m->ts.need("input_projection.conv.weight", {C, m->Cin, 1});
    if (!w) return DMEL_EMISSING;
    DMEL_TRY(pack_pointwise_T(m->in_dx, w, C, m->Cin));
    m->recipe(&m->in_dx, {"input_projection.conv.weight", 1, m->Cin, 0, 0});
    slot("input_projection.conv.weight", (int64_t)C * m->Cin);
    slot("input_projection.conv.bias", C);
  }
  for (int i = 0; i < m->L; ++i) {
    const std::string p = "residual_layers." + std::to_string(i) + ".";
    const int dil = m->cycle ? 1 << (i % m->cycle) : 1;
    dmel_wavenet::TrainLayer& t = m->tl[i];
    t.dil = dil;
    const HostTensor* cw = m->ts.need(p + "conv_layer.conv.weight", {2 * C, C, 3});
    const HostTensor* cb = m->ts.need(p + "conv_layer.conv.bias", {2 * C});
    const HostTensor* ow = m->ts.need(p + "output_projection.conv.weight", {2 * C, C, 1});
    const HostTensor* ob = m->ts.need(p + "output_projection.conv.bias", {2 * C});
    if (!cw || !cb || !ow || !ob) return DMEL_EMISSING;
    const HostTensor *qw = nullptr, *qb = nullptr;
    if (Cc) {
      qw = m->ts.need(p + "condition_projection.conv.weight", {2 * C, Cc, 1});
      qb = m->ts.need(p + "condition_projection.conv.bias", {2 * C});
      if (!qw || !qb) return DMEL_EMISSING;
    }
    PackDesc d;      // pre-activation of the gate, rows in their natural order (wavenet.py:121-127)
    d.mode = EPI_LINEAR; d.C = 2 * C; d.nseg = Cc ? 2 : 1;
    d.seg[0].Cin = C; d.seg[0].taps = 3; d.seg[0].dil = dil; d.seg[0].pad_left = dil;
    d.seg[1].Cin = Cc;
    DMEL_TRY(pack_conv(t.pre_lin, d,
                       [&](int sg, int row, int ci, int tap) {
                         return sg == 0 ? cw->v[((size_t)row * C + ci) * 3 + tap] : qw->v[(size_t)row * Cc + ci];
                       },
                       [&](int row) { return cb->v[row] + (qb ? qb->v[row] : 0.f); }));
    PackDesc e;
    e.mode = EPI_LINEAR; e.C = 2 * C; e.nseg = 1; e.seg[0].Cin = C;
    DMEL_TRY(pack_conv(t.out_lin, e, [&](int, int row, int ci, int) { return ow->v[(size_t)row * C + ci]; },
                       [&](int row) { return ob->v[row]; }));
    PackDesc g;      // d pre -> d x: transposed, tap-reversed dilated conv
    g.mode = EPI_LINEAR; g.C = C; g.nseg = 1;
    g.seg[0].Cin = 2 * C; g.seg[0].taps = 3; g.seg[0].dil = dil; g.seg[0].pad_left = dil;
    DMEL_TRY(pack_conv(t.pre_dx, g, [&](int, int row, int cc, int tap) { return cw->v[((size_t)cc * C + row) * 3 + (2 - tap)]; },
                       [&](int) { return 0.f; }));
    if (Cc) DMEL_TRY(pack_pointwise_T(t.pre_dc, qw, 2 * C, Cc));
    DMEL_TRY(pack_pointwise_T(t.out_dz, ow, 2 * C, C));
    m->recipe(&t.pre_lin, {p + "conv_layer.conv.weight", 3 * C, 3, 1, 0},
              Cc ? dmel_wavenet::Recipe::Seg{p + "condition_projection.conv.weight", Cc, 1, 0, 0} : dmel_wavenet::Recipe::Seg{"", 0, 0, 0, 0},
              p + "conv_layer.conv.bias", Cc ? p + "condition_projection.conv.bias" : "");
    m->recipe(&t.out_lin, {p + "output_projection.conv.weight", C, 1, 0, 0}, {"", 0, 0, 0, 0}, p + "output_projection.conv.bias");
    m->recipe(&t.pre_dx, {p + "conv_layer.conv.weight", 3, 3 * C, 1, 1});
    if (Cc) m->recipe(&t.pre_dc, {p + "condition_projection.conv.weight", 1, Cc, 0, 0});
    m->recipe(&t.out_dz, {p + "output_projection.conv.weight", 1, C, 0, 0});
    slot(p + "conv_layer.conv.weight", (int64_t)2 * C * C * 3);
    slot(p + "conv_layer.conv.bias", 2 * C);
    if (Cc) {
      slot(p + "condition_projection.conv.weight", (int64_t)2 * C * Cc);
      slot(p + "condition_projection.conv.bias", 2 * C);
    }
    slot(p + "output_projection.conv.weight", (int64_t)2 * C * C);
    slot(p + "output_projection.conv.bias", 2 * C);
  }
  {
    const HostTensor* w = m->ts.need("skip_projection.conv.weight", {C, C, 1});
    if (!w) return DMEL_EMISSING;
    DMEL_TRY(pack_pointwise_T(m->skip_dx, w, C, C));
    m->recipe(&m->skip_dx, {"skip_projection.conv.weight", 1, C, 0, 0});
    slot("skip_projection.conv.weight", (int64_t)C * C);
    slot("skip_projection.conv.bias", C);
  }
  if (m->has_out) {
    const HostTensor* w = m->ts.need("output_projection.conv.weight", {m->Cout, C, 1});
    if (!w) return DMEL_EMISSING;
    DMEL_TRY(pack_pointwise_T(m->out_dx, w, m->Cout, C));
    m->recipe(&m->out_dx, {"output_projection.conv.weight", 1, C, 0, 0});
    slot("output_projection.conv.weight", (int64_t)m->Cout * C);
    slot("output_projection.conv.bias", m->Cout);
  }
  m->grad_floats = off;
  m->train_ready = true;
  return DMEL_OK;
}

extern "C" int dmel_wavenet_set_train_precision(dmel_wavenet* m, int precision) {
  DMEL_CHECK_ARG(m && (precision == DMEL_PRECISION_FP32 || precision == DMEL_PRECISION_BF16), "wavenet_set_train_precision: DMEL_PRECISION_FP32 or DMEL_PRECISION_BF16");
  m->train_precision = precision;
  return DMEL_OK;
}
extern "C" int dmel_wavenet_enable_training(dmel_wavenet* m, int on) {
  DMEL_CHECK_ARG(m, "NULL handle");
  m->train = on != 0;
  m->ready = false;          // takes effect at the next finalize (the host tensors are needed)
  m->train_ready = false;
  return DMEL_OK;
}

extern "C" int dmel_wavenet_finalize(dmel_wavenet* m) {
  DMEL_CHECK_ARG(m, "NULL handle");
  const int C = m->C;
  m->recipes.clear();
  if (m->has_in) {
    DMEL_TRY(pack_pointwise(m->in_proj, m->ts, "input_projection.conv.", C, m->Cin));
    m->recipe(&m->in_proj, {"input_projection.conv.weight", m->Cin, 1, 0, 0}, {"", 0, 0, 0, 0}, "input_projection.conv.bias");
  }
  m->gate.clear(); m->resskip.clear();
  m->gate.resize(m->L); m->resskip.resize(m->L);
  for (int i = 0; i < m->L; ++i) {
    const std::string p = "residual_layers." + std::to_string(i) + ".";
    const int dil = m->cycle ? 1 << (i % m->cycle) : 1;   // wavenet.py:169
    const HostTensor* cw = m->ts.need(p + "conv_layer.conv.weight", {2 * C, C, 3});
    const HostTensor* cb = m->ts.need(p + "conv_layer.conv.bias", {2 * C});
    const HostTensor* ow = m->ts.need(p + "output_projection.conv.weight", {2 * C, C, 1});
    const HostTensor* ob = m->ts.need(p + "output_projection.conv.bias", {2 * C});
    if (!cw || !cb || !ow || !ob) return DMEL_EMISSING;
    const HostTensor *qw = nullptr, *qb = nullptr;
    if (m->Ccond) {
      qw = m->ts.need(p + "condition_projection.conv.weight", {2 * C, m->Ccond, 1});
      qb = m->ts.need(p + "condition_projection.conv.bias", {2 * C});
      if (!qw || !qb) return DMEL_EMISSING;
    }
    PackDesc d;
    d.mode = EPI_GATE; d.C = C; d.nseg = m->Ccond ? 2 : 1;
    d.seg[0].Cin = C; d.seg[0].taps = 3; d.seg[0].dil = dil; d.seg[0].pad_left = dil;
    d.seg[1].Cin = m->Ccond;
    const int Cc = m->Ccond;
    DMEL_TRY(pack_conv(m->gate[i], d,
                       [&](int sg, int row, int ci, int tap) {
                         return sg == 0 ? cw->v[((size_t)row * C + ci) * 3 + tap] : qw->v[(size_t)row * Cc + ci];
                       },
                       [&](int row) { return cb->v[row] + (qb ? qb->v[row] : 0.f); }));
    PackDesc e;
    e.mode = EPI_RESSKIP; e.C = C; e.nseg = 1; e.seg[0].Cin = C;
    DMEL_TRY(pack_conv(m->resskip[i], e, [&](int, int row, int ci, int) { return ow->v[(size_t)row * C + ci]; },
                       [&](int row) { return ob->v[row]; }));
    m->recipe(&m->gate[i], {p + "conv_layer.conv.weight", 3 * C, 3, 1, 0},
              Cc ? dmel_wavenet::Recipe::Seg{p + "condition_projection.conv.weight", Cc, 1, 0, 0} : dmel_wavenet::Recipe::Seg{"", 0, 0, 0, 0},
              p + "conv_layer.conv.bias", Cc ? p + "condition_projection.conv.bias" : "");
    m->recipe(&m->resskip[i], {p + "output_projection.conv.weight", C, 1, 0, 0}, {"", 0, 0, 0, 0}, p + "output_projection.conv.bias");
  }
  DMEL_TRY(pack_pointwise(m->skip_proj, m->ts, "skip_projection.conv.", C, C));
  m->recipe(&m->skip_proj, {"skip_projection.conv.weight", C, 1, 0, 0}, {"", 0, 0, 0, 0}, "skip_projection.conv.bias");
  if (m->has_out) {
    DMEL_TRY(pack_pointwise(m->out_proj, m->ts, "output_projection.conv.", m->Cout, C));
    m->recipe(&m->out_proj, {"output_projection.conv.weight", C, 1, 0, 0}, {"", 0, 0, 0, 0}, "output_projection.conv.bias");
  }
  if (m->train) DMEL_TRY(wavenet_pack_training(m));
  // whole-stack kernel (wavenet_fused.hip): narrow, unconditioned, no output projection, dilations <= 8
  m->fused.ok = false;
  if (C > 32 && C <= 80 && !m->Ccond && !m->has_out && (!m->has_in || m->Cin <= 16) && (m->cycle == 0 || m->cycle <= 4)) {
    WaveNetFused& f = m->fused;
    f.Cin = m->Cin; f.C = C; f.L = m->L; f.cycle = m->cycle; f.has_in = m->has_in ? 1 : 0;
    f.skip_scale = (float)(1.0 / std::sqrt((double)m->L));
    f.in_w = m->has_in ? m->in_proj.w48.p : nullptr;
    f.in_b = m->has_in ? m->in_proj.bias.as<float>() : nullptr;
    f.skip_w = m->skip_proj.w48.p; f.skip_b = m->skip_proj.bias.as<float>();
    std::vector<const void*> tab((size_t)4 * m->L);
    for (int i = 0; i < m->L; ++i) {
      tab[i] = m->gate[i].w48.p;
      tab[m->L + i] = m->gate[i].bias.p;
      tab[2 * m->L + i] = m->resskip[i].w48.p;
      tab[3 * m->L + i] = m->resskip[i].bias.p;
    }
    DMEL_TRY(f.table.upload(tab.data(), tab.size() * sizeof(void*)));
    f.ok = true;
  }
  m->ts.t.clear();
  m->ready = true;
  return DMEL_OK;
}

// Folded batch: short items (1 s clips = 92 frames) leave the per-item tiling with 96-column tiles that do not divide the chip evenly
// (decoder of cfg 2: 35 row tiles x 32 items = 1120 wave tiles on 1024 SIMDs -> two rounds, the second 9 % full).  Laying the N items
// side by side on ONE time axis, each followed by max-dilation zero columns (the convolution's zero padding, shared by neighbours),
// turns every layer into a single long-row GEMM whose 128-column tiles pack the chip in one round (35 x 25 = 875 wave tiles).
struct FoldGeom {
  bool on = false;
  int P = 0;                 // item pitch = T + gap
  int64_t cols = 0, pitch = 0;
};
static FoldGeom wavenet_fold(const dmel_wavenet* m, int N, int64_t T) {
  // Opt-in (DMEL_WAVENET_FOLD=1, read per call so one process can A/B it): measured on the decoder of cfg 2 (profiles/r02_wavenet_fold.txt)
  // the folded launches are NOT faster with the current conv kernel -- 4.98 ms vs 4.12 ms per forward at the kernel's own tile choice,
  // 4.11 ms when forced onto the same 128 x 96 tile: a lone wave per SIMD runs the K loop at ~45 % of its MFMA time, so trading two
  // half-empty rounds for one full round of single waves gains nothing; what this shape needs is overlap inside a wave, not balance.
  const char* e = getenv("DMEL_WAVENET_FOLD");
  const int mode = e ? atoi(e) : 0;   // 0 off (default), 1 on
  const int max_t = 384;
  FoldGeom f;
  const int maxdil = m->cycle ? 1 << std::min(m->L - 1, m->cycle - 1) : 1;
  f.P = (int)T + maxdil;
  f.cols = (int64_t)N * f.P;
  f.pitch = (int64_t)align_up((size_t)f.cols, 32);
  f.on = N >= 2 && f.cols < ((int64_t)1 << 28) && mode == 1 && T <= max_t;
  return f;
}

struct WavePlan { float *xb, *zb, *sb, *tb, *cf, *xin, *yf; void *xp, *zp, *cp; size_t bytes; };
static WavePlan wavenet_plan(const dmel_wavenet* m, int N, int64_t T, void* ws) {
  Arena a(ws, (size_t)-1);
  WavePlan p{};
  const FoldGeom f = wavenet_fold(m, N, T);
  const size_t n = f.on ? (size_t)m->C * f.pitch : (size_t)N * m->C * T;
  p.xb = a.take<float>(n);
  p.zb = a.take<float>(n);
  p.sb = a.take<float>(n);
  p.tb = m->has_out ? a.take<float>(n) : nullptr;
  // pre-split operand planes of the fp16-split layered path (as many bytes as the fp32 tensors they shadow)
  p.xp = a.take<float>(n);
  p.zp = a.take<float>(n);
  p.cp = m->Ccond ? a.take<float>((size_t)N * m->Ccond * T) : nullptr;
  if (f.on) {
    p.cf = m->Ccond ? a.take<float>((size_t)m->Ccond * f.pitch) : nullptr;
    p.xin = m->has_in ? a.take<float>((size_t)m->Cin * f.pitch) : nullptr;
    p.yf = m->has_out ? a.take<float>((size_t)m->Cout * f.pitch) : nullptr;
  }
  p.bytes = align_up(a.off, 256);
  return p;
}

extern "C" size_t dmel_wavenet_workspace_bytes(const dmel_wavenet* m, int N, int64_t T) {
  if (!m || N <= 0 || T <= 0) return 0;
  return wavenet_plan(m, N, T, nullptr).bytes;
}

// one folded launch: input rows (Cin, pitch), output rows (Cout, pitch), all N items in one "batch item" of f.cols columns
static ConvRun run_folded(const FoldGeom& f, const float* x, float* y, int64_t T) {
  ConvRun r;
  r.seg[0].x = x; r.seg[0].bstride = 0; r.seg[0].cstride = f.pitch; r.seg[0].Tin = f.cols;
  r.B = 1; r.Tcols = f.cols; r.y = y; r.y_bs = 0; r.y_cs = f.pitch; r.Tout = f.cols;
  r.fold_pitch = f.P; r.fold_valid = (int)T;
  return r;
}

static int wavenet_forward_folded(const dmel_wavenet* m, const FoldGeom& f, const WavePlan& p, const float* x, const float* condition,
                                  float* y, int N, int64_t T, const int64_t* in_lengths, const int64_t* out_lengths, int div,
                                  hipStream_t st) {
  const int C = m->C;
  if (m->Ccond) DMEL_TRY(launch_fold(condition, p.cf, nullptr, 1, N, m->Ccond, T, f.P, f.pitch, st));
  if (m->has_in) {  // wavenet.py:205-207
    DMEL_TRY(launch_fold(x, p.xin, in_lengths, div, N, m->Cin, T, f.P, f.pitch, st));
    ConvRun r = run_folded(f, p.xin, p.xb, T);
    r.act = ACT_SILU; r.precision = m->precision;
    DMEL_TRY(launch_conv(m->in_proj, r, st));
  } else {
    DMEL_TRY(launch_fold(x, p.xb, in_lengths, div, N, C, T, f.P, f.pitch, st));
  }
  for (int i = 0; i < m->L; ++i) {  // wavenet.py:116-135
    ConvRun g = run_folded(f, p.xb, p.zb, T);
    if (m->Ccond) { g.seg[1].x = p.cf; g.seg[1].bstride = 0; g.seg[1].cstride = f.pitch; g.seg[1].Tin = f.cols; }
    g.precision = m->precision;
    DMEL_TRY(launch_conv(m->gate[i], g, st));
    ConvRun r = run_folded(f, p.zb, p.xb, T);
    r.skip = p.sb; r.skip_first = (i == 0);
    r.precision = m->precision;
    DMEL_TRY(launch_conv(m->resskip[i], r, st));
  }
  {  // wavenet.py:218-223
    float* proj = m->has_out ? p.tb : p.zb;
    ConvRun r = run_folded(f, p.sb, proj, T);
    r.seg[0].in_scale = (float)(1.0 / std::sqrt((double)m->L));
    if (m->has_out) r.act = ACT_SILU;
    r.precision = m->precision;
    DMEL_TRY(launch_conv(m->skip_proj, r, st));
    if (m->has_out) {
      ConvRun o = run_folded(f, p.tb, p.yf, T);
      o.precision = m->precision;
      DMEL_TRY(launch_conv(m->out_proj, o, st));
      proj = p.yf;
    }
    DMEL_TRY(launch_unfold(proj, y, out_lengths, div, N, m->Cout, T, f.P, f.pitch, st));
  }
  return DMEL_OK;
}

extern "C" int dmel_wavenet_forward(const dmel_wavenet* m, const float* x, const float* condition, float* y, int N, int64_t T,
                                    const int64_t* in_lengths, const int64_t* out_lengths, int group_repeat, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  DMEL_CHECK_ARG(m && x && y && workspace, "wavenet_forward: NULL argument");
  if (!m->ready) { set_error("wavenet_forward: handle not finalized"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG((m->Ccond != 0) == (condition != nullptr), "wavenet_forward: condition tensor does not match the configuration");
  DMEL_CHECK_ARG(N > 0 && T > 0, "wavenet_forward: bad shape");
  const WavePlan p = wavenet_plan(m, N, T, workspace);
  float *xb = p.xb, *zb = p.zb, *sb = p.sb, *tb = p.tb;
  DMEL_CHECK_ARG(workspace_bytes >= p.bytes, "wavenet_forward: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
  hipStream_t st = (hipStream_t)stream;
  const int C = m->C, div = group_repeat > 0 ? group_repeat : 1;
  {  // DMEL_WAVENET_FUSED=0 keeps the layered path (A/B); read per call
    const char* e = getenv("DMEL_WAVENET_FUSED");
    if (m->fused.ok && T <= 96 && m->precision == DMEL_PRECISION_FP32 && !(e && e[0] == '0') && !getenv("DMEL_CONV_FP32_MFMA"))
      return launch_wavenet_fused(m->fused, x, y, in_lengths, out_lengths, div, N, T, st);
  }
  const FoldGeom fold = wavenet_fold(m, N, T);
  if (fold.on) return wavenet_forward_folded(m, fold, p, x, condition, y, N, T, in_lengths, out_lengths, div, st);

  if (m->has_in) {  // wavenet.py:205-207: 1x1 projection + SiLU
    ConvRun r = run_1seg(x, m->Cin, T, xb, C, T, N);
    r.seg[0].in_len = in_lengths; r.len_div = div; r.act = ACT_SILU;
    r.precision = m->precision;
    DMEL_TRY(launch_conv(m->in_proj, r, st));
  } else {
    DMEL_TRY(launch_masked_copy(x, xb, in_lengths, div, N, C, T, st));
  }
  // DMEL_WAVENET_PRESPLIT=1 (round-3 experiment, off by default): at the fp16-split precision every convolution of the stack reads its
  // input as PRE-SPLIT fp16 planes -- the condition is split once per forward instead of once per layer and row block (20 x 9 times for
  // the 560-channel decoder), the gate's output leaves its epilogue as planes only, the residual stream as fp32 + planes -- so that the
  // staging pass copies 16-byte units instead of converting.  Bit-identical (tests), and NOT faster: gate 114.9 vs 112.8 us, residual /
  // skip 46.5 vs 49.6 us per launch at 32 x 92 frames (rocprofv3, profiles/r03_conv_experiments.txt).  The operand conversion is not
  // what these kernels wait for either.
  const char* ps_env = getenv("DMEL_WAVENET_PRESPLIT");
  const bool presplit = m->precision == DMEL_PRECISION_FP32_F16X2 && C % 8 == 0 && (m->Ccond % 8) == 0 &&
                        (m->cycle ? 1 << std::min(m->L - 1, m->cycle - 1) : 1) <= 8 && ps_env && ps_env[0] == '1' && !getenv("DMEL_CONV_FP32_MFMA");
  const int64_t units = (int64_t)N * (C / 8) * T, cunits = (int64_t)N * (m->Ccond / 8) * T;
  if (presplit) {
    DMEL_TRY(launch_split_planes(xb, p.xp, units, nullptr, 1, N, C, T, st));
    if (m->Ccond) DMEL_TRY(launch_split_planes(condition, p.cp, cunits, nullptr, 1, N, m->Ccond, T, st));
  }
  for (int i = 0; i < m->L; ++i) {  // wavenet.py:116-135
    ConvRun g = run_1seg(xb, C, T, zb, C, T, N);
    if (m->Ccond) {
      g.seg[1].x = condition; g.seg[1].bstride = (int64_t)m->Ccond * T; g.seg[1].cstride = T; g.seg[1].Tin = T;
    }
    g.precision = m->precision;
    if (presplit) {
      g.seg[0].xp = p.xp; g.seg[0].xp_plane = units;
      if (m->Ccond) { g.seg[1].xp = p.cp; g.seg[1].xp_plane = cunits; }
      g.yp = p.zp; g.yp_plane = units; g.yp_only = 1;
    }
    DMEL_TRY(launch_conv(m->gate[i], g, st));
    ConvRun r = run_1seg(zb, C, T, xb, C, T, N);
    r.skip = sb; r.skip_first = (i == 0);
    r.precision = m->precision;
    if (presplit) {
      r.seg[0].xp = p.zp; r.seg[0].xp_plane = units;
      if (i + 1 < m->L) { r.yp = p.xp; r.yp_plane = units; }
    }
    DMEL_TRY(launch_conv(m->resskip[i], r, st));
  }
  {  // wavenet.py:218-223
    ConvRun r = run_1seg(sb, C, T, m->has_out ? tb : y, C, T, N);
    r.seg[0].in_scale = (float)(1.0 / std::sqrt((double)m->L));
    if (m->has_out) r.act = ACT_SILU;
    else { r.out_len = out_lengths; r.len_div = div; }
    r.precision = m->precision;
    DMEL_TRY(launch_conv(m->skip_proj, r, st));
    if (m->has_out) {
      ConvRun o = run_1seg(tb, C, T, y, m->Cout, T, N);
      o.out_len = out_lengths; o.len_div = div;
      o.precision = m->precision;
    DMEL_TRY(launch_conv(m->out_proj, o, st));
    }
  }
  return DMEL_OK;
}

// ---- incremental forward (include/dmel_hip.h: dmel_wavenet_stream_step) ----------------------------------------------------
extern "C" int dmel_wavenet_stream_step(const dmel_wavenet* m, float* hist, float* skip, const float* cond, float* y, float* scratch,
                                        int N, int64_t cap, const int64_t* prev, const int64_t* next, void* stream) {
  DMEL_CHECK_ARG(m && hist && skip && y && scratch && prev && next, "wavenet_stream_step: NULL argument");
  if (!m->ready) { set_error("wavenet_stream_step: handle not finalized"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG((m->Ccond != 0) == (cond != nullptr), "wavenet_stream_step: condition tensor does not match the configuration");
  if (m->has_in) { set_error("wavenet_stream_step: models with an input projection are not supported (the decoder has none)"); return DMEL_EUNSUPPORTED; }
  DMEL_CHECK_ARG(N > 0 && cap > 0, "wavenet_stream_step: bad shape");
  const int C = m->C, L = m->L;
  const bool final_step = next[L] == next[0];
  for (int l = 0; l <= L; ++l) {
    DMEL_CHECK_ARG(prev[l] >= 0 && prev[l] <= next[l] && next[l] <= cap, "wavenet_stream_step: level %d: need 0 <= prev <= next <= cap", l);
    if (l > 0) {
      const int dil = m->cycle ? 1 << ((l - 1) % m->cycle) : 1;
      DMEL_CHECK_ARG(final_step ? next[l] == next[0] : (next[l] == prev[l] || next[l] + dil <= next[l - 1]),
                     "wavenet_stream_step: level %d runs ahead of its input (next %lld, input next %lld, dilation %d)", l,
                     (long long)next[l], (long long)next[l - 1], dil);
      DMEL_CHECK_ARG(prev[l] <= prev[l - 1], "wavenet_stream_step: level %d is ahead of level %d", l, l - 1);
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t bs = (int64_t)C * cap, lvl = (int64_t)N * bs;
  float* zb = scratch;                        // gate output of the block being computed, (N, C, cap)
  auto sub = [&](const float* x, int Cin, int64_t valid, int64_t p, float* out, int Cout, int64_t cols) {
    ConvRun r;
    r.seg[0].x = x; r.seg[0].bstride = (int64_t)Cin * cap; r.seg[0].cstride = cap; r.seg[0].Tin = valid; r.seg[0].tshift = p;
    r.B = N; r.Tcols = cols; r.y = out + p; r.y_bs = (int64_t)Cout * cap; r.y_cs = cap; r.Tout = cols;
    r.precision = m->precision;
    return r;
  };
  for (int i = 0; i < L; ++i) {  // wavenet.py:116-135 on the new columns of block i + 1
    const int64_t p = prev[i + 1], cols = next[i + 1] - p;
    if (cols <= 0) continue;
    const float* xin = hist + (int64_t)i * lvl;
    float* xout = hist + (int64_t)(i + 1) * lvl;
    ConvRun g = sub(xin, C, next[i], p, zb, C, cols);
    if (m->Ccond) {
      g.seg[1].x = cond; g.seg[1].bstride = (int64_t)m->Ccond * cap; g.seg[1].cstride = cap; g.seg[1].Tin = next[0]; g.seg[1].tshift = p;
    }
    DMEL_TRY(launch_conv(m->gate[i], g, st));
    // the residual update is in place in the whole-sequence kernel ((x + r) / sqrt2 over x): here block i's input must survive as
    // history, so its new columns are first copied into block i + 1's rows and updated there
    DMEL_HIP(hipMemcpy2DAsync(xout + p, (size_t)cap * sizeof(float), xin + p, (size_t)cap * sizeof(float), (size_t)cols * sizeof(float),
                              (size_t)N * C, hipMemcpyDeviceToDevice, st));
    ConvRun r = sub(zb, C, next[i + 1], p, xout, C, cols);
    r.skip = skip + p; r.skip_first = (i == 0);
    DMEL_TRY(launch_conv(m->resskip[i], r, st));
  }
  {  // wavenet.py:218-223 on the columns whose skip sum is complete
    const int64_t p = prev[L], cols = next[L] - p;
    if (cols > 0) {
      float* tb = scratch + (int64_t)N * bs;    // second half of the scratch
      ConvRun r = sub(skip, C, next[L], p, m->has_out ? tb : y, C, cols);
      r.seg[0].in_scale = (float)(1.0 / std::sqrt((double)m->L));
      if (m->has_out) r.act = ACT_SILU;
      DMEL_TRY(launch_conv(m->skip_proj, r, st));
      if (m->has_out) {
        ConvRun o = sub(tb, C, next[L], p, y, m->Cout, cols);
        DMEL_TRY(launch_conv(m->out_proj, o, st));
      }
    }
  }
  return DMEL_OK;
}

// ---- WaveNet training path --------------------------------------------------------------------------------------
// forward_train computes exactly wavenet.py:204-225 but unfused, keeping what backward needs (every block's input, gate
// pre-activation and gated output; the SiLU inputs); backward is reverse-mode differentiation of that graph -- what
// `manual_backward(loss)` (codec_lit_modules.py:236,315) does through autograd -- built from: backward-data = the forward
// conv kernel on transposed weights, backward-weight = conv_wgrad_kernel, and the elementwise kernels of train_ops.hip.
namespace {
struct TrainPlan {
  float *X, *PRE, *Z, *U, *XS, *P, *Q, *SK, *O;          // saved by forward (X: L+1 slabs, PRE: L x 2n, Z: L slabs)
  float *GXa, *GXb, *GS, *GO, *GXS, *DZ, *DPRE, *DP, *DQ;   // backward temporaries
  size_t n, bytes;
};
TrainPlan train_plan(const dmel_wavenet* m, int N, int64_t T, void* ws) {
  TrainPlan p{};
  Arena a(ws, (size_t)-1);
  const size_t n = (size_t)N * m->C * T;
  p.n = n;
  p.X = a.take<float>(n * (m->L + 1));
  p.PRE = a.take<float>(2 * n * m->L);
  p.Z = a.take<float>(n * m->L);
  p.U = m->has_in ? a.take<float>(n) : nullptr;
  p.XS = a.take<float>(n);
  p.P = a.take<float>(n);
  p.Q = m->has_out ? a.take<float>(n) : nullptr;
  p.SK = a.take<float>(n);
  p.O = a.take<float>(2 * n);
  p.GXa = a.take<float>(n);
  p.GXb = a.take<float>(n);
  p.GS = a.take<float>(n);
  p.GO = a.take<float>(2 * n);
  p.GXS = a.take<float>(n);
  p.DZ = a.take<float>(n);
  p.DPRE = a.take<float>(2 * n);
  p.DP = a.take<float>(n);
  p.DQ = m->has_out ? a.take<float>(n) : nullptr;
  p.bytes = align_up(a.off, 256);
  return p;
}
int train_precision(const dmel_wavenet* m) { return exact_precision(m->precision); }
}  // namespace

extern "C" int dmel_wavenet_refresh(dmel_wavenet* m, int n, const char* const* keys, const float* const* device_tensors,
                                    void* stream) {
  DMEL_CHECK_ARG(m && keys && device_tensors && n > 0, "wavenet_refresh: bad argument");
  if (!m->ready) { set_error("wavenet_refresh: handle not finalized"); return DMEL_EMISSING; }
  std::map<std::string, const float*> dev;
  for (int i = 0; i < n; ++i) {
    DMEL_CHECK_ARG(keys[i] && device_tensors[i], "wavenet_refresh: NULL entry %d", i);
    dev[keys[i]] = device_tensors[i];
  }
  auto find = [&](const std::string& k, const float** out) -> int {
    if (k.empty()) { *out = nullptr; return DMEL_OK; }
    auto it = dev.find(k);
    if (it == dev.end()) { set_error("wavenet_refresh: tensor '%s' was not provided", k.c_str()); return DMEL_EMISSING; }
    *out = it->second;
    return DMEL_OK;
  };
  RepackBatch batch((hipStream_t)stream);      // all convolutions of the network in one launch
  for (auto& r : m->recipes) {
    RepackSrc src;
    for (int sgi = 0; sgi < 2; ++sgi) {
      DMEL_TRY(find(r.seg[sgi].key, &src.seg[sgi].w));
      src.seg[sgi].rs = r.seg[sgi].rs; src.seg[sgi].cs = r.seg[sgi].cs; src.seg[sgi].ts = r.seg[sgi].ts; src.seg[sgi].rev = r.seg[sgi].rev;
    }
    DMEL_TRY(find(r.b0, &src.b0));
    DMEL_TRY(find(r.b1, &src.b1));
    DMEL_TRY(launch_repack(*r.pc, src, (hipStream_t)stream));
  }
  return batch.flush();
}

extern "C" size_t dmel_wavenet_train_workspace_bytes(const dmel_wavenet* m, int N, int64_t T) {
  if (!m || N <= 0 || T <= 0) return 0;
  return train_plan(m, N, T, nullptr).bytes;
}
extern "C" int64_t dmel_wavenet_grad_floats(const dmel_wavenet* m) { return m && m->train_ready ? m->grad_floats : 0; }
extern "C" int dmel_wavenet_grad_slot(const dmel_wavenet* m, const char* key, int64_t* offset, int64_t* numel) {
  DMEL_CHECK_ARG(m && key && offset && numel, "wavenet_grad_slot: NULL argument");
  if (!m->train_ready) { set_error("wavenet_grad_slot: training was not enabled before finalize"); return DMEL_EMISSING; }
  for (const auto& s : m->slots)
    if (s.key == key) { *offset = s.offset; *numel = s.numel; return DMEL_OK; }
  set_error("wavenet_grad_slot: '%s' is not a trained parameter of this WaveNet", key);
  return DMEL_EINVAL;
}

extern "C" int dmel_wavenet_forward_train(const dmel_wavenet* m, const float* x, const float* condition, float* y, int N, int64_t T,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  TrainPrecisionScope train_scope(m ? m->train_precision : 0);
  DMEL_CHECK_ARG(m && x && y && workspace, "wavenet_forward_train: NULL argument");
  if (!m->ready || !m->train_ready) { set_error("wavenet_forward_train: enable_training + finalize first"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG((m->Ccond != 0) == (condition != nullptr), "wavenet_forward_train: condition tensor does not match the configuration");
  DMEL_CHECK_ARG(N > 0 && T > 0, "wavenet_forward_train: bad shape");
  const TrainPlan p = train_plan(m, N, T, workspace);
  DMEL_CHECK_ARG(workspace_bytes >= p.bytes, "wavenet_forward_train: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
  hipStream_t st = (hipStream_t)stream;
  // the forward of a CONDITIONED WaveNet (the decoder: nothing discrete downstream) runs on the three-product fp16 split like its inference
  // forward; an unconditioned one (the encoder: its output is quantised) and every backward pass keep the six-product split
  static const bool fwd_f16 = [] { const char* e = getenv("DMEL_TRAIN_FWD_F16X2"); return !(e && e[0] == '0'); }();
  const int C = m->C, prec = (train_precision(m) == DMEL_PRECISION_FP32 && m->Ccond > 0 && fwd_f16) ? DMEL_PRECISION_FP32_F16X2 : train_precision(m);
  const size_t n = p.n;
  if (m->has_in) {  // wavenet.py:205-207
    ConvRun r = run_1seg(x, m->Cin, T, p.U, C, T, N);
    r.precision = prec;
    DMEL_TRY(launch_conv(m->in_proj, r, st));
    DMEL_TRY(launch_silu_fwd(p.U, p.X, (int64_t)n, st));
  } else {
    DMEL_HIP(hipMemcpyAsync(p.X, x, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  for (int i = 0; i < m->L; ++i) {  // wavenet.py:116-135
    const float* xi = p.X + (size_t)i * n;
    float* pre = p.PRE + (size_t)i * 2 * n;
    float* z = p.Z + (size_t)i * n;
    ConvRun g = run_1seg(xi, C, T, pre, 2 * C, T, N);
    if (m->Ccond) { g.seg[1].x = condition; g.seg[1].bstride = (int64_t)m->Ccond * T; g.seg[1].cstride = T; g.seg[1].Tin = T; }
    g.precision = prec;
    DMEL_TRY(launch_conv(m->tl[i].pre_lin, g, st));
    DMEL_TRY(launch_gate_fwd(pre, z, N, C, T, st));
    ConvRun o = run_1seg(z, C, T, p.O, 2 * C, T, N);
    o.precision = prec;
    DMEL_TRY(launch_conv(m->tl[i].out_lin, o, st));
    DMEL_TRY(launch_resskip_fwd(xi, p.O, p.X + (size_t)(i + 1) * n, p.SK, i == 0, N, C, T, st));
  }
  DMEL_TRY(launch_scale(p.SK, p.XS, (float)(1.0 / std::sqrt((double)m->L)), (int64_t)n, st));   // wavenet.py:218
  {
    ConvRun r = run_1seg(p.XS, C, T, m->has_out ? p.P : y, C, T, N);
    r.precision = prec;
    DMEL_TRY(launch_conv(m->skip_proj, r, st));
  }
  if (m->has_out) {  // wavenet.py:221-223
    DMEL_TRY(launch_silu_fwd(p.P, p.Q, (int64_t)n, st));
    ConvRun o = run_1seg(p.Q, C, T, y, m->Cout, T, N);
    o.precision = prec;
    DMEL_TRY(launch_conv(m->out_proj, o, st));
  }
  return DMEL_OK;
}

extern "C" int dmel_wavenet_backward(const dmel_wavenet* m, const float* x, const float* condition, const float* dy, float* dx,
                                     float* dcondition, float* grads, int N, int64_t T, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  return dmel_wavenet_backward_hooked(m, x, condition, dy, dx, dcondition, grads, N, T, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int dmel_wavenet_backward_hooked(const dmel_wavenet* m, const float* x, const float* condition, const float* dy, float* dx,
                                            float* dcondition, float* grads, int N, int64_t T, void* workspace,
                                            size_t workspace_bytes, void* stream, dmel_grad_ready_fn on_ready, void* user) {
  TrainPrecisionScope train_scope(m ? m->train_precision : 0);
  DMEL_CHECK_ARG(m && x && dy && grads && workspace, "wavenet_backward: NULL argument");
  if (!m->ready || !m->train_ready) { set_error("wavenet_backward: enable_training + finalize first"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG((m->Ccond != 0) == (condition != nullptr), "wavenet_backward: condition tensor does not match the configuration");
  DMEL_CHECK_ARG(N > 0 && T > 0, "wavenet_backward: bad shape");
  const TrainPlan p = train_plan(m, N, T, workspace);
  DMEL_CHECK_ARG(workspace_bytes >= p.bytes, "wavenet_backward: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
  hipStream_t st = (hipStream_t)stream;
  const int C = m->C, Cc = m->Ccond, prec = train_precision(m);
  const size_t n = p.n;
  ClearedRange cleared(grads, (size_t)m->grad_floats * sizeof(float), st);      // one memset for the ~120 gradient tensors of the network
  DMEL_TRY(cleared.error());
  auto G = [&](const std::string& key) -> float* {
    for (const auto& s : m->slots)
      if (s.key == key) return grads + s.offset;
    return nullptr;
  };
  // contiguous region of the flat gradient buffer that belongs to the slots whose key starts with `prefix` (slots are laid out
  // input_projection | residual_layers.0 | ... | residual_layers.L-1 | skip_projection | output_projection)
  auto ready = [&](const std::string& prefix, const std::string& prefix2 = std::string()) {
    if (!on_ready) return;
    int64_t lo = -1, hi = -1;
    for (const auto& s : m->slots) {
      const bool hit = s.key.compare(0, prefix.size(), prefix) == 0 || (!prefix2.empty() && s.key.compare(0, prefix2.size(), prefix2) == 0);
      if (!hit) continue;
      if (lo < 0) lo = s.offset;
      hi = s.offset + s.numel;
    }
    if (lo >= 0) on_ready(user, lo, hi - lo);
  };
  // ---- tail: y = out_proj(silu(P)), P = skip_proj(XS), XS = SK / sqrt(L) ----
  const float* dP = dy;
  if (m->has_out) {
    DMEL_TRY(launch_conv_wgrad(p.Q, dy, G("output_projection.conv.weight"), G("output_projection.conv.bias"), m->Cout, C, 1, 1, N, T, st));
    ConvRun r = run_1seg(dy, m->Cout, T, p.DQ, C, T, N);
    r.precision = prec;
    DMEL_TRY(launch_conv(m->out_dx, r, st));
    DMEL_TRY(launch_silu_bwd(p.DQ, p.P, p.DP, (int64_t)n, st));
    dP = p.DP;
  }
  DMEL_TRY(launch_conv_wgrad(p.XS, dP, G("skip_projection.conv.weight"), G("skip_projection.conv.bias"), C, C, 1, 1, N, T, st));
  {
    ConvRun r = run_1seg(dP, C, T, p.DZ, C, T, N);      // DZ as scratch: d XS
    r.precision = prec;
    DMEL_TRY(launch_conv(m->skip_dx, r, st));
    DMEL_TRY(launch_scale(p.DZ, p.GS, (float)(1.0 / std::sqrt((double)m->L)), (int64_t)n, st));   // d skip of every block
  }
  ready("skip_projection.", "output_projection.");
  // ---- blocks in reverse; gx = d loss / d x_{i+1}; x_L feeds nothing but the (discarded) last residual: gx starts at 0 ----
  float* gx = p.GXa;
  float* gx_prev = p.GXb;
  DMEL_HIP(hipMemsetAsync(gx, 0, n * sizeof(float), st));
  for (int i = m->L - 1; i >= 0; --i) {
    const std::string pk = "residual_layers." + std::to_string(i) + ".";
    const dmel_wavenet::TrainLayer& t = m->tl[i];
    const float* xi = p.X + (size_t)i * n;
    const float* pre = p.PRE + (size_t)i * 2 * n;
    const float* z = p.Z + (size_t)i * n;
    DMEL_TRY(launch_resskip_bwd(gx, p.GS, p.GO, p.GXS, N, C, T, st));
    DMEL_TRY(launch_conv_wgrad(z, p.GO, G(pk + "output_projection.conv.weight"), G(pk + "output_projection.conv.bias"), 2 * C, C, 1, 1, N,
                               T, st));
    {
      ConvRun r = run_1seg(p.GO, 2 * C, T, p.DZ, C, T, N);
      r.precision = prec;
      DMEL_TRY(launch_conv(t.out_dz, r, st));
    }
    DMEL_TRY(launch_gate_bwd(p.DZ, pre, p.DPRE, N, C, T, st));
    float* db = G(pk + "conv_layer.conv.bias");
    DMEL_TRY(launch_conv_wgrad(xi, p.DPRE, G(pk + "conv_layer.conv.weight"), db, 2 * C, C, 3, t.dil, N, T, st));
    if (Cc) {
      DMEL_TRY(launch_conv_wgrad(condition, p.DPRE, G(pk + "condition_projection.conv.weight"), nullptr, 2 * C, Cc, 1, 1, N, T, st));
      DMEL_HIP(hipMemcpyAsync(G(pk + "condition_projection.conv.bias"), db, (size_t)2 * C * sizeof(float), hipMemcpyDeviceToDevice, st));
      if (dcondition) {
        ConvRun r = run_1seg(p.DPRE, 2 * C, T, dcondition, Cc, T, N);
        r.accumulate = (i != m->L - 1);
        r.precision = prec;
        DMEL_TRY(launch_conv(t.pre_dc, r, st));
      }
    }
    {
      ConvRun r = run_1seg(p.DPRE, 2 * C, T, gx_prev, C, T, N);     // d x_i = gx / sqrt2 + W^T d pre
      r.res = p.GXS; r.res_bs = (int64_t)C * T; r.res_cs = T;
      r.precision = prec;
      DMEL_TRY(launch_conv(t.pre_dx, r, st));
    }
    ready(pk);      // every gradient of block i is enqueued: its bucket can leave while blocks i-1 ... 0 still run
    std::swap(gx, gx_prev);
  }
  // ---- head: x_0 = silu(U), U = in_proj(x) ----
  if (m->has_in) {
    DMEL_TRY(launch_silu_bwd(gx, p.U, p.DP, (int64_t)n, st));
    DMEL_TRY(launch_conv_wgrad(x, p.DP, G("input_projection.conv.weight"), G("input_projection.conv.bias"), C, m->Cin, 1, 1, N, T, st));
    if (dx) {
      ConvRun r = run_1seg(p.DP, C, T, dx, m->Cin, T, N);
      r.precision = prec;
      DMEL_TRY(launch_conv(m->in_dx, r, st));
    }
  } else if (dx) {
    DMEL_HIP(hipMemcpyAsync(dx, gx, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  ready("input_projection.");
  return DMEL_OK;
}

// =====================================================================================================
// DownsampleFiniteScalarQuantize (is_dmel)          models/modules/dowmsample_fsq.py:124-147
// =====================================================================================================
namespace {
struct ConvNeXt {                // firefly.py:337-402
  DevBuf dw_w, dw_b, ln_w, ln_b, gamma;
  PackedConv pw1, pw2;
};
int build_convnext(ConvNeXt& cx, const TensorStore& ts, const std::string& p, int C) {
  const HostTensor* dw = ts.need(p + "dwconv.weight", {C, 1, 7});
  const HostTensor* db = ts.need(p + "dwconv.bias", {C});
  const HostTensor* lw = ts.need(p + "norm.weight", {C});
  const HostTensor* lb = ts.need(p + "norm.bias", {C});
  const HostTensor* w1 = ts.need(p + "pwconv1.weight", {4 * C, C});
  const HostTensor* b1 = ts.need(p + "pwconv1.bias", {4 * C});
  const HostTensor* w2 = ts.need(p + "pwconv2.weight", {C, 4 * C});
  const HostTensor* b2 = ts.need(p + "pwconv2.bias", {C});
  const HostTensor* ga = ts.need(p + "gamma", {C});
  if (!dw || !db || !lw || !lb || !w1 || !b1 || !w2 || !b2 || !ga) return DMEL_EMISSING;
  DMEL_TRY(upload_vec(cx.dw_w, dw->v)); DMEL_TRY(upload_vec(cx.dw_b, db->v));
  DMEL_TRY(upload_vec(cx.ln_w, lw->v)); DMEL_TRY(upload_vec(cx.ln_b, lb->v));
  DMEL_TRY(upload_vec(cx.gamma, ga->v));
  PackDesc d;
  d.mode = EPI_LINEAR; d.nseg = 1; d.C = 4 * C; d.seg[0].Cin = C;
  DMEL_TRY(pack_conv(cx.pw1, d, [&](int, int row, int ci, int) { return w1->v[(size_t)row * C + ci]; },
                     [&](int row) { return b1->v[row]; }));
  d.C = C; d.seg[0].Cin = 4 * C;
  DMEL_TRY(pack_conv(cx.pw2, d, [&](int, int row, int ci, int) { return w2->v[(size_t)row * 4 * C + ci]; },
                     [&](int row) { return b2->v[row]; }));
  return DMEL_OK;
}
// y = x + gamma * pwconv2(gelu(pwconv1(LN(dwconv(x)))))      x, y: (N, C, T); y may alias x; h1: (N,C,T), h2: (N,4C,T)
int run_convnext(const ConvNeXt& cx, const float* x, float* y, float* h1, float* h2, int N, int C, int64_t T, hipStream_t st) {
  DMEL_TRY(launch_dwconv_ln(x, h1, cx.dw_w.as<float>(), cx.dw_b.as<float>(), cx.ln_w.as<float>(), cx.ln_b.as<float>(), N, C, T, st));
  ConvRun a = run_1seg(h1, C, T, h2, 4 * C, T, N);
  a.act = ACT_GELU;
  DMEL_TRY(launch_conv(cx.pw1, a, st));
  ConvRun b = run_1seg(h2, 4 * C, T, y, C, T, N);
  b.row_scale = cx.gamma.as<float>();
  b.res = x; b.res_bs = (int64_t)C * T; b.res_cs = T;
  return launch_conv(cx.pw2, b, st);
}
}  // namespace

// ---- standalone ConvNeXtBlock handle (firefly.py:337-402), inference and training -----------------------------------
struct dmel_convnext {
  int train_precision = 0;     // DMEL_PRECISION_BF16: bf16 training mode
  int C = 0;
  TensorStore ts;
  bool ready = false, train = false, train_ready = false;
  ConvNeXt cx;
  PackedConv pw1T, pw2T;          // transposed images: backward-data of the two Linear layers
  struct GradSlot { std::string key; int64_t offset, numel; };
  std::vector<GradSlot> slots;
  int64_t grad_floats = 0;
};

namespace {
int pack_convnext_T(PackedConv& pw1T, PackedConv& pw2T, const TensorStore& ts, const std::string& p, int C);
}

extern "C" int dmel_convnext_create(dmel_convnext** out, int dim) {
  DMEL_CHECK_ARG(out && dim > 0, "convnext_create: bad argument");
  auto* m = new dmel_convnext();
  m->C = dim;
  *out = m;
  return DMEL_OK;
}
extern "C" void dmel_convnext_destroy(dmel_convnext* m) { delete m; }
extern "C" int dmel_convnext_set_tensor(dmel_convnext* m, const char* key, const float* data, const int64_t* shape, int ndim) {
  DMEL_CHECK_ARG(m, "NULL handle");
  m->ready = false;
  return m->ts.set(key, data, shape, ndim);
}
extern "C" int dmel_convnext_set_train_precision(dmel_convnext* m, int precision) {
  DMEL_CHECK_ARG(m && (precision == DMEL_PRECISION_FP32 || precision == DMEL_PRECISION_BF16), "convnext_set_train_precision: DMEL_PRECISION_FP32 or DMEL_PRECISION_BF16");
  m->train_precision = precision;
  return DMEL_OK;
}
extern "C" int dmel_convnext_enable_training(dmel_convnext* m, int on) {
  DMEL_CHECK_ARG(m, "NULL handle");
  m->train = on != 0;
  m->ready = false;
  m->train_ready = false;
  return DMEL_OK;
}
extern "C" int dmel_convnext_finalize(dmel_convnext* m) {
  DMEL_CHECK_ARG(m, "NULL handle");
  const int C = m->C;
  DMEL_TRY(build_convnext(m->cx, m->ts, "", C));
  if (m->train) {
    DMEL_TRY(pack_convnext_T(m->pw1T, m->pw2T, m->ts, "", C));
    m->slots.clear();
    int64_t off = 0;
    auto slot = [&](const char* key, int64_t numel) { m->slots.push_back({key, off, numel}); off += numel; };
    slot("dwconv.weight", (int64_t)C * 7); slot("dwconv.bias", C); slot("norm.weight", C); slot("norm.bias", C);
    slot("pwconv1.weight", (int64_t)4 * C * C); slot("pwconv1.bias", 4 * C); slot("pwconv2.weight", (int64_t)4 * C * C);
    slot("pwconv2.bias", C); slot("gamma", C);
    m->grad_floats = off;
    m->train_ready = true;
  }
  m->ts.t.clear();
  m->ready = true;
  return DMEL_OK;
}
extern "C" size_t dmel_convnext_workspace_bytes(const dmel_convnext* m, int N, int64_t T) {
  if (!m || N <= 0 || T <= 0) return 0;
  return align_up((size_t)5 * N * m->C * T * sizeof(float), 256);
}
extern "C" int dmel_convnext_forward(const dmel_convnext* m, const float* x, float* y, int N, int64_t T, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  DMEL_CHECK_ARG(m && x && y && workspace, "convnext_forward: NULL argument");
  if (!m->ready) { set_error("convnext_forward: handle not finalized"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG(N > 0 && T > 0 && workspace_bytes >= dmel_convnext_workspace_bytes(m, N, T), "convnext_forward: bad shape or workspace");
  float* h1 = reinterpret_cast<float*>(workspace);
  return run_convnext(m->cx, x, y, h1, h1 + (size_t)N * m->C * T, N, m->C, T, (hipStream_t)stream);
}

namespace {
struct CxPlan { float *H0, *H1, *U, *G, *V, *DV, *DG, *DH1, *DH0; size_t n, bytes; };
CxPlan cx_plan_at(Arena& a, size_t n) {
  CxPlan p{};
  p.n = n;
  p.H0 = a.take<float>(n); p.H1 = a.take<float>(n); p.U = a.take<float>(4 * n); p.G = a.take<float>(4 * n); p.V = a.take<float>(n);
  p.DV = a.take<float>(n); p.DG = a.take<float>(4 * n); p.DH1 = a.take<float>(n); p.DH0 = a.take<float>(n);
  return p;
}
CxPlan cx_plan(const dmel_convnext* m, int N, int64_t T, void* ws) {
  Arena a(ws, (size_t)-1);
  CxPlan p = cx_plan_at(a, (size_t)N * m->C * T);
  p.bytes = align_up(a.off, 256);
  return p;
}
struct CxGrads { float *dw_w, *dw_b, *ln_w, *ln_b, *w1, *b1, *w2, *b2, *gamma; };

// firefly.py:383-402 unfused, keeping the pre-norm, normed, pre-GELU, post-GELU and pre-scale tensors for backward
int convnext_train_fwd(const ConvNeXt& cx, const CxPlan& p, const float* x, float* y, int N, int C, int64_t T, hipStream_t st) {
  DMEL_TRY(launch_dwconv_ln(x, p.H1, cx.dw_w.as<float>(), cx.dw_b.as<float>(), cx.ln_w.as<float>(), cx.ln_b.as<float>(), N, C, T, st, p.H0));
  ConvRun a = run_1seg(p.H1, C, T, p.U, 4 * C, T, N);
  DMEL_TRY(launch_conv(cx.pw1, a, st));
  DMEL_TRY(launch_gelu_fwd(p.U, p.G, (int64_t)(4 * p.n), st));
  ConvRun b = run_1seg(p.G, 4 * C, T, p.V, C, T, N);
  DMEL_TRY(launch_conv(cx.pw2, b, st));
  return launch_layerscale_res_fwd(x, p.V, cx.gamma.as<float>(), y, N, C, T, st);
}
int convnext_bwd(const ConvNeXt& cx, const PackedConv& pw1T, const PackedConv& pw2T, const CxPlan& p, const CxGrads& g, const float* x,
                 const float* dy, float* dx, int N, int C, int64_t T, hipStream_t st) {
  DMEL_TRY(launch_layerscale_bwd(dy, p.V, cx.gamma.as<float>(), p.DV, g.gamma, N, C, T, st));          // y = x + gamma * V
  DMEL_TRY(launch_conv_wgrad(p.G, p.DV, g.w2, g.b2, C, 4 * C, 1, 1, N, T, st));                         // V = pwconv2(G)
  {
    ConvRun r = run_1seg(p.DV, C, T, p.DG, 4 * C, T, N);
    DMEL_TRY(launch_conv(pw2T, r, st));
  }
  DMEL_TRY(launch_gelu_bwd(p.DG, p.U, p.DG, (int64_t)(4 * p.n), st));                                   // G = gelu(U); d U overwrites d G
  DMEL_TRY(launch_conv_wgrad(p.H1, p.DG, g.w1, g.b1, 4 * C, C, 1, 1, N, T, st));                        // U = pwconv1(H1)
  {
    ConvRun r = run_1seg(p.DG, 4 * C, T, p.DH1, C, T, N);
    DMEL_TRY(launch_conv(pw1T, r, st));
  }
  // H1 = LayerNorm(H0), H0 = dwconv(x); dx = dy (identity path) + dwconv^T(d H0)
  DMEL_TRY(launch_ln_bwd(p.DH1, p.H0, cx.ln_w.as<float>(), p.DH0, g.ln_w, g.ln_b, N, C, T, st));
  return launch_dwconv_bwd(p.DH0, x, cx.dw_w.as<float>(), dy, dx, g.dw_w, g.dw_b, N, C, T, st);
}
int pack_convnext_T(PackedConv& pw1T, PackedConv& pw2T, const TensorStore& ts, const std::string& p, int C) {
  const HostTensor* w1 = ts.need(p + "pwconv1.weight", {4 * C, C});
  const HostTensor* w2 = ts.need(p + "pwconv2.weight", {C, 4 * C});
  if (!w1 || !w2) return DMEL_EMISSING;
  PackDesc d;
  d.mode = EPI_LINEAR; d.nseg = 1; d.C = C; d.seg[0].Cin = 4 * C;          // d u (4C) -> d h1 (C): W1^T
  DMEL_TRY(pack_conv(pw1T, d, [&](int, int row, int cc, int) { return w1->v[(size_t)cc * C + row]; }, [&](int) { return 0.f; }));
  d.C = 4 * C; d.seg[0].Cin = C;                                           // d v (C) -> d g (4C): W2^T
  return pack_conv(pw2T, d, [&](int, int row, int cc, int) { return w2->v[(size_t)cc * 4 * C + row]; }, [&](int) { return 0.f; });
}
}  // namespace

extern "C" size_t dmel_convnext_train_workspace_bytes(const dmel_convnext* m, int N, int64_t T) {
  if (!m || N <= 0 || T <= 0) return 0;
  return cx_plan(m, N, T, nullptr).bytes;
}
extern "C" int64_t dmel_convnext_grad_floats(const dmel_convnext* m) { return m && m->train_ready ? m->grad_floats : 0; }
extern "C" int dmel_convnext_grad_slot(const dmel_convnext* m, const char* key, int64_t* offset, int64_t* numel) {
  DMEL_CHECK_ARG(m && key && offset && numel, "convnext_grad_slot: NULL argument");
  if (!m->train_ready) { set_error("convnext_grad_slot: training was not enabled before finalize"); return DMEL_EMISSING; }
  for (const auto& s : m->slots)
    if (s.key == key) { *offset = s.offset; *numel = s.numel; return DMEL_OK; }
  set_error("convnext_grad_slot: '%s' is not a parameter of ConvNeXtBlock", key);
  return DMEL_EINVAL;
}

extern "C" int dmel_convnext_forward_train(const dmel_convnext* m, const float* x, float* y, int N, int64_t T, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  TrainPrecisionScope train_scope(m ? m->train_precision : 0);
  DMEL_CHECK_ARG(m && x && y && workspace, "convnext_forward_train: NULL argument");
  if (!m->ready || !m->train_ready) { set_error("convnext_forward_train: enable_training + finalize first"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG(N > 0 && T > 0, "convnext_forward_train: bad shape");
  const CxPlan p = cx_plan(m, N, T, workspace);
  DMEL_CHECK_ARG(workspace_bytes >= p.bytes, "convnext_forward_train: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
  return convnext_train_fwd(m->cx, p, x, y, N, m->C, T, (hipStream_t)stream);
}

extern "C" int dmel_convnext_backward(const dmel_convnext* m, const float* x, const float* dy, float* dx, float* grads, int N, int64_t T,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  TrainPrecisionScope train_scope(m ? m->train_precision : 0);
  DMEL_CHECK_ARG(m && x && dy && dx && grads && workspace, "convnext_backward: NULL argument");
  if (!m->ready || !m->train_ready) { set_error("convnext_backward: enable_training + finalize first"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG(N > 0 && T > 0, "convnext_backward: bad shape");
  const CxPlan p = cx_plan(m, N, T, workspace);
  DMEL_CHECK_ARG(workspace_bytes >= p.bytes, "convnext_backward: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
  auto G = [&](const char* key) -> float* {
    for (const auto& s : m->slots)
      if (s.key == key) return grads + s.offset;
    return nullptr;
  };
  CxGrads g{G("dwconv.weight"), G("dwconv.bias"), G("norm.weight"), G("norm.bias"), G("pwconv1.weight"), G("pwconv1.bias"),
            G("pwconv2.weight"), G("pwconv2.bias"), G("gamma")};
  return convnext_bwd(m->cx, m->pw1T, m->pw2T, p, g, x, dy, dx, N, m->C, T, (hipStream_t)stream);
}

struct dmel_quantizer {
  int train_precision = 0;     // DMEL_PRECISION_BF16: bf16 training mode
  int dim, G, Cg, D, nf;
  int factors[4];
  int levels[4];
  FsqConst fk;
  TensorStore ts;
  bool ready = false;
  std::vector<PackedConv> down, up;
  std::vector<ConvNeXt> down_cx, up_cx;
  DevBuf w_in, b_in, w_out, b_out;
  // training path
  bool train = false, train_ready = false;
  std::vector<PackedConv> down_dx, up_dx;                      // backward-data images of the k2s2 conv / transposed conv
  std::vector<PackedConv> down_pw1T, down_pw2T, up_pw1T, up_pw2T;
  struct GradSlot { std::string key; int64_t offset, numel; };
  std::vector<GradSlot> slots;
  int64_t grad_floats = 0;
  int64_t slot_of(const std::string& key) const {
    for (const auto& s : slots)
      if (s.key == key) return s.offset;
    return -1;
  }
};

extern "C" int dmel_quantizer_create(dmel_quantizer** out, int input_dim, int n_groups, const int* levels, int n_levels,
                                     const int* downsample_factor, int n_factors, int fsq_prebound) {
  DMEL_CHECK_ARG(out && levels && downsample_factor, "NULL argument");
  DMEL_CHECK_ARG(n_groups > 0 && input_dim > 0 && input_dim % n_groups == 0, "quantizer: input_dim %d not divisible by groups %d",
                 input_dim, n_groups);
  DMEL_CHECK_ARG(n_factors >= 1 && n_factors <= 4, "quantizer: 1..4 downsample stages supported");
  for (int i = 0; i < n_factors; ++i)
    if (downsample_factor[i] != 2) {
      set_error("quantizer: downsample factor %d unsupported (only 2)", downsample_factor[i]);
      return DMEL_EUNSUPPORTED;
    }
  auto* q = new dmel_quantizer();
  q->dim = input_dim; q->G = n_groups; q->Cg = input_dim / n_groups; q->D = n_levels; q->nf = n_factors;
  for (int i = 0; i < n_factors; ++i) q->factors[i] = downsample_factor[i];
  int rc = make_fsq_const(q->fk, levels, n_levels, fsq_prebound);
  if (rc != DMEL_OK) { delete q; return rc; }
  for (int i = 0; i < n_levels; ++i) q->levels[i] = levels[i];
  *out = q;
  return DMEL_OK;
}
extern "C" int dmel_quantizer_set_strict(dmel_quantizer* q, int on) {
  DMEL_CHECK_ARG(q, "NULL handle");
  q->fk.strict = on != 0;
  return DMEL_OK;
}
extern "C" void dmel_quantizer_destroy(dmel_quantizer* q) { delete q; }
extern "C" int dmel_quantizer_set_tensor(dmel_quantizer* q, const char* key, const float* data, const int64_t* shape, int ndim) {
  DMEL_CHECK_ARG(q, "NULL handle");
  q->ready = false;
  return q->ts.set(key, data, shape, ndim);
}

extern "C" int dmel_quantizer_finalize(dmel_quantizer* q) {
  DMEL_CHECK_ARG(q, "NULL handle");
  const int C = q->Cg, G = q->G, D = q->D;
  q->down.clear(); q->up.clear(); q->down_cx.clear(); q->up_cx.clear();
  q->down.resize(q->nf); q->up.resize(q->nf); q->down_cx.resize(q->nf); q->up_cx.resize(q->nf);
  for (int i = 0; i < q->nf; ++i) {
    {  // downsample.{i}.0 = Conv1d(C, C, k=2, stride=2): two 1-tap segments reading x[2s] and x[2s+1]
      const std::string p = "downsample." + std::to_string(i) + ".0.";
      const HostTensor* w = q->ts.need(p + "weight", {C, C, 2});
      const HostTensor* b = q->ts.need(p + "bias", {C});
      if (!w || !b) return DMEL_EMISSING;
      PackDesc d;
      d.mode = EPI_LINEAR; d.nseg = 2; d.C = C;
      for (int s = 0; s < 2; ++s) { d.seg[s].Cin = C; d.seg[s].tstride = 2; d.seg[s].toff = s; }
      DMEL_TRY(pack_conv(q->down[i], d, [&](int sg, int row, int ci, int) { return w->v[((size_t)row * C + ci) * 2 + sg]; },
                         [&](int row) { return b->v[row]; }));
      DMEL_TRY(build_convnext(q->down_cx[i], q->ts, "downsample." + std::to_string(i) + ".1.", C));
    }
    {  // upsample.{i}.0 = ConvTranspose1d(C, C, k=2, stride=2): weight (Cin, Cout, 2); out[2q+ph] = W[:, :, ph]^T x[q]
      const std::string p = "upsample." + std::to_string(i) + ".0.";
      const HostTensor* w = q->ts.need(p + "weight", {C, C, 2});
      const HostTensor* b = q->ts.need(p + "bias", {C});
      if (!w || !b) return DMEL_EMISSING;
      PackDesc d;
      d.mode = EPI_LINEAR; d.nseg = 1; d.C = C; d.phases = 2; d.seg[0].Cin = C;
      DMEL_TRY(pack_conv(q->up[i], d,
                         [&](int, int row, int ci, int) { int ph = row / C, co = row % C; return w->v[((size_t)ci * C + co) * 2 + ph]; },
                         [&](int row) { return b->v[row % C]; }));
      DMEL_TRY(build_convnext(q->up_cx[i], q->ts, "upsample." + std::to_string(i) + ".1.", C));
    }
  }
  std::vector<float> wi((size_t)G * D * C), bi((size_t)G * D), wo((size_t)G * C * D), bo((size_t)G * C);
  for (int g = 0; g < G; ++g) {
    const std::string p = "residual_fsq.rvqs." + std::to_string(g) + ".";
    const HostTensor* a = q->ts.need(p + "project_in.weight", {D, C});
    const HostTensor* ab = q->ts.need(p + "project_in.bias", {D});
    const HostTensor* o = q->ts.need(p + "project_out.weight", {C, D});
    const HostTensor* ob = q->ts.need(p + "project_out.bias", {C});
    if (!a || !ab || !o || !ob) return DMEL_EMISSING;
    std::copy(a->v.begin(), a->v.end(), wi.begin() + (size_t)g * D * C);
    std::copy(ab->v.begin(), ab->v.end(), bi.begin() + (size_t)g * D);
    std::copy(o->v.begin(), o->v.end(), wo.begin() + (size_t)g * C * D);
    std::copy(ob->v.begin(), ob->v.end(), bo.begin() + (size_t)g * C);
  }
  DMEL_TRY(upload_vec(q->w_in, wi)); DMEL_TRY(upload_vec(q->b_in, bi));
  DMEL_TRY(upload_vec(q->w_out, wo)); DMEL_TRY(upload_vec(q->b_out, bo));
  if (q->train) {
    q->down_dx.clear(); q->up_dx.clear(); q->down_pw1T.clear(); q->down_pw2T.clear(); q->up_pw1T.clear(); q->up_pw2T.clear();
    q->down_dx.resize(q->nf); q->up_dx.resize(q->nf);
    q->down_pw1T.resize(q->nf); q->down_pw2T.resize(q->nf); q->up_pw1T.resize(q->nf); q->up_pw2T.resize(q->nf);
    q->slots.clear();
    int64_t off = 0;
    auto slot = [&](const std::string& key, int64_t numel) { q->slots.push_back({key, off, numel}); off += numel; };
    const char* cxk[9] = {"dwconv.weight", "dwconv.bias", "norm.weight", "norm.bias", "pwconv1.weight", "pwconv1.bias", "pwconv2.weight",
                          "pwconv2.bias", "gamma"};
    const int64_t cxn[9] = {(int64_t)C * 7, C, C, C, (int64_t)4 * C * C, 4 * C, (int64_t)4 * C * C, C, C};
    for (int i = 0; i < q->nf; ++i) {
      const std::string pd = "downsample." + std::to_string(i) + ".", pu = "upsample." + std::to_string(i) + ".";
      const HostTensor* wd = q->ts.need(pd + "0.weight", {C, C, 2});
      const HostTensor* wu = q->ts.need(pu + "0.weight", {C, C, 2});
      if (!wd || !wu) return DMEL_EMISSING;
      {  // d x[ci, 2q + k] = sum_co W[co, ci, k] d y[co, q]: a k2s2 transposed conv (phase-major rows)
        PackDesc d;
        d.mode = EPI_LINEAR; d.nseg = 1; d.C = C; d.phases = 2; d.seg[0].Cin = C;
        DMEL_TRY(pack_conv(q->down_dx[i], d,
                           [&](int, int row, int cc, int) { int ph = row / C, ci = row % C; return wd->v[((size_t)cc * C + ci) * 2 + ph]; },
                           [&](int) { return 0.f; }));
      }
      {  // d x[ci, q] = sum_{co, k} W[ci, co, k] d y[co, 2q + k]: a k2s2 conv (two strided 1-tap segments)
        PackDesc d;
        d.mode = EPI_LINEAR; d.nseg = 2; d.C = C;
        for (int sgi = 0; sgi < 2; ++sgi) { d.seg[sgi].Cin = C; d.seg[sgi].tstride = 2; d.seg[sgi].toff = sgi; }
        DMEL_TRY(pack_conv(q->up_dx[i], d, [&](int sg, int row, int cc, int) { return wu->v[((size_t)row * C + cc) * 2 + sg]; },
                           [&](int) { return 0.f; }));
      }
      DMEL_TRY(pack_convnext_T(q->down_pw1T[i], q->down_pw2T[i], q->ts, pd + "1.", C));
      DMEL_TRY(pack_convnext_T(q->up_pw1T[i], q->up_pw2T[i], q->ts, pu + "1.", C));
      slot(pd + "0.weight", (int64_t)C * C * 2); slot(pd + "0.bias", C);
      for (int k = 0; k < 9; ++k) slot(pd + "1." + cxk[k], cxn[k]);
      slot(pu + "0.weight", (int64_t)C * C * 2); slot(pu + "0.bias", C);
      for (int k = 0; k < 9; ++k) slot(pu + "1." + cxk[k], cxn[k]);
    }
    // FSQ parameter gradients come out in the packed (G, ...) layouts: group g's tensor is the g-th block of each region
    const int64_t o_wi = off; off += (int64_t)G * D * C;
    const int64_t o_bi = off; off += (int64_t)G * D;
    const int64_t o_wo = off; off += (int64_t)G * C * D;
    const int64_t o_bo = off; off += (int64_t)G * C;
    for (int g = 0; g < G; ++g) {
      const std::string p = "residual_fsq.rvqs." + std::to_string(g) + ".";
      q->slots.push_back({p + "project_in.weight", o_wi + (int64_t)g * D * C, (int64_t)D * C});
      q->slots.push_back({p + "project_in.bias", o_bi + (int64_t)g * D, D});
      q->slots.push_back({p + "project_out.weight", o_wo + (int64_t)g * C * D, (int64_t)C * D});
      q->slots.push_back({p + "project_out.bias", o_bo + (int64_t)g * C, C});
    }
    q->slots.push_back({"#fsq.w_in", o_wi, (int64_t)G * D * C});
    q->slots.push_back({"#fsq.b_in", o_bi, (int64_t)G * D});
    q->slots.push_back({"#fsq.w_out", o_wo, (int64_t)G * C * D});
    q->slots.push_back({"#fsq.b_out", o_bo, (int64_t)G * C});
    q->grad_floats = off;
    q->train_ready = true;
  }
  q->ts.t.clear();
  q->ready = true;
  return DMEL_OK;
}

extern "C" int dmel_quantizer_set_train_precision(dmel_quantizer* q, int precision) {
  DMEL_CHECK_ARG(q && (precision == DMEL_PRECISION_FP32 || precision == DMEL_PRECISION_BF16), "quantizer_set_train_precision: DMEL_PRECISION_FP32 or DMEL_PRECISION_BF16");
  q->train_precision = precision;
  return DMEL_OK;
}
extern "C" int dmel_quantizer_enable_training(dmel_quantizer* q, int on) {
  DMEL_CHECK_ARG(q, "NULL handle");
  q->train = on != 0;
  q->ready = false;
  q->train_ready = false;
  return DMEL_OK;
}

static size_t quantizer_plan(const dmel_quantizer* q, int B, int64_t Tmax, void* ws, float** a, float** b, float** h1, float** h2) {
  Arena ar(ws, (size_t)-1);
  const size_t n = (size_t)B * q->G * q->Cg * Tmax;
  float* pa = ar.take<float>(n);
  float* pb = ar.take<float>(n);
  float* p1 = ar.take<float>(n);
  float* p2 = ar.take<float>(4 * n);
  if (a) { *a = pa; *b = pb; *h1 = p1; *h2 = p2; }
  return align_up(ar.off, 256);
}

extern "C" size_t dmel_quantizer_workspace_bytes(const dmel_quantizer* q, int B, int64_t T) {
  if (!q || B <= 0 || T <= 0) return 0;
  return quantizer_plan(q, B, T, nullptr, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int dmel_quantizer_encode(const dmel_quantizer* q, const float* z, int32_t* ids, float* prequant, int B, int64_t T,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  return dmel_quantizer_encode_ex(q, z, ids, prequant, nullptr, B, T, workspace, workspace_bytes, stream);
}

extern "C" int dmel_quantizer_encode_ex(const dmel_quantizer* q, const float* z, int32_t* ids, float* prequant, float* latents, int B,
                                        int64_t T, void* workspace, size_t workspace_bytes, void* stream) {
  DMEL_CHECK_ARG(q && z && ids && workspace, "quantizer_encode: NULL argument");
  if (!q->ready) { set_error("quantizer_encode: handle not finalized"); return DMEL_EMISSING; }
  int64_t Tq = T;
  for (int i = 0; i < q->nf; ++i) Tq /= 2;
  DMEL_CHECK_ARG(B > 0 && Tq > 0, "quantizer_encode: sequence too short for the downsampling (T=%lld)", (long long)T);
  float *pa, *pb, *h1, *h2;
  const size_t need = quantizer_plan(q, B, T, workspace, &pa, &pb, &h1, &h2);
  DMEL_CHECK_ARG(workspace_bytes >= need, "quantizer_encode: workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const int N = B * q->G, C = q->Cg;
  const float* cur = z;
  int64_t Tc = T;
  float* bufs[2] = {pa, pb};
  for (int i = 0; i < q->nf; ++i) {  // dowmsample_fsq.py:49-62
    const int64_t Tn = Tc / 2;
    float* o = bufs[i & 1];
    ConvRun r;
    for (int s = 0; s < 2; ++s) { r.seg[s].x = cur; r.seg[s].bstride = (int64_t)C * Tc; r.seg[s].cstride = Tc; r.seg[s].Tin = Tc; }
    r.B = N; r.Tcols = Tn; r.y = o; r.y_bs = (int64_t)C * Tn; r.y_cs = Tn; r.Tout = Tn;
    DMEL_TRY(launch_conv(q->down[i], r, st));
    DMEL_TRY(run_convnext(q->down_cx[i], o, o, h1, h2, N, C, Tn, st));
    cur = o;
    Tc = Tn;
  }
  if (latents) DMEL_HIP(hipMemcpyAsync(latents, cur, (size_t)N * C * Tc * sizeof(float), hipMemcpyDeviceToDevice, st));
  // "(b g) f t -> b (g f) t" is a view; FSQ per group (dowmsample_fsq.py:127-132)
  return launch_fsq_encode(cur, q->w_in.as<float>(), q->b_in.as<float>(), ids, prequant, q->fk, B, q->G, C, Tc, st);
}

extern "C" int dmel_quantizer_decode(const dmel_quantizer* q, const int32_t* ids, float* zout, int B, int64_t T4, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  DMEL_CHECK_ARG(q && ids && zout && workspace, "quantizer_decode: NULL argument");
  if (!q->ready) { set_error("quantizer_decode: handle not finalized"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG(B > 0 && T4 > 0, "quantizer_decode: bad shape");
  int64_t Tfull = T4;
  for (int i = 0; i < q->nf; ++i) Tfull *= 2;
  float *pa, *pb, *h1, *h2;
  const size_t need = quantizer_plan(q, B, Tfull, workspace, &pa, &pb, &h1, &h2);
  DMEL_CHECK_ARG(workspace_bytes >= need, "quantizer_decode: workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const int N = B * q->G, C = q->Cg;
  DMEL_TRY(launch_fsq_decode(ids, q->w_out.as<float>(), q->b_out.as<float>(), pa, q->fk, B, q->G, C, T4, st));
  const float* cur = pa;
  int64_t Tc = T4;
  float* bufs[2] = {pb, pa};
  for (int j = 0; j < q->nf; ++j) {  // dowmsample_fsq.py:64-77 (upsample.{j}: reversed factor order, all factors are 2)
    const int64_t Tn = Tc * 2;
    float* o = (j == q->nf - 1) ? zout : bufs[j & 1];
    ConvRun r = run_1seg(cur, C, Tc, o, C, Tn, N);
    r.Tcols = Tc; r.out_tstride = 2; r.Tout = Tn;
    DMEL_TRY(launch_conv(q->up[j], r, st));
    DMEL_TRY(run_convnext(q->up_cx[j], o, o, h1, h2, N, C, Tn, st));
    cur = o;
    Tc = Tn;
  }
  return DMEL_OK;
}

// Re-pack every weight image and parameter buffer of a finalized quantiser from device tensors (after an optimiser step).
extern "C" int dmel_quantizer_refresh(dmel_quantizer* q, int n, const char* const* keys, const float* const* device_tensors, void* stream) {
  DMEL_CHECK_ARG(q && keys && device_tensors && n > 0, "quantizer_refresh: bad argument");
  if (!q->ready) { set_error("quantizer_refresh: handle not finalized"); return DMEL_EMISSING; }
  std::map<std::string, const float*> dev;
  for (int i = 0; i < n; ++i) {
    DMEL_CHECK_ARG(keys[i] && device_tensors[i], "quantizer_refresh: NULL entry %d", i);
    dev[keys[i]] = device_tensors[i];
  }
  bool missing = false;
  auto T = [&](const std::string& k) -> const float* {
    auto it = dev.find(k);
    if (it == dev.end()) { set_error("quantizer_refresh: tensor '%s' was not provided", k.c_str()); missing = true; return nullptr; }
    return it->second;
  };
  hipStream_t st = (hipStream_t)stream;
  RepackBatch batch(st);
  const int C = q->Cg, G = q->G, D = q->D;
  auto copy = [&](DevBuf& dst, const float* src, size_t count) -> int {
    if (!src) return DMEL_EMISSING;
    DMEL_HIP(hipMemcpyAsync(dst.p, src, count * sizeof(float), hipMemcpyDeviceToDevice, st));
    return DMEL_OK;
  };
  auto seg = [](const float* w, int64_t rs, int64_t cs, int64_t ps = 0, int pC = 0) {
    RepackSeg s;
    s.w = w; s.rs = rs; s.cs = cs; s.ps = ps; s.pC = pC;
    return s;
  };
  auto refresh_cx = [&](ConvNeXt& cx, PackedConv* pw1T, PackedConv* pw2T, const std::string& p) -> int {
    const float *w1 = T(p + "pwconv1.weight"), *b1 = T(p + "pwconv1.bias"), *w2 = T(p + "pwconv2.weight"), *b2 = T(p + "pwconv2.bias");
    const float *dw = T(p + "dwconv.weight"), *db = T(p + "dwconv.bias"), *lw = T(p + "norm.weight"), *lb = T(p + "norm.bias");
    const float* ga = T(p + "gamma");
    if (missing) return DMEL_EMISSING;
    DMEL_TRY(copy(cx.dw_w, dw, (size_t)C * 7)); DMEL_TRY(copy(cx.dw_b, db, C)); DMEL_TRY(copy(cx.ln_w, lw, C));
    DMEL_TRY(copy(cx.ln_b, lb, C)); DMEL_TRY(copy(cx.gamma, ga, C));
    RepackSrc a; a.seg[0] = seg(w1, C, 1); a.b0 = b1;
    DMEL_TRY(launch_repack(cx.pw1, a, st));
    RepackSrc b; b.seg[0] = seg(w2, 4 * C, 1); b.b0 = b2;
    DMEL_TRY(launch_repack(cx.pw2, b, st));
    if (pw1T) {
      RepackSrc c; c.seg[0] = seg(w1, 1, C);
      DMEL_TRY(launch_repack(*pw1T, c, st));
      RepackSrc d; d.seg[0] = seg(w2, 1, 4 * C);
      DMEL_TRY(launch_repack(*pw2T, d, st));
    }
    return DMEL_OK;
  };
  for (int i = 0; i < q->nf; ++i) {
    const std::string pd = "downsample." + std::to_string(i) + ".", pu = "upsample." + std::to_string(i) + ".";
    const float *wd = T(pd + "0.weight"), *bd = T(pd + "0.bias"), *wu = T(pu + "0.weight"), *bu = T(pu + "0.bias");
    if (missing) return DMEL_EMISSING;
    {  // value(sg, row, ci) = wd[(row*C + ci)*2 + sg]
      RepackSrc a; a.seg[0] = seg(wd, 2 * C, 2); a.seg[1] = seg(wd + 1, 2 * C, 2); a.b0 = bd;
      DMEL_TRY(launch_repack(q->down[i], a, st));
    }
    {  // value(row = (ph, co), ci) = wu[(ci*C + co)*2 + ph]; bias[co]
      RepackSrc a; a.seg[0] = seg(wu, 2, 2 * C, 1, C); a.b0 = bu; a.bias_mod = C;
      DMEL_TRY(launch_repack(q->up[i], a, st));
    }
    DMEL_TRY(refresh_cx(q->down_cx[i], q->train_ready ? &q->down_pw1T[i] : nullptr, q->train_ready ? &q->down_pw2T[i] : nullptr, pd + "1."));
    DMEL_TRY(refresh_cx(q->up_cx[i], q->train_ready ? &q->up_pw1T[i] : nullptr, q->train_ready ? &q->up_pw2T[i] : nullptr, pu + "1."));
    if (q->train_ready) {
      {  // value(row = (ph, ci), cc) = wd[(cc*C + ci)*2 + ph]
        RepackSrc a; a.seg[0] = seg(wd, 2, 2 * C, 1, C);
        DMEL_TRY(launch_repack(q->down_dx[i], a, st));
      }
      {  // value(sg, row, cc) = wu[(row*C + cc)*2 + sg]
        RepackSrc a; a.seg[0] = seg(wu, 2 * C, 2); a.seg[1] = seg(wu + 1, 2 * C, 2);
        DMEL_TRY(launch_repack(q->up_dx[i], a, st));
      }
    }
  }
  for (int g = 0; g < G; ++g) {
    const std::string p = "residual_fsq.rvqs." + std::to_string(g) + ".";
    const float *a = T(p + "project_in.weight"), *ab = T(p + "project_in.bias"), *o = T(p + "project_out.weight"), *ob = T(p + "project_out.bias");
    if (missing) return DMEL_EMISSING;
    DMEL_HIP(hipMemcpyAsync(q->w_in.as<float>() + (size_t)g * D * C, a, (size_t)D * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    DMEL_HIP(hipMemcpyAsync(q->b_in.as<float>() + (size_t)g * D, ab, (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, st));
    DMEL_HIP(hipMemcpyAsync(q->w_out.as<float>() + (size_t)g * C * D, o, (size_t)C * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    DMEL_HIP(hipMemcpyAsync(q->b_out.as<float>() + (size_t)g * C, ob, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  return batch.flush();
}

// ---- quantiser training path: DownsampleFiniteScalarQuantize.forward (dowmsample_fsq.py:86-122) and its backward ----------
namespace {
struct QStage { float* in; float* conv; CxPlan cx; int64_t Tin, Tout; };      // input of the (transposed) conv, its output = ConvNeXt input
struct QPlan {
  QStage down[4], up[4];
  float *lat, *fsq_out, *ga, *gb, *scratch;
  int32_t* ids;
  int64_t T4;
  size_t bytes;
};
QPlan q_plan(const dmel_quantizer* q, int B, int64_t T, void* ws) {
  QPlan p{};
  Arena a(ws, (size_t)-1);
  const size_t NC = (size_t)B * q->G * q->Cg;
  int64_t Tc = T;
  for (int i = 0; i < q->nf; ++i) {
    const int64_t Tn = Tc / 2;
    p.down[i].Tin = Tc; p.down[i].Tout = Tn;
    p.down[i].in = i == 0 ? nullptr : a.take<float>(NC * Tc);     // stage 0 reads the caller's z
    p.down[i].conv = a.take<float>(NC * Tn);
    p.down[i].cx = cx_plan_at(a, NC * Tn);
    Tc = Tn;
  }
  p.T4 = Tc;
  p.lat = a.take<float>(NC * Tc);
  p.ids = a.take<int32_t>((size_t)B * q->G * Tc);
  p.fsq_out = a.take<float>(NC * Tc);
  p.scratch = a.take<float>((size_t)2 * B * q->G * Tc * q->D);
  for (int j = 0; j < q->nf; ++j) {
    const int64_t Tn = Tc * 2;
    p.up[j].Tin = Tc; p.up[j].Tout = Tn;
    p.up[j].in = j == 0 ? p.fsq_out : a.take<float>(NC * Tc);
    p.up[j].conv = a.take<float>(NC * Tn);
    p.up[j].cx = cx_plan_at(a, NC * Tn);
    Tc = Tn;
  }
  p.ga = a.take<float>(NC * T);
  p.gb = a.take<float>(NC * T);
  p.bytes = align_up(a.off, 256);
  return p;
}
CxGrads cx_grads(const dmel_quantizer* q, float* grads, const std::string& p) {
  auto G = [&](const char* k) { return grads + q->slot_of(p + k); };
  return CxGrads{G("dwconv.weight"), G("dwconv.bias"), G("norm.weight"), G("norm.bias"), G("pwconv1.weight"), G("pwconv1.bias"),
                 G("pwconv2.weight"), G("pwconv2.bias"), G("gamma")};
}
}  // namespace

extern "C" size_t dmel_quantizer_train_workspace_bytes(const dmel_quantizer* q, int B, int64_t T) {
  if (!q || B <= 0 || T <= 0) return 0;
  return q_plan(q, B, T, nullptr).bytes;
}
extern "C" int64_t dmel_quantizer_grad_floats(const dmel_quantizer* q) { return q && q->train_ready ? q->grad_floats : 0; }
extern "C" int dmel_quantizer_grad_slot(const dmel_quantizer* q, const char* key, int64_t* offset, int64_t* numel) {
  DMEL_CHECK_ARG(q && key && offset && numel, "quantizer_grad_slot: NULL argument");
  if (!q->train_ready) { set_error("quantizer_grad_slot: training was not enabled before finalize"); return DMEL_EMISSING; }
  for (const auto& s : q->slots)
    if (s.key == key) { *offset = s.offset; *numel = s.numel; return DMEL_OK; }
  set_error("quantizer_grad_slot: '%s' is not a trained parameter of this quantiser", key);
  return DMEL_EINVAL;
}

// z (B*G, Cg, T) -> zq (B*G, Cg, T) [= (B, G*Cg, T)], ids (B, G, T4), latents (B*G, Cg, T4); zq is zero-padded from 2^nf * T4 to T
// (left = diff / 2) exactly as dowmsample_fsq.py:113-120.
extern "C" int dmel_quantizer_forward_train(const dmel_quantizer* q, const float* z, float* zq, int32_t* ids, float* latents, int B,
                                            int64_t T, void* workspace, size_t workspace_bytes, void* stream) {
  TrainPrecisionScope train_scope(q ? q->train_precision : 0);
  DMEL_CHECK_ARG(q && z && zq && workspace, "quantizer_forward_train: NULL argument");
  if (!q->ready || !q->train_ready) { set_error("quantizer_forward_train: enable_training + finalize first"); return DMEL_EMISSING; }
  const QPlan p = q_plan(q, B, T, workspace);
  DMEL_CHECK_ARG(B > 0 && p.T4 > 0, "quantizer_forward_train: sequence too short for the downsampling (T=%lld)", (long long)T);
  DMEL_CHECK_ARG(workspace_bytes >= p.bytes, "quantizer_forward_train: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
  hipStream_t st = (hipStream_t)stream;
  const int N = B * q->G, C = q->Cg;
  const size_t NC = (size_t)N * C;
  const float* cur = z;
  for (int i = 0; i < q->nf; ++i) {
    const QStage& sgq = p.down[i];
    ConvRun r;
    for (int s = 0; s < 2; ++s) { r.seg[s].x = cur; r.seg[s].bstride = (int64_t)C * sgq.Tin; r.seg[s].cstride = sgq.Tin; r.seg[s].Tin = sgq.Tin; }
    r.B = N; r.Tcols = sgq.Tout; r.y = sgq.conv; r.y_bs = (int64_t)C * sgq.Tout; r.y_cs = sgq.Tout; r.Tout = sgq.Tout;
    DMEL_TRY(launch_conv(q->down[i], r, st));
    float* o = (i == q->nf - 1) ? p.lat : p.down[i + 1].in;
    DMEL_TRY(convnext_train_fwd(q->down_cx[i], sgq.cx, sgq.conv, o, N, C, sgq.Tout, st));
    cur = o;
  }
  DMEL_TRY(launch_fsq_encode(p.lat, q->w_in.as<float>(), q->b_in.as<float>(), p.ids, nullptr, q->fk, B, q->G, C, p.T4, st));
  DMEL_TRY(launch_fsq_decode(p.ids, q->w_out.as<float>(), q->b_out.as<float>(), p.fsq_out, q->fk, B, q->G, C, p.T4, st));
  if (ids) DMEL_HIP(hipMemcpyAsync(ids, p.ids, (size_t)B * q->G * p.T4 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  if (latents) DMEL_HIP(hipMemcpyAsync(latents, p.lat, NC * p.T4 * sizeof(float), hipMemcpyDeviceToDevice, st));
  cur = p.fsq_out;
  for (int j = 0; j < q->nf; ++j) {
    const QStage& sgq = p.up[j];
    ConvRun r = run_1seg(cur, C, sgq.Tin, sgq.conv, C, sgq.Tout, N);
    r.Tcols = sgq.Tin; r.out_tstride = 2; r.Tout = sgq.Tout;
    DMEL_TRY(launch_conv(q->up[j], r, st));
    float* o = (j == q->nf - 1) ? p.ga : p.up[j + 1].in;
    DMEL_TRY(convnext_train_fwd(q->up_cx[j], sgq.cx, sgq.conv, o, N, C, sgq.Tout, st));
    cur = o;
  }
  const int64_t Tfull = p.up[q->nf - 1].Tout, diff = T - Tfull, left = diff / 2;
  if (diff == 0) {
    DMEL_HIP(hipMemcpyAsync(zq, p.ga, NC * T * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else {
    DMEL_HIP(hipMemsetAsync(zq, 0, NC * T * sizeof(float), st));
    DMEL_HIP(hipMemcpy2DAsync(zq + left, (size_t)T * sizeof(float), p.ga, (size_t)Tfull * sizeof(float), (size_t)Tfull * sizeof(float), NC,
                              hipMemcpyDeviceToDevice, st));
  }
  return DMEL_OK;
}

extern "C" int dmel_quantizer_backward(const dmel_quantizer* q, const float* z, const float* dzq, float* dz, float* grads, int B, int64_t T,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  TrainPrecisionScope train_scope(q ? q->train_precision : 0);
  DMEL_CHECK_ARG(q && z && dzq && dz && grads && workspace, "quantizer_backward: NULL argument");
  if (!q->ready || !q->train_ready) { set_error("quantizer_backward: enable_training + finalize first"); return DMEL_EMISSING; }
  const QPlan p = q_plan(q, B, T, workspace);
  DMEL_CHECK_ARG(B > 0 && p.T4 > 0 && workspace_bytes >= p.bytes, "quantizer_backward: bad shape or workspace");
  hipStream_t st = (hipStream_t)stream;
  const int N = B * q->G, C = q->Cg;
  const size_t NC = (size_t)N * C;
  ClearedRange cleared(grads, (size_t)q->grad_floats * sizeof(float), st);
  DMEL_TRY(cleared.error());
  const int64_t Tfull = p.up[q->nf - 1].Tout, diff = T - Tfull, left = diff / 2;
  // un-pad: gradient of the cropped region only
  float* g = p.gb;
  if (diff == 0) {
    DMEL_HIP(hipMemcpyAsync(g, dzq, NC * T * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else {
    DMEL_HIP(hipMemcpy2DAsync(g, (size_t)Tfull * sizeof(float), dzq + left, (size_t)T * sizeof(float), (size_t)Tfull * sizeof(float), NC,
                              hipMemcpyDeviceToDevice, st));
  }
  float* other = p.ga;
  for (int j = q->nf - 1; j >= 0; --j) {      // upsample.{j}: ConvTranspose1d(k2, s2) -> ConvNeXtBlock
    const QStage& sgq = p.up[j];
    const std::string pu = "upsample." + std::to_string(j) + ".";
    DMEL_TRY(convnext_bwd(q->up_cx[j], q->up_pw1T[j], q->up_pw2T[j], sgq.cx, cx_grads(q, grads, pu + "1."), sgq.conv, g, other, N, C,
                          sgq.Tout, st));
    std::swap(g, other);                       // g = d (transposed conv output), length Tout
    float* dw = grads + q->slot_of(pu + "0.weight");
    DMEL_TRY(zero_unless_cleared(dw, (size_t)C * C * 2 * sizeof(float), st));
    for (int k = 0; k < 2; ++k)                // d W[ci, co, k] = sum x[ci, q] d y[co, 2q + k]
      DMEL_TRY(launch_conv_wgrad_strided(sgq.in, g, dw, C, C, 2, k, sgq.Tin, sgq.Tout, 2, k, N, st));
    DMEL_TRY(launch_conv_bgrad(g, grads + q->slot_of(pu + "0.bias"), C, N, sgq.Tout, st));
    {
      ConvRun r;
      for (int s = 0; s < 2; ++s) { r.seg[s].x = g; r.seg[s].bstride = (int64_t)C * sgq.Tout; r.seg[s].cstride = sgq.Tout; r.seg[s].Tin = sgq.Tout; }
      r.B = N; r.Tcols = sgq.Tin; r.y = other; r.y_bs = (int64_t)C * sgq.Tin; r.y_cs = sgq.Tin; r.Tout = sgq.Tin;
      DMEL_TRY(launch_conv(q->up_dx[j], r, st));
    }
    std::swap(g, other);                       // g = d (stage input), length Tin
  }
  // FSQ (straight-through)
  DMEL_TRY(launch_fsq_backward(p.lat, g, q->w_in.as<float>(), q->b_in.as<float>(), q->w_out.as<float>(), other,
                               grads + q->slot_of("#fsq.w_in"), grads + q->slot_of("#fsq.b_in"), grads + q->slot_of("#fsq.w_out"),
                               grads + q->slot_of("#fsq.b_out"), p.scratch, q->fk, B, q->G, C, p.T4, st));
  std::swap(g, other);
  for (int i = q->nf - 1; i >= 0; --i) {      // downsample.{i}: Conv1d(k2, s2) -> ConvNeXtBlock
    const QStage& sgq = p.down[i];
    const std::string pd = "downsample." + std::to_string(i) + ".";
    DMEL_TRY(convnext_bwd(q->down_cx[i], q->down_pw1T[i], q->down_pw2T[i], sgq.cx, cx_grads(q, grads, pd + "1."), sgq.conv, g, other, N, C,
                          sgq.Tout, st));
    std::swap(g, other);                       // g = d (conv output), length Tout
    const float* xin = i == 0 ? z : sgq.in;
    float* dw = grads + q->slot_of(pd + "0.weight");
    DMEL_TRY(zero_unless_cleared(dw, (size_t)C * C * 2 * sizeof(float), st));
    for (int k = 0; k < 2; ++k)                // d W[co, ci, k] = sum d y[co, q] x[ci, 2q + k]
      DMEL_TRY(launch_conv_wgrad_strided(g, xin, dw, C, C, 2, k, sgq.Tout, sgq.Tin, 2, k, N, st));
    DMEL_TRY(launch_conv_bgrad(g, grads + q->slot_of(pd + "0.bias"), C, N, sgq.Tout, st));
    float* dst = i == 0 ? dz : other;
    if (sgq.Tin != 2 * sgq.Tout) DMEL_HIP(hipMemsetAsync(dst, 0, NC * sgq.Tin * sizeof(float), st));   // odd length: the last sample is unused
    {
      ConvRun r = run_1seg(g, C, sgq.Tout, dst, C, sgq.Tin, N);
      r.Tcols = sgq.Tout; r.out_tstride = 2; r.Tout = sgq.Tin;
      DMEL_TRY(launch_conv(q->down_dx[i], r, st));
    }
    if (i > 0) std::swap(g, other);
  }
  return DMEL_OK;
}

// =====================================================================================================
// Discriminator                                     models/modules/discriminator.py:6-35
// =====================================================================================================
// Six weight-normed Conv2d over the mel image (B, 1, H = n_mels, W = frames), kernels (3, kw), stride (1, sw), SiLU in between.
// A (3, kw) Conv2d is three 1-D convolutions over W, one per kernel row dh, summed: y[:, h] += conv1d_dh(x[:, h + dh - 1]).
// Activations are kept in a flattened, zero-padded image layout (see DPlan below) in which row h + dh - 1 of the input is the same
// buffer displaced by one row pitch, so the implicit-GEMM conv kernel runs unchanged over B long items.  A stride-2 convolution
// over W is the sum of its two polyphase branches (even taps on x[2w'], odd taps on x[2w' + 1]): two K segments.
namespace {
struct DLayer {
  int Cin, Cout, kw, sw, pw;
  PackedConv fwd[3];                      // per kernel row dh
  std::vector<float> w;                   // folded weight (Cout, Cin, 3, kw), kept for the training images
  PackedConv bwd[3][2];                   // backward-data per kernel row (and per output phase for the stride-2 layers)
  DevBuf g_dev, v_dev;                    // weight-norm parameters on the device (for d g / d v)
};
constexpr int kDiscLayers = 6;
const int kDiscCfg[kDiscLayers][5] = {{1, 64, 9, 1, 4}, {64, 128, 9, 2, 4}, {128, 256, 9, 2, 4}, {256, 512, 9, 2, 4}, {512, 1024, 3, 1, 1},
                                       {1024, 1, 3, 1, 1}};
int64_t disc_out_w(const DLayer& l, int64_t W) { return (W + 2 * l.pw - l.kw) / l.sw + 1; }
}  // namespace

struct dmel_discriminator {
  int train_precision = 0;     // DMEL_PRECISION_BF16: bf16 training mode
  TensorStore ts;
  bool ready = false, train = false, train_ready = false;
  DLayer layer[kDiscLayers];
  struct GradSlot { std::string key; int64_t offset, numel; };
  std::vector<GradSlot> slots;
  int64_t grad_floats = 0;
  int64_t slot_of(const std::string& key) const {
    for (const auto& sl : slots)
      if (sl.key == key) return sl.offset;
    return -1;
  }
};

extern "C" int dmel_discriminator_create(dmel_discriminator** out) {
  DMEL_CHECK_ARG(out, "NULL out");
  auto* d = new dmel_discriminator();
  for (int i = 0; i < kDiscLayers; ++i) {
    d->layer[i].Cin = kDiscCfg[i][0]; d->layer[i].Cout = kDiscCfg[i][1]; d->layer[i].kw = kDiscCfg[i][2]; d->layer[i].sw = kDiscCfg[i][3];
    d->layer[i].pw = kDiscCfg[i][4];
  }
  *out = d;
  return DMEL_OK;
}
extern "C" void dmel_discriminator_destroy(dmel_discriminator* d) { delete d; }
extern "C" int dmel_discriminator_set_tensor(dmel_discriminator* d, const char* key, const float* data, const int64_t* shape, int ndim) {
  DMEL_CHECK_ARG(d, "NULL handle");
  d->ready = false;
  return d->ts.set(key, data, shape, ndim);
}

static int disc_pack_forward(DLayer& l, const std::vector<float>& bias) {
  const int Cin = l.Cin, kw = l.kw;
  const std::vector<float>& w = l.w;
  for (int dh = 0; dh < 3; ++dh) {
    PackDesc d;
    d.mode = EPI_LINEAR; d.C = l.Cout;
    auto W = [&, dh](int row, int ci, int dw) { return w[(((size_t)row * Cin + ci) * 3 + dh) * kw + dw]; };
    auto B = [&, dh](int row) { return dh == 1 ? bias[row] : 0.f; };     // the bias rides the centre row's launch
    if (l.sw == 1) {
      d.nseg = 1;
      d.seg[0].Cin = Cin; d.seg[0].taps = kw; d.seg[0].dil = 1; d.seg[0].pad_left = l.pw;
      DMEL_TRY(pack_conv(l.fwd[dh], d, [&](int, int row, int ci, int tap) { return W(row, ci, tap); }, B));
    } else {
      d.nseg = 2;                                                       // even taps on x[2w' + ...], odd taps on x[2w' + 1 + ...]
      for (int sg = 0; sg < 2; ++sg) {
        d.seg[sg].Cin = Cin; d.seg[sg].taps = (kw + 1 - sg) / 2; d.seg[sg].dil = 1; d.seg[sg].tstride = 2; d.seg[sg].toff = sg;
        d.seg[sg].pad_left = l.pw;
      }
      DMEL_TRY(pack_conv(l.fwd[dh], d, [&](int sg, int row, int ci, int tap) { return W(row, ci, 2 * tap + sg); }, B));
    }
  }
  return DMEL_OK;
}

// backward-data images.  y[n] (+)= W_dh * x[n + dh - 1]  =>  d x[m] (+)= W_dh^T * d y[m - dh + 1]:
//   stride 1: the same convolution with the taps reversed (kw = 2 pw + 1, so the padding is unchanged);
//   stride 2: output phase ph of d x takes the taps dw = ph (mod 2): d x[2q + ph] = sum_t W[.., 8 - 2t - ph] d y[q + t - (2 - ph)].
static int disc_pack_backward(DLayer& l) {
  const int Cin = l.Cin, kw = l.kw;
  const std::vector<float>& w = l.w;
  for (int dh = 0; dh < 3; ++dh) {
    auto W = [&, dh](int co, int ci, int dw) { return w[(((size_t)co * Cin + ci) * 3 + dh) * kw + dw]; };
    if (l.sw == 1) {
      PackDesc d;
      d.mode = EPI_LINEAR; d.C = Cin; d.nseg = 1;
      d.seg[0].Cin = l.Cout; d.seg[0].taps = kw; d.seg[0].dil = 1; d.seg[0].pad_left = l.pw;
      DMEL_TRY(pack_conv(l.bwd[dh][0], d, [&](int, int row, int cc, int tap) { return W(cc, row, kw - 1 - tap); }, [&](int) { return 0.f; }));
    } else {
      for (int ph = 0; ph < 2; ++ph) {
        PackDesc d;
        d.mode = EPI_LINEAR; d.C = Cin; d.nseg = 1;
        d.seg[0].Cin = l.Cout; d.seg[0].taps = 5 - ph; d.seg[0].dil = 1; d.seg[0].pad_left = 2 - ph;
        DMEL_TRY(pack_conv(l.bwd[dh][ph], d, [&](int, int row, int cc, int tap) { return W(cc, row, 8 - 2 * tap - ph); },
                           [&](int) { return 0.f; }));
      }
    }
  }
  return DMEL_OK;
}

extern "C" int dmel_discriminator_finalize(dmel_discriminator* d) {
  DMEL_CHECK_ARG(d, "NULL handle");
  for (int i = 0; i < kDiscLayers; ++i) {
    DLayer& l = d->layer[i];
    const std::string p = "blocks." + std::to_string(2 * i) + ".";
    if (!d->ts.conv_weight(p, {l.Cout, l.Cin, 3, l.kw}, l.w)) return DMEL_EMISSING;
    const HostTensor* b = d->ts.need(p + "bias", {l.Cout});
    if (!b) return DMEL_EMISSING;
    DMEL_TRY(disc_pack_forward(l, b->v));
    if (d->train) {
      const HostTensor* g0 = d->ts.need(p + "parametrizations.weight.original0", {l.Cout, 1, 1, 1});
      const HostTensor* v0 = d->ts.need(p + "parametrizations.weight.original1", {l.Cout, l.Cin, 3, l.kw});
      if (!g0 || !v0) { set_error("discriminator training needs the weight-normed form (parametrizations.weight.original0|1) of '%s'", p.c_str()); return DMEL_EMISSING; }
      DMEL_TRY(upload_vec(l.g_dev, g0->v));
      DMEL_TRY(upload_vec(l.v_dev, v0->v));
      DMEL_TRY(disc_pack_backward(l));
    }
  }
  if (d->train) {
    d->slots.clear();
    int64_t off = 0;
    for (int i = 0; i < kDiscLayers; ++i) {
      const DLayer& l = d->layer[i];
      const std::string p = "blocks." + std::to_string(2 * i) + ".";
      d->slots.push_back({p + "bias", off, l.Cout}); off += l.Cout;
      d->slots.push_back({p + "parametrizations.weight.original0", off, l.Cout}); off += l.Cout;
      const int64_t nw = (int64_t)l.Cout * l.Cin * 3 * l.kw;
      d->slots.push_back({p + "parametrizations.weight.original1", off, nw}); off += nw;
    }
    d->grad_floats = off;
    d->train_ready = true;
  }
  d->ts.t.clear();
  d->ready = true;
  return DMEL_OK;
}

extern "C" int dmel_discriminator_set_train_precision(dmel_discriminator* d, int precision) {
  DMEL_CHECK_ARG(d && (precision == DMEL_PRECISION_FP32 || precision == DMEL_PRECISION_BF16), "discriminator_set_train_precision: DMEL_PRECISION_FP32 or DMEL_PRECISION_BF16");
  d->train_precision = precision;
  return DMEL_OK;
}
extern "C" int dmel_discriminator_enable_training(dmel_discriminator* d, int on) {
  DMEL_CHECK_ARG(d, "NULL handle");
  d->train = on != 0;
  d->ready = false;
  d->train_ready = false;
  return DMEL_OK;
}

namespace {
// Flattened image layout.  An activation (B, C, H, W) is stored as B items of C rows of (H + 2) * P floats: image row h sits at
// [(h + 1) * P, (h + 1) * P + W), the two pad rows and the P - W gap columns of every row are zero.  With that, a kernel row dh of the
// 3 x kw convolution is a 1-D convolution over the flat axis of the same buffer displaced by (dh - 1) * P floats (horizontal taps that
// leave the image land in the zero gap, vertical ones in a zero pad row), so every launch sees 32 long items instead of B * (H + 2)
// rows of 12-94 frames and the column tiles are full.  A stride-2 layer halves the pitch: flat 2 n + dw - 4 <-> (h, 2 w + dw - 4).
// Outputs are computed on the whole flat axis; the SiLU pass that follows rewrites pads and gaps with zeros.
struct DPlan {
  float* act[kDiscLayers + 1];        // act[0]: embedded input image; act[i+1]: output of layer i (after SiLU for i < 5)
  float* pre[kDiscLayers];            // pre-activation of layer i (training keeps it; inference aliases act[i+1])
  int64_t W[kDiscLayers + 1], P[kDiscLayers + 1];     // valid width and row pitch of act[i]
  int Hp;
  int64_t guard;                      // floats in front of / behind every buffer that displaced reads may touch
  size_t bytes;
};
int64_t disc_pitch0(const dmel_discriminator* d, int64_t W) {
  for (int64_t P0 = (W + 4 + 15) / 16 * 16;; P0 += 16) {      // multiples of 16: the pitch stays even after three halvings
    int64_t w = W, P = P0;
    bool ok = true;
    for (int i = 0; i < kDiscLayers && ok; ++i) {
      const DLayer& l = d->layer[i];
      ok = P - w >= l.pw;                                   // forward taps of layer i over act[i]
      w = disc_out_w(l, w);
      if (l.sw == 2) P /= 2;
      ok = ok && P - w >= (l.sw == 2 ? 2 : l.pw);           // backward-data taps of layer i over d act[i+1]
    }
    if (ok) return P0;
  }
}
DPlan disc_plan(const dmel_discriminator* d, int B, int H, int64_t W, bool train, void* ws) {
  DPlan p{};
  Arena a(ws, (size_t)-1);
  p.Hp = H + 2;
  p.W[0] = W; p.P[0] = disc_pitch0(d, W);
  for (int i = 0; i < kDiscLayers; ++i) {
    p.W[i + 1] = disc_out_w(d->layer[i], p.W[i]);
    p.P[i + 1] = p.P[i] / d->layer[i].sw;
  }
  p.guard = p.P[0] + 64;
  auto buf = [&](int C, int64_t P) { return a.take<float>((size_t)B * C * p.Hp * P + 2 * p.guard) + p.guard; };
  p.act[0] = buf(1, p.P[0]);
  for (int i = 0; i < kDiscLayers; ++i) {
    p.act[i + 1] = buf(d->layer[i].Cout, p.P[i + 1]);
    p.pre[i] = (train && i < kDiscLayers - 1) ? buf(d->layer[i].Cout, p.P[i + 1]) : p.act[i + 1];
  }
  p.bytes = align_up(a.off, 256);
  return p;
}
__device__ __forceinline__ bool disc_valid(int col, int Hp, int P, int Wv) {
  const int h = col / P, w = col - h * P;
  return h >= 1 && h < Hp - 1 && w < Wv;
}
// grid (rows = B * C, ceil(Hp * P / 256)); y may alias u
__global__ __launch_bounds__(256) void disc_silu_fwd_kernel(const float* u, float* y, int Hp, int P, int Wv) {
  const int col = blockIdx.y * 256 + threadIdx.x, len = Hp * P;
  if (col >= len) return;
  const int64_t i = (int64_t)blockIdx.x * len + col;
  const float v = u[i];
  y[i] = disc_valid(col, Hp, P, Wv) ? v / (1.f + __expf(-v)) : 0.f;
}
// g <- g * silu'(u) on the image, 0 on pads and gaps (in place)
__global__ __launch_bounds__(256) void disc_silu_bwd_kernel(float* g, const float* __restrict__ u, int Hp, int P, int Wv) {
  const int col = blockIdx.y * 256 + threadIdx.x, len = Hp * P;
  if (col >= len) return;
  const int64_t i = (int64_t)blockIdx.x * len + col;
  const float v = u[i], sg = 1.f / (1.f + __expf(-v));
  g[i] = disc_valid(col, Hp, P, Wv) ? g[i] * sg * (1.f + v * (1.f - sg)) : 0.f;
}
// compact (B, H, Wv) -> flat (B, 1, Hp, P), zeros on pads and gaps; grid (B, ceil(Hp * P / 256))
__global__ __launch_bounds__(256) void disc_embed_kernel(const float* __restrict__ src, float* __restrict__ dst, int Hp, int P, int Wv) {
  const int col = blockIdx.y * 256 + threadIdx.x, len = Hp * P;
  if (col >= len) return;
  const int h = col / P, w = col - h * P;
  const bool ok = h >= 1 && h < Hp - 1 && w < Wv;
  dst[(int64_t)blockIdx.x * len + col] = ok ? src[((int64_t)blockIdx.x * (Hp - 2) + (h - 1)) * Wv + w] : 0.f;
}
// flat (B, 1, Hp, P) -> compact (B, H, Wv); grid (B, ceil(H * Wv / 256))
__global__ __launch_bounds__(256) void disc_extract_kernel(const float* __restrict__ src, float* __restrict__ dst, int Hp, int P, int Wv) {
  const int j = blockIdx.y * 256 + threadIdx.x, n = (Hp - 2) * Wv;
  if (j >= n) return;
  const int h = j / Wv, w = j - h * Wv;
  dst[(int64_t)blockIdx.x * n + j] = src[(int64_t)blockIdx.x * Hp * P + (int64_t)(h + 1) * P + w];
}
inline dim3 disc_grid(int64_t rows, int64_t len) { return dim3((unsigned)rows, (unsigned)((len + 255) / 256)); }

// layer i forward into `out` (pre-activation): three launches, centre row first
int disc_layer_forward(const dmel_discriminator* d, const DPlan& p, int i, int B, const float* in, float* out, hipStream_t st) {
  const DLayer& l = d->layer[i];
  const int64_t Tin = p.Hp * p.P[i], Tout = p.Hp * p.P[i + 1];
  const int order[3] = {1, 0, 2};
  for (int k = 0; k < 3; ++k) {
    const int dh = order[k];
    ConvRun r;
    const int nseg = l.sw == 1 ? 1 : 2;
    for (int sg = 0; sg < nseg; ++sg) {
      r.seg[sg].x = in + (int64_t)(dh - 1) * p.P[i]; r.seg[sg].bstride = l.Cin * Tin; r.seg[sg].cstride = Tin; r.seg[sg].Tin = Tin;
    }
    r.B = B; r.Tcols = Tout; r.y = out; r.y_bs = l.Cout * Tout; r.y_cs = Tout; r.Tout = Tout;
    r.accumulate = k > 0;
    // forward activations (log-mel images, SiLU outputs) sit in the fp16 split's range; the backward pass keeps the six-product split
    r.precision = DMEL_PRECISION_FP32_F16X2;
    DMEL_TRY(launch_conv(l.fwd[dh], r, st));
  }
  return DMEL_OK;
}
int disc_forward_common(const dmel_discriminator* d, const DPlan& p, const float* x, int B, int H, int64_t W, hipStream_t st) {
  // displaced reads run up to one row pitch past either end of an activation buffer: those guards must hold finite values
  auto zero_guards = [&](float* base, int C, int64_t P) -> hipError_t {
    hipError_t e = hipMemsetAsync(base - p.guard, 0, (size_t)p.guard * sizeof(float), st);
    if (e != hipSuccess) return e;
    return hipMemsetAsync(base + (size_t)B * C * p.Hp * P, 0, (size_t)p.guard * sizeof(float), st);
  };
  DMEL_HIP(zero_guards(p.act[0], 1, p.P[0]));
  hipLaunchKernelGGL(disc_embed_kernel, disc_grid(B, p.Hp * p.P[0]), dim3(256), 0, st, x, p.act[0], p.Hp, (int)p.P[0], (int)W);
  DMEL_HIP(hipGetLastError());
  for (int i = 0; i < kDiscLayers; ++i) {
    const DLayer& l = d->layer[i];
    DMEL_TRY(disc_layer_forward(d, p, i, B, p.act[i], p.pre[i], st));
    if (i < kDiscLayers - 1) {
      DMEL_HIP(zero_guards(p.act[i + 1], l.Cout, p.P[i + 1]));
      hipLaunchKernelGGL(disc_silu_fwd_kernel, disc_grid((int64_t)B * l.Cout, p.Hp * p.P[i + 1]), dim3(256), 0, st, p.pre[i], p.act[i + 1], p.Hp,
                         (int)p.P[i + 1], (int)p.W[i + 1]);
      DMEL_HIP(hipGetLastError());
    }
  }
  return DMEL_OK;
}
int disc_extract(const float* flat, float* compact, int B, int Hp, int64_t P, int64_t Wv, hipStream_t st) {
  hipLaunchKernelGGL(disc_extract_kernel, disc_grid(B, (int64_t)(Hp - 2) * Wv), dim3(256), 0, st, flat, compact, Hp, (int)P, (int)Wv);
  DMEL_HIP(hipGetLastError());
  return DMEL_OK;
}
bool disc_shape_ok(const dmel_discriminator* d, int B, int H, int64_t W) {
  if (B <= 0 || H <= 0 || W <= 0 || (int64_t)B * 1024 >= ((int64_t)1 << 31)) return false;
  return (int64_t)(H + 2) * disc_pitch0(d, W) < ((int64_t)1 << 23);          // flat rows stay far inside the kernels' 32-bit offsets
}
}  // namespace

extern "C" int64_t dmel_discriminator_out_frames(const dmel_discriminator* d, int64_t W) {
  if (!d || W <= 0) return 0;
  for (int i = 0; i < kDiscLayers; ++i) W = disc_out_w(d->layer[i], W);
  return W;
}
extern "C" size_t dmel_discriminator_workspace_bytes(const dmel_discriminator* d, int B, int H, int64_t W) {
  if (!d || B <= 0 || H <= 0 || W <= 0) return 0;
  return disc_plan(d, B, H, W, false, nullptr).bytes;
}
// x (B, H, W) -> logits (B, H, W_out)           discriminator.py:34-35
extern "C" int dmel_discriminator_forward(const dmel_discriminator* d, const float* x, float* y, int B, int H, int64_t W, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  DMEL_CHECK_ARG(d && x && y && workspace, "discriminator_forward: NULL argument");
  if (!d->ready) { set_error("discriminator_forward: handle not finalized"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG(disc_shape_ok(d, B, H, W), "discriminator_forward: bad shape");
  const DPlan p = disc_plan(d, B, H, W, false, workspace);
  DMEL_CHECK_ARG(workspace_bytes >= p.bytes, "discriminator_forward: workspace too small (%zu < %zu)", workspace_bytes, p.bytes);
  hipStream_t st = (hipStream_t)stream;
  DMEL_TRY(disc_forward_common(d, p, x, B, H, W, st));
  return disc_extract(p.act[kDiscLayers], y, B, p.Hp, p.P[kDiscLayers], p.W[kDiscLayers], st);
}

// ---- discriminator training path -------------------------------------------------------------------------------------------
namespace {
// per output channel: d g = <dW, v> / |v|;  d v = g / |v| * (dW - v <dW, v> / |v|^2)        (torch._weight_norm, dim 0)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                              const float* __restrict__ g, float* __restrict__ dg, float* __restrict__ dv,
                                                              int64_t inner) {
  __shared__ float part[2][4];
  const int co = blockIdx.x;
  const float* dwr = dw + (int64_t)co * inner;
  const float* vr = v + (int64_t)co * inner;
  float dot = 0.f, ss = 0.f;
  for (int64_t i = threadIdx.x; i < inner; i += 256) {
    dot = fmaf(dwr[i], vr[i], dot);
    ss = fmaf(vr[i], vr[i], ss);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { dot += __shfl_xor(dot, o, 64); ss += __shfl_xor(ss, o, 64); }
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = dot; part[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  dot = part[0][0] + part[0][1] + part[0][2] + part[0][3];
  ss = part[1][0] + part[1][1] + part[1][2] + part[1][3];
  const float nrm = sqrtf(ss), gg = g[co];
  if (threadIdx.x == 0) dg[co] = dot / nrm;
  const float k1 = gg / nrm, k2 = dot / ss;
  for (int64_t i = threadIdx.x; i < inner; i += 256) dv[(int64_t)co * inner + i] = k1 * (dwr[i] - vr[i] * k2);
}
struct DTrainPlan {
  DPlan f;
  float *ga, *gb, *dwfold;            // two gradient buffers in the flat layout (ping-pong down the layers), folded-weight gradient
  size_t bytes;
};
DTrainPlan disc_train_plan(const dmel_discriminator* d, int B, int H, int64_t W, void* ws) {
  DTrainPlan t{};
  t.f = disc_plan(d, B, H, W, true, ws);
  Arena a(ws, (size_t)-1);
  a.off = t.f.bytes;
  size_t mx = (size_t)B * t.f.Hp * t.f.P[0], mw = 0;
  for (int i = 0; i < kDiscLayers; ++i) {
    mx = std::max(mx, (size_t)B * d->layer[i].Cout * t.f.Hp * t.f.P[i + 1]);
    mw = std::max(mw, (size_t)d->layer[i].Cout * d->layer[i].Cin * 3 * d->layer[i].kw);
  }
  t.ga = a.take<float>(mx + 2 * t.f.guard) + t.f.guard;
  t.gb = a.take<float>(mx + 2 * t.f.guard) + t.f.guard;
  t.dwfold = a.take<float>(mw);
  t.bytes = align_up(a.off, 256);
  return t;
}
}  // namespace

namespace {
// folded weight on the device: W[co, :] = g[co] * v[co, :] / |v[co, :]|      (torch._weight_norm, dim 0)
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ w,
                                                              int64_t inner) {
  __shared__ float part[4];
  const int co = blockIdx.x;
  const float* vr = v + (int64_t)co * inner;
  float ss = 0.f;
  for (int64_t i = threadIdx.x; i < inner; i += 256) ss = fmaf(vr[i], vr[i], ss);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float scale = g[co] / sqrtf(part[0] + part[1] + part[2] + part[3]);
  for (int64_t i = threadIdx.x; i < inner; i += 256) w[(int64_t)co * inner + i] = vr[i] * scale;
}
}  // namespace

// Re-pack every weight image from the device-resident bias / weight-norm g / v (after an optimiser step); `scratch` holds the folded
// weight of the largest layer (1024 * 512 * 9 floats).  (The host-side fold at finalize sums |v|^2 in double, this kernel in float:
// images agree to rounding, not bitwise.)
extern "C" int dmel_discriminator_refresh(dmel_discriminator* d, int n, const char* const* keys, const float* const* device_tensors,
                                          void* stream) {
  DMEL_CHECK_ARG(d && keys && device_tensors && n > 0, "discriminator_refresh: bad argument");
  if (!d->ready) { set_error("discriminator_refresh: handle not finalized"); return DMEL_EMISSING; }
  std::map<std::string, const float*> dev;
  for (int i = 0; i < n; ++i) {
    DMEL_CHECK_ARG(keys[i] && device_tensors[i], "discriminator_refresh: NULL entry %d", i);
    dev[keys[i]] = device_tensors[i];
  }
  hipStream_t st = (hipStream_t)stream;
  DevBuf& fold = *thread_scratch(2, st);
  const size_t need = (size_t)1024 * 512 * 9 * sizeof(float);
  if (fold.bytes < need) {
    fold.release();
    DMEL_HIP(hipMalloc(&fold.p, need));
    fold.bytes = need;
  }
  for (int i = 0; i < kDiscLayers; ++i) {
    DLayer& l = d->layer[i];
    const std::string p = "blocks." + std::to_string(2 * i) + ".";
    auto get = [&](const std::string& k) -> const float* {
      auto it = dev.find(k);
      return it == dev.end() ? nullptr : it->second;
    };
    const float *b = get(p + "bias"), *g = get(p + "parametrizations.weight.original0"), *v = get(p + "parametrizations.weight.original1");
    if (!b || !g || !v) { set_error("discriminator_refresh: tensors of '%s' were not provided", p.c_str()); return DMEL_EMISSING; }
    const int64_t inner = (int64_t)l.Cin * 3 * l.kw;
    float* w = fold.as<float>();
    hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3((unsigned)l.Cout), dim3(256), 0, st, v, g, w, inner);
    DMEL_HIP(hipGetLastError());
    if (l.g_dev.p) {
      DMEL_HIP(hipMemcpyAsync(l.g_dev.p, g, (size_t)l.Cout * sizeof(float), hipMemcpyDeviceToDevice, st));
      DMEL_HIP(hipMemcpyAsync(l.v_dev.p, v, (size_t)l.Cout * inner * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    RepackBatch batch(st);      // the 3-9 images of this layer in one launch
    for (int dh = 0; dh < 3; ++dh) {
      {
        RepackSrc src;
        const int nseg = l.sw == 1 ? 1 : 2;
        for (int sg = 0; sg < nseg; ++sg) {           // value(row = co, ci, tap) = W[co][ci][dh][sw * tap + sg]
          src.seg[sg].w = w + dh * l.kw + sg; src.seg[sg].rs = inner; src.seg[sg].cs = 3 * l.kw; src.seg[sg].ts = l.sw;
        }
        src.b0 = dh == 1 ? b : nullptr;
        DMEL_TRY(launch_repack(l.fwd[dh], src, st));
      }
      if (d->train_ready) {
        const int nph = l.sw == 1 ? 1 : 2;
        for (int ph = 0; ph < nph; ++ph) {             // value(row = ci, cc = co, tap) = W[co][ci][dh][kw-1-tap | 8 - 2 tap - ph]
          RepackSrc src;
          src.seg[0].rs = 3 * l.kw; src.seg[0].cs = inner;
          if (l.sw == 1) { src.seg[0].w = w + dh * l.kw; src.seg[0].ts = 1; src.seg[0].rev = 1; }
          else { src.seg[0].w = w + dh * l.kw + 8 - ph; src.seg[0].ts = -2; }
          DMEL_TRY(launch_repack(l.bwd[dh][ph], src, st));
        }
      }
    }
    DMEL_TRY(batch.flush());
    // the folded weight buffer is reused by the next layer: the launches above are ordered on the same stream
  }
  return DMEL_OK;
}

extern "C" size_t dmel_discriminator_train_workspace_bytes(const dmel_discriminator* d, int B, int H, int64_t W) {
  if (!d || B <= 0 || H <= 0 || W <= 0) return 0;
  return disc_train_plan(d, B, H, W, nullptr).bytes;
}
extern "C" int64_t dmel_discriminator_grad_floats(const dmel_discriminator* d) { return d && d->train_ready ? d->grad_floats : 0; }
extern "C" int dmel_discriminator_grad_slot(const dmel_discriminator* d, const char* key, int64_t* offset, int64_t* numel) {
  DMEL_CHECK_ARG(d && key && offset && numel, "discriminator_grad_slot: NULL argument");
  if (!d->train_ready) { set_error("discriminator_grad_slot: training was not enabled before finalize"); return DMEL_EMISSING; }
  for (const auto& sl : d->slots)
    if (sl.key == key) { *offset = sl.offset; *numel = sl.numel; return DMEL_OK; }
  set_error("discriminator_grad_slot: '%s' is not a parameter of the discriminator", key);
  return DMEL_EINVAL;
}

extern "C" int dmel_discriminator_forward_train(const dmel_discriminator* d, const float* x, float* y, int B, int H, int64_t W,
                                                void* workspace, size_t workspace_bytes, void* stream) {
  TrainPrecisionScope train_scope(d ? d->train_precision : 0);
  DMEL_CHECK_ARG(d && x && y && workspace, "discriminator_forward_train: NULL argument");
  if (!d->ready || !d->train_ready) { set_error("discriminator_forward_train: enable_training + finalize first"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG(disc_shape_ok(d, B, H, W), "discriminator_forward_train: bad shape");
  const DTrainPlan t = disc_train_plan(d, B, H, W, workspace);
  DMEL_CHECK_ARG(workspace_bytes >= t.bytes, "discriminator_forward_train: workspace too small (%zu < %zu)", workspace_bytes, t.bytes);
  hipStream_t st = (hipStream_t)stream;
  DMEL_TRY(disc_forward_common(d, t.f, x, B, H, W, st));
  return disc_extract(t.f.act[kDiscLayers], y, B, t.f.Hp, t.f.P[kDiscLayers], t.f.W[kDiscLayers], st);
}

// dy (B, H, W_out) -> dx (B, H, W) (nullable) and the parameter gradients (bias, weight-norm g and v of every layer) in `grads`
extern "C" int dmel_discriminator_backward(const dmel_discriminator* d, const float* dy, float* dx, float* grads, int B, int H, int64_t W,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  TrainPrecisionScope train_scope(d ? d->train_precision : 0);
  DMEL_CHECK_ARG(d && dy && grads && workspace, "discriminator_backward: NULL argument");
  if (!d->ready || !d->train_ready) { set_error("discriminator_backward: enable_training + finalize first"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG(disc_shape_ok(d, B, H, W), "discriminator_backward: bad shape");
  const DTrainPlan t = disc_train_plan(d, B, H, W, workspace);
  DMEL_CHECK_ARG(workspace_bytes >= t.bytes, "discriminator_backward: workspace too small (%zu < %zu)", workspace_bytes, t.bytes);
  hipStream_t st = (hipStream_t)stream;
  const DPlan& p = t.f;
  const int Hp = p.Hp;
  // gradients live in the flat layout of the activation they belong to.  Their guards are only ever multiplied into pad / gap outputs,
  // which the SiLU backward pass rewrites with zeros, but they are cleared once so that every read is of defined memory.
  DMEL_HIP(hipMemsetAsync(t.ga - p.guard, 0, (size_t)p.guard * sizeof(float), st));
  DMEL_HIP(hipMemsetAsync(t.gb - p.guard, 0, (size_t)p.guard * sizeof(float), st));
  float* g = t.ga;
  hipLaunchKernelGGL(disc_embed_kernel, disc_grid(B, Hp * p.P[kDiscLayers]), dim3(256), 0, st, dy, g, Hp, (int)p.P[kDiscLayers],
                     (int)p.W[kDiscLayers]);
  DMEL_HIP(hipGetLastError());
  bool g_is_a = true;
  for (int i = kDiscLayers - 1; i >= 0; --i) {
    const DLayer& l = d->layer[i];
    const std::string pk = "blocks." + std::to_string(2 * i) + ".";
    const int64_t Tin = Hp * p.P[i], Tout = Hp * p.P[i + 1];
    // g: d act[i+1] (B, Cout, Hp * P).  Through the SiLU (not after the last layer), in place, zero on pads and gaps.
    if (i < kDiscLayers - 1) {
      hipLaunchKernelGGL(disc_silu_bwd_kernel, disc_grid((int64_t)B * l.Cout, Tout), dim3(256), 0, st, g, p.pre[i], Hp, (int)p.P[i + 1],
                         (int)p.W[i + 1]);
      DMEL_HIP(hipGetLastError());
    }
    // bias and weight gradients: one launch per kernel row, the kw horizontal taps on the grid.  The fp16-split kernels that read g (the
    // weight gradients and, below, backward-data) scale it by its max, found once per layer.
    DMEL_TRY(launch_conv_bgrad(g, grads + d->slot_of(pk + "bias"), l.Cout, B, Tout, st));
    DMEL_HIP(hipMemsetAsync(t.dwfold, 0, (size_t)l.Cout * l.Cin * 3 * l.kw * sizeof(float), st));
    const uint32_t* g_absmax = nullptr;
    DMEL_TRY(launch_absmax(g, (int64_t)B * l.Cout * Tout, st, &g_absmax));
    for (int dh = 0; dh < 3; ++dh)
      DMEL_TRY(launch_conv_wgrad_strided(g, p.act[i] + (int64_t)(dh - 1) * p.P[i], t.dwfold, l.Cout, l.Cin, l.sw, -l.pw, Tout, Tin, 3 * l.kw,
                                         dh * l.kw, B, st, l.kw, g_absmax));
    hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((unsigned)l.Cout), dim3(256), 0, st, t.dwfold, l.v_dev.as<float>(), l.g_dev.as<float>(),
                       grads + d->slot_of(pk + "parametrizations.weight.original0"),
                       grads + d->slot_of(pk + "parametrizations.weight.original1"), (int64_t)l.Cin * 3 * l.kw);
    DMEL_HIP(hipGetLastError());
    if (i == 0 && !dx) break;
    // backward-data into the other buffer (B, Cin, Hp * P[i])
    float* dst = g_is_a ? t.gb : t.ga;
    const int order[3] = {1, 0, 2};
    for (int k = 0; k < 3; ++k) {
      const int dh = order[k];
      const int nph = l.sw == 1 ? 1 : 2;
      for (int ph = 0; ph < nph; ++ph) {
        ConvRun r = run_1seg(g - (int64_t)(dh - 1) * p.P[i + 1], l.Cout, Tout, dst, l.Cin, Tin, B);
        r.accumulate = k > 0;
        static const bool dgrad_f16 = [] { const char* e = getenv("DMEL_DGRAD_F16X2"); return !(e && e[0] == '0'); }();
        if (g_absmax && dgrad_f16) { r.precision = DMEL_PRECISION_FP32_F16X2; r.seg[0].in_absmax = g_absmax; }
        if (l.sw == 2) {
          r.out_tstride = 2; r.phase_base = ph; r.Tcols = Tin / 2; r.Tout = Tin;
        }
        DMEL_TRY(launch_conv(l.bwd[dh][ph], r, st));
      }
    }
    g = dst;
    g_is_a = !g_is_a;
  }
  if (dx) DMEL_TRY(disc_extract(g, dx, B, Hp, p.P[0], W, st));      // layer 0 has no SiLU in front: its pads / gaps are simply not read
  return DMEL_OK;
}

// =====================================================================================================
// BigVGAN                                          models/modules/bigvgan/bigvgan.py:244-407
// =====================================================================================================
namespace {
struct SnakeP {
  DevBuf alpha, beta;
};
struct AmpBlock {             // AMPBlock1, bigvgan.py:31-147
  int k;
  int dil[3];
  PackedConv c1[3], c2[3];
  SnakeP act[6];
};
struct UpStage {
  int u, k, Cin, Cout;
  PackedConv lo, hi;          // phases [0,u/2) with taps d={-1,0}; phases [u/2,u) with taps d={0,+1}
};
// ConvTranspose1d(Cin, Cout, k = 2u, stride u, padding u/2) as two groups of u/2 phase sub-convolutions with two taps each:
//   y[co, u q + ph] = sum_ci sum_d W[ci, co, ph + u/2 - u d] x[ci, q + d]                    (bigvgan.py:320-334, :371-374)
// w: (Cin, Cout, k) host floats, bias: Cout host floats.  us.u / k / Cin / Cout are set by the caller.
int pack_up_stage(UpStage& us, const float* w, const float* bias) {
  const int u = us.u, k = us.k, hu = u / 2, Co = us.Cout;
  for (int half = 0; half < 2; ++half) {
    PackDesc d;
    d.mode = EPI_LINEAR; d.nseg = 1; d.C = Co; d.phases = hu;
    d.seg[0].Cin = us.Cin; d.seg[0].taps = 2; d.seg[0].dil = 1; d.seg[0].pad_left = half == 0 ? 1 : 0;
    DMEL_TRY(pack_conv(half == 0 ? us.lo : us.hi, d,
                       [&](int, int row, int ci, int tap) {
                         const int ph = row / Co + half * hu, co = row % Co;
                         const int dd = tap - (half == 0 ? 1 : 0);
                         const int kk = ph + hu - u * dd;
                         return (kk >= 0 && kk < k) ? w[((size_t)ci * Co + co) * k + kk] : 0.f;
                       },
                       [&](int row) { return bias ? bias[row % Co] : 0.f; }));
  }
  return DMEL_OK;
}
// x (B, Cin, T) -> y (B, Cout, u T)
int launch_up_stage(const UpStage& us, const float* x, float* y, int B, int64_t T, int precision, hipStream_t st) {
  for (int half = 0; half < 2; ++half) {
    ConvRun r = run_1seg(x, us.Cin, T, y, us.Cout, T * us.u, B);
    r.Tcols = T; r.out_tstride = us.u; r.phase_base = half * (us.u / 2); r.Tout = T * us.u;
    r.precision = precision;
    DMEL_TRY(launch_conv(half == 0 ? us.lo : us.hi, r, st));
  }
  return DMEL_OK;
}
}  // namespace

// ---- standalone transposed convolution / output convolution (C-ABI rows convT1d, conv_post of SURVEY section 8(b)) ---------------
struct dmel_conv_transpose {
  UpStage us;
  int precision = 0;
};
extern "C" int dmel_conv_transpose1d_create(dmel_conv_transpose** out, const float* w_host, const float* bias_host, int Cin, int Cout, int k,
                                            int stride) {
  DMEL_CHECK_ARG(out && w_host, "conv_transpose1d_create: NULL argument");
  DMEL_CHECK_ARG(Cin > 0 && Cout > 0 && stride >= 2 && (stride % 2) == 0, "conv_transpose1d_create: bad shape (even stride >= 2 required)");
  if (k != 2 * stride) {
    set_error("conv_transpose1d: only k == 2 * stride with padding stride / 2 (every BigVGAN up-sampler) is built, got k %d stride %d", k, stride);
    return DMEL_EUNSUPPORTED;
  }
  auto* h = new dmel_conv_transpose();
  h->us.u = stride; h->us.k = k; h->us.Cin = Cin; h->us.Cout = Cout;
  const int rc = pack_up_stage(h->us, w_host, bias_host);
  if (rc != DMEL_OK) { delete h; return rc; }
  *out = h;
  return DMEL_OK;
}
extern "C" void dmel_conv_transpose1d_destroy(dmel_conv_transpose* h) { delete h; }
extern "C" int dmel_conv_transpose1d_set_precision(dmel_conv_transpose* h, int precision) {
  DMEL_CHECK_ARG(h && valid_precision(precision), "conv_transpose1d_set_precision: not a DMEL_PRECISION_* value");
  h->precision = precision;
  return DMEL_OK;
}
extern "C" int dmel_conv_transpose1d_forward(const dmel_conv_transpose* h, const float* x, float* y, int B, int64_t T, void* stream) {
  DMEL_CHECK_ARG(h && x && y, "conv_transpose1d_forward: NULL argument");
  DMEL_CHECK_ARG(B > 0 && T > 0, "conv_transpose1d_forward: bad shape");
  return launch_up_stage(h->us, x, y, B, T, h->precision, (hipStream_t)stream);
}
extern "C" int dmel_conv_post_f32(const float* x, const float* w_dev, float bias, int act, float* y, int B, int C, int K, int64_t T, void* stream) {
  DMEL_CHECK_ARG(x && w_dev && y, "conv_post: NULL argument");
  DMEL_CHECK_ARG(act == 0 || act == 2 || act == 3, "conv_post: act must be 0 (none), 2 (tanh) or 3 (clamp to [-1, 1])");
  return launch_conv_post(x, y, w_dev, bias, act == 0 ? ACT_NONE : act == 2 ? ACT_TANH : ACT_CLAMP1, B, C, K, T, (hipStream_t)stream);
}

struct dmel_bigvgan {
  dmel_bigvgan_config cfg;
  TensorStore ts;
  bool ready = false;
  PackedConv conv_pre;
  DevBuf post_w;            // conv_post weight (1, C, 7) on the device: a single output row runs as a reduction kernel
  float post_bias = 0.f;
  int post_c = 0;
  std::vector<UpStage> ups;
  std::vector<AmpBlock> blocks;
  SnakeP act_post;
  float taps_up[12], taps_dn[12];   // UpSample1d.filter / DownSample1d.lowpass.filter (the reference registers the same 12 taps for both)
  int64_t total_up = 1;
  // The AMP blocks of a stage are independent until their outputs are averaged: they run on the caller's stream plus
  // up to two library-owned side streams, forked and joined with events around every stage, so the VALU-bound
  // anti-alias activations of one block overlap the MFMA-bound convolutions of another.  Semantics on the caller's
  // stream are unchanged (everything is ordered behind what was on it and finished when the final join is reached).
  static constexpr int kSide = 2;
  hipStream_t side[kSide] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[kSide] = {nullptr, nullptr}, ev_chain[3] = {nullptr, nullptr, nullptr},
             ev_stag[3] = {nullptr, nullptr, nullptr};
  bool multi = false;
  int precision = 0;
  ~dmel_bigvgan() {
    for (int i = 0; i < kSide; ++i) {
      if (side[i]) (void)hipStreamDestroy(side[i]);
      if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
    }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (auto& e : ev_chain) if (e) (void)hipEventDestroy(e);
    for (auto& e : ev_stag) if (e) (void)hipEventDestroy(e);
  }
};

extern "C" int dmel_bigvgan_create(dmel_bigvgan** out, const dmel_bigvgan_config* cfg) {
  DMEL_CHECK_ARG(out && cfg, "NULL argument");
  DMEL_CHECK_ARG(cfg->num_upsamples >= 1 && cfg->num_upsamples <= 8 && cfg->num_kernels >= 1 && cfg->num_kernels <= 8,
                 "bigvgan: bad stage/kernel count");
  DMEL_CHECK_ARG(cfg->num_mels > 0 && cfg->upsample_initial_channel > 0, "bigvgan: bad channel counts");
  DMEL_CHECK_ARG(cfg->resblock_type >= 0 && cfg->resblock_type <= 2, "bigvgan: resblock_type must be 1 (AMPBlock1) or 2 (AMPBlock2)");
  int64_t up = 1;
  for (int i = 0; i < cfg->num_upsamples; ++i) {
    const int u = cfg->upsample_rates[i], k = cfg->upsample_kernel_sizes[i];
    if (k != 2 * u || (u % 2) != 0) {
      set_error("bigvgan: upsample stage %d (rate %d, kernel %d) unsupported: kernel must be 2*rate with an even rate", i, u, k);
      return DMEL_EUNSUPPORTED;
    }
    DMEL_CHECK_ARG((cfg->upsample_initial_channel >> (i + 1)) > 0, "bigvgan: channel count underflows at stage %d", i);
    up *= u;
  }
  for (int j = 0; j < cfg->num_kernels; ++j) {
    DMEL_CHECK_ARG(cfg->resblock_kernel_sizes[j] % 2 == 1, "bigvgan: even resblock kernel");
    for (int l = 0; l < 3; ++l)
      if ((cfg->resblock_kernel_sizes[j] - 1) * cfg->resblock_dilations[j][l] > 64) {
        set_error("bigvgan: resblock kernel %d dilation %d exceeds the 64-sample halo", cfg->resblock_kernel_sizes[j],
                  cfg->resblock_dilations[j][l]);
        return DMEL_EUNSUPPORTED;
      }
  }
  auto* m = new dmel_bigvgan();
  m->cfg = *cfg;
  m->total_up = up;
  // kaiser_sinc_filter1d(0.25, 0.3, 12) as the reference computes it in fp32 (filter.py:30-62); overridden by any
  // "*.filter" buffer found in the state dict.
  static const float kTaps[12] = {0.0020289647f, 0.0093894657f, -0.0255434588f, -0.0576573834f, 0.1285725832f, 0.4432097971f,
                                  0.4432097971f, 0.1285725832f, -0.0576573834f, -0.0255434588f, 0.0093894657f, 0.0020289647f};
  std::memcpy(m->taps_up, kTaps, sizeof(kTaps));
  std::memcpy(m->taps_dn, kTaps, sizeof(kTaps));
  *out = m;
  return DMEL_OK;
}
extern "C" void dmel_bigvgan_destroy(dmel_bigvgan* m) { delete m; }
extern "C" int dmel_bigvgan_set_precision(dmel_bigvgan* m, int precision) {
  DMEL_CHECK_ARG(m && valid_precision(precision), "bigvgan_set_precision: not a DMEL_PRECISION_* value");
  m->precision = precision;
  return DMEL_OK;
}
extern "C" int dmel_bigvgan_set_tensor(dmel_bigvgan* m, const char* key, const float* data, const int64_t* shape, int ndim) {
  DMEL_CHECK_ARG(m, "NULL handle");
  m->ready = false;
  return m->ts.set(key, data, shape, ndim);
}

static int pack_same_conv(PackedConv& pc, const TensorStore& ts, const std::string& prefix, int Cout, int Cin, int k, int dil,
                          bool has_bias) {
  std::vector<float> w;
  if (!ts.conv_weight(prefix, {Cout, Cin, k}, w)) return DMEL_EMISSING;
  const HostTensor* b = nullptr;
  if (has_bias) {
    b = ts.need(prefix + "bias", {Cout});
    if (!b) return DMEL_EMISSING;
  }
  PackDesc d;
  d.mode = EPI_LINEAR; d.nseg = 1; d.C = Cout;
  d.seg[0].Cin = Cin; d.seg[0].taps = k; d.seg[0].dil = dil; d.seg[0].pad_left = dil * (k - 1) / 2;  // get_padding, utils.py:57-58
  return pack_conv(pc, d, [&](int, int row, int ci, int tap) { return w[((size_t)row * Cin + ci) * k + tap]; },
                   [&](int row) { return b ? b->v[row] : 0.f; });
}

static int load_snake(SnakeP& s, const TensorStore& ts, const std::string& prefix, int C, bool snake) {
  const HostTensor* a = ts.need(prefix + "act.alpha", {C});
  if (!a) return DMEL_EMISSING;
  DMEL_TRY(upload_vec(s.alpha, a->v));
  if (!snake) {
    const HostTensor* b = ts.need(prefix + "act.beta", {C});
    if (!b) return DMEL_EMISSING;
    DMEL_TRY(upload_vec(s.beta, b->v));
  }
  return DMEL_OK;
}

extern "C" int dmel_bigvgan_finalize(dmel_bigvgan* m) {
  DMEL_CHECK_ARG(m, "NULL handle");
  const dmel_bigvgan_config& c = m->cfg;
  const bool snake = c.activation_snake != 0;
  // anti-alias taps: every Activation1d registers two buffers ("...upsample.filter", "...downsample.lowpass.filter"); all up filters
  // must agree with each other and all down filters with each other (one launch-constant pair per model); up and down may differ
  bool have_up = false, have_dn = false;
  for (auto& kv : m->ts.t) {
    const std::string& k = kv.first;
    const bool is_f = k.size() > 7 && k.compare(k.size() - 7, 7, ".filter") == 0;
    if (!is_f) continue;
    if (kv.second.numel() != 12) { set_error("bigvgan: filter '%s' is not 12 taps", k.c_str()); return DMEL_EUNSUPPORTED; }
    const bool is_dn = k.find("downsample") != std::string::npos;
    float* dst = is_dn ? m->taps_dn : m->taps_up;
    bool& have = is_dn ? have_dn : have_up;
    if (!have) { std::memcpy(dst, kv.second.v.data(), 12 * sizeof(float)); have = true; }
    else if (std::memcmp(dst, kv.second.v.data(), 12 * sizeof(float)) != 0) {
      set_error("bigvgan: anti-alias filter '%s' differs from the other %s filters (per-activation filters unsupported)", k.c_str(),
                is_dn ? "down-sampling" : "up-sampling");
      return DMEL_EUNSUPPORTED;
    }
  }
  const int C0 = c.upsample_initial_channel;
  DMEL_TRY(pack_same_conv(m->conv_pre, m->ts, "conv_pre.", C0, c.num_mels, 7, 1, true));
  m->ups.clear(); m->blocks.clear();
  m->ups.resize(c.num_upsamples);
  m->blocks.resize((size_t)c.num_upsamples * c.num_kernels);
  int ch = C0;
  for (int i = 0; i < c.num_upsamples; ++i) {
    UpStage& us = m->ups[i];
    us.u = c.upsample_rates[i]; us.k = c.upsample_kernel_sizes[i]; us.Cin = C0 >> i; us.Cout = C0 >> (i + 1);
    ch = us.Cout;
    const std::string p = "ups." + std::to_string(i) + ".0.";
    std::vector<float> w;  // ConvTranspose1d weight (Cin, Cout, k)
    if (!m->ts.conv_weight(p, {us.Cin, us.Cout, us.k}, w)) return DMEL_EMISSING;
    const HostTensor* b = m->ts.need(p + "bias", {us.Cout});
    if (!b) return DMEL_EMISSING;
    DMEL_TRY(pack_up_stage(us, w.data(), b->v.data()));
    for (int j = 0; j < c.num_kernels; ++j) {
      AmpBlock& ab = m->blocks[(size_t)i * c.num_kernels + j];
      ab.k = c.resblock_kernel_sizes[j];
      const std::string bp = "resblocks." + std::to_string(i * c.num_kernels + j) + ".";
      for (int l = 0; l < 3; ++l) {
        ab.dil[l] = c.resblock_dilations[j][l];
        if (c.resblock_type == 2) {   // AMPBlock2: one dilated conv per layer (bigvgan.py:176-191), three activations
          DMEL_TRY(pack_same_conv(ab.c1[l], m->ts, bp + "convs." + std::to_string(l) + ".", ch, ch, ab.k, ab.dil[l], true));
        } else {
          DMEL_TRY(pack_same_conv(ab.c1[l], m->ts, bp + "convs1." + std::to_string(l) + ".", ch, ch, ab.k, ab.dil[l], true));
          DMEL_TRY(pack_same_conv(ab.c2[l], m->ts, bp + "convs2." + std::to_string(l) + ".", ch, ch, ab.k, 1, true));
        }
      }
      for (int a = 0; a < (c.resblock_type == 2 ? 3 : 6); ++a)
        DMEL_TRY(load_snake(ab.act[a], m->ts, bp + "activations." + std::to_string(a) + ".", ch, snake));
    }
  }
  if (!m->ev_fork) {
    const char* e = getenv("DMEL_BIGVGAN_STREAMS");
    m->multi = !(e && atoi(e) == 1) && c.num_kernels <= 3 && c.num_kernels > 1;
    DMEL_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) DMEL_HIP(hipEventCreateWithFlags(&m->ev_chain[i], hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) DMEL_HIP(hipEventCreateWithFlags(&m->ev_stag[i], hipEventDisableTiming));
    for (int i = 0; i < dmel_bigvgan::kSide; ++i) {
      DMEL_HIP(hipStreamCreateWithFlags(&m->side[i], hipStreamNonBlocking));
      DMEL_HIP(hipEventCreateWithFlags(&m->ev_join[i], hipEventDisableTiming));
    }
  }
  DMEL_TRY(load_snake(m->act_post, m->ts, "activation_post.", ch, snake));
  {
    std::vector<float> pw;
    if (!m->ts.conv_weight("conv_post.", {1, ch, 7}, pw)) return DMEL_EMISSING;
    m->post_bias = 0.f;
    if (c.use_bias_at_final) {
      const HostTensor* pb = m->ts.need("conv_post.bias", {1});
      if (!pb) return DMEL_EMISSING;
      m->post_bias = pb->v[0];
    }
    DMEL_TRY(upload_vec(m->post_w, pw));
    m->post_c = ch;
  }
  m->ts.t.clear();
  m->ready = true;
  return DMEL_OK;
}

constexpr int kBigvganBufs = 12;   // x, xu, xs + 3 x (xj, ua, vb)
extern "C" int dmel_bigvgan_set_streams(dmel_bigvgan* m, int n_streams) {
  DMEL_CHECK_ARG(m, "NULL handle");
  DMEL_CHECK_ARG(n_streams == 1 || n_streams == 3, "bigvgan: 1 or 3 streams supported");
  m->multi = n_streams == 3 && m->cfg.num_kernels > 1 && m->cfg.num_kernels <= 3 && m->ev_fork != nullptr;
  return DMEL_OK;
}

static size_t bigvgan_plan(const dmel_bigvgan* m, int B, int64_t T, void* ws, float* bufs[kBigvganBufs]) {
  const dmel_bigvgan_config& c = m->cfg;
  size_t mx = (size_t)c.upsample_initial_channel * T;
  int64_t Tc = T;
  for (int i = 0; i < c.num_upsamples; ++i) {
    Tc *= c.upsample_rates[i];
    mx = std::max(mx, (size_t)(c.upsample_initial_channel >> (i + 1)) * Tc);
  }
  Arena a(ws, (size_t)-1);
  for (int i = 0; i < kBigvganBufs; ++i) {
    float* p = a.take<float>(mx * B);
    if (bufs) bufs[i] = p;
  }
  return align_up(a.off, 256);
}

extern "C" size_t dmel_bigvgan_workspace_bytes(const dmel_bigvgan* m, int B, int64_t T) {
  if (!m || B <= 0 || T <= 0) return 0;
  return bigvgan_plan(m, B, T, nullptr, nullptr);
}

extern "C" int dmel_bigvgan_forward(const dmel_bigvgan* m, const float* mel, float* audio, int B, int64_t T, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  DMEL_CHECK_ARG(m && mel && audio && workspace, "bigvgan_forward: NULL argument");
  if (!m->ready) { set_error("bigvgan_forward: handle not finalized"); return DMEL_EMISSING; }
  DMEL_CHECK_ARG(B > 0 && T > 0, "bigvgan_forward: bad shape");
  float* bufs[kBigvganBufs];
  const size_t need = bigvgan_plan(m, B, T, workspace, bufs);
  DMEL_CHECK_ARG(workspace_bytes >= need, "bigvgan_forward: workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const dmel_bigvgan_config& c = m->cfg;
  // "fp32-grade, library's choice": nothing downstream of the vocoder is discrete, so its default is the three-product fp16 split
  // (include/dmel_hip.h); DMEL_VOCODER_F16X2=0 is the A/B switch back to the six-product bf16 split
  const char* f16_env = getenv("DMEL_VOCODER_F16X2");
  const int prec = m->precision == DMEL_PRECISION_FP32 && !(f16_env && f16_env[0] == '0') ? DMEL_PRECISION_FP32_F16X2 : m->precision;
  const int logscale = c.snake_logscale;
  float *x = bufs[0], *xu = bufs[1], *xs = bufs[2];
  float* ua = bufs[4];   // block 0's activation scratch doubles as the post-activation buffer

  {  // conv_pre (bigvgan.py:369)
    ConvRun r = run_1seg(mel, c.num_mels, T, x, c.upsample_initial_channel, T, B);
    r.precision = prec;
    DMEL_TRY(launch_conv(m->conv_pre, r, st));
  }
  int64_t Tc = T;
  int ch = c.upsample_initial_channel;
  for (int i = 0; i < c.num_upsamples; ++i) {
    const UpStage& us = m->ups[i];
    const int64_t Tn = Tc * us.u;
    DMEL_TRY(launch_up_stage(us, x, xu, B, Tc, prec, st));  // transposed conv as two phase groups (bigvgan.py:371-374)
    ch = us.Cout;
    Tc = Tn;
    const int64_t bs = (int64_t)ch * Tc;
    const bool multi = m->multi;
    if (multi) {  // fork: the side streams start behind the transposed conv
      DMEL_HIP(hipEventRecord(m->ev_fork, st));
      for (int k = 0; k < dmel_bigvgan::kSide; ++k) DMEL_HIP(hipStreamWaitEvent(m->side[k], m->ev_fork, 0));
    }
    for (int j = 0; j < c.num_kernels; ++j) {  // AMPBlock1.forward (bigvgan.py:132-141), summed and averaged (:376-382)
      const AmpBlock& ab = m->blocks[(size_t)i * c.num_kernels + j];
      hipStream_t sj = (multi && j > 0) ? m->side[j - 1] : st;
      float *xj = bufs[3 + 3 * j], *uj = bufs[4 + 3 * j], *vj = bufs[5 + 3 * j];
      const float* xin = xu;
      static const int stagger = [] { const char* e = getenv("DMEL_BIGVGAN_STAGGER"); return e ? atoi(e) : 1; }();
      // DMEL_FUSE_SNAKE=1: every act -> conv pair as ONE kernel (conv_snake.hip: producer waves compute the activation, consumer waves run
      // the MFMA loop).  Bit-identical to the two-kernel form and measured SLOWER on every vocoder shape (0.54-1.03x,
      // profiles/r03_fused_vs_two_kernels.txt; DESIGN.md section 4 has the probes that explain it), so the two-kernel form stays the
      // default.  Read per call so that a test can flip it inside one process.
      const char* fuse_env = getenv("DMEL_FUSE_SNAKE");
      const bool fuse = fuse_env && fuse_env[0] == '1';
      for (int l = 0; l < 3; ++l) {
        // stagger the branches by one kernel: started together they run snake|snake|snake then conv|conv|conv in lockstep
        // and the VALU-bound activations never meet the matrix-pipe-bound convolutions on a CU
        if (l == 0 && multi && stagger && j > 0) DMEL_HIP(hipStreamWaitEvent(sj, m->ev_stag[j - 1], 0));
        if (c.resblock_type == 2) {   // AMPBlock2.forward (bigvgan.py:232-237): xt = a(x); xt = c(xt); x = xt + x
          ConvRun r2 = run_1seg(uj, ch, Tc, l < 2 ? xj : xs, ch, Tc, B);
          r2.res = xin; r2.res_bs = bs; r2.res_cs = Tc;
          r2.precision = prec;
          const bool fuse2 = fuse && conv_snake_eligible(ab.c1[l], r2);
          if (!fuse2) DMEL_TRY(launch_aa_snake(xin, uj, ab.act[l].alpha.as<float>(), ab.act[l].beta.as<float>(), m->taps_up, m->taps_dn, logscale, B, ch, Tc, sj));
          if (!fuse2 && l == 0 && multi && stagger && j + 1 < c.num_kernels) DMEL_HIP(hipEventRecord(m->ev_stag[j], sj));
          if (l == 2) {
            r2.accumulate = j > 0;
            if (j == c.num_kernels - 1) r2.out_div = (float)c.num_kernels;
            if (multi && j > 0) DMEL_HIP(hipStreamWaitEvent(sj, m->ev_chain[j - 1], 0));
          }
          if (fuse2) {
            r2.seg[0].x = xin;
            DMEL_TRY(launch_conv_snake(ab.c1[l], r2, ab.act[l].alpha.as<float>(), ab.act[l].beta.as<float>(), m->taps_up, m->taps_dn, logscale, sj));
            if (l == 0 && multi && stagger && j + 1 < c.num_kernels) DMEL_HIP(hipEventRecord(m->ev_stag[j], sj));
          } else {
            DMEL_TRY(launch_conv(ab.c1[l], r2, sj));
          }
          if (l == 2 && multi && j + 1 < c.num_kernels) DMEL_HIP(hipEventRecord(m->ev_chain[j], sj));
          xin = xj;
          continue;
        }
        ConvRun r1 = run_1seg(uj, ch, Tc, vj, ch, Tc, B);
        r1.precision = prec;
        if (fuse && conv_snake_eligible(ab.c1[l], r1)) {       // act -> conv as ONE kernel: the activated tensor never exists in HBM
          r1.seg[0].x = xin;
          DMEL_TRY(launch_conv_snake(ab.c1[l], r1, ab.act[2 * l].alpha.as<float>(), ab.act[2 * l].beta.as<float>(), m->taps_up, m->taps_dn, logscale, sj));
          if (l == 0 && multi && stagger && j + 1 < c.num_kernels) DMEL_HIP(hipEventRecord(m->ev_stag[j], sj));
        } else {
          DMEL_TRY(launch_aa_snake(xin, uj, ab.act[2 * l].alpha.as<float>(), ab.act[2 * l].beta.as<float>(), m->taps_up, m->taps_dn, logscale, B, ch, Tc, sj));
          if (l == 0 && multi && stagger && j + 1 < c.num_kernels) DMEL_HIP(hipEventRecord(m->ev_stag[j], sj));
          DMEL_TRY(launch_conv(ab.c1[l], r1, sj));
        }
        ConvRun r2 = run_1seg(uj, ch, Tc, l < 2 ? xj : xs, ch, Tc, B);
        r2.res = xin; r2.res_bs = bs; r2.res_cs = Tc;
        r2.precision = prec;
        const bool fuse2 = fuse && conv_snake_eligible(ab.c2[l], r2);
        if (!fuse2) DMEL_TRY(launch_aa_snake(vj, uj, ab.act[2 * l + 1].alpha.as<float>(), ab.act[2 * l + 1].beta.as<float>(), m->taps_up, m->taps_dn, logscale, B, ch, Tc, sj));
        if (l == 2) {
          // xs = ((out_0 + out_1) + out_2) / 3, in the reference's order: block j's last conv runs behind block j-1's
          r2.accumulate = j > 0;
          if (j == c.num_kernels - 1) r2.out_div = (float)c.num_kernels;
          if (multi && j > 0) DMEL_HIP(hipStreamWaitEvent(sj, m->ev_chain[j - 1], 0));
        }
        if (fuse2) {
          r2.seg[0].x = vj;
          DMEL_TRY(launch_conv_snake(ab.c2[l], r2, ab.act[2 * l + 1].alpha.as<float>(), ab.act[2 * l + 1].beta.as<float>(), m->taps_up, m->taps_dn, logscale, sj));
        } else {
          DMEL_TRY(launch_conv(ab.c2[l], r2, sj));
        }
        if (l == 2 && multi && j + 1 < c.num_kernels) DMEL_HIP(hipEventRecord(m->ev_chain[j], sj));
        xin = xj;
      }
    }
    if (multi) {  // join: the caller's stream continues behind every side stream
      for (int k = 0; k < dmel_bigvgan::kSide && k + 1 < c.num_kernels; ++k) {
        DMEL_HIP(hipEventRecord(m->ev_join[k], m->side[k]));
        DMEL_HIP(hipStreamWaitEvent(st, m->ev_join[k], 0));
      }
    }
    std::swap(x, xs);
  }
  // activation_post, conv_post, tanh | clamp (bigvgan.py:385-391)
  DMEL_TRY(launch_aa_snake(x, ua, m->act_post.alpha.as<float>(), m->act_post.beta.as<float>(), m->taps_up, m->taps_dn, logscale, B, ch, Tc, st));
  return launch_conv_post(ua, audio, m->post_w.as<float>(), m->post_bias, c.use_tanh_at_final ? ACT_TANH : ACT_CLAMP1, B, ch, 7,
                          Tc, st);
}
