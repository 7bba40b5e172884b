// Error string + launch-timing plumbing of libdmel_hip.so.
#include "common.h"

#include <map>
#include <tuple>
#include <atomic>
#include <mutex>

namespace dmel {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

// ---- per-family hipEvent timing --------------------------------------------------------------
struct ProfRec {
  hipEvent_t start, stop;
  int family;
  double flops, bytes, issue;
};
struct ProfState {
  std::mutex mu;
  std::atomic<bool> on{false};   // read by every ProfScope constructor without the lock
  std::vector<std::string> families;
  std::vector<ProfRec> recs;
};
static ProfState& prof() {
  static ProfState s;
  return s;
}

ProfScope::ProfScope(const char* family, hipStream_t s, double flops, double bytes, double issue_flops) : slot(-1), stream(s) {
  ProfState& p = prof();
  if (!p.on) return;
  std::lock_guard<std::mutex> lk(p.mu);
  int fam = -1;
  for (size_t i = 0; i < p.families.size(); ++i)
    if (p.families[i] == family) fam = (int)i;
  if (fam < 0) {
    p.families.push_back(family);
    fam = (int)p.families.size() - 1;
  }
  ProfRec r;
  r.family = fam;
  r.flops = flops;
  r.bytes = bytes;
  r.issue = issue_flops;
  if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  (void)hipEventRecord(r.start, s);
  p.recs.push_back(r);
  slot = (int)p.recs.size() - 1;
}

ProfScope::~ProfScope() {
  if (slot < 0) return;
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  (void)hipEventRecord(p.recs[slot].stop, stream);
}

}  // namespace dmel

using namespace dmel;

extern "C" const char* dmel_last_error(void) { return g_err.c_str(); }
namespace dmel {
DevBuf* thread_scratch(int which, hipStream_t stream) {
  // deliberately leaked (common.h): no hipFree from a thread_local destructor after the runtime is gone
  static thread_local auto* bufs = new std::map<std::tuple<int, int, hipStream_t>, DevBuf>();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return &(*bufs)[std::make_tuple(dev, which, stream)];
}
namespace {
struct ClearedState { const char* lo = nullptr; const char* hi = nullptr; };
ClearedState& cleared_state() {
  static thread_local ClearedState s;
  return s;
}
}  // namespace
ClearedRange::ClearedRange(void* p, size_t bytes, hipStream_t s) : lo((const char*)p), hi((const char*)p + bytes), err(DMEL_OK) {
  ClearedState& st = cleared_state();
  prev_lo = st.lo; prev_hi = st.hi;
  if (hipMemsetAsync(p, 0, bytes, s) != hipSuccess) {
    set_error("gradient buffer: hipMemsetAsync of %zu bytes failed", bytes);
    err = DMEL_EINVAL;
    lo = hi = nullptr;
  }
  st.lo = lo; st.hi = hi;
}
ClearedRange::~ClearedRange() {
  ClearedState& st = cleared_state();
  st.lo = prev_lo; st.hi = prev_hi;
}
int zero_unless_cleared(void* p, size_t bytes, hipStream_t s) {
  const ClearedState& st = cleared_state();
  const char* c = (const char*)p;
  if (st.lo && c >= st.lo && c + bytes <= st.hi) return DMEL_OK;
  DMEL_HIP(hipMemsetAsync(p, 0, bytes, s));
  return DMEL_OK;
}
int& train_precision_override() {
  static thread_local int v = -1;
  return v;
}
}  // namespace dmel

extern "C" int dmel_abi_version(void) { return 2; }   // 2: dmel_aa_snake_* take separate up / down filters; hooked backward; strict encode

extern "C" int dmel_prof_enable(int on) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  p.on = on != 0;
  return DMEL_OK;
}

extern "C" int dmel_prof_reset(void) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  for (auto& r : p.recs) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  p.recs.clear();
  return DMEL_OK;
}

extern "C" int dmel_prof_read_ex(const char* family, int64_t* launches, double* total_ms, double* total_flops, double* total_bytes,
                                 double* total_issue_flops) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> lk(p.mu);
  int64_t n = 0;
  double ms = 0, fl = 0, by = 0, is = 0;
  for (auto& r : p.recs) {
    if (p.families[r.family] != family) continue;
    if (hipEventSynchronize(r.stop) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.start, r.stop) != hipSuccess) continue;
    ++n;
    ms += t;
    fl += r.flops;
    by += r.bytes;
    is += r.issue;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (total_bytes) *total_bytes = by;
  if (total_issue_flops) *total_issue_flops = is;
  return DMEL_OK;
}
extern "C" int dmel_prof_read(const char* family, int64_t* launches, double* total_ms, double* total_flops,
                              double* total_bytes) {
  return dmel_prof_read_ex(family, launches, total_ms, total_flops, total_bytes, nullptr);
}
