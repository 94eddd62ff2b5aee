// ABI bookkeeping for libnsdp_hip.so: version, thread-local last-error string, device probe, and the
// optional HIP-event kernel profiler used by bench.py.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "common.h"
#include "prof.h"

namespace nsdp {
static thread_local char g_last_error[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

namespace prof {

namespace {
struct Record {
  int kind;
  hipEvent_t start, stop;
  double flops, bytes;
};
std::atomic<unsigned> g_enabled{0};   // bit k: kernels of Kind k are timed
std::mutex g_mu;
std::vector<Record> g_records;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

const char *kind_name(int kind) {
  static const char *names[kNumKinds] = {"linear_nt_kernel",  "linear_wgrad_kernel", "fps_kernel",
                                         "knn_kernel",        "gather_rows_kernel",  "scatter_add_rows_kernel",
                                         "attn_fwd_kernels",  "attn_bwd_kernels",    "batch_norm_kernels",
                                         "decoder_fwd_kernel", "linear_bf16x3_kernel", "wgrad_bf16x3_kernel",
                                         "linear_bf16_kernel", "wgrad_bf16_kernel"};
  return (kind >= 0 && kind < kNumKinds) ? names[kind] : "?";
}

Scope::Scope(Kind kind, hipStream_t stream, double flops, double bytes) : slot_(-1), stream_(stream) {
  if (!((g_enabled.load(std::memory_order_relaxed) >> static_cast<int>(kind)) & 1u)) return;
  std::lock_guard<std::mutex> lock(g_mu);
  Record r{static_cast<int>(kind), get_event(), get_event(), flops, bytes};
  if (!r.start || !r.stop) return;
  (void)hipEventRecord(r.start, stream);
  g_records.push_back(r);
  slot_ = static_cast<int>(g_records.size()) - 1;
}

namespace {
std::atomic<int> g_trace{0};
std::set<std::string> g_trace_names;
}  // namespace

bool trace_on() { return g_trace.load(std::memory_order_relaxed) != 0; }

void trace_note(const char *fmt, ...) {
  char buf[160];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  std::lock_guard<std::mutex> lock(g_mu);
  g_trace_names.insert(buf);
}

Scope::~Scope() {
  if (slot_ < 0) return;
  std::lock_guard<std::mutex> lock(g_mu);
  if (slot_ < static_cast<int>(g_records.size())) (void)hipEventRecord(g_records[slot_].stop, stream_);
}

}  // namespace prof
}  // namespace nsdp

extern "C" {

int nsdp_abi_version(void) { return 7; }

const char *nsdp_last_error(void) { return nsdp::g_last_error; }

int nsdp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

void nsdp_prof_enable(int on) { nsdp_prof_enable_kinds(on ? ~0u : 0u); }

void nsdp_prof_enable_kinds(unsigned mask) {
  using namespace nsdp::prof;
  std::lock_guard<std::mutex> lock(g_mu);
  const int on = mask != 0;
  if (on) {
    for (auto &r : g_records) {
      g_pool.push_back(r.start);
      g_pool.push_back(r.stop);
    }
    g_records.clear();
  }
  g_enabled.store(mask);
}

void nsdp_trace_enable(int on) {
  using namespace nsdp::prof;
  std::lock_guard<std::mutex> lock(g_mu);
  if (on) g_trace_names.clear();
  g_trace.store(on ? 1 : 0);
}

int nsdp_trace_read(char *buf, int capacity) {
  using namespace nsdp::prof;
  std::lock_guard<std::mutex> lock(g_mu);
  std::string all;
  for (const auto &n : g_trace_names) {
    if (!all.empty()) all += '\n';
    all += n;
  }
  if (buf && capacity > 0) {
    const int n = static_cast<int>(all.size()) < capacity - 1 ? static_cast<int>(all.size()) : capacity - 1;
    memcpy(buf, all.data(), n);
    buf[n] = 0;
  }
  return static_cast<int>(all.size()) + 1;
}

int nsdp_prof_num_kinds(void) { return nsdp::prof::kNumKinds; }

const char *nsdp_prof_name(int kind) { return nsdp::prof::kind_name(kind); }

int nsdp_prof_collect(int kind, long long *launches, double *total_ms, double *flops, double *bytes) {
  using namespace nsdp::prof;
  std::lock_guard<std::mutex> lock(g_mu);
  long long n = 0;
  double ms = 0, fl = 0, by = 0;
  for (auto &r : g_records) {
    if (r.kind != kind) continue;
    if (hipEventSynchronize(r.stop) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.start, r.stop) != hipSuccess) continue;
    ++n;
    ms += t;
    fl += r.flops;
    by += r.bytes;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  return 0;
}

}  // extern "C"
