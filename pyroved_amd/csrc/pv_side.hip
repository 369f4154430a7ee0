// pv_side.hip — the library's second HIP stream (one per device) for work that is off a step's dependent chain: the
// convolutional encoders' weight gradients, their split-order reductions, the step's weight tilings.  A step forks work
// onto it with pv_stream_after(side, main) and joins with pv_stream_after(main, side) before anything that depends on
// it (and always before the entry point returns: nothing is left running on the side stream that the caller's stream
// does not wait for).  Events come from a small ring per device; the stream has the lowest priority the device offers,
// so the dependent chain on the caller's stream wins the dispatcher when both have workgroups pending.
#include "pv_common.h"
#include "pv_side.h"
#include <atomic>
#include <mutex>
#include <stdlib.h>

namespace {

constexpr int kMaxDev = 16, kEvents = 64;
struct Side {
  hipStream_t stream = nullptr, stream2 = nullptr;
  hipEvent_t ev[kEvents] = {};
  int next = 0;
  bool tried = false;
};
Side g_side[kMaxDev];
std::mutex g_mu;

// per plan: PV_PLAN_NO_SIDE_STREAM (pv_side_stream_for); PV_NO_SIDE=1 in the experiments build
bool side_enabled() {
  static const bool on = !pv_exp_int("PV_NO_SIDE", 0);
  return on;
}

Side* side_of_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
  Side& S = g_side[dev];
  std::lock_guard<std::mutex> lk(g_mu);
  if (!S.tried) {
    S.tried = true;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStream_t st = nullptr;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    for (int i = 0; i < kEvents; ++i)
      if (hipEventCreateWithFlags(&S.ev[i], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        for (int k = 0; k < i; ++k) (void)hipEventDestroy(S.ev[k]);
        (void)hipStreamDestroy(st);
        return nullptr;
      }
    S.stream = st;
    // a second one of the same priority (two independent families of side work; null when the device refuses)
    if (hipStreamCreateWithPriority(&S.stream2, hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); S.stream2 = nullptr; }
  }
  return S.stream ? &S : nullptr;
}

}  // namespace

bool pv_stream_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}

hipStream_t pv_side_stream() {
  if (!side_enabled()) return nullptr;
  Side* S = side_of_current_device();
  return S ? S->stream : nullptr;
}

hipStream_t pv_side_stream2() {
  if (!side_enabled()) return nullptr;
  Side* S = side_of_current_device();
  return S ? S->stream2 : nullptr;
}

namespace {
thread_local hipEvent_t t_armed = nullptr;
thread_local bool t_taken = false;
hipEvent_t next_event(Side* S) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t e = S->ev[S->next];
  S->next = (S->next + 1) % kEvents;
  return e;
}
}  // namespace

void pv_fork_arm() {
  Side* S = side_of_current_device();
  t_armed = S ? next_event(S) : nullptr;
  t_taken = false;
}
void pv_fork_disarm() { t_armed = nullptr; t_taken = false; }
hipEvent_t pv_fork_take() {
  if (!t_armed || t_taken) return nullptr;
  t_taken = true;
  return t_armed;
}
bool pv_fork_taken() { return t_armed && t_taken; }
int pv_fork_to(hipStream_t side, hipStream_t main, hipStream_t side_b) {
  if (t_armed && t_taken) {
    hipError_t rc = hipStreamWaitEvent(side, t_armed, 0);
    if (rc == hipSuccess && side_b && side_b != side) rc = hipStreamWaitEvent(side_b, t_armed, 0);
    pv_fork_disarm();
    return rc == hipSuccess ? 0 : (int)rc;
  }
  pv_fork_disarm();
  if (side_b && side_b != side) {                     // one marker for both
    if (side == main && side_b == main) return 0;
    Side* S = side_of_current_device();
    if (!S) return PV_EINVAL;
    hipEvent_t e = next_event(S);
    hipError_t rc = hipEventRecord(e, main);
    if (rc == hipSuccess && side != main) rc = hipStreamWaitEvent(side, e, 0);
    if (rc == hipSuccess && side_b != main) rc = hipStreamWaitEvent(side_b, e, 0);
    return rc == hipSuccess ? 0 : (int)rc;
  }
  return pv_stream_after(side, main);
}

// ---- roctx ranges (pv_common.h: PV_RANGE) ----
#include <dlfcn.h>
namespace {
typedef int (*RangePushFn)(const char*);
typedef int (*RangePopFn)();
RangePushFn g_push = nullptr;
RangePopFn g_pop = nullptr;
std::atomic<int> g_roctx{-1};                         // -1: not resolved yet, 0: absent, 1: present
void resolve_roctx() {
  void* push = dlsym(RTLD_DEFAULT, "roctxRangePushA");
  void* pop = dlsym(RTLD_DEFAULT, "roctxRangePop");
  const char* want = getenv("PV_ROCTX");
  if ((!push || !pop) && want && atoi(want) != 0) {
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
      if (!h) continue;
      push = dlsym(h, "roctxRangePushA");
      pop = dlsym(h, "roctxRangePop");
      if (push && pop) break;
    }
  }
  std::lock_guard<std::mutex> lk(g_mu);
  if (push && pop) { g_push = (RangePushFn)push; g_pop = (RangePopFn)pop; g_roctx.store(1); }
  else g_roctx.store(0);
}
}  // namespace
void pv_range_push(const char* name) {
  int st = g_roctx.load(std::memory_order_acquire);
  if (st < 0) { resolve_roctx(); st = g_roctx.load(std::memory_order_acquire); }
  if (st == 1) (void)g_push(name);
}
void pv_range_pop() {
  if (g_roctx.load(std::memory_order_acquire) == 1) (void)g_pop();
}
// test hook: 1 when the marker library was found (after the first range)
extern "C" int pv_debug_roctx_state() { pv_range_push("pv_debug_roctx_state"); pv_range_pop(); return g_roctx.load(); }

// ---- dynamic-LDS opt-in per (device, kernel) ----
namespace {
struct LdsKey { const void* fn; int bytes; };
LdsKey g_lds[kMaxDev][64];
int g_lds_n[kMaxDev] = {};
int g_lds_limit[kMaxDev] = {};
}  // namespace
int pv_set_dynamic_lds(const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
    (void)hipGetLastError();
    return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  }
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < g_lds_n[dev]; ++i)
      if (g_lds[dev][i].fn == fn && g_lds[dev][i].bytes >= bytes) return 0;
  }
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < g_lds_n[dev]; ++i)
    if (g_lds[dev][i].fn == fn) { g_lds[dev][i].bytes = bytes; return 0; }
  if (g_lds_n[dev] < 64) g_lds[dev][g_lds_n[dev]++] = LdsKey{fn, bytes};
  return 0;
}
int pv_device_lds_limit() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) { (void)hipGetLastError(); return 0; }
  if (g_lds_limit[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); v = 0; }
    g_lds_limit[dev] = v > 0 ? v : -1;
  }
  return g_lds_limit[dev] > 0 ? g_lds_limit[dev] : 0;
}

int pv_stream_after(hipStream_t waiter, hipStream_t signaller) {
  if (waiter == signaller) return 0;
  Side* S = side_of_current_device();
  if (!S) return PV_EINVAL;
  hipEvent_t e = next_event(S);
  hipError_t rc = hipEventRecord(e, signaller);
  if (rc == hipSuccess) rc = hipStreamWaitEvent(waiter, e, 0);
  return rc == hipSuccess ? 0 : (int)rc;
}
