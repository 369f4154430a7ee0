// pv_side.h — fork / join onto the library's low-priority side stream (pv_side.hip)
#pragma once
#include "pv_common.h"
#include <hip/hip_ext.h>
// the side stream of the calling thread's current device; null: disabled (PV_NO_SIDE=1) or not available
hipStream_t pv_side_stream();
hipStream_t pv_side_stream2();                         // a second one (same priority); null when pv_side_stream() is
// is `s` being captured into a graph?  (Forks with stop events and launches that hand data over through per-call flag values
// are not replayable: the entry points fall back to their one-stream, one-kernel-per-stage forms under capture.)
bool pv_stream_capturing(hipStream_t s);
// pv_side_stream() unless `s` is being captured
inline hipStream_t pv_side_stream_for(hipStream_t s, int plan_flags = 0) {
  return ((plan_flags & 2 /* PV_PLAN_NO_SIDE_STREAM */) || pv_stream_capturing(s)) ? nullptr : pv_side_stream();
}
// everything enqueued on `signaller` so far happens before whatever is enqueued on `waiter` from now on
// (an event record on `signaller`: a marker packet that costs that stream ~5 us, scripts/ubench/event_cost.hip)
int pv_stream_after(hipStream_t waiter, hipStream_t signaller);

// The cheap fork: the kernel that PRODUCES what the side stream waits for carries the event as its stop event
// (hipExtLaunchKernelGGL: no marker packet, ~0-1 us).  pv_fork_arm() before calling the producer; its final launch goes
// through PV_LAUNCH_FORK, which attaches the armed event; pv_fork_to(side, main) then makes `side` wait for that launch —
// or, when no launch took the event (another code path ran) or it was disarmed, for everything on `main` so far.
// The state is per host thread.  pv_fork_disarm(): something else was enqueued on the main stream after the producer.
void pv_fork_arm();
void pv_fork_disarm();
hipEvent_t pv_fork_take();
bool pv_fork_taken();                                  // did a launch take the armed event?
int pv_fork_to(hipStream_t side, hipStream_t main, hipStream_t side_b = nullptr);   // side_b: a second waiter on the same event
#define PV_LAUNCH_FORK(KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                            \
  do {                                                                                                    \
    hipEvent_t fe__ = pv_fork_take();                                                                     \
    if (fe__) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, nullptr, fe__, 0, __VA_ARGS__);     \
    else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                               \
  } while (0)

// An entry point that forks work onto the side stream holds one of these: if it returns early (an error) after a fork, the
// caller's stream still waits for the side stream — nothing is left running on memory the caller may release.
struct PvSideJoin {
  hipStream_t main = nullptr, side = nullptr;
  bool forked = false;
  void fork(hipStream_t m, hipStream_t sd) { main = m; side = sd; forked = sd != nullptr; }
  void joined() { forked = false; }
  ~PvSideJoin() {
    if (forked) (void)pv_stream_after(main, side);
    pv_fork_disarm();
  }
};
