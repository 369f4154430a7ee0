// pv_dist.hip — the data-parallel step's collective INSIDE the library (ABI v16; SURVEY section 8b, "Collective boundary": "a direct
// ncclAllReduce through the same C layer").  The reference has no distributed code at all; what is sharded
// is trainers/svi.py:104-113's `self.svi.step(x)` — its loss is a SUM over the plate (models/ivae.py:177,215), so the global
// gradient is the plain sum of the shards' gradients: ONE ncclAllReduce(SUM) over [flat gradient | 4 loss scalars].
//
// Why here and not torch.distributed.all_reduce: ProcessGroupNCCL runs its collectives on an internal stream and brackets each with
// an event record + stream wait on both sides (~10 us of hand-offs around a ~30-50 us latency-bound 0.6 MB all-reduce, on a step
// of 104 us).  pv_ivae_dp_step enqueues  loss_and_grads' launches -> ncclAllReduce -> pv_adam_step_hist  on the CALLER's stream in
// one library call: no hand-off, no second stream, nothing between the last gradient launch and the collective's first kernel,
// capturable as a hipGraph (RCCL's kernels capture like any other launch).
//
// RCCL is NOT a link dependency of libpyroved_amd.so (a single-GPU user never loads it): the caller names the RCCL library its
// process already holds — PyTorch ships its own librccl.so, and two copies of RCCL in one process would each open the fabric —
// through pv_dist_load(path); the five entry points used are looked up there once.  The communicator is the caller's
// (ncclCommInitRank on the same library; pyroved_amd/dist.py: NativeComm) and is handed in as an opaque pointer per call.
#include "pv_common.h"
#include <dlfcn.h>
#include <mutex>
#include <stdio.h>
#include <string.h>

namespace {
// ncclResult_t / ncclDataType_t / ncclRedOp_t are plain ints in RCCL's C API (rccl.h: ncclSuccess = 0, ncclFloat32 = 7, ncclSum = 0)
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
typedef int (*nccl_count_fn)(void*, int*);
typedef int (*nccl_async_fn)(void*, int*);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;

struct Rccl {
  void* handle = nullptr;
  nccl_allreduce_fn allreduce = nullptr;
  nccl_errstr_fn errstr = nullptr;
  nccl_count_fn count = nullptr, user_rank = nullptr;
  nccl_async_fn async_err = nullptr;
  char path[512] = {0};
};
// once-initialised (guarded) cache of the resolved entry points: set by the first successful pv_dist_load, never changed
std::mutex g_mu;
Rccl g_rccl;

int load_locked(const char* path) {
  if (g_rccl.handle) return (path && path[0] && strcmp(path, g_rccl.path) != 0) ? PV_EINVAL : 0;   // one RCCL per process
  const char* cands[] = {path && path[0] ? path : nullptr, "librccl.so.1", "librccl.so"};
  void* h = nullptr;
  const char* used = nullptr;
  for (const char* c : cands) {
    if (!c) continue;
    h = dlopen(c, RTLD_NOW | RTLD_NOLOAD);               // what the process already holds, first
    if (!h) h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
    if (h) { used = c; break; }
    if (path && path[0]) break;                            // an explicit path that does not load is an error, not a hint
  }
  if (!h) return PV_ECOLL;
  Rccl r;
  r.handle = h;
  r.allreduce = (nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
  r.errstr = (nccl_errstr_fn)dlsym(h, "ncclGetErrorString");
  r.count = (nccl_count_fn)dlsym(h, "ncclCommCount");
  r.user_rank = (nccl_count_fn)dlsym(h, "ncclCommUserRank");
  r.async_err = (nccl_async_fn)dlsym(h, "ncclCommGetAsyncError");
  if (!r.allreduce || !r.count || !r.user_rank) { dlclose(h); return PV_ECOLL; }
  snprintf(r.path, sizeof r.path, "%s", used);
  g_rccl = r;
  return 0;
}

const Rccl* rccl() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_rccl.handle ? &g_rccl : nullptr;
}
}  // namespace

extern "C" int pv_dist_load(const char* rccl_path) {
  std::lock_guard<std::mutex> lk(g_mu);
  return load_locked(rccl_path);
}

extern "C" const char* pv_dist_library(void) {
  const Rccl* r = rccl();
  return r ? r->path : "";
}

extern "C" int pv_dist_comm_info(void* comm, int32_t* rank, int32_t* world) {
  const Rccl* r = rccl();
  if (!r || !comm) return PV_EINVAL;
  int n = 0, k = 0;
  if (r->count(comm, &n) != 0 || r->user_rank(comm, &k) != 0) return PV_ECOLL;
  if (rank) *rank = k;
  if (world) *world = n;
  return 0;
}

extern "C" int pv_dist_allreduce_sum(void* comm, float* buf, int64_t n, void* stream) {
  PV_RANGE("pv_dist_allreduce_sum");
  const Rccl* r = rccl();
  if (!r || !comm || !buf || n < 0) return PV_EINVAL;
  if (n == 0) return 0;
  // in place; fp32 SUM in RCCL's fixed ring / tree order for (count, ranks): every rank ends with the same bits
  const int rc = r->allreduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, comm, (hipStream_t)stream);
  return rc == 0 ? 0 : PV_ECOLL;
}

// SVI.step of one data-parallel replica as ONE enqueue on ONE stream (trainers/svi.py:107 under sharding):
//   Trace_ELBO.loss_and_grads on this rank's shard  ->  all-reduce(SUM) of [grads | scalars]  ->  Adam + zero_grads + the reduced
//   loss scalars into hist_dst (pv_adam_step_hist).  plan->scalars must be plan->grads + plan->n_params (the flat gradient
//   buffer's 4 trailing slots: DESIGN.md section 3), so that the loss rides in the same collective.
extern "C" int pv_ivae_dp_step(const pv_ivae_plan* plan, void* comm, float* hist_dst, void* stream) {
  PV_RANGE("pv_ivae_dp_step");
  if (!plan || !comm || !plan->grads || !plan->adam_m || !plan->adam_v || plan->adam_step < 1 || plan->n_params <= 0)
    return PV_EINVAL;
  if (plan->scalars != plan->grads + plan->n_params) return PV_EINVAL;
  if (plan->ext_encoder || plan->ext_decoder) return PV_EINVAL;   // (their gradients live in the caller's framework)
  if (!rccl()) return PV_EINVAL;
  PV_TRY(pv_ivae_loss_and_grads(plan, 1, stream));
  PV_TRY(pv_dist_allreduce_sum(comm, plan->grads, plan->n_params + 4, stream));
  return pv_adam_step_hist(plan->params, plan->grads, plan->adam_m, plan->adam_v, plan->n_params, plan->lr, plan->adam_beta1,
                           plan->adam_beta2, plan->adam_eps, plan->adam_step, plan->scalars, hist_dst, hist_dst ? 4 : 0, stream);
}

// The same for VED (models/ved.py:122-163): pv_ved_plan carries no optimizer fields, so Adam's arrive as arguments.
extern "C" int pv_ved_dp_step(const pv_ved_plan* plan, void* comm, float lr, float beta1, float beta2, float eps,
                              int32_t adam_step, float* hist_dst, void* stream) {
  PV_RANGE("pv_ved_dp_step");
  if (!plan || !comm || !plan->grads || !plan->adam_m || !plan->adam_v || adam_step < 1 || plan->n_params <= 0) return PV_EINVAL;
  if (plan->scalars != plan->grads + plan->n_params) return PV_EINVAL;
  if (!rccl()) return PV_EINVAL;
  PV_TRY(pv_ved_loss_and_grads(plan, 1, stream));
  PV_TRY(pv_dist_allreduce_sum(comm, plan->grads, plan->n_params + 4, stream));
  return pv_adam_step_hist(plan->params, plan->grads, plan->adam_m, plan->adam_v, plan->n_params, lr, beta1, beta2, eps,
                           adam_step, plan->scalars, hist_dst, hist_dst ? 4 : 0, stream);
}
