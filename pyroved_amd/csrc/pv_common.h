// pv_common.h — shared device helpers for libpyroved_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pyroved_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PV_LAUNCH_CHECK()                      \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

// ---- roctx ranges around the ABI entry points (tracing).  The marker library is NOT a link dependency: the
// two symbols are looked up once — in what the process already holds (a profiler's preload, an application that links roctx),
// else by dlopen of librocprofiler-sdk-roctx.so / libroctx64.so when PV_ROCTX=1 — and a missing library costs one branch per
// call.  rocprofv3 --marker-trace then cuts a trainer epoch by step and phase (pv_ivae_step, pv_ved_loss_and_grads, ...).
void pv_range_push(const char* name);   // pv_side.hip
void pv_range_pop();
struct PvRange {
  explicit PvRange(const char* name) { pv_range_push(name); }
  ~PvRange() { pv_range_pop(); }
  PvRange(const PvRange&) = delete;
  PvRange& operator=(const PvRange&) = delete;
};
#define PV_RANGE(name) PvRange pv_range_guard__(name)

// hipFuncAttributeMaxDynamicSharedMemorySize for `fn`, once per (device, function): the attribute is per device, and a process
// may drive several (a process-wide "configured" flag left the second device's launches failing).  0 or a hipError_t.
int pv_set_dynamic_lds(const void* fn, int bytes);   // pv_side.hip
// the current device's LDS per workgroup (bytes; 0 when unknown)
int pv_device_lds_limit();

// ---- experiment switches.  The shipped library has NO environment switches besides PV_ROCTX (pv_side.hip) and no mutable
// process-wide state: every A/B knob of the kernel experiments (NOTES.md) is pv_exp_int(name, default), which is the
// compile-time constant `default` here and reads the environment only in the -DPV_EXPERIMENTS build
// (`make experiments` -> libpyroved_amd_exp.so, loaded through PV_LIB_PATH by scripts/ and by the tests of dropped variants).
#ifdef PV_EXPERIMENTS
#include <stdlib.h>
static inline const char* pv_exp_str(const char* name) { return getenv(name); }
static inline int pv_exp_int(const char* name, int dflt) { const char* e = pv_exp_str(name); return e ? atoi(e) : dflt; }
static inline long long pv_exp_ll(const char* name, long long dflt) { const char* e = pv_exp_str(name); return e ? atoll(e) : dflt; }
#else
static inline constexpr const char* pv_exp_str(const char*) { return nullptr; }
static inline constexpr int pv_exp_int(const char*, int dflt) { return dflt; }
static inline constexpr long long pv_exp_ll(const char*, long long dflt) { return dflt; }
#endif

#define PV_TRY(expr)                           \
  do {                                         \
    int r__ = (expr);                          \
    if (r__ != 0) return r__;                  \
  } while (0)

static inline int64_t pv_align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// ---- activations (utils/nn.py:118-124) -------------------------------------------------------
__device__ __forceinline__ float pv_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float pv_softplus(float x) {
  // torch.nn.Softplus(beta=1, threshold=20)
  return x > 20.0f ? x : log1pf(expf(x));
}

__device__ __forceinline__ float pv_act_fwd(float x, int act) {
  switch (act) {
    case PV_ACT_TANH: return tanhf(x);
    case PV_ACT_RELU: return x > 0.0f ? x : 0.0f;
    case PV_ACT_LRELU: return x > 0.0f ? x : 0.01f * x;
    case PV_ACT_SOFTPLUS: return pv_softplus(x);
    case PV_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    case PV_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
    default: return x;
  }
}

// derivative of the activation expressed through its output y (and, for GELU, its input).
__device__ __forceinline__ float pv_act_grad(float y, float pre, int act) {
  switch (act) {
    case PV_ACT_TANH: return 1.0f - y * y;
    case PV_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
    case PV_ACT_LRELU: return y > 0.0f ? 1.0f : 0.01f;
    case PV_ACT_SOFTPLUS: return 1.0f - expf(-y);   // sigmoid(x) with y = softplus(x)
    case PV_ACT_GELU:
      return 0.5f * (1.0f + erff(pre * 0.70710678118654752440f)) +
             pre * expf(-0.5f * pre * pre) * 0.39894228040143267794f;
    case PV_ACT_SIGMOID: return y * (1.0f - y);
    default: return 1.0f;
  }
}

// Epilogue helpers.  pv_act_fwd / pv_act_grad inlined per VALUE put every activation's code (tanhf, erff, log1pf ...) behind a
// run-time switch at each of a lane's 16 outputs: ~500 instructions and ~36 branches per value, 6-8 k of a GEMM kernel's 9 k
// instructions.  pv_act_lin: the piecewise-linear activations (none / relu / lrelu) as v > 0 ? v : slope v, branch-free;
// pv_act_pair_slow: ONE out-of-line copy per translation unit of the full switch, act(v) * act_aux'(y, pr).
__device__ __forceinline__ bool pv_act_is_lin(int act) { return act == PV_ACT_NONE || act == PV_ACT_RELU || act == PV_ACT_LRELU; }
__device__ __forceinline__ float pv_act_slope(int act) { return act == PV_ACT_NONE ? 1.0f : act == PV_ACT_RELU ? 0.0f : 0.01f; }
static __device__ __noinline__ float pv_act_pair_slow(float v, int act, bool has_aux, float y, float pr, int act_aux) {
  v = pv_act_fwd(v, act);
  if (has_aux) v *= pv_act_grad(y, pr, act_aux);
  return v;
}

// pv_act_fwd2 / pv_act_grad2: the same functions with the cheap cases inline and the transcendental ones behind one
// out-of-line copy (a call costs less than the inlined switch's branches, and the kernel's code shrinks several-fold)
static __device__ __noinline__ float pv_act_fwd_slow(float x, int act) { return pv_act_fwd(x, act); }
static __device__ __noinline__ float pv_act_grad_slow(float y, float pre, int act) { return pv_act_grad(y, pre, act); }
__device__ __forceinline__ float pv_act_fwd2(float x, int act) {
  if (pv_act_is_lin(act)) return x > 0.0f ? x : x * pv_act_slope(act);
  return pv_act_fwd_slow(x, act);
}
__device__ __forceinline__ float pv_act_grad2(float y, float pre, int act) {
  if (act == PV_ACT_SOFTPLUS || act == PV_ACT_GELU) return pv_act_grad_slow(y, pre, act);
  const float lin = y > 0.0f ? 1.0f : pv_act_slope(act);
  return act == PV_ACT_TANH ? 1.0f - y * y : act == PV_ACT_SIGMOID ? y * (1.0f - y) : lin;
}

// torch.distributions.ContinuousBernoulli(probs = sigmoid(a)).log_prob(x) (utils/prob.py:27) and d(-log_prob)/da.
//   probs -> clamp_probs; logits = log(p) - log1p(-p); log_prob = -BCEWithLogits(logits, x) + log C(p), with
//   log C(p) = log|log1p(-p) - log p| - log|1 - 2p| outside (0.499, 0.501] and its Taylor expansion
//   log 2 + (4/3 + 104/45 u) u, u = (p - 1/2)^2, inside (continuous_bernoulli.py: _cont_bern_log_norm).
__device__ __forceinline__ void pv_cbern(float a, float x, float& ll, float& dlda, float& pr_out) {
  const float eps = 1.1920928955078125e-07f;
  const float pr = 1.0f / (1.0f + expf(-a));
  const float pc = fminf(fmaxf(pr, eps), 1.0f - eps);
  const float lg = logf(pc) - log1pf(-pc);
  const float bce = fmaxf(lg, 0.0f) - lg * x + log1pf(expf(-fabsf(lg)));
  const float t = 1.0f - 2.0f * pc, t2 = t * t;               // log C = log(2 atanh(t) / t)
  float logc, dlogc_da;                                    // d log C / da = (d log C / dp) p (1 - p)
  if (fabsf(t) < 0.3f) {
    // torch's closed forms (log|log1p(-p) - log p| - log|1 - 2p| for the value, -1/L + 2p(1-p)/(1-2p) for the
    // derivative) cancel catastrophically near p = 1/2 — where every pixel of a fresh model sits — so their fp32
    // value depends on the last bit of the log routines (the reference's own fp32 numbers carry that noise, ~1e-6
    // per pixel).  Series of the same functions in t^2 (atanh(t)/t = sum t^2k / (2k+1)), exact to fp32 rounding;
    // inside torch's "unstable region" |p - 1/2| <= 1e-3 they agree with its Taylor branch to 1e-9.
    const float sp = 2.0f / 3 + t2 * (2.0f / 15 + t2 * (2.0f / 35 + t2 * (2.0f / 63 + t2 * (2.0f / 99 + t2 * (2.0f / 143 +
                     t2 * (2.0f / 195 + t2 * (2.0f / 255)))))));
    const float am1 = t2 * (1.0f / 3 + t2 * (1.0f / 5 + t2 * (1.0f / 7 + t2 * (1.0f / 9 + t2 * (1.0f / 11 +
                      t2 * (1.0f / 13 + t2 * (1.0f / 15)))))));
    const float at = 1.0f + am1;
    logc = 0.69314718055994531f + log1pf(am1);
    dlogc_da = -t * sp / (2.0f * at);
  } else {
    const float L = log1pf(-pc) - logf(pc);
    logc = logf(fabsf(L)) - (pc <= 0.5f ? log1pf(-2.0f * pc) : logf(2.0f * pc - 1.0f));
    dlogc_da = -1.0f / L + 2.0f * pc * (1.0f - pc) / t;
  }
  const float mask = (pr >= eps && pr <= 1.0f - eps) ? 1.0f : 0.0f;      // clamp's gradient
  ll = -bce + logc;
  dlda = ((1.0f / (1.0f + expf(-lg)) - x) - dlogc_da) * mask;
  pr_out = pr;
}

// one element of the observation likelihood (fc.py:143-152, prob.py:15-30): a = the decoder's output before the output
// non-linearity, x = the target -> log p(x | a), dL/da of the NEGATIVE ELBO's likelihood term, loc (torch.finfo(float32).eps clamp as clamp_probs)
__device__ __forceinline__ void pv_lik_one(float av, float xv, int lik, int sigmoid_out, float sig, float& ll, float& d, float& lv) {
  if (lik == PV_LIK_BERNOULLI) {
    const float pr = 1.0f / (1.0f + expf(-av));
    const float pc = fminf(fmaxf(pr, 1.1920928955078125e-07f), 1.0f - 1.1920928955078125e-07f);
    const float lg = logf(pc) - log1pf(-pc);
    ll = -(fmaxf(lg, 0.0f) - lg * xv + log1pf(expf(-fabsf(lg))));
    const float mask = (pr >= 1.1920928955078125e-07f && pr <= 1.0f - 1.1920928955078125e-07f) ? 1.0f : 0.0f;
    d = (1.0f / (1.0f + expf(-lg)) - xv) * mask;
    lv = pr;
  } else if (lik == PV_LIK_CBERNOULLI) {
    pv_cbern(av, xv, ll, d, lv);
  } else {
    const float pr = sigmoid_out ? 1.0f / (1.0f + expf(-av)) : av;
    const float df = xv - pr;
    ll = -(df * df) / (2.0f * sig * sig) - logf(sig) - 0.91893853320467274178f;
    d = -df / (sig * sig) * (sigmoid_out ? pr * (1.0f - pr) : 1.0f);
    lv = pr;
  }
}


// v summed over the four 16-lane rows of the wave (lanes r, r + 16, r + 32, r + 48), result in every lane: gfx950's row / half swaps
// (v_permlane16_swap, v_permlane32_swap: plain VALU) instead of two ds_bpermute round trips through the LDS pipeline.  The pairings
// and the order of the two additions are those of `v += shfl_xor(v, 16); v += shfl_xor(v, 32)`: the same bits.
__device__ __forceinline__ float pv_sum_rows(float v) {
  unsigned b = __float_as_uint(v);
  auto s16 = __builtin_amdgcn_permlane16_swap(b, b, false, false);
  v = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
  b = __float_as_uint(v);
  auto s32 = __builtin_amdgcn_permlane32_swap(b, b, false, false);
  return __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
}
// max over the wave of a non-negative value, uniform result: four DPP steps inside each row of 16 lanes (quad xor 1, xor 2, half-row
// mirror, row mirror), then the four rows' values through scalar registers — non-negative floats order like their bit patterns.
// (__shfl_xor compiles to ds_bpermute: six dependent LDS round trips per value, ~700 cycles at the end of every tile.)
__device__ __forceinline__ float pv_wave_max_nonneg(float v) {
  int b = __float_as_int(v);
#define PV_DPP_MAX(CTRL) b = __float_as_int(fmaxf(__int_as_float(b), __int_as_float(__builtin_amdgcn_update_dpp(b, b, (CTRL), 0xF, 0xF, false))))
  PV_DPP_MAX(0xB1);                                   // quad_perm [1,0,3,2]
  PV_DPP_MAX(0x4E);                                   // quad_perm [2,3,0,1]
  PV_DPP_MAX(0x141);                                  // row_half_mirror
  PV_DPP_MAX(0x140);                                  // row_mirror
#undef PV_DPP_MAX
  const unsigned r0 = (unsigned)__builtin_amdgcn_readlane(b, 0), r1 = (unsigned)__builtin_amdgcn_readlane(b, 16);
  const unsigned r2 = (unsigned)__builtin_amdgcn_readlane(b, 32), r3 = (unsigned)__builtin_amdgcn_readlane(b, 48);
  const unsigned m01 = r0 > r1 ? r0 : r1, m23 = r2 > r3 ? r2 : r3;
  return __uint_as_float(m01 > m23 ? m01 : m23);
}
// the butterfly sum over the wave (v += v[lane ^ 32], ^ 16, ^ 8, ^ 4, ^ 2, ^ 1: every lane ends with the total).  __shfl_xor compiles
// to ds_bpermute — six dependent round trips through the LDS pipeline, ~700 cycles in the small latency-bound kernels that call this
// several times in a row.  The same pairs in the same order (the same bits) without the LDS: the half / row swaps of gfx950 for
// 32 and 16, DPP for the rest (row_ror:8 is lane ^ 8 inside a row of 16; lane ^ 4 is row_half_mirror followed by the quad reversal).
__device__ __forceinline__ float pv_wave_sum(float v) {
  unsigned b = __float_as_uint(v);
  auto s32 = __builtin_amdgcn_permlane32_swap(b, b, false, false);
  v = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
  b = __float_as_uint(v);
  auto s16 = __builtin_amdgcn_permlane16_swap(b, b, false, false);
  v = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
#define PV_DPP_F(X, CTRL) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), (CTRL), 0xF, 0xF, false))
  v += PV_DPP_F(v, 0x128);                            // row_ror:8            lane ^ 8
  { const float t = PV_DPP_F(v, 0x141); v += PV_DPP_F(t, 0x1B); }   // row_half_mirror, quad_perm [3,2,1,0]   lane ^ 4
  v += PV_DPP_F(v, 0x4E);                             // quad_perm [2,3,0,1]  lane ^ 2
  v += PV_DPP_F(v, 0xB1);                             // quad_perm [1,0,3,2]  lane ^ 1
#undef PV_DPP_F
  return v;
}

// internal launchers shared between translation units -------------------------------------------
struct PvGemm {
  const float* A; int64_t a_rs, a_cs;   // A(m,k) = A[m*a_rs + k*a_cs]
  const float* B; int64_t b_rs, b_cs;   // B(k,n) = B[k*b_rs + n*b_cs]
  float* C; int64_t ldc;                // C(m,n) = C[m*ldc + n]
  int M, N, K;
  const float* bias;                    // [N] or null
  int act;                              // applied after bias
  float* pre;                           // optional pre-activation store (ldc stride)
  const float* aux; const float* auxpre; int64_t ldaux; int act_aux;   // C *= act'(aux)
  float* rowsumA;                       // optional: rowsumA[m] = sum_k A(m,k)  (bias gradient of a wgrad GEMM)
  // implicit im2col operands (convolutions, kernel 3 / padding 1 / stride 1 over channels-last [B][H][W][C]): the
  // operand is never materialised; element (row = (b, y, x), j = ci*KK + tap) is gathered from the activation.
  //   conv_a: A(m, k) = im2col(A)[row = m][j = k]          (forward / dgrad: rows are the GEMM's M)
  //   conv_b: B(k, n) = im2col(B)[row = k][j = n]          (wgrad: rows are the GEMM's K)
  int conv_a, conv_b;
  int cH, cW, cC, cnd;                  // activation geometry (cnd = 1: W = 1, 3 taps; cnd = 2: 9 taps)
};
// Runs C = epilogue(A*B).  splits > 1 => partial sums through ws (needs splits*M*N floats).
int pv_gemm(const PvGemm& g, int splits, void* ws, int64_t ws_bytes, hipStream_t s);
int pv_gemm_pick_splits(int M, int N, int K);
// up to 4 plain wgrad problems (short contraction, wide output) in one launch, one wave per 16x16 tile (pv_wgrad.hip)
// Adam fused into a gradient-producing launch (pv_ivae_step): the kernel that finalises a gradient element applies
// torch.optim.Adam's update to that element right away (and leaves the zeroed gradient pyro's zero_grads leaves);
// guest workgroups of the same launch update every element the launch does not produce (those were final before it).
struct PvAdamFuse {
  float* p; float* g; float* m; float* v;
  int64_t n;
  float b1, b2, eps, step_size, bc2_sqrt;
};
__device__ __forceinline__ void pv_adam_update(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, int64_t i, float gi, float b1, float b2, float eps,
                                               float step_size, float bc2_sqrt) {
  float mi = m[i], vi = v[i];
  mi = mi + (gi - mi) * (1.0f - b1);                 // exp_avg.lerp_(grad, 1 - beta1)
  vi = vi * b2 + (1.0f - b2) * gi * gi;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - step_size * (mi / denom);            // param.addcdiv_(exp_avg, denom, value=-step_size)
  m[i] = mi; v[i] = vi;
  g[i] = 0.0f;                                       // pyro.infer.util.zero_grads
}
// fin: the step's loss scalars finished by one more guest workgroup of the launch (pv_finish_scalars)
struct PvFinishArgs { const float* llb; int B; float* scalars; const float* kl_part; int n_part; float beta; };
int pv_wgrad_small(const PvGemm* gs, int n, hipStream_t s, const PvAdamFuse* adam = nullptr,
                   const PvFinishArgs* fin = nullptr);
// out[i] = sum_p part[p*stride + i], p ascending (deterministic)
int pv_reduce_partials(const float* part, int nparts, int64_t stride, float* out, int64_t n, hipStream_t s);
// Workgroup barrier for kernels whose waves communicate through LDS only: waits for the wave's LDS traffic, not for
// its outstanding global stores (__syncthreads() fences those too — a ~1 us store round trip per barrier in the
// latency-bound encoder kernels, whose global outputs are consumed by LATER launches only).
__device__ __forceinline__ void pv_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

int pv_colsum(const float* x, int64_t ldx, int64_t M, int N, float* out, void* ws, int64_t ws_bytes, hipStream_t s);
int64_t pv_colsum_ws(int64_t M, int N);
