// pv_ss.hip — the pieces the semi-supervised models (models/ssivae.py, models/ss_reg_ivae.py, trainers/auxsvi.py of the
// reference) add to the iVAE step:
//   * pv_mlp_*: a plain fully-connected network with a softmax (fcClassifierNet, nets/fc.py:240-271) or a linear
//     (fcRegressorNet, nets/fc.py:274-304) output — forward keeping the activations, backward from dloss/d(output);
//   * the small per-sample kernels of the semi-supervised objectives (enumerated-label expectation, auxiliary
//     supervised loss, reparameterised continuous label).
// The encoder_z / decoder part of every step is pv_ivae_loss_and_grads with y = the label vector (given, enumerated or
// sampled), per-sample weights (plan->row_w) and the extra outputs plan->row_elbo / plan->dy.
#include "pv_common.h"
#include "pv_kernels.h"
#include "pv_linear.h"

#define LOG_SQRT_2PI 0.91893853320467274178f

namespace {

struct MlpLayout {
  float* act[PV_MAX_LAYERS]; float* pre[PV_MAX_LAYERS]; float* logits; float* dbuf[2];
  void* scratch; int64_t scratch_bytes; int64_t total;
};

bool mlp_valid(const pv_mlp_plan* p) {
  if (!p || p->batch <= 0 || p->in_dim <= 0 || p->n_layers < 1 || p->n_layers > PV_MAX_LAYERS) return false;
  if (p->out_kind != PV_MLP_LINEAR && p->out_kind != PV_MLP_SOFTMAX) return false;
  int64_t w = p->in_dim;
  for (int i = 0; i < p->n_layers; ++i) {
    if (p->layers[i].in_dim != w || p->layers[i].out_dim <= 0) return false;
    w = p->layers[i].out_dim;
  }
  return p->out.in_dim == w && p->out.out_dim > 0;
}

void mlp_carve(const pv_mlp_plan* p, char* base, MlpLayout& L) {
  int64_t off = 0;
  auto take = [&](int64_t nfloats) {
    float* q = base ? (float*)(base + off) : nullptr;
    off += pv_align_up(nfloats * (int64_t)sizeof(float), 256);
    return q;
  };
  const int64_t B = p->batch;
  int64_t maxw = p->out.out_dim, scratch = 0;
  auto upd = [&](int64_t v) { if (v > scratch) scratch = v; };
  for (int i = 0; i < p->n_layers; ++i) {
    const pv_layer& l = p->layers[i];
    L.act[i] = take(B * l.out_dim);
    L.pre[i] = l.act == PV_ACT_GELU ? take(B * l.out_dim) : nullptr;
    if (l.out_dim > maxw) maxw = l.out_dim;
    upd(gemm_ws_need(B, l.out_dim, l.in_dim));
    upd(gemm_ws_need(l.out_dim, l.in_dim, B));
    upd(gemm_ws_need(B, l.in_dim, l.out_dim));
    upd(pv_colsum_ws(B, l.out_dim));
  }
  upd(gemm_ws_need(B, p->out.out_dim, p->out.in_dim));
  upd(gemm_ws_need(p->out.out_dim, p->out.in_dim, B));
  upd(gemm_ws_need(B, p->out.in_dim, p->out.out_dim));
  upd(pv_colsum_ws(B, p->out.out_dim));
  L.logits = take(B * p->out.out_dim);
  L.dbuf[0] = take(B * maxw);
  L.dbuf[1] = take(B * maxw);
  L.scratch_bytes = pv_align_up(scratch, 256);
  L.scratch = base ? (void*)(base + off) : nullptr;
  off += L.scratch_bytes;
  L.total = off;
}

// dlogits[b][:] = alpha[b][:] * (dalpha[b][:] - sum_k alpha[b][k] dalpha[b][k])   (softmax backward)
__global__ void softmax_bwd_rows_kernel(const float* __restrict__ alpha, const float* __restrict__ dalpha, int B, int K,
                                        float* __restrict__ dlogits) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* a = alpha + (int64_t)b * K;
  const float* d = dalpha + (int64_t)b * K;
  float dot = 0.0f;
  for (int k = 0; k < K; ++k) dot += a[k] * d[k];
  for (int k = 0; k < K; ++k) dlogits[(int64_t)b * K + k] = a[k] * (d[k] - dot);
}

// one workgroup: deterministic sum of per-sample values computed by f(b)
template <class F>
__device__ __forceinline__ float block_total(int B, float* sm, F f) {
  float v = 0.0f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) v += f(b);
  return pv_block_sum(v, sm);
}

// TraceEnum_ELBO over the guide-enumerated label (ssivae.py:172-176, 192-195; trainers/auxsvi.py:73-77): with
// e[k][b] the per-sample ELBO term of the pass that assumed label k,
//   loss = -sum_b sum_k alpha_bk (e_kb + log(1/K) - log alpha_bk);   dalpha_bk = -(e_kb - log K - log alpha_bk - 1)
// scal[0] = loss_add = sum_bk alpha_bk (log alpha_bk + log K)   (the weighted -e part is pv_ivae's scalars[0])
__global__ __launch_bounds__(256) void ss_enum_kernel(const float* __restrict__ alpha, const float* __restrict__ e, int B, int K,
                                                       float* __restrict__ dalpha, float* __restrict__ scal) {
  __shared__ float sm[16];
  const float lK = logf((float)K);
  const float tot = block_total(B, sm, [&](int b) {
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float a = alpha[(int64_t)b * K + k];
      // OneHotCategorical(probs).log_prob: logits = log(clamp(probs)) (torch.distributions.utils.probs_to_logits)
      const float la = logf(fminf(fmaxf(a, 1.1920928955078125e-07f), 1.0f - 1.1920928955078125e-07f));
      acc += a * (la + lK);
      if (dalpha) dalpha[(int64_t)b * K + k] = -(e[(int64_t)k * B + b] - lK - la - 1.0f);
    }
    return acc;
  });
  if (threadIdx.x == 0) scal[0] = tot;
}

// model_aux (ssivae.py:215-228 / ss_reg_ivae.py: same place): loss_aux = -mult * sum_b log p(y_b | encoder_y(x_b))
//   classification: OneHotCategorical(probs = alpha): log p = sum_k y_bk log alpha_bk ; dalpha = -mult y / alpha
//   regression:     Normal(c, sig).to_event(1):       log p = sum_i -(y - c)^2 / (2 sig^2) - log sig - log sqrt(2 pi)
__global__ __launch_bounds__(256) void ss_aux_kernel(int task, const float* __restrict__ out, const float* __restrict__ y, int B,
                                                      int D, float mult, float sig, float* __restrict__ dout,
                                                      float* __restrict__ scal) {
  __shared__ float sm[16];
  const float tot = block_total(B, sm, [&](int b) {
    float acc = 0.0f;
    for (int i = 0; i < D; ++i) {
      const int64_t e = (int64_t)b * D + i;
      if (task == PV_SS_CLASSIFICATION) {
        const float a = out[e];
        const float ac = fminf(fmaxf(a, 1.1920928955078125e-07f), 1.0f - 1.1920928955078125e-07f);
        acc += y[e] * logf(ac);
        if (dout) dout[e] = (a == ac) ? -mult * y[e] / a : 0.0f;       // (the clamp's gradient)
      } else {
        const float d = y[e] - out[e];
        acc += -(d * d) / (2.0f * sig * sig) - logf(sig) - LOG_SQRT_2PI;
        if (dout) dout[e] = -mult * d / (sig * sig);
      }
    }
    return acc;
  });
  if (threadIdx.x == 0) scal[0] = -mult * tot;
}

// ss_reg_iVAE's continuous label (ss_reg_ivae.py:176-181, 196-199): guide y = c + sig*eps (reparameterised), model
// scores it under Normal(0, sig).  mode 0: ys = c + sig*eps.  mode 1 (after the iVAE step returned dloss/dy):
//   loss_add = -sum (log p(y) - log q(y)) = sum y^2/(2 sig^2) - eps^2/2 ;  dc = dy + y / sig^2
// mode 2 (observed label): loss_add = -sum log Normal(y; 0, sig)
__global__ __launch_bounds__(256) void ss_reg_kernel(int mode, const float* __restrict__ c, const float* __restrict__ eps,
                                                      float* __restrict__ ys, const float* __restrict__ dy, int B, int D,
                                                      float sig, float* __restrict__ dc, float* __restrict__ scal) {
  __shared__ float sm[16];
  const float tot = block_total(B, sm, [&](int b) {
    float acc = 0.0f;
    for (int i = 0; i < D; ++i) {
      const int64_t e = (int64_t)b * D + i;
      if (mode == 0) {
        ys[e] = c[e] + sig * eps[e];
      } else if (mode == 1) {
        const float y = ys[e], ep = eps[e];
        // log p(y) = -y^2/(2 sig^2) - log sig - C ; log q(y) = -((y - c)/sig)^2/2 - log sig - C with y - c = sig*eps
        const float d = y - c[e];
        acc += (y * y) / (2.0f * sig * sig) - (d * d) / (2.0f * sig * sig);
        (void)ep;
        dc[e] = dy[e] + y / (sig * sig);
      } else {
        const float y = ys[e];
        acc += (y * y) / (2.0f * sig * sig) + logf(sig) + LOG_SQRT_2PI;
      }
    }
    return acc;
  });
  if (threadIdx.x == 0 && scal) scal[0] = tot;
}

}  // namespace

extern "C" int64_t pv_mlp_workspace_bytes(const pv_mlp_plan* plan) {
  if (!mlp_valid(plan)) return PV_EINVAL;
  MlpLayout L;
  mlp_carve(plan, nullptr, L);
  return L.total;
}

extern "C" int pv_mlp_forward(const pv_mlp_plan* plan, float* out, void* stream) {
  PV_RANGE("pv_mlp_forward");
  if (!mlp_valid(plan) || !plan->params || !plan->x || !plan->ws || !out) return PV_EINVAL;
  MlpLayout L;
  mlp_carve(plan, (char*)plan->ws, L);
  if (plan->ws_bytes < L.total) return PV_EWS;
  hipStream_t s = (hipStream_t)stream;
  const int64_t B = plan->batch;
  const float* in = plan->x;
  int64_t ldin = plan->in_dim;
  const float* P = plan->params;
  for (int i = 0; i < plan->n_layers; ++i) {
    const pv_layer& l = plan->layers[i];
    PV_TRY(linear_fwd(in, ldin, P + l.w_off, l.b_off >= 0 ? P + l.b_off : nullptr, L.act[i], L.pre[i], l.out_dim, B,
                      l.in_dim, l.out_dim, l.act, L.scratch, L.scratch_bytes, s));
    in = L.act[i]; ldin = l.out_dim;
  }
  const pv_layer& o = plan->out;
  float* dst = plan->out_kind == PV_MLP_SOFTMAX ? L.logits : out;
  PV_TRY(linear_fwd(in, ldin, P + o.w_off, o.b_off >= 0 ? P + o.b_off : nullptr, dst, nullptr, o.out_dim, B, o.in_dim,
                    o.out_dim, PV_ACT_NONE, L.scratch, L.scratch_bytes, s));
  if (plan->out_kind == PV_MLP_SOFTMAX) PV_TRY(pv_softmax_rows(L.logits, o.out_dim, (int)B, o.out_dim, out, s));
  return 0;
}

extern "C" int pv_mlp_backward(const pv_mlp_plan* plan, const float* out, const float* dout, void* stream) {
  PV_RANGE("pv_mlp_backward");
  if (!mlp_valid(plan) || !plan->params || !plan->grads || !plan->x || !plan->ws || !dout) return PV_EINVAL;
  if (plan->out_kind == PV_MLP_SOFTMAX && !out) return PV_EINVAL;
  MlpLayout L;
  mlp_carve(plan, (char*)plan->ws, L);
  if (plan->ws_bytes < L.total) return PV_EWS;
  hipStream_t s = (hipStream_t)stream;
  const int64_t B = plan->batch;
  const float* P = plan->params;
  float* G = plan->grads;
  const pv_layer& o = plan->out;
  const int nl = plan->n_layers;
  float* cur = L.dbuf[0];
  float* oth = L.dbuf[1];
  const float* dpre = dout;                         // dL/d(pre-activation) of the output layer
  if (plan->out_kind == PV_MLP_SOFTMAX) {
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, out, dout, (int)B,
                       (int)o.out_dim, cur);
    PV_LAUNCH_CHECK();
    dpre = cur;
    float* t = cur; cur = oth; oth = t;
  }
  const float* hlast = L.act[nl - 1];
  PV_TRY(linear_wgrad(dpre, o.out_dim, hlast, o.in_dim, G + o.w_off, o.b_off >= 0 ? G + o.b_off : nullptr, B, o.in_dim,
                      o.out_dim, L.scratch, L.scratch_bytes, s));
  PV_TRY(linear_dgrad(dpre, o.out_dim, P + o.w_off, cur, o.in_dim, hlast, L.pre[nl - 1], o.in_dim, plan->layers[nl - 1].act,
                      B, o.in_dim, o.out_dim, L.scratch, L.scratch_bytes, s));
  for (int i = nl - 1; i >= 0; --i) {
    const pv_layer& l = plan->layers[i];
    const float* in = i > 0 ? L.act[i - 1] : plan->x;
    const int64_t ldin = l.in_dim;
    PV_TRY(linear_wgrad(cur, l.out_dim, in, ldin, G + l.w_off, l.b_off >= 0 ? G + l.b_off : nullptr, B, l.in_dim, l.out_dim,
                        L.scratch, L.scratch_bytes, s));
    if (i > 0) {
      PV_TRY(linear_dgrad(cur, l.out_dim, P + l.w_off, oth, l.in_dim, L.act[i - 1], L.pre[i - 1], l.in_dim,
                          plan->layers[i - 1].act, B, l.in_dim, l.out_dim, L.scratch, L.scratch_bytes, s));
      float* t = cur; cur = oth; oth = t;
    }
  }
  return 0;
}

extern "C" int pv_ss_enum_terms(const float* alpha, const float* row_elbo, int64_t batch, int32_t n_classes, float* dalpha,
                                float* loss_add, void* stream) {
  if (!alpha || !row_elbo || !loss_add || batch <= 0 || n_classes <= 0) return PV_EINVAL;
  hipLaunchKernelGGL(ss_enum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, alpha, row_elbo, (int)batch, (int)n_classes,
                     dalpha, loss_add);
  PV_LAUNCH_CHECK();
  return 0;
}

extern "C" int pv_ss_aux_loss(int32_t task, const float* out, const float* y, int64_t batch, int32_t dim, float multiplier,
                              float reg_sig, float* dout, float* loss, void* stream) {
  if ((task != PV_SS_CLASSIFICATION && task != PV_SS_REGRESSION) || !out || !y || !loss || batch <= 0 || dim <= 0)
    return PV_EINVAL;
  hipLaunchKernelGGL(ss_aux_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (int)task, out, y, (int)batch, (int)dim,
                     multiplier, reg_sig, dout, loss);
  PV_LAUNCH_CHECK();
  return 0;
}

extern "C" int pv_ss_reg_sample(const float* c, const float* eps, int64_t batch, int32_t dim, float reg_sig, float* ys,
                                void* stream) {
  if (!c || !eps || !ys || batch <= 0 || dim <= 0) return PV_EINVAL;
  hipLaunchKernelGGL(ss_reg_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, 0, c, eps, ys, (const float*)nullptr,
                     (int)batch, (int)dim, reg_sig, (float*)nullptr, (float*)nullptr);
  PV_LAUNCH_CHECK();
  return 0;
}

extern "C" int pv_ss_reg_terms(const float* c, const float* eps, const float* ys, const float* dy, int64_t batch, int32_t dim,
                               float reg_sig, float* dc, float* loss_add, void* stream) {
  if (!ys || !loss_add || batch <= 0 || dim <= 0) return PV_EINVAL;
  const bool sampled = c != nullptr;
  if (sampled && (!eps || !dy || !dc)) return PV_EINVAL;
  hipLaunchKernelGGL(ss_reg_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sampled ? 1 : 2, c, eps, const_cast<float*>(ys),
                     dy, (int)batch, (int)dim, reg_sig, dc, loss_add);
  PV_LAUNCH_CHECK();
  return 0;
}
