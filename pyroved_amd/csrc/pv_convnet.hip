// pv_convnet.hip — a stand-alone convolutional stack (nets/conv.py:150-262: FeatureExtractor / Upsampler used outside a
// model) behind the C ABI: forward and backward of the op sequence on the executor of pv_convstack.h, tensors at the
// boundary in the reference's channels-first layout.  The forward leaves every activation in the caller's workspace;
// the backward reads them from there.
#include "pv_common.h"
#include "pv_kernels.h"
#include "pv_convstack.h"

namespace {
using pvcs::Shape;

struct NLayout {
  Shape sh[PV_MAX_OPS + 1];
  float* a[PV_MAX_OPS + 1];
  float* x_nsc; float* out_nsc; float* g[2];
  pvcs::Scratch sc; pvcs::WtPlan wtp; char* wt;
  char* fin_ws; int64_t fin_bytes;
  int64_t total;
};

struct NCarver {
  char* base; int64_t off;
  float* take(int64_t n) {
    float* p = base ? (float*)(base + off) : nullptr;
    off += pv_align_up((n > 0 ? n : 1) * (int64_t)sizeof(float), 256);
    return p;
  }
};

// the conv mode the stack runs with: the plan's, with mode 4 (the cheaper backward) size-gated exactly as in the iVAE and VED
// plans (pv_convstack.h: conv_mode_for) — a direct ABI caller asking for 4 at a small batch gets the three-product backward
static int net_conv_mode(const pv_convnet_plan* p) {
  return pvcs::conv_mode_for(p->conv_bf16, p->ops, p->n_ops, p->ndim, p->batch, p->in_ch, p->in_dim);
}

bool ncarve(const pv_convnet_plan* p, char* base, NLayout& L) {
  if (!p || p->batch <= 0 || (p->ndim != 1 && p->ndim != 2) || p->in_ch < 1 || p->n_ops < 1 || p->n_ops > PV_MAX_OPS) return false;
  // flags: only PV_PLAN_NO_SIDE_STREAM is defined for this plan, and it is what this entry point does anyway (every launch on
  // the caller's stream); any other bit is a caller's mistake
  if (p->flags & ~PV_PLAN_NO_SIDE_STREAM) return false;
  const int cmode = net_conv_mode(p);
  NCarver c{base, 0};
  const int64_t B = p->batch;
  pvcs::Needs nd;
  L.sh[0] = Shape{p->in_dim[0], p->ndim == 2 ? p->in_dim[1] : 1, p->in_ch};
  if (L.sh[0].H < 1 || L.sh[0].W < 1) return false;
  if (!pvcs::stack_shapes(p->ops, p->n_ops, p->ndim, B, L.sh, nd)) return false;
  L.x_nsc = p->in_ch > 1 ? c.take(L.sh[0].elems(B)) : nullptr;
  L.a[0] = nullptr;
  for (int i = 0; i < p->n_ops; ++i) L.a[i + 1] = c.take(L.sh[i + 1].elems(B));
  L.out_nsc = nullptr;
  L.g[0] = c.take(nd.maxact); L.g[1] = c.take(nd.maxact);
  L.sc.col = c.take(nd.maxcol);
  L.sc.bn = c.take(pvcs::bn_floats(nd)); L.sc.bn_maxC = nd.bn_maxC; L.sc.bn_eval = p->bn_eval;
  L.sc.conv_bf16 = cmode;
  pvcs::wt_layout(p->ops, p->n_ops, p->ndim, 0, cmode, p->need_dx != 0, L.wtp);
  L.wt = reinterpret_cast<char*>(c.take((L.wtp.bytes + 3) / 4));
  // (the fused first block has no input gradient: only when the caller will not ask for dL/dx)
  L.sc.code = (nd.code_bytes && !p->need_dx) ? reinterpret_cast<unsigned char*>(c.take((nd.code_bytes + 3) / 4)) : nullptr;
  L.sc.code2 = nd.code2_bytes ? reinterpret_cast<unsigned char*>(c.take((nd.code2_bytes + 3) / 4)) : nullptr;
  L.sc.ws_bytes = pv_align_up(nd.scratch, 256);
  L.sc.ws = base ? (void*)(base + c.off) : nullptr;
  c.off += L.sc.ws_bytes;
  L.fin_bytes = pv_align_up(nd.wg_sum, 256);
  L.fin_ws = base ? base + c.off : nullptr;
  c.off += L.fin_bytes;
  L.sc.wt = L.wt; L.sc.wtp = &L.wtp;
  L.total = c.off;
  return true;
}
}  // namespace

extern "C" int64_t pv_convnet_workspace_bytes(const pv_convnet_plan* p) {
  NLayout L;
  if (!ncarve(p, nullptr, L)) return PV_EINVAL;
  return L.total;
}

extern "C" int pv_convnet_out_shape(const pv_convnet_plan* p, int32_t* out_shape) {
  NLayout L;
  if (!out_shape || !ncarve(p, nullptr, L)) return PV_EINVAL;
  const Shape& o = L.sh[p->n_ops];
  out_shape[0] = o.C; out_shape[1] = o.H;
  if (p->ndim == 2) out_shape[2] = o.W;
  return 0;
}

extern "C" int pv_convnet_forward(const pv_convnet_plan* p, const float* x, float* out, void* stream) {
  PV_RANGE("pv_convnet_forward");
  NLayout L;
  if (!p || !p->params || !p->ws || !x || !out || !ncarve(p, (char*)p->ws, L)) return PV_EINVAL;
  if (p->ws_bytes < L.total) return PV_EWS;
  hipStream_t s = (hipStream_t)stream;
  const int64_t B = p->batch;
  const Shape& s0 = L.sh[0];
  if (p->in_ch > 1) { PV_TRY(pv_ncs_to_nsc(x, L.x_nsc, B, p->in_ch, (int64_t)s0.H * s0.W, s)); x = L.x_nsc; }
  L.a[0] = const_cast<float*>(x);
  PV_TRY(pvcs::wt_prep(p->params, p->ops, p->n_ops, p->ndim, 0, net_conv_mode(p), L.wtp, L.wt, false, s));
  PV_TRY(pvcs::stack_fwd(p->params, p->ops, p->n_ops, p->ndim, (int)B, L.a, L.sh, L.sc, s));
  const Shape& so = L.sh[p->n_ops];
  return pv_nsc_to_ncs(L.a[p->n_ops], out, B, so.C, (int64_t)so.H * so.W, s);
}

extern "C" int pv_convnet_backward(const pv_convnet_plan* p, const float* x, const float* dout, float* dx, void* stream) {
  PV_RANGE("pv_convnet_backward");
  NLayout L;
  if (!p || !p->params || !p->grads || !p->ws || !x || !dout || !ncarve(p, (char*)p->ws, L)) return PV_EINVAL;
  if (p->ws_bytes < L.total) return PV_EWS;
  if (dx && !p->need_dx) return PV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int64_t B = p->batch;
  const Shape& s0 = L.sh[0];
  L.a[0] = p->in_ch > 1 ? L.x_nsc : const_cast<float*>(x);      // (x_nsc still holds the forward's transposed input)
  const Shape& so = L.sh[p->n_ops];
  int pp = 0;
  float* g = L.g[pp];
  PV_TRY(pv_ncs_to_nsc(dout, g, B, so.C, (int64_t)so.H * so.W, s));
  pp ^= 1;
  PV_TRY(pvcs::wt_prep(p->params, p->ops, p->n_ops, p->ndim, 0, net_conv_mode(p), L.wtp, L.wt, true, s));
  float* gout = nullptr;
  PvFinishList fin{};
  fin.base = L.fin_ws; fin.cap = L.fin_bytes;
  L.sc.fin = &fin;
  PV_TRY(pvcs::stack_bwd(p->params, p->grads, p->ops, p->n_ops, p->ndim, (int)B, L.a, L.sh, g, L.g, pp, dx != nullptr, &gout,
                         L.sc, s));
  PV_TRY(pv_wgrad_finish_all(&fin, s));
  if (dx) PV_TRY(pv_nsc_to_ncs(gout, dx, B, p->in_ch, (int64_t)s0.H * s0.W, s));
  return 0;
}
