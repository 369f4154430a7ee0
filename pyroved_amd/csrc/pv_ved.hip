// pv_ved.hip — host-side orchestration of one SVI step of models.VED behind the C ABI (include/pyroved_amd.h):
//   convEncoderNet(x) -> (mu, sigma) -> z = mu + sigma*eps -> convDecoderNet(z) -> log p(y | z) -> ELBO -> gradients
// (models/ved.py:122-163, nets/conv.py).  The networks arrive as op sequences (pv_op); every convolution is a GEMM
// over channels-last activations (im2col for kernel 3, the activation itself for kernel 1) on the f32-input MFMA
// GEMM of pv_gemm.hip with bias + activation fused; pooling / upsampling / layout changes are the gathers of
// pv_conv.hip.  Backward walks the same sequence in reverse, re-creating each im2col instead of keeping it.
// No allocation, no synchronisation, no retained state: buffers are carved from the caller's workspace.
#include "pv_common.h"
#include "pv_kernels.h"
#include "pv_convstack.h"
#include "pv_side.h"
#include "pv_dec1d.h"
#include <stdlib.h>

namespace {

using pvcs::Shape;

struct VCarver {
  char* base; int64_t off;
  float* take(int64_t n) {
    float* p = base ? (float*)(base + off) : nullptr;
    off += pv_align_up((n > 0 ? n : 1) * (int64_t)sizeof(float), 256);
    return p;
  }
};

struct VLayout {
  Shape es[PV_MAX_OPS + 1], ds[PV_MAX_OPS + 1];       // activation shapes: es[0] = input, ds[0] = decoder seed
  float* ea[PV_MAX_OPS + 1]; float* da[PV_MAX_OPS + 1];
  float* x_nsc; float* y_nsc; float* loc_nsc;
  float* feat; float* head; float* dhead; float* z; float* z_scale; float* dzc;
  float* f0; float* df0;
  const float* head_part = nullptr; int head_nseg = 0;            // the conv head's partial sums when its finish rides in the decoder's launch
  float* llrow; float* dlda; float* llb; float* kl_part;          // kl_part (2 B): per-sample KL terms when the head rides in the decoder's launch
  float* g[2];                                         // gradient ping-pong (largest activation)
  float* dg[PV_MAX_OPS + 1];                           // the decoder's per-op gradients dL/d(da[i]) (kept for the batched weight gradients)
  float* eg[PV_MAX_OPS + 1];                           // the encoder's per-op gradients dL/d(ea[i]) (their weight gradients run on the side stream)
  pvcs::Scratch sc;                                    // im2col / dcol scratch + GEMM split-K scratch
  pvcs::WtPlan wtp; char* wt;                          // the step's tiled conv weights (both stacks, both orientations)
  float* head_wt;                                      // features2latent's weight re-indexed channels-last (null: GEMM path)
  char* fin_ws; int64_t fin_bytes;                     // every weight gradient's partials until the one finish launch
  float* l2f_wt;                                       // latent2features' weight re-indexed channels-last (null: GEMM path)
  float* d1_wt;                                        // the 1-D decoder's weights tiled for its fused launches (pv_dec1d.hip; null: not that shape)
  int64_t F;                                           // flattened feature size C*S of the encoder output
  int64_t total;
};

bool valid_ved(const pv_ved_plan* p) {
  if (!p || p->batch <= 0 || p->z_dim <= 0 || p->z_dim > 256) return false;
  if ((p->ndim_in != 1 && p->ndim_in != 2) || (p->ndim_out != 1 && p->ndim_out != 2)) return false;
  if (p->in_ch < 1 || p->out_ch < 1 || p->n_enc_ops < 1 || p->n_enc_ops > PV_MAX_OPS || p->n_dec_ops < 1 ||
      p->n_dec_ops > PV_MAX_OPS)
    return false;
  if (p->lik != PV_LIK_BERNOULLI && p->lik != PV_LIK_GAUSSIAN && p->lik != PV_LIK_CBERNOULLI) return false;
  if (p->lik != PV_LIK_GAUSSIAN && !p->sigmoid_out) return false;
  if (p->head.out_dim != 2 * p->z_dim || p->l2f.in_dim != p->z_dim) return false;
  return true;
}

// shapes + workspace carving; false on an inconsistent plan
// the conv mode this call runs with: the plan's, except that mode 4 (the cheaper backward) needs gradient sums long enough for
// its one-piece operands' rounding to average out — decided over BOTH stacks (the one mode drives the encoder's and the decoder's
// convolutions: a 1-D encoder in front of a 2-D decoder — spec2im — has no encoder convolution that could trip the gate, and
// the decoder's early 4x4 / 8x8 layers at a small batch are exactly the short sums the gate exists for)
static int ved_conv_mode(const pv_ved_plan* p) {
  const int enc = pvcs::conv_mode_for(p->conv_bf16, p->enc, p->n_enc_ops, p->ndim_in, p->batch, p->in_ch, p->in_dim);
  if (enc != 4) return enc;
  return pvcs::conv_mode_for(p->conv_bf16, p->dec, p->n_dec_ops, p->ndim_out, p->batch, p->dec_c0, p->dec_dim0);
}

bool vcarve(const pv_ved_plan* p, char* base, VLayout& L) {
  VCarver c{base, 0};
  const int64_t B = p->batch, z = p->z_dim;
  pvcs::Needs nd;
  // ---- encoder ----
  L.es[0] = Shape{p->in_dim[0], p->ndim_in == 2 ? p->in_dim[1] : 1, p->in_ch};
  if (!pvcs::stack_shapes(p->enc, p->n_enc_ops, p->ndim_in, B, L.es, nd)) return false;
  const int64_t enc_code2 = nd.code2_bytes;           // (the encoder's fused convolution + max-pool pairs; stack 0 only)
  L.x_nsc = p->in_ch > 1 ? c.take(L.es[0].elems(B)) : nullptr;
  L.ea[0] = nullptr;                                   // = x (or x_nsc), set by the caller
  for (int i = 0; i < p->n_enc_ops; ++i) L.ea[i + 1] = c.take(L.es[i + 1].elems(B));
  const Shape& fe = L.es[p->n_enc_ops];
  L.F = (int64_t)fe.H * fe.W * fe.C;
  if (p->head.in_dim != L.F) return false;
  L.head_wt = pv_convhead_supported(L.F, 2 * p->z_dim) ? c.take(2 * z * L.F) : nullptr;
  if (L.head_wt || !base) pvcs::upd(nd.scratch, pv_convhead_ws((int)B, L.F, 2 * p->z_dim));
  L.feat = c.take(B * L.F);
  L.head = c.take(B * 2 * z); L.dhead = c.take(B * 2 * z);
  L.z = c.take(B * z); L.z_scale = c.take(B * z); L.dzc = c.take(B * z);
  pvcs::upd(nd.scratch, gemm_ws_need(B, 2 * z, L.F)); pvcs::upd(nd.scratch, gemm_ws_need(2 * z, L.F, B));
  pvcs::upd(nd.scratch, gemm_ws_need(B, L.F, 2 * z));
  // ---- decoder ----
  L.ds[0] = Shape{p->dec_dim0[0], p->ndim_out == 2 ? p->dec_dim0[1] : 1, p->dec_c0};
  const int64_t F0 = (int64_t)L.ds[0].H * L.ds[0].W * L.ds[0].C;
  if (p->l2f.out_dim != F0) return false;
  if (!pvcs::stack_shapes(p->dec, p->n_dec_ops, p->ndim_out, B, L.ds, nd)) return false;
  L.f0 = c.take(B * F0); L.df0 = c.take(B * F0);
  L.l2f_wt = pv_l2f_supported(F0, (int)z, L.ds[0].C) ? c.take(z * F0) : nullptr;
  if (L.l2f_wt || !base) pvcs::upd(nd.scratch, pv_convhead_ws((int)B, F0, (int)z));     // (its input gradient = pv_convhead_fwd)
  L.da[0] = c.take(B * F0);
  L.d1_wt = pv_dec1d_supported(p->dec, p->n_dec_ops, p->ndim_out, L.ds[0].H, L.ds[0].C) ? c.take(pv_dec1d_wt_floats(p->dec, p->n_dec_ops)) : nullptr;
  pvcs::upd(nd.scratch, gemm_ws_need(B, F0, z)); pvcs::upd(nd.scratch, gemm_ws_need(F0, z, B));
  pvcs::upd(nd.scratch, gemm_ws_need(B, z, F0));
  for (int i = 0; i < p->n_dec_ops; ++i) L.da[i + 1] = c.take(L.ds[i + 1].elems(B));
  const Shape& od = L.ds[p->n_dec_ops];
  if (od.H != p->out_dim[0] || od.W != (p->ndim_out == 2 ? p->out_dim[1] : 1) || od.C != p->out_ch) return false;
  const int64_t OUT = od.elems(B);
  L.y_nsc = p->out_ch > 1 ? c.take(OUT) : nullptr;
  L.loc_nsc = p->out_ch > 1 ? c.take(OUT) : nullptr;
  L.llrow = c.take(OUT); L.dlda = c.take(OUT); L.llb = c.take(B); L.kl_part = c.take(2 * B);
  L.g[0] = c.take(nd.maxact); L.g[1] = c.take(nd.maxact);
  for (int i = 0; i <= p->n_dec_ops; ++i) L.dg[i] = c.take(L.ds[i].elems(B));
  {
    const bool c1 = pvcs::c1pool_fusable(p->enc, p->n_enc_ops, p->ndim_in, L.es[0]);     // (its backward never writes dL/d(ea[1]))
    L.eg[0] = nullptr;
    for (int i = 1; i <= p->n_enc_ops; ++i) L.eg[i] = (i == 1 && c1) || i == p->n_enc_ops ? nullptr : c.take(L.es[i].elems(B));
  }
  L.sc.col = c.take(nd.maxcol);
  pvcs::wt_layout(p->enc, p->n_enc_ops, p->ndim_in, 0, ved_conv_mode(p), false, L.wtp);
  pvcs::wt_layout(p->dec, p->n_dec_ops, p->ndim_out, 1, ved_conv_mode(p), true, L.wtp);
  L.wt = reinterpret_cast<char*>(c.take((L.wtp.bytes + 3) / 4));
  L.sc.code = nd.code_bytes ? reinterpret_cast<unsigned char*>(c.take((nd.code_bytes + 3) / 4)) : nullptr;
  L.sc.code2 = enc_code2 ? reinterpret_cast<unsigned char*>(c.take((enc_code2 + 3) / 4)) : nullptr;
  L.sc.bn = c.take(pvcs::bn_floats(nd)); L.sc.bn_maxC = nd.bn_maxC; L.sc.bn_eval = p->bn_eval;
  L.sc.conv_bf16 = ved_conv_mode(p);
  L.sc.ws_bytes = pv_align_up(nd.scratch, 256);
  L.sc.ws = base ? (void*)(base + c.off) : nullptr;
  c.off += L.sc.ws_bytes;
  L.fin_bytes = pv_align_up(nd.wg_sum, 256);
  L.fin_ws = base ? base + c.off : nullptr;
  c.off += L.fin_bytes;
  L.total = c.off;
  L.sc.wt = L.wt; L.sc.wtp = &L.wtp;
  return true;
}

// the decoder as one forward / one input-gradient launch: a 1-D stack pv_dec1d.hip takes, its kernel-1 + upsample pairs fused
// the way the recorded weight gradients expect
bool dec1d_active(const pv_ved_plan* p, const VLayout& L) {
  if (!L.d1_wt || (p->flags & PV_PLAN_NO_DEC1D) || !pv_dec1d_enabled() || !pvcs::k1_lean() || !pvcs::k3_lean_1d(p->ndim_out)) return false;
  for (int i = 0; i + 1 < p->n_dec_ops; ++i)
    if (p->dec[i + 1].kind == PV_OP_UPSAMPLE2 && !pvcs::k1up_fusable(p->dec, p->n_dec_ops, p->ndim_out, i)) return false;
  return true;
}

// the head (reparameterised sample, its KL terms, its backward) in the decoder's launches: with latent_to_features there
bool head_folded(const pv_ved_plan* p, const VLayout& L) {
  return dec1d_active(p, L) && L.l2f_wt && pv_dec1d_l2f_ok(p->z_dim) && !pv_exp_str("PV_NO_HEADFOLD");
}

// tile the conv weights the coming launches need: stack 0 / 1 / both, with or without the input-gradient orientation
int ved_wt_prep(const pv_ved_plan* p, VLayout& L, bool enc, bool dec, bool with_dgrad, hipStream_t s) {
  PvWprepEntry e[4 * PV_MAX_OPS + 8];                  // both stacks' tilings: one launch
  int ne = 0;
  if (enc) {
    const Shape& fe = L.es[p->n_enc_ops];
    const PvWprepEntry he = pvcs::head_entry(p->params + p->head.w_off, L.head_wt, 2 * p->z_dim, fe.C, (int64_t)fe.H * fe.W);
    pvcs::wt_entries(p->params, p->enc, p->n_enc_ops, p->ndim_in, 0, ved_conv_mode(p), L.wtp, L.wt, with_dgrad, e, ne, &he,
                     L.head_wt ? 1 : 0);
  }
  if (dec) {
    PvWprepEntry le = pvcs::head_entry(p->params + p->l2f.w_off, L.l2f_wt, p->z_dim, L.ds[0].C, (int64_t)L.ds[0].H * L.ds[0].W);
    le.kind = 7;                                       // wt[k][s*C + c] = w[c*S + s][k]
    if (dec1d_active(p, L)) {                          // (its own tilings instead of the layer kernels')
      if (L.l2f_wt) e[ne++] = le;
      pv_dec1d_wt_entries(p->params, p->dec, p->n_dec_ops, L.d1_wt, e, ne);
    } else {
      pvcs::wt_entries(p->params, p->dec, p->n_dec_ops, p->ndim_out, 1, ved_conv_mode(p), L.wtp, L.wt, with_dgrad, e, ne, &le,
                       L.l2f_wt ? 1 : 0);
    }
  }
  return ne ? pv_conv_wprep_table(e, ne, s) : 0;
}

// encoder forward up to (head, z, z_scale[, KL scalars]); eps == null: inference (z = unused)
int ved_encoder_fwd(const pv_ved_plan* p, VLayout& L, float* z_loc_out, float* z_scale_out, bool with_kl, hipStream_t s) {
  const int64_t B = p->batch;
  const Shape& s0 = L.es[0];
  const float* x = p->x;
  if (p->in_ch > 1) { PV_TRY(pv_ncs_to_nsc(p->x, L.x_nsc, B, p->in_ch, (int64_t)s0.H * s0.W, s)); x = L.x_nsc; }
  L.ea[0] = const_cast<float*>(x);
  if (p->conv_ev_start && p->conv_ev_stop) {          // measurement: events around the heaviest kernel-3 convolution
    double fl = 0.0;
    L.sc.ev_op = pvcs::heaviest_conv(p->enc, p->n_enc_ops, p->ndim_in, (int)B, L.es, &fl);
    L.sc.ev_start = p->conv_ev_start; L.sc.ev_stop = p->conv_ev_stop;
    if (p->conv_ev_flops) *p->conv_ev_flops = fl;
  }
  PV_TRY(pvcs::stack_fwd(p->params, p->enc, p->n_enc_ops, p->ndim_in, (int)B, L.ea, L.es, L.sc, s));
  L.sc.ev_op = -1;
  const Shape& fe = L.es[p->n_enc_ops];
  // torch flattens (C, spatial): features2latent sees channels-first order — the weight is re-indexed, not the features
  L.head_part = nullptr;
  if (L.head_wt && with_kl && head_folded(p, L) && !pv_exp_str("PV_NO_HEADPART") &&
      pv_convhead_fwd_partials(L.ea[p->n_enc_ops], L.head_wt, (int)B, L.F, 2 * p->z_dim, L.sc.ws, L.sc.ws_bytes, s, &L.head_part,
                               &L.head_nseg) == 0) {
    // (the partial sums meet in the decoder's launch, which writes L.head)
  } else if (L.head_wt) {
    PV_TRY(pv_convhead_fwd(L.ea[p->n_enc_ops], L.head_wt, p->head.b_off >= 0 ? p->params + p->head.b_off : nullptr, L.head,
                           (int)B, L.F, 2 * p->z_dim, L.sc.ws, L.sc.ws_bytes, s));
  } else {
    PV_TRY(pv_nsc_to_ncs(L.ea[p->n_enc_ops], L.feat, B, fe.C, (int64_t)fe.H * fe.W, s));
    PV_TRY(linear_fwd(L.feat, L.F, p->params + p->head.w_off, p->head.b_off >= 0 ? p->params + p->head.b_off : nullptr,
                      L.head, nullptr, 2 * p->z_dim, B, L.F, 2 * p->z_dim, PV_ACT_NONE, L.sc.ws, L.sc.ws_bytes, s));
  }
  if (with_kl && head_folded(p, L)) return 0;          // (z is drawn in the decoder's launch: ved_decoder_fwd)
  PvHead h{};
  h.head = L.head; h.eps = with_kl ? p->eps : L.z_scale; h.z = L.z; h.z_scale = L.z_scale;
  h.z_loc_out = z_loc_out; h.z_scale_out = z_scale_out;
  h.scalars = with_kl ? p->scalars : L.dhead;       // (inference: the KL slots land in scratch)
  h.B = (int)B; h.z_dim = p->z_dim; h.beta = with_kl ? p->beta : 0.0f;
  return pv_head_fwd(h, s);
}

// decoder forward from z (B, z_dim) to the logits / pre-sigmoid output in L.da[n_dec_ops]
// arm_fork: the one-launch form carries a fork event (pv_fork_taken() tells the caller whether it ran)
int ved_decoder_fwd(const pv_ved_plan* p, VLayout& L, const float* z, hipStream_t s, const PvD1Lik* lk = nullptr, bool* lik_done = nullptr,
                    bool with_head = false, bool arm_fork = false) {
  const int64_t B = p->batch;
  const Shape& d0 = L.ds[0];
  const int64_t F0 = (int64_t)d0.H * d0.W * d0.C;
  if (L.l2f_wt && dec1d_active(p, L) && pv_dec1d_l2f_ok(p->z_dim)) {       // the Linear rides in the decoder's launch
    const PvD1L2f lf{z, L.l2f_wt, p->l2f.b_off >= 0 ? p->params + p->l2f.b_off : nullptr, nullptr, p->z_dim};
    PvD1Head hd{L.head, p->eps, L.z, L.z_scale, p->z_loc, p->z_scale, L.kl_part, nullptr, 2 * p->z_dim, p->beta};
    if (L.head_part) { hd.part = L.head_part; hd.bias = p->head.b_off >= 0 ? p->params + p->head.b_off : nullptr; hd.head_out = L.head; hd.nseg = L.head_nseg; }
    if (lik_done) *lik_done = lk != nullptr;
    if (arm_fork) pv_fork_arm();
    return pv_dec1d_fwd(p->params, p->dec, p->n_dec_ops, L.d1_wt, (int)B, d0.H, d0.C, L.da, s, &lf, lk,
                        with_head && head_folded(p, L) ? &hd : nullptr);
  }
  if (L.l2f_wt) {                                      // Linear + view(-1, C0, *dims), written channels-last directly
    PV_TRY(pv_l2f_fwd(z, L.l2f_wt, p->l2f.b_off >= 0 ? p->params + p->l2f.b_off : nullptr, L.da[0], (int)B, d0.H * d0.W, d0.C,
                      p->z_dim, s));
  } else {
    PV_TRY(linear_fwd(z, p->z_dim, p->params + p->l2f.w_off, p->l2f.b_off >= 0 ? p->params + p->l2f.b_off : nullptr,
                      L.f0, nullptr, F0, B, p->z_dim, F0, PV_ACT_NONE, L.sc.ws, L.sc.ws_bytes, s));
    PV_TRY(pv_ncs_to_nsc(L.f0, L.da[0], B, d0.C, (int64_t)d0.H * d0.W, s));     // view(-1, C0, *dims) -> channels-last
  }
  if (dec1d_active(p, L)) {
    if (lik_done) *lik_done = lk != nullptr;
    return pv_dec1d_fwd(p->params, p->dec, p->n_dec_ops, L.d1_wt, (int)B, d0.H, d0.C, L.da, s, nullptr, lk);
  }
  return pvcs::stack_fwd(p->params, p->dec, p->n_dec_ops, p->ndim_out, (int)B, L.da, L.ds, L.sc, s, 1);
}

}  // namespace

extern "C" int64_t pv_ved_workspace_bytes(const pv_ved_plan* plan) {
  if (!valid_ved(plan)) return PV_EINVAL;
  VLayout L;
  if (!vcarve(plan, nullptr, L)) return PV_EINVAL;
  return L.total;
}

// test hook (not in include/; pv_convstack.h: conv_trace): the encoder stack's stored activations and max-pool winners
extern "C" int pv_debug_ved_conv_trace(const pv_ved_plan* p, int64_t* out) {
  if (!valid_ved(p) || !p->ws || !out) return PV_EINVAL;
  VLayout L;
  if (!vcarve(p, (char*)p->ws, L) || p->ws_bytes < L.total) return PV_EINVAL;
  pvcs::conv_trace(p->enc, p->n_enc_ops, p->ndim_in, p->batch, L.es, L.ea, L.sc.code, L.sc.code2, (const char*)p->ws, out);
  return 0;
}

extern "C" int pv_ved_loss_and_grads(const pv_ved_plan* p, int want_grads, void* stream) {
  PV_RANGE("pv_ved_loss_and_grads");
  if (!valid_ved(p) || !p->params || !p->x || !p->y || !p->eps || !p->scalars || !p->ws) return PV_EINVAL;
  if (want_grads && !p->grads) return PV_EINVAL;
  VLayout L;
  if (!vcarve(p, (char*)p->ws, L)) return PV_EINVAL;
  if (p->ws_bytes < L.total) return PV_EWS;
  hipStream_t s = (hipStream_t)stream;
  const int64_t B = p->batch, z = p->z_dim;
  pv_fork_disarm();                                    // (no fork state of an earlier, failed call)
  PvSideJoin sj;                                       // joins the side stream on every return path
  // The step's weight tilings run on the side stream next to the fused first block (which reads the raw weights); the
  // encoder's stack joins before its first tiled convolution.
  hipStream_t side = pv_side_stream_for(s, p->flags);
  bool wt_join = false;
  static const int wprep_side = pv_exp_int("PV_SIDE_WPREP", 0) ? 1 : 0;   // (measured: the join costs more than the overlap returns)
  if (wprep_side && side && pvcs::c1pool_fusable(p->enc, p->n_enc_ops, p->ndim_in, L.es[0]) && L.sc.code) {
    PV_TRY(pv_stream_after(side, s));                  // (the previous step's Adam wrote the weights on s)
    PV_TRY(ved_wt_prep(p, L, true, true, want_grads != 0, side));
    wt_join = true;
    L.sc.side = side; L.sc.wt_join = &wt_join;
  } else {
    PV_TRY(ved_wt_prep(p, L, true, true, want_grads != 0, s));
  }
  PV_TRY(ved_encoder_fwd(p, L, p->z_loc, p->z_scale, true, s));
  if (wt_join) { wt_join = false; PV_TRY(pv_stream_after(s, side)); }    // (a stack that never joined)
  L.sc.wt_join = nullptr; L.sc.side = nullptr;
  // ---- decoder + likelihood of the target (ved.py:141-145); one output channel: the likelihood rides in the decoder's launch ----
  const Shape& od = L.ds[p->n_dec_ops];
  const int64_t OUT = od.elems(B), per = OUT / B, S = (int64_t)od.H * od.W;
  const PvD1Lik lk{p->y, p->loc, want_grads ? L.dlda : nullptr, L.llb, p->lik, p->sigmoid_out, p->decoder_sig};
  bool lik_done = false;
  // a step with gradients and a side stream: the loss scalars are summed there, next to the backward's first launch
  static const int fin_side_env = pv_exp_int("PV_FIN_SIDE", 1);
  hipStream_t side_f = (want_grads != 0 && fin_side_env != 0) ? pv_side_stream_for(s, p->flags) : nullptr;
  PV_TRY(ved_decoder_fwd(p, L, L.z, s, p->out_ch == 1 ? &lk : nullptr, &lik_done, true, side_f != nullptr));
  const float* y = p->y;
  if (!lik_done && p->out_ch > 1) { PV_TRY(pv_ncs_to_nsc(p->y, L.y_nsc, B, p->out_ch, S, s)); y = L.y_nsc; }
  float* loc = p->loc ? (p->out_ch > 1 ? L.loc_nsc : p->loc) : nullptr;
  if (lik_done) {
    // (nothing to launch)
  } else if (B >= 64 || per <= 4096) {                 // one workgroup per sample: element terms and their sum in one launch
    PV_TRY(pv_lik_rows(L.da[p->n_dec_ops], y, B, per, p->lik, p->sigmoid_out, p->decoder_sig, loc, want_grads ? L.dlda : nullptr,
                       L.llb, s));
    if (p->loc && p->out_ch > 1) PV_TRY(pv_nsc_to_ncs(L.loc_nsc, p->loc, B, p->out_ch, S, s));
  } else {
    PV_TRY(pv_lik_elem(L.da[p->n_dec_ops], y, OUT, p->lik, p->sigmoid_out, p->decoder_sig, loc, L.llrow,
                       want_grads ? L.dlda : nullptr, s));
    if (p->loc && p->out_ch > 1) PV_TRY(pv_nsc_to_ncs(L.loc_nsc, p->loc, B, p->out_ch, S, s));
    PV_TRY(pv_segsum(L.llrow, B, per, L.llb, s));
  }
  hipStream_t sf = s;
  if (side_f && lik_done && pv_fork_taken()) { PV_TRY(pv_fork_to(side_f, s)); sj.fork(s, side_f); sf = side_f; }
  else pv_fork_disarm();
  if (head_folded(p, L)) PV_TRY(pv_finish_scalars(L.llb, (int)B, p->scalars, L.kl_part, (int)B, 1.0f /* partials come scaled */, sf));
  else PV_TRY(pv_finish_scalars(L.llb, (int)B, p->scalars, nullptr, 0, p->beta, sf));
  if (!want_grads) return 0;

  PvFinishList fin{};                                  // the conv stacks' weight-gradient reductions: one launch at the end
  fin.base = L.fin_ws; fin.cap = L.fin_bytes;
  L.sc.fin = &fin;
  // ---- backward: decoder ops in reverse ----
  float* g = nullptr;                                  // (dlda = dL/d(output of the last op), loss = -ELBO)
  int pp = 0;
  // the decoder's register-fed weight gradients (kernel-1 family, Conv1d kernel 3) are recorded and run as ONE launch after
  // the input-gradient chain: every layer keeps its own gradient buffer (L.dg) until then.  PV_NO_K1BATCH=1: one launch each.
  static const int k1b_env = pv_exp_int("PV_NO_K1BATCH", 0) ? 0 : 1;
  PvK1Batch k1b{};
  if (k1b_env) fin.k1b = &k1b;
  hipStream_t side2 = k1b_env ? pv_side_stream_for(s, p->flags) : nullptr;      // (k1b_env: every decoder gradient has its own buffer)
  bool dz_done = false, head_done = false;
  if (k1b_env && dec1d_active(p, L)) {
    // every input gradient of the decoder in one launch (the fork event rides on it), then the weight gradients are recorded
    if (side2) pv_fork_arm();
    dz_done = L.l2f_wt && pv_dec1d_l2f_ok(p->z_dim);   // the latent gradient rides in the same launch
    const PvD1L2f lf{nullptr, L.l2f_wt, nullptr, L.dzc, p->z_dim};
    head_done = dz_done && head_folded(p, L);          // ... and the head's backward
    const PvD1Head hd{L.head, p->eps, L.z, L.z_scale, nullptr, nullptr, nullptr, L.dhead, 2 * p->z_dim, p->beta};
    PV_TRY(pv_dec1d_bwd(p->dec, p->n_dec_ops, L.d1_wt, (int)B, L.ds[0].H, L.ds[0].C, L.da, L.dlda, L.dg, s, dz_done ? &lf : nullptr,
                        head_done ? &hd : nullptr));
    PV_TRY(pvcs::stack_wgrads(p->params, p->grads, p->dec, p->n_dec_ops, p->ndim_out, (int)B, L.da, L.ds, L.dlda, L.dg, L.sc, s, 1));
    g = L.dg[0];
  } else {
    L.sc.fork_after = side2 != nullptr;
    L.sc.side = side2;                                 // (chunks of the recorded weight gradients run next to the chain)
    PV_TRY(pvcs::stack_bwd(p->params, p->grads, p->dec, p->n_dec_ops, p->ndim_out, (int)B, L.da, L.ds, L.dlda, L.g, pp, true,
                           &g, L.sc, s, 1, false, k1b_env ? L.dg : nullptr));
    L.sc.fork_after = false; L.sc.side = nullptr;
  }
  // The recorded decoder weight gradients and everything else off the dependent chain from here on (latent_to_features'
  // weight gradient, the encoder's kernel-3 weight gradients) go to the side stream; the chain — latent gradient, head,
  // the encoder's input gradients — stays on s.  Joined before the finish.
  const bool two = side2 != nullptr;
  side = side2;
  if (two) sj.fork(s, side);
  // the decoder's recorded weight gradients on a stream of their own: the encoder's kernel-3 weight gradients then start as
  // soon as their dL/dy exists instead of queueing behind them (round 5, `gpurun_out/r05y`: VED at batch 256 0.806-0.812 ->
  // 0.792 ms; PV_K1_STREAM3=0 in the experiments build: both families on the one side stream)
  static const int k1_own_env = pv_exp_int("PV_K1_STREAM3", 1);
  hipStream_t side3 = (two && k1_own_env) ? pv_side_stream2() : nullptr;
  struct Join3 { hipStream_t m, s3; ~Join3() { if (s3) (void)pv_stream_after(m, s3); } } j3{s, side3};
  hipStream_t sw = two ? (side3 ? side3 : side) : s;
  if (two) PV_TRY(pv_fork_to(side, s, side3));         // after the chain's last launch (its stop event when it took one)
  PV_TRY(pv_k1_wgrad_flush(&k1b, sw));
  fin.k1b = nullptr;
  for (int k = 0; k < fin.n; ++k) fin.st[k] = sw;      // (recorded on s, written by the launch above)
  const Shape& d0 = L.ds[0];
  const int64_t F0 = (int64_t)d0.H * d0.W * d0.C;
  if (L.l2f_wt) {                                      // straight from the channels-last gradient
    PV_TRY(pv_l2f_wgrad(g, L.z, p->grads + p->l2f.w_off, p->l2f.b_off >= 0 ? p->grads + p->l2f.b_off : nullptr, (int)B,
                        d0.H * d0.W, d0.C, (int)z, sw));
    if (!dz_done) PV_TRY(pv_convhead_fwd(g, L.l2f_wt, nullptr, L.dzc, (int)B, F0, (int)z, L.sc.ws, L.sc.ws_bytes, s));
  } else {
    PV_TRY(pv_nsc_to_ncs(g, L.df0, B, d0.C, (int64_t)d0.H * d0.W, s));
    PV_TRY(linear_wgrad(L.df0, F0, L.z, z, p->grads + p->l2f.w_off, p->l2f.b_off >= 0 ? p->grads + p->l2f.b_off : nullptr, B,
                        z, F0, L.sc.ws, L.sc.ws_bytes, s));
    PV_TRY(linear_dgrad(L.df0, F0, p->params + p->l2f.w_off, L.dzc, z, nullptr, nullptr, 0, PV_ACT_NONE, B, z, F0,
                        L.sc.ws, L.sc.ws_bytes, s));
  }
  // ---- reparameterised sample + sampled KL -> head ----
  PvHeadBwd hb{};
  hb.dzc = L.dzc; hb.ldzc = z; hb.z = L.z; hb.z_scale = L.z_scale; hb.eps = p->eps; hb.head = L.head; hb.dhead = L.dhead;
  hb.B = (int)B; hb.z_dim = (int)z; hb.coord_dim = 0; hb.beta = p->beta;
  if (!head_done) PV_TRY(pv_head_bwd(hb, s));
  const Shape& fe = L.es[p->n_enc_ops];
  const pv_op& last = p->enc[p->n_enc_ops - 1];
  bool g_is_pre = false;
  if (L.head_wt) {
    // dL/d(features) straight in channels-last order, with the last convolution's activation derivative folded in
    const bool fold = last.kind == PV_OP_CONV && last.act != PV_ACT_GELU;
    // the head's weight gradient needs dhead only: when the decoder's backward launch wrote it (the side streams already wait for
    // that launch) it runs on the side stream next to the input gradient below instead of in front of it on the chain
    static const int head_side_env = pv_exp_int("PV_VED_HEAD_SIDE", 1);
    hipStream_t hs = (two && head_done && head_side_env && !pv_convhead_wgrad_uses_ws()) ? side : s;
    PV_TRY(pv_convhead_wgrad(L.dhead, L.ea[p->n_enc_ops], p->grads + p->head.w_off,
                             p->head.b_off >= 0 ? p->grads + p->head.b_off : nullptr, (int)B, fe.H * fe.W, fe.C, (int)(2 * z),
                             L.sc.ws, L.sc.ws_bytes, hs));
    g = L.g[pp];
    if (two) pv_fork_arm();                            // (the last convolution's weight gradient forks off this launch)
    PV_TRY(pv_convhead_bwd(L.dhead, L.head_wt, L.ea[p->n_enc_ops], fold ? last.act : PV_ACT_NONE, g, (int)B, L.F, (int)(2 * z), s));
    pp ^= 1;
    g_is_pre = fold;
  } else {
    PV_TRY(linear_wgrad(L.dhead, 2 * z, L.feat, L.F, p->grads + p->head.w_off,
                        p->head.b_off >= 0 ? p->grads + p->head.b_off : nullptr, B, L.F, 2 * z, L.sc.ws, L.sc.ws_bytes, s));
    float* dfeat = L.g[pp];
    PV_TRY(linear_dgrad(L.dhead, 2 * z, p->params + p->head.w_off, dfeat, L.F, nullptr, nullptr, 0, PV_ACT_NONE, B, L.F,
                        2 * z, L.sc.ws, L.sc.ws_bytes, s));
    pp ^= 1;
    g = L.g[pp];
    PV_TRY(pv_ncs_to_nsc(dfeat, g, B, fe.C, (int64_t)fe.H * fe.W, s));
    pp ^= 1;
  }
  // ---- encoder ops in reverse (no input gradient for the first one) ----
  bool joined = false;
  L.sc.side = two ? side : nullptr; L.sc.side_joined = &joined;
  PV_TRY(pvcs::stack_bwd(p->params, p->grads, p->enc, p->n_enc_ops, p->ndim_in, (int)B, L.ea, L.es, g, L.g, pp, false,
                         nullptr, L.sc, s, 0, g_is_pre, two ? L.eg : nullptr));
  L.sc.side = nullptr; L.sc.side_joined = nullptr;
  if (two && !joined) PV_TRY(pv_stream_after(s, side));
  else if (!two && sf != s) PV_TRY(pv_stream_after(s, sf));   // (no second fork: the loss scalars alone ran on the side stream)
  sj.joined();
  if (side3) { PV_TRY(pv_stream_after(s, side3)); j3.s3 = nullptr; }
  return pv_wgrad_finish_all(&fin, s);
}

extern "C" int pv_ved_encode(const pv_ved_plan* p, float* z_loc, float* z_scale, void* stream) {
  PV_RANGE("pv_ved_encode");
  if (!valid_ved(p) || !p->params || !p->x || !p->ws || !z_loc || !z_scale) return PV_EINVAL;
  VLayout L;
  if (!vcarve(p, (char*)p->ws, L)) return PV_EINVAL;
  if (p->ws_bytes < L.total) return PV_EWS;
  PV_TRY(ved_wt_prep(p, L, true, false, false, (hipStream_t)stream));
  return ved_encoder_fwd(p, L, z_loc, z_scale, false, (hipStream_t)stream);
}

extern "C" int pv_ved_decode(const pv_ved_plan* p, const float* z, float* loc, void* stream) {
  PV_RANGE("pv_ved_decode");
  if (!valid_ved(p) || !p->params || !p->ws || !z || !loc) return PV_EINVAL;
  VLayout L;
  if (!vcarve(p, (char*)p->ws, L)) return PV_EINVAL;
  if (p->ws_bytes < L.total) return PV_EWS;
  hipStream_t s = (hipStream_t)stream;
  const int64_t B = p->batch;
  PV_TRY(ved_wt_prep(p, L, false, true, false, s));
  PV_TRY(ved_decoder_fwd(p, L, z, s));
  const Shape& od = L.ds[p->n_dec_ops];
  const int64_t OUT = od.elems(B), S = (int64_t)od.H * od.W;
  float* out = p->out_ch > 1 ? L.loc_nsc : loc;
  // the output non-linearity only (no likelihood): reuse lik_elem's `loc` output
  PV_TRY(pv_lik_elem(L.da[p->n_dec_ops], L.da[p->n_dec_ops], OUT, PV_LIK_GAUSSIAN, p->sigmoid_out, 1.0f, out, nullptr,
                     nullptr, s));
  if (p->out_ch > 1) PV_TRY(pv_nsc_to_ncs(L.loc_nsc, loc, B, p->out_ch, S, s));
  return 0;
}
