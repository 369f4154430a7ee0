// pv_plan.hip — host-side orchestration of one iVAE SVI step behind the C ABI (include/pyroved_amd.h).
// Enqueues, on the caller's stream, the kernel sequence that replaces
//   SVItrainer.train's `self.svi.step(x)` (trainers/svi.py:104-113)
//     -> Trace_ELBO.loss_and_grads(iVAE.model, iVAE.guide) (models/ivae.py:165-221)
//     -> pyro.optim.Adam -> zero_grads.
// No allocation, no synchronisation, no retained state: everything lives in the caller's plan.
#include "pv_common.h"
#include <algorithm>
#include <stdlib.h>
#include "pv_kernels.h"
#include "pv_sdec_fused.h"
#include "pv_linear.h"
#include "pv_convstack.h"
#include "pv_side.h"
#include "pv_conv.h"

namespace {

struct Carver {
  char* base;
  int64_t off;
  float* take(int64_t nfloats) {
    float* p = base ? (float*)(base + off) : nullptr;
    off += pv_align_up(nfloats * (int64_t)sizeof(float), 256);
    return p;
  }
};

}  // namespace

int64_t gemm_ws_need(int64_t M, int64_t N, int64_t K) {
  const int s = pv_gemm_pick_splits((int)M, (int)N, (int)K);
  return s > 1 ? (int64_t)s * M * (N + 1) * (int64_t)sizeof(float) : 0;     // + row-sum partials
}

namespace {

struct Layout {
  // encoder
  float* xin; float* eact[PV_MAX_LAYERS]; float* epre[PV_MAX_LAYERS];
  float* head; float* dhead; float* z; float* z_scale; float* tp; float* zy;
  float* edp[PV_MAX_LAYERS];               // dL/d(pre-activation) of every encoder hidden layer
  bool enc_compact; float* kl_part; int kl_blocks;   // compact encoder kernels (pv_encoder.hip)
  unsigned* enc_flags;                               // ... their merged launch's tile flags (8 per row block; any content)
  unsigned* coop_flags;                              // the shared first layer's hand-off tags (PvEncFold::coop_flags; any content)
  // decoder
  float* hz; float* h0; float* dact[PV_MAX_LAYERS]; float* dpre_[PV_MAX_LAYERS];
  float* logits; float* llrow; float* llb; float* dbuf[2];
  float* part_dwo; float* part_dbo; float* part_hz; float* part_wc; float* part_tp;
  float* dhz; float* dtp; float* dzc;
  float* row_ll;                           // (B) unweighted ll_b (plan->row_elbo)
  float* alpha; float* sw; float* llkb;    // jiVAE: class probabilities (B, K), decoder row weights (K*B), ll per (k, b)
  float* jfix;                              // jiVAE without enumeration: correction of the guide's discrete log-prob sum
  // convolutional encoder (plan->n_enc_ops > 0): activation shapes / buffers, flattened features, gradient ping-pong
  bool enc_ext;                            // external encoder: (z_loc, z_scale) given, gradients handed back
  bool enc_conv; pvcs::Shape ces[PV_MAX_OPS + 1]; float* cea[PV_MAX_OPS + 1]; float* cfeat; float* cg[2];
  float* ceg[PV_MAX_OPS + 1];                              // per-op gradients dL/d(cea[i]) (the weight gradients run on the side stream)
  float* ccol; int64_t cF; float* cbn; int cbn_maxC;
  pvcs::WtPlan cwtp; char* cwt;                            // the step's tiled conv-encoder weights
  unsigned char* ccode;                                    // max-pool winners of the fused first block
  unsigned char* ccode2;                                   // ... of the convolution + max-pool pairs further up
  float* chead_wt;                                         // the head's weight re-indexed channels-last (null: GEMM path)
  char* cfin_ws; int64_t cfin_bytes;                       // the conv encoder's weight-gradient partials until the one finish launch
  void* scratch; int64_t scratch_bytes;    // split-K partials / colsum partials (used by one call at a time)
  int64_t rows;                            // decoder rows: B*N (spatial) or B (vanilla)
  int nchunk, rows_per_chunk;
  // fused persistent decoder path
  bool fused; int f_grid, f_kmax;
  float* f_part; float* f_part_hz; float* f_rowtp; float* f_wimg; float* f_park;
  int64_t total;
};

// jiVAE (discrete_dim = K > 0): K decoder samples per input, ordered [k][b]; head = [mu | softplus input | class logits]
static inline int64_t plan_K(const pv_ivae_plan* p) { return p->discrete_dim > 0 ? p->discrete_dim : 0; }
// conv mode of the convolutional encoder (pv_convstack.h: 0 fp32-class fp16 pieces, 1 mixed, 2 fp32-class three bf16 pieces,
// 3 one fp16 piece).  fused == 3 is the throughput precision; conv_wide (a weight left fp16's range) selects the range-free
// bf16 forms of either precision
static inline int plan_conv_mode(const pv_ivae_plan* p) {
  if (p->fused == 3) return p->conv_wide ? 1 : 3;
  if (p->conv_wide) return 2;
  if (p->flags & PV_PLAN_CONV_X3) return 0;
  // round 5: the cheaper backward (pv_convstack.h mode 4) where the stack's gradient sums are long enough
  return pvcs::conv_mode_for(4, p->enc_ops, p->n_enc_ops, p->enc_ndim, p->batch, 1, p->enc_in_dim);
}
static inline int64_t plan_S(const pv_ivae_plan* p) { return (plan_K(p) > 0 ? plan_K(p) : 1) * (int64_t)p->batch; }
static inline int64_t plan_head_w(const pv_ivae_plan* p) { return 2 * (int64_t)p->z_dim + plan_K(p); }
static inline int64_t plan_lat_in(const pv_ivae_plan* p) {
  return (p->coord_dim > 0 ? p->latent_dim : p->z_dim) + p->c_dim + plan_K(p);
}

bool valid_plan(const pv_ivae_plan* p) {
  if (!p || p->batch <= 0 || p->n_pix <= 0 || p->z_dim <= 0) return false;
  if (p->coord_dim < 0 || p->coord_dim > 2) return false;
  if (p->n_enc_ops < 0 || p->n_enc_ops > PV_MAX_OPS) return false;
  if (p->n_enc_ops > 0 && (p->c_dim != 0 || p->discrete_dim != 0 || (p->enc_ndim != 1 && p->enc_ndim != 2))) return false;
  if (p->ext_encoder && (p->c_dim != 0 || p->discrete_dim != 0)) return false;
  if (!p->ext_encoder && p->n_enc_ops == 0 && (p->n_enc < 1 || p->n_enc > PV_MAX_LAYERS)) return false;
  if (p->n_dec < 1 || p->n_dec > PV_MAX_LAYERS) return false;
  if (p->discrete_dim < 0 || (!p->ext_encoder && p->head.out_dim != plan_head_w(p))) return false;
  if (p->discrete_dim > 0 && p->c_dim != 0) return false;                            // jiVAE: no conditioning vector
  if ((p->row_w || p->row_elbo || p->dy) && (p->discrete_dim > 0 || p->n_enc_ops > 0 || p->ext_encoder)) return false;
  if (p->dy && p->c_dim == 0) return false;
  if (p->lik != PV_LIK_BERNOULLI && p->lik != PV_LIK_GAUSSIAN && p->lik != PV_LIK_CBERNOULLI) return false;
  if (p->lik != PV_LIK_GAUSSIAN && !p->sigmoid_out) return false;   // probs outside (0,1): unsupported
  if (p->coord_dim > 0 && p->out.out_dim != 1) return false;
  if (p->coord_dim == 0 && p->out.out_dim != p->n_pix) return false;
  if (p->dec_kernel != 0 && !pv_sdec_fused_sel_valid(p->fused, p->dec_kernel)) return false;
  return true;
}

// inference_only: encode / decode process B samples (jiVAE's K-fold enumeration exists only in the training step)
// encode_only (with inference_only): none of the decoder's per-row buffers is touched, so they get no room
void carve(const pv_ivae_plan* p, char* base, Layout& L, bool inference_only = false, bool encode_only = false) {
  Carver c{base, 0};
  const int64_t B = p->batch, N = encode_only ? 0 : p->n_pix, z = p->z_dim;
  const int64_t lat_in = plan_lat_in(p), K = plan_K(p), S = inference_only ? p->batch : plan_S(p), hw = plan_head_w(p);
  L.rows = p->coord_dim > 0 ? S * N : S;
  const int64_t R = L.rows;
  L.xin = p->c_dim > 0 ? c.take(B * (p->n_pix + p->c_dim)) : nullptr;
  int64_t maxe = 0;
  L.enc_ext = p->ext_encoder != 0;
  L.enc_conv = !L.enc_ext && p->n_enc_ops > 0;
  const int n_enc = (L.enc_conv || L.enc_ext) ? 0 : p->n_enc;
  pvcs::Needs cnd;
  for (auto& e : L.ceg) e = nullptr;
  L.cfeat = L.cg[0] = L.cg[1] = L.ccol = L.cbn = nullptr; L.cF = 0; L.cbn_maxC = 0; L.cwt = nullptr; L.ccode = nullptr; L.ccode2 = nullptr; L.chead_wt = nullptr; L.cfin_ws = nullptr; L.cfin_bytes = 0;
  if (L.enc_conv) {
    L.ces[0] = pvcs::Shape{p->enc_in_dim[0], p->enc_ndim == 2 ? p->enc_in_dim[1] : 1, 1};
    if ((int64_t)L.ces[0].H * L.ces[0].W == p->n_pix && pvcs::stack_shapes(p->enc_ops, p->n_enc_ops, p->enc_ndim, B, L.ces, cnd)) {
      L.cea[0] = nullptr;
      for (int i = 0; i < p->n_enc_ops; ++i) L.cea[i + 1] = c.take(L.ces[i + 1].elems(B));
      const pvcs::Shape& fe = L.ces[p->n_enc_ops];
      L.cF = (int64_t)fe.H * fe.W * fe.C;
      L.cfeat = c.take(B * L.cF);
      if (pv_convhead_supported(L.cF, p->head.out_dim)) {
        L.chead_wt = c.take((int64_t)p->head.out_dim * L.cF);
        pvcs::upd(cnd.scratch, pv_convhead_ws((int)B, L.cF, p->head.out_dim));
      }
      L.cg[0] = c.take(cnd.maxact); L.cg[1] = c.take(cnd.maxact);
      if (!inference_only) {
        const bool c1 = pvcs::c1pool_fusable(p->enc_ops, p->n_enc_ops, p->enc_ndim, L.ces[0]);   // (its backward never writes dL/d(cea[1]))
        for (int i = 1; i < p->n_enc_ops; ++i) L.ceg[i] = (i == 1 && c1) ? nullptr : c.take(L.ces[i].elems(B));
      }
      L.ccol = c.take(cnd.maxcol);
      pvcs::wt_layout(p->enc_ops, p->n_enc_ops, p->enc_ndim, 0, plan_conv_mode(p), false, L.cwtp);
      L.cwt = reinterpret_cast<char*>(c.take((L.cwtp.bytes + 3) / 4));
      L.ccode = cnd.code_bytes ? reinterpret_cast<unsigned char*>(c.take((cnd.code_bytes + 3) / 4)) : nullptr;
      L.ccode2 = cnd.code2_bytes ? reinterpret_cast<unsigned char*>(c.take((cnd.code2_bytes + 3) / 4)) : nullptr;
      L.cbn = c.take(pvcs::bn_floats(cnd)); L.cbn_maxC = cnd.bn_maxC;
    } else {
      L.cF = -1;                                   // inconsistent op sequence: rejected by the entry points
    }
  }
  for (int i = 0; i < n_enc; ++i) {
    L.eact[i] = c.take(B * p->enc[i].out_dim);
    L.epre[i] = p->enc[i].act == PV_ACT_GELU ? c.take(B * p->enc[i].out_dim) : nullptr;
    if (p->enc[i].out_dim > maxe) maxe = p->enc[i].out_dim;
  }
  L.head = c.take(B * hw);
  L.dhead = c.take(B * hw);
  L.z = c.take(B * z);
  L.z_scale = c.take(B * z);
  L.tp = c.take(S * 8);
  L.zy = (p->c_dim > 0 || K > 0) ? c.take(S * lat_in) : nullptr;
  L.alpha = K > 0 ? c.take(B * K) : nullptr;
  L.sw = K > 0 ? c.take(S) : nullptr;
  for (int i = 0; i < n_enc; ++i) L.edp[i] = c.take(B * p->enc[i].out_dim);
  L.enc_compact = !L.enc_conv && !L.enc_ext && pv_enc_compact_supported(p);
  L.kl_blocks = (int)((B + 15) / 16);
  L.kl_part = c.take(2 * (B > L.kl_blocks ? B : L.kl_blocks));   // (per 16-row block — or per sample when the guide rides in the decoder launch)
  L.enc_flags = reinterpret_cast<unsigned*>(c.take(8 * L.kl_blocks));     // (the merged encoder launch's tile flags)
  L.coop_flags = reinterpret_cast<unsigned*>(c.take(1024 + 16));
  int64_t maxd = 0;
  L.fused = p->fused && pv_sdec_fused_supported(p) && !(K > 0 && !L.enc_compact);   // (jiVAE + generic encoder: layered)
  L.f_grid = L.f_kmax = 0;
  L.f_part = L.f_part_hz = L.f_rowtp = L.f_wimg = L.f_park = nullptr;
  for (int i = 0; i < p->n_dec; ++i) {
    L.dact[i] = L.fused ? nullptr : c.take(R * p->dec[i].out_dim);
    L.dpre_[i] = (!L.fused && p->dec[i].act == PV_ACT_GELU) ? c.take(R * p->dec[i].out_dim) : nullptr;
    if (p->dec[i].out_dim > maxd) maxd = p->dec[i].out_dim;
  }
  int64_t scratch = 0;
  auto upd = [&](int64_t v) { if (v > scratch) scratch = v; };
  if (p->coord_dim > 0) {
    const int64_t H0 = p->fc_coord.out_dim;
    if (H0 > maxd) maxd = H0;
    L.hz = c.take(S * H0);
    L.h0 = L.fused ? nullptr : c.take(R * H0);
    if (L.fused) {
      const int64_t units = R / FD_UNIT;
      L.f_grid = pv_sdec_fused_grid(units);
      L.f_kmax = pv_sdec_fused_kmax((int)N, units, L.f_grid);
      if (p->fused >= 2) L.f_kmax *= pv_sdec_fused_bf16_waves(p->fused == 2, units, p->dec_kernel);   // the bf16 kernels publish dL/d(hz) per wave
      L.f_part = c.take((int64_t)L.f_grid * FD_REC);
      L.f_part_hz = c.take(S * L.f_kmax * (H0 + PV_RS_W));        // (+ the row-sum slots behind it: PvFused::part_rs, one zero fill)
      L.f_rowtp = c.take(4 * R);
      L.f_wimg = c.take(FB_WIMG_BYTES / (int64_t)sizeof(float));
      const int64_t park = (p->fused == 2 && !inference_only) ? pv_sdec_fused_bf16_park_bytes(true, units, L.f_grid, p->dec_kernel) : 0;
      L.f_park = park ? c.take(park / (int64_t)sizeof(float)) : nullptr;
      upd(pv_colsum_ws(B, (int)H0));
    }
    L.logits = nullptr;
    L.llrow = c.take(R);
    const int64_t ob = pv_out_lik_blocks(R);
    L.part_dwo = c.take(ob * p->out.in_dim);
    L.part_dbo = c.take(ob);
    // row chunks per sample for the coord_latent backward: >= ~1024 workgroups overall
    int nchunk = (int)((1024 + B - 1) / B);
    if (nchunk > (N + 31) / 32) nchunk = (int)((N + 31) / 32);
    if (nchunk < 1) nchunk = 1;
    L.rows_per_chunk = (int)((N + nchunk - 1) / nchunk);
    if (L.rows_per_chunk < 1) L.rows_per_chunk = 1;
    L.nchunk = (int)((N + L.rows_per_chunk - 1) / L.rows_per_chunk);
    L.part_hz = c.take(S * L.nchunk * H0);
    L.part_wc = c.take(S * L.nchunk * H0 * p->coord_dim);
    L.part_tp = c.take(S * L.nchunk * 4);
    L.dhz = c.take(S * H0);
    L.dtp = c.take(S * 4);
    L.dzc = c.take(S * lat_in);
    upd(gemm_ws_need(S, H0, lat_in));          // hz
    upd(gemm_ws_need(H0, lat_in, S));          // dWz
    upd(gemm_ws_need(S, lat_in, H0));          // dzc
  } else {
    L.hz = L.h0 = nullptr;
    L.logits = c.take(S * N);
    L.llrow = c.take(S * N);
    L.part_dwo = L.part_dbo = L.part_hz = L.part_wc = L.part_tp = L.dhz = L.dtp = nullptr;
    L.nchunk = L.rows_per_chunk = 0;
    L.dzc = c.take(S * lat_in);
    if (N > maxd) maxd = N;
    upd(gemm_ws_need(S, N, p->out.in_dim));
    upd(gemm_ws_need(N, p->out.in_dim, S));
    upd(gemm_ws_need(S, p->out.in_dim, N));
    upd(pv_colsum_ws(S, (int)N));
  }
  L.llb = c.take(B);
  L.row_ll = c.take(B);
  L.llkb = K > 0 ? c.take(S) : nullptr;
  L.jfix = K > 0 ? c.take(4) : nullptr;
  L.dbuf[0] = L.fused ? nullptr : c.take(R * maxd);
  L.dbuf[1] = L.fused ? nullptr : c.take(R * maxd);
  // scratch: the largest split-K / colsum requirement of any single call
  upd(cnd.scratch);
  for (int i = 0; i < n_enc; ++i) {
    upd(gemm_ws_need(B, p->enc[i].out_dim, p->enc[i].in_dim));
    upd(gemm_ws_need(p->enc[i].out_dim, p->enc[i].in_dim, B));
    upd(gemm_ws_need(B, p->enc[i].in_dim, p->enc[i].out_dim));
    upd(pv_colsum_ws(B, p->enc[i].out_dim));
  }
  if (!L.enc_ext) {
    upd(gemm_ws_need(B, hw, p->head.in_dim));
    upd(gemm_ws_need(hw, p->head.in_dim, B));
    upd(gemm_ws_need(B, p->head.in_dim, hw));
    upd(pv_colsum_ws(B, (int)hw));
  }
  for (int i = 0; i < p->n_dec && !L.fused; ++i) {
    upd(gemm_ws_need(R, p->dec[i].out_dim, p->dec[i].in_dim));
    upd(gemm_ws_need(p->dec[i].out_dim, p->dec[i].in_dim, R));
    upd(gemm_ws_need(R, p->dec[i].in_dim, p->dec[i].out_dim));
    upd(pv_colsum_ws(R, p->dec[i].out_dim));
  }
  L.scratch_bytes = pv_align_up(scratch, 256);
  L.scratch = base ? (void*)(base + c.off) : nullptr;
  c.off += L.scratch_bytes;
  if (L.enc_conv && L.cF >= 0) {
    L.cfin_bytes = pv_align_up(cnd.wg_sum, 256);
    L.cfin_ws = base ? base + c.off : nullptr;
    c.off += L.cfin_bytes;
  }
  L.total = c.off;
}

}  // namespace

// ---- generic nn.Linear building blocks (exported to other translation units through pv_linear.h) ----
// y = act(x W^T + b)
int linear_fwd(const float* x, int64_t ldx, const float* W, const float* b, float* y, float* pre, int64_t ldy,
               int64_t M, int64_t K, int64_t N, int act, void* ws, int64_t wsb, hipStream_t s) {
  PvGemm g{};
  g.A = x; g.a_rs = ldx; g.a_cs = 1;
  g.B = W; g.b_rs = 1; g.b_cs = K;            // B(k,n) = W[n][k]
  g.C = y; g.ldc = ldy; g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.bias = b; g.act = act; g.pre = pre;
  return pv_gemm(g, pv_gemm_pick_splits((int)M, (int)N, (int)K), ws, wsb, s);
}

// dx = (dpre W) * act'(xact)
int linear_dgrad(const float* dpre, int64_t lddp, const float* W, float* dx, int64_t lddx, const float* xact,
                 const float* xpre, int64_t ldxa, int act_prev, int64_t M, int64_t K, int64_t N, void* ws, int64_t wsb,
                 hipStream_t s) {
  PvGemm g{};
  g.A = dpre; g.a_rs = lddp; g.a_cs = 1;      // (M, N)
  g.B = W; g.b_rs = K; g.b_cs = 1;            // B(n,k) = W[n][k]
  g.C = dx; g.ldc = lddx; g.M = (int)M; g.N = (int)K; g.K = (int)N;
  g.act = PV_ACT_NONE;
  if (act_prev != PV_ACT_NONE) { g.aux = xact; g.auxpre = xpre; g.ldaux = ldxa; g.act_aux = act_prev; }
  return pv_gemm(g, pv_gemm_pick_splits((int)M, (int)K, (int)N), ws, wsb, s);
}

// dw = dpre^T x ; db = colsum(dpre)
int linear_wgrad(const float* dpre, int64_t lddp, const float* x, int64_t ldx, float* dw, float* db, int64_t M,
                 int64_t K, int64_t N, void* ws, int64_t wsb, hipStream_t s) {
  if (dw) {
    PvGemm g{};
    g.A = dpre; g.a_rs = 1; g.a_cs = lddp;    // A(n, row) = dpre[row][n]
    g.B = x; g.b_rs = ldx; g.b_cs = 1;        // B(row, k) = x[row][k]
    g.C = dw; g.ldc = K; g.M = (int)N; g.N = (int)K; g.K = (int)M;
    g.act = PV_ACT_NONE;
    g.rowsumA = db;                            // db[n] = sum_rows dpre[row][n], fused into the same pass
    PV_TRY(pv_gemm(g, pv_gemm_pick_splits((int)N, (int)K, (int)M), ws, wsb, s));
    return 0;
  }
  if (db) PV_TRY(pv_colsum(dpre, lddp, M, (int)N, db, ws, wsb, s));
  return 0;
}

// ---- convolutions (kernel 3, padding 1, stride 1; channels-last) as GEMMs over an IMPLICIT im2col operand ----
// y[(b,y,x)][co] = act(sum_j patch[(b,y,x)][j] W[co][j] + b[co]),  j = ci*KK + tap;  W = the torch weight as it lies
int conv3_fwd(const float* in, int B, int H, int W_, int C, int nd, const float* W, const float* b, float* y, int Cout,
              int act, void* ws, int64_t wsb, hipStream_t s) {
  const int64_t rows = (int64_t)B * H * W_, K = (int64_t)C * (nd == 2 ? 9 : 3);
  PvGemm g{};
  g.A = in; g.a_rs = K; g.a_cs = 1; g.conv_a = 1; g.cH = H; g.cW = W_; g.cC = C; g.cnd = nd;
  g.B = W; g.b_rs = 1; g.b_cs = K;
  g.C = y; g.ldc = Cout; g.M = (int)rows; g.N = Cout; g.K = (int)K;
  g.bias = b; g.act = act;
  return pv_gemm(g, pv_gemm_pick_splits((int)rows, Cout, (int)K), ws, wsb, s);
}

// dw[co][j] = sum_rows dpre[row][co] patch[row][j] ; db[co] = sum_rows dpre[row][co]
int conv3_wgrad(const float* dpre, const float* in, int B, int H, int W_, int C, int nd, float* dw, float* db, int Cout,
                void* ws, int64_t wsb, hipStream_t s) {
  const int64_t rows = (int64_t)B * H * W_, K = (int64_t)C * (nd == 2 ? 9 : 3);
  PvGemm g{};
  g.A = dpre; g.a_rs = 1; g.a_cs = Cout;
  g.B = in; g.b_rs = K; g.b_cs = 1; g.conv_b = 1; g.cH = H; g.cW = W_; g.cC = C; g.cnd = nd;
  g.C = dw; g.ldc = K; g.M = Cout; g.N = (int)K; g.K = (int)rows;
  g.act = PV_ACT_NONE;
  g.rowsumA = db;
  return pv_gemm(g, pv_gemm_pick_splits(Cout, (int)K, (int)rows), ws, wsb, s);
}

namespace {

// convolutional encoder: x viewed as (B, 1, *enc_in_dim) -> op sequence -> flatten (C, spatial) -> L.head
// prep != null: the spatial decoder's weight images are written by the same tiling launch (pv_conv_wprep_table)
// hp != null: the caller's next launch (pv_head_fwd) sums the conv head's partial sums itself — *hp is filled when that form ran
struct PvHeadPart { const float* part = nullptr; const float* bias = nullptr; int nseg = 0, out = 0; };
int conv_encoder_fwd(const pv_ivae_plan* p, const Layout& L, hipStream_t s, const PvFbPrep* prep = nullptr, PvHeadPart* hp = nullptr) {
  const int64_t B = p->batch;
  if (L.cF < 0 || p->head.in_dim != L.cF) return PV_EINVAL;
  float* a[PV_MAX_OPS + 1];
  a[0] = const_cast<float*>(p->x);                  // one input channel: (B, 1, H, W) is already channels-last
  for (int i = 1; i <= p->n_enc_ops; ++i) a[i] = L.cea[i];
  pvcs::Scratch sc{L.ccol, L.scratch, L.scratch_bytes, L.cbn, L.cbn_maxC, p->bn_eval, plan_conv_mode(p)};
  sc.wt = L.cwt; sc.wtp = &L.cwtp;                  // (both orientations: the backward of the same step reuses them)
  sc.code = L.ccode; sc.code2 = L.ccode2;
  if (p->conv_ev_start && p->conv_ev_stop) {
    double fl = 0.0;
    sc.ev_op = pvcs::heaviest_conv(p->enc_ops, p->n_enc_ops, p->enc_ndim, (int)p->batch, L.ces, &fl);
    sc.ev_start = p->conv_ev_start; sc.ev_stop = p->conv_ev_stop;
    if (p->conv_ev_flops) *p->conv_ev_flops = fl;
  }
  const pvcs::Shape& fe0 = L.ces[p->n_enc_ops];
  const bool hfused = pv_convhead_supported(L.cF, p->head.out_dim) && L.chead_wt;
  const PvWprepEntry he = pvcs::head_entry(p->params + p->head.w_off, L.chead_wt, p->head.out_dim, fe0.C, (int64_t)fe0.H * fe0.W);
  // the weight tilings next to the fused first block (raw weights) on the side stream; stack_fwd joins before its first tiled op
  hipStream_t side = pv_side_stream_for(s, p->flags);
  bool wt_join = false;
  static const int wprep_side = pv_exp_int("PV_SIDE_WPREP", 0) ? 1 : 0;   // (measured: the join costs more than the overlap returns)
  if (wprep_side && side && sc.code && pvcs::c1pool_fusable(p->enc_ops, p->n_enc_ops, p->enc_ndim, L.ces[0])) {
    PV_TRY(pv_stream_after(side, s));
    PV_TRY(pvcs::wt_prep(p->params, p->enc_ops, p->n_enc_ops, p->enc_ndim, 0, plan_conv_mode(p), L.cwtp, L.cwt, true, side, &he,
                         hfused ? 1 : 0, prep));
    wt_join = true;
    sc.side = side; sc.wt_join = &wt_join;
  } else {
    PV_TRY(pvcs::wt_prep(p->params, p->enc_ops, p->n_enc_ops, p->enc_ndim, 0, plan_conv_mode(p), L.cwtp, L.cwt, true, s, &he,
                         hfused ? 1 : 0, prep));
  }
  PV_TRY(pvcs::stack_fwd(p->params, p->enc_ops, p->n_enc_ops, p->enc_ndim, (int)B, a, L.ces, sc, s));
  if (wt_join) { wt_join = false; PV_TRY(pv_stream_after(s, side)); }
  const pvcs::Shape& fe = L.ces[p->n_enc_ops];
  if (hfused && hp) {
    const float* part = nullptr;
    int nseg = 0;
    if (pv_convhead_fwd_partials(L.cea[p->n_enc_ops], L.chead_wt, (int)B, L.cF, p->head.out_dim, L.scratch, L.scratch_bytes, s, &part,
                                 &nseg) == 0) {
      hp->part = part; hp->bias = p->head.b_off >= 0 ? p->params + p->head.b_off : nullptr; hp->nseg = nseg; hp->out = p->head.out_dim;
      return 0;
    }
  }
  if (hfused)      // (the weight is re-indexed channels-last, not the feature map: pv_convhead.hip)
    return pv_convhead_fwd(L.cea[p->n_enc_ops], L.chead_wt, p->head.b_off >= 0 ? p->params + p->head.b_off : nullptr, L.head,
                           (int)B, L.cF, p->head.out_dim, L.scratch, L.scratch_bytes, s);
  PV_TRY(pv_nsc_to_ncs(L.cea[p->n_enc_ops], L.cfeat, B, fe.C, (int64_t)fe.H * fe.W, s));
  return linear_fwd(L.cfeat, L.cF, p->params + p->head.w_off, p->head.b_off >= 0 ? p->params + p->head.b_off : nullptr,
                    L.head, nullptr, p->head.out_dim, B, L.cF, p->head.out_dim, PV_ACT_NONE, L.scratch, L.scratch_bytes, s);
}

int encoder_fwd(const pv_ivae_plan* p, const Layout& L, hipStream_t s, const PvFbPrep* prep = nullptr, PvHeadPart* hp = nullptr) {
  if (L.enc_conv) return conv_encoder_fwd(p, L, s, prep, hp);
  const int64_t B = p->batch;
  const float* in = p->x;
  int64_t ldin = p->n_pix;
  if (p->c_dim > 0) {
    if (!p->y) return PV_EINVAL;
    PV_TRY(pv_concat(p->x, p->n_pix, p->n_pix, p->y, p->c_dim, p->c_dim, L.xin, B, s));
    in = L.xin; ldin = p->n_pix + p->c_dim;
  }
  for (int i = 0; i < p->n_enc; ++i) {
    const pv_layer& l = p->enc[i];
    if (l.in_dim != ldin) return PV_EINVAL;
    PV_TRY(linear_fwd(in, ldin, p->params + l.w_off, l.b_off >= 0 ? p->params + l.b_off : nullptr, L.eact[i],
                      L.epre[i], l.out_dim, B, l.in_dim, l.out_dim, l.act, L.scratch, L.scratch_bytes, s));
    in = L.eact[i]; ldin = l.out_dim;
  }
  const pv_layer& h = p->head;
  if (h.in_dim != ldin) return PV_EINVAL;
  return linear_fwd(in, ldin, p->params + h.w_off, h.b_off >= 0 ? p->params + h.b_off : nullptr, L.head, nullptr,
                    h.out_dim, B, h.in_dim, h.out_dim, PV_ACT_NONE, L.scratch, L.scratch_bytes, s);
}

// decoder forward from the decoder's latent input zin (B, lat_in); tp must be filled for coord_dim > 0.
// leaves the last hidden activation in L.dact[n_dec-1]; for coord_dim == 0 also the logits in L.logits
int decoder_hidden_fwd(const pv_ivae_plan* p, const Layout& L, const float* zin, int64_t ldz, int64_t lat_in,
                       hipStream_t s) {
  const int64_t R = L.rows;
  const int64_t B = p->coord_dim > 0 ? R / p->n_pix : R;       // decoder samples (jiVAE: K per input)
  const float* in;
  int64_t ldin;
  if (p->coord_dim > 0) {
    const int64_t H0 = p->fc_coord.out_dim;
    if (p->fc_latent.in_dim != lat_in || p->fc_latent.out_dim != H0 || p->fc_coord.in_dim != p->coord_dim)
      return PV_EINVAL;
    PV_TRY(linear_fwd(zin, ldz, p->params + p->fc_latent.w_off, nullptr, L.hz, nullptr, H0, B, lat_in, H0,
                      PV_ACT_NONE, L.scratch, L.scratch_bytes, s));
    PvCoordLat c{};
    c.grid = p->grid; c.tp = L.tp; c.Wc = p->params + p->fc_coord.w_off; c.bc = p->params + p->fc_coord.b_off;
    c.hz = L.hz; c.h0 = L.h0; c.M = R; c.N = p->n_pix; c.cd = p->coord_dim; c.H0 = (int)H0;
    PV_TRY(pv_coordlat_fwd(c, s));
    in = L.h0; ldin = H0;
  } else {
    in = zin; ldin = ldz;
    if (p->dec[0].in_dim != lat_in) return PV_EINVAL;
  }
  for (int i = 0; i < p->n_dec; ++i) {
    const pv_layer& l = p->dec[i];
    if (i > 0 || p->coord_dim > 0) { if (l.in_dim != ldin) return PV_EINVAL; }
    // instrumentation: the last hidden layer's forward GEMM is the layered path's representative kernel
    const bool timed = (i == p->n_dec - 1) && p->ev_start && p->ev_stop;
    if (timed) (void)hipEventRecord((hipEvent_t)p->ev_start, s);
    PV_TRY(linear_fwd(in, ldin, p->params + l.w_off, l.b_off >= 0 ? p->params + l.b_off : nullptr, L.dact[i],
                      L.dpre_[i], l.out_dim, R, l.in_dim, l.out_dim, l.act, L.scratch, L.scratch_bytes, s));
    if (timed) (void)hipEventRecord((hipEvent_t)p->ev_stop, s);
    in = L.dact[i]; ldin = l.out_dim;
  }
  if (p->out.in_dim != ldin) return PV_EINVAL;
  if (p->coord_dim == 0) {
    PV_TRY(linear_fwd(in, ldin, p->params + p->out.w_off, p->out.b_off >= 0 ? p->params + p->out.b_off : nullptr,
                      L.logits, nullptr, p->n_pix, R, p->out.in_dim, p->n_pix, PV_ACT_NONE, L.scratch,
                      L.scratch_bytes, s));
  }
  return 0;
}

// wgrad problem of one Linear: dw[N,K] = dpre[M,N]^T x[M,K], db[N] = colsum(dpre)
PvGemm wgrad_problem(const float* dpre, int64_t lddp, const float* x, int64_t ldx, float* dw, float* db, int64_t M,
                     int64_t K, int64_t N) {
  PvGemm g{};
  g.A = dpre; g.a_rs = 1; g.a_cs = lddp;
  g.B = x; g.b_rs = ldx; g.b_cs = 1;
  g.C = dw; g.ldc = K; g.M = (int)N; g.N = (int)K; g.K = (int)M;
  g.act = PV_ACT_NONE;
  g.rowsumA = db;
  return g;
}

// encoder backward from dL/d(head pre-activations) (L.dhead): the dgrad chain through the hidden layers, then
// every weight gradient of the encoder (plus `extra`, e.g. fc_latent's) in one multi-GEMM launch per 4 problems
typedef PvFinishArgs PvFinish;

// adam / adam_done: pv_ivae_step's optimizer update, applied inside the weight-gradient launch when that single launch
// finalises every encoder gradient (compact encoder, <= 4 problems); *adam_done tells the caller whether it was
// dgrad_done: the compact encoder's dgrad chain already ran (inside pv_latent_bwd_reduce); the loss scalars then ride
// in the weight-gradient launch
int encoder_bwd(const pv_ivae_plan* p, const Layout& L, const PvGemm* extra, int n_extra, hipStream_t s,
                const PvFinish* fin = nullptr, const PvAdamFuse* adam = nullptr, bool* adam_done = nullptr,
                bool dgrad_done = false, bool head_side = false) {
  const int64_t B = p->batch;
  float* G = p->grads;
  void* ws = L.scratch;
  const int64_t wsb = L.scratch_bytes;
  const int ne = p->n_enc;
  const pv_layer& hd = p->head;
  if (L.enc_ext) {                                   // dhead already sits in the caller's ext_dhead
    if (fin) PV_TRY(pv_finish_scalars(fin->llb, fin->B, fin->scalars, fin->kl_part, fin->n_part, fin->beta, s));
    for (int i = 0; i < n_extra; i += 4) PV_TRY(pv_wgrad_small(extra + i, n_extra - i < 4 ? n_extra - i : 4, s));
    return 0;
  }
  if (L.enc_conv) {
    // head (features2latent.fc_latent) backward, then the op sequence in reverse; the other small wgrads ride along
    const pvcs::Shape& fe = L.ces[p->n_enc_ops];
    const pv_op& last = p->enc_ops[p->n_enc_ops - 1];
    bool g_is_pre = false;
    hipStream_t side = pv_side_stream_for(s, p->flags);
    PvSideJoin sj;                                    // joins the side stream on an early return
    if (pv_convhead_supported(L.cF, hd.out_dim) && L.chead_wt) {
      // dL/d(features) straight in channels-last order into cg[1], the last convolution's activation derivative folded in
      g_is_pre = last.kind == PV_OP_CONV && last.act != PV_ACT_GELU;
      // the head's weight gradient needs dhead only: on the side stream (forked off the launch that wrote dhead) next to the
      // input gradient below
      hipStream_t hs = s;
      if (head_side && side) { PV_TRY(pv_fork_to(side, s)); sj.fork(s, side); hs = side; }
      else pv_fork_disarm();
      PV_TRY(pv_convhead_wgrad(L.dhead, L.cea[p->n_enc_ops], G + hd.w_off, hd.b_off >= 0 ? G + hd.b_off : nullptr, (int)B,
                               fe.H * fe.W, fe.C, hd.out_dim, ws, wsb, hs));
      // ... and so do the small weight gradients the caller hands over (fc_latent's: they need the latent-backward launch's
      // results only) with the loss scalars riding: behind the head's on the side stream instead of at the very end of the step.
      // The optimizer update is then a launch of its own after the last reduction (round 5: the closing launch 13.6 -> ~6 us).
      static const int ab_early = pv_exp_int("PV_EXTRA_EARLY", 1);
      if (ab_early && hs == side && side && n_extra > 0 && n_extra <= 4) {
        PV_TRY(pv_wgrad_small(extra, n_extra, side, nullptr, fin));
        n_extra = 0; fin = nullptr;
      }
      if (side) pv_fork_arm();                                        // (the last convolution's weight gradient forks off this launch)
      PV_TRY(pv_convhead_bwd(L.dhead, L.chead_wt, L.cea[p->n_enc_ops], g_is_pre ? last.act : PV_ACT_NONE, L.cg[1], (int)B, L.cF,
                             hd.out_dim, s));
    } else {
      PV_TRY(linear_wgrad(L.dhead, hd.out_dim, L.cfeat, L.cF, G + hd.w_off, hd.b_off >= 0 ? G + hd.b_off : nullptr, B, L.cF,
                          hd.out_dim, ws, wsb, s));
      PV_TRY(linear_dgrad(L.dhead, hd.out_dim, p->params + hd.w_off, L.cg[0], L.cF, nullptr, nullptr, 0, PV_ACT_NONE, B,
                          L.cF, hd.out_dim, ws, wsb, s));
      PV_TRY(pv_ncs_to_nsc(L.cg[0], L.cg[1], B, fe.C, (int64_t)fe.H * fe.W, s));
    }
    float* a[PV_MAX_OPS + 1];
    a[0] = const_cast<float*>(p->x);
    for (int i = 1; i <= p->n_enc_ops; ++i) a[i] = L.cea[i];
    pvcs::Scratch sc{L.ccol, ws, wsb, L.cbn, L.cbn_maxC, p->bn_eval, plan_conv_mode(p)};
    sc.wt = L.cwt; sc.wtp = &L.cwtp;                  // tiled by this step's conv_encoder_fwd
    sc.code = L.ccode; sc.code2 = L.ccode2;
    PvFinishList wfin{};                              // the weight gradients' reductions: one launch after the stack
    wfin.base = L.cfin_ws; wfin.cap = L.cfin_bytes;
    sc.fin = &wfin;
    int pp = 0;                                       // g = cg[1]; first free ping-pong buffer = cg[0]
    // kernel-3 weight gradients on the side stream, the input-gradient chain on s (every op's gradient in its own buffer);
    // joined before the finish
    bool joined = false;
    sj.fork(s, side);
    sc.side = side; sc.side_joined = &joined;
    PV_TRY(pvcs::stack_bwd(p->params, G, p->enc_ops, p->n_enc_ops, p->enc_ndim, (int)B, a, L.ces, L.cg[1], L.cg, pp, false,
                           nullptr, sc, s, 0, g_is_pre, side ? L.ceg : nullptr));
    if (side && !joined) PV_TRY(pv_stream_after(s, side));
    sj.joined();
    PV_TRY(pv_wgrad_finish_all(&wfin, s));
    if (fin && n_extra < 1) PV_TRY(pv_finish_scalars(fin->llb, fin->B, fin->scalars, fin->kl_part, fin->n_part, fin->beta, s));
    // (the loss scalars ride in the first of these launches; every other gradient is final by now, so pv_ivae_step's Adam update
    //  rides in the last one: its own outputs in its epilogue, the rest of the flat buffer by guest workgroups)
    static const int ab_adam = pv_exp_int("PV_CONV_ADAM_RIDE", 1);
    for (int i = 0; i < n_extra; i += 4) {
      const bool last = i + 4 >= n_extra;
      const bool ride = last && adam && adam_done && ab_adam && B <= 4096;
      PV_TRY(pv_wgrad_small(extra + i, n_extra - i < 4 ? n_extra - i : 4, s, ride ? adam : nullptr, i == 0 ? fin : nullptr));
      if (ride) *adam_done = true;
    }
    return 0;
  }
  const float* elast = L.eact[ne - 1];
  if (L.enc_compact && dgrad_done) {
    // nothing to launch here
  } else if (L.enc_compact) {
    PvEncDgrad d{};
    d.params = p->params; d.n_enc = ne; d.B = (int)B; d.head = hd; d.dhead = L.dhead;
    for (int i = 0; i < ne; ++i) { d.enc[i] = p->enc[i]; d.eact[i] = L.eact[i]; d.edp[i] = L.edp[i]; }
    if (fin) {
      d.fin_llb = fin->llb; d.fin_scalars = fin->scalars; d.fin_kl_part = fin->kl_part; d.fin_n_part = fin->n_part;
      d.fin_beta = fin->beta;
    }
    PV_TRY(pv_enc_dgrad(d, s));
  } else {
    PV_TRY(linear_dgrad(L.dhead, hd.out_dim, p->params + hd.w_off, L.edp[ne - 1], hd.in_dim, elast, L.epre[ne - 1],
                        hd.in_dim, p->enc[ne - 1].act, B, hd.in_dim, hd.out_dim, ws, wsb, s));
    for (int i = ne - 1; i > 0; --i) {
      const pv_layer& l = p->enc[i];
      PV_TRY(linear_dgrad(L.edp[i], l.out_dim, p->params + l.w_off, L.edp[i - 1], l.in_dim, L.eact[i - 1],
                          L.epre[i - 1], p->enc[i - 1].out_dim, p->enc[i - 1].act, B, l.in_dim, l.out_dim, ws, wsb,
                          s));
    }
  }
  // (the same list, as a pure function, for the launch that closes an own-sample step: compact_wgrad_problems below — keep the two in step)
  PvGemm probs[PV_MAX_LAYERS + 4];
  int np = 0;
  for (int i = 0; i < n_extra; ++i) {
    // a long contraction over a few output tiles (jiVAE's fc_latent: K*B decoder samples onto 128 x lat_in) would sit on
    // a handful of workgroups in the one-tile-per-workgroup launch: split-K GEMM instead, finished before that launch
    // so that its fused Adam guests see the final gradient
    const PvGemm& e = extra[i];
    const int64_t tiles = (int64_t)((e.M + 15) / 16) * ((e.N + 15) / 16);
    const int sp = pv_gemm_pick_splits(e.M, e.N, e.K);
    if (e.K > 1024 && tiles < 128 && sp > 1 &&
        (int64_t)sp * e.M * (e.N + 1) * (int64_t)sizeof(float) <= wsb) PV_TRY(pv_gemm(e, sp, ws, wsb, s));
    else probs[np++] = e;
  }
  probs[np++] = wgrad_problem(L.dhead, hd.out_dim, elast, hd.in_dim, G + hd.w_off,
                              hd.b_off >= 0 ? G + hd.b_off : nullptr, B, hd.in_dim, hd.out_dim);
  const float* xin = p->c_dim > 0 ? L.xin : p->x;
  const int64_t ldx = p->n_pix + p->c_dim;
  for (int i = ne - 1; i >= 0; --i) {
    const pv_layer& l = p->enc[i];
    const float* in = i > 0 ? L.eact[i - 1] : xin;
    const int64_t ldin = i > 0 ? p->enc[i - 1].out_dim : ldx;
    probs[np++] = wgrad_problem(L.edp[i], l.out_dim, in, ldin, G + l.w_off, l.b_off >= 0 ? G + l.b_off : nullptr, B,
                                l.in_dim, l.out_dim);
  }
  const PvFinish* fin_w = (L.enc_compact && dgrad_done) ? fin : nullptr;     // (otherwise pv_enc_dgrad hosted it)
  if (B <= 4096 && np <= 4 && adam && adam_done && L.enc_compact) {
    PV_TRY(pv_wgrad_small(probs, np, s, adam, fin_w));
    *adam_done = true;
  } else if (B <= 4096) {
    for (int i = 0; i < np; i += 4)
      PV_TRY(pv_wgrad_small(probs + i, np - i < 4 ? np - i : 4, s, nullptr, i == 0 ? fin_w : nullptr));
  } else {                                   // long contractions: split-K GEMMs, one launch pair each
    if (fin_w) PV_TRY(pv_finish_scalars(fin_w->llb, fin_w->B, fin_w->scalars, fin_w->kl_part, fin_w->n_part, fin_w->beta, s));
    for (int i = 0; i < np; ++i)
      PV_TRY(pv_gemm(probs[i], pv_gemm_pick_splits(probs[i].M, probs[i].N, probs[i].K), ws, wsb, s));
  }
  return 0;
}

// the compact (fc) encoder's weight-gradient problems behind `extra` — what encoder_bwd hands its one-tile-per-workgroup launch;
// -1 when one of them wants the split-K GEMM or they are more than that launch takes
int compact_wgrad_problems(const pv_ivae_plan* p, const Layout& L, const PvGemm* extra, int n_extra, PvGemm (&probs)[4]) {
  const int64_t B = p->batch;
  const int ne = p->n_enc;
  if (B > 4096 || n_extra + 1 + ne > 4) return -1;
  float* G = p->grads;
  const pv_layer& hd = p->head;
  int np = 0;
  for (int i = 0; i < n_extra; ++i) {
    const PvGemm& e = extra[i];
    const int64_t tiles = (int64_t)((e.M + 15) / 16) * ((e.N + 15) / 16);
    if (e.K > 1024 && tiles < 128 && pv_gemm_pick_splits(e.M, e.N, e.K) > 1) return -1;
    probs[np++] = e;
  }
  probs[np++] = wgrad_problem(L.dhead, hd.out_dim, L.eact[ne - 1], hd.in_dim, G + hd.w_off, hd.b_off >= 0 ? G + hd.b_off : nullptr, B,
                              hd.in_dim, hd.out_dim);
  const float* xin = p->c_dim > 0 ? L.xin : p->x;
  const int64_t ldx = p->n_pix + p->c_dim;
  for (int i = ne - 1; i >= 0; --i) {
    const pv_layer& l = p->enc[i];
    probs[np++] = wgrad_problem(L.edp[i], l.out_dim, i > 0 ? L.eact[i - 1] : xin, i > 0 ? p->enc[i - 1].out_dim : ldx, G + l.w_off,
                                l.b_off >= 0 ? G + l.b_off : nullptr, B, l.in_dim, l.out_dim);
  }
  return np;
}

// plan->row_w / row_elbo on the paths that form ll_b with pv_segsum: keep the unweighted ll_b, weight what the loss sums
int weigh_llb(const pv_ivae_plan* p, const Layout& L, hipStream_t s) {
  if (p->row_elbo) {
    hipError_t e = hipMemcpyAsync(L.row_ll, L.llb, (size_t)p->batch * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
  }
  if (p->row_w) PV_TRY(pv_scale_rows(L.llb, p->row_w, p->batch, 1, s));
  return 0;
}

// plan->row_elbo (per-sample ELBO terms) and plan->dy (dloss/dy = the y columns of dL/d(encoder input) and of
// dL/d(decoder latent input) `dzc`), after the step's other work
int extra_outputs(const pv_ivae_plan* p, const Layout& L, const float* dzc, int64_t lat_in, hipStream_t s) {
  const int64_t B = p->batch;
  if (p->row_elbo)
    PV_TRY(pv_row_elbo(L.row_ll, L.z, L.head, L.z_scale, (int)B, p->z_dim, (int)plan_head_w(p), p->beta, p->row_elbo, s));
  if (p->dy && dzc) {
    const pv_layer& l0 = p->enc[0];
    const int64_t N = p->n_pix, c = p->c_dim;
    if (l0.in_dim != N + c) return PV_EINVAL;
    PvGemm g{};
    g.A = L.edp[0]; g.a_rs = l0.out_dim; g.a_cs = 1;                         // (B, H)
    g.B = p->params + l0.w_off + N; g.b_rs = l0.in_dim; g.b_cs = 1;          // B(j, i) = W0[j][N + i]
    g.C = p->dy; g.ldc = c; g.M = (int)B; g.N = (int)c; g.K = l0.out_dim;
    g.act = PV_ACT_NONE;
    PV_TRY(pv_gemm(g, 1, L.scratch, L.scratch_bytes, s));
    PV_TRY(pv_add_cols(p->dy, c, dzc + (lat_in - c), lat_in, B, (int)c, s));
  }
  return 0;
}

// dL/dz from the decoder (dzc: content/y columns; dtp: phi, scale, tx, ty) -> head -> encoder
int latent_encoder_bwd(const pv_ivae_plan* p, const Layout& L, int64_t lat_in, int dtp_sb, int dtp_sc,
                       hipStream_t s) {
  PvHeadBwd hb{};
  hb.dzc = L.dzc; hb.ldzc = lat_in; hb.dtp = L.dtp; hb.dtp_sb = dtp_sb; hb.dtp_sc = dtp_sc;
  hb.z = L.z; hb.z_scale = L.z_scale; hb.eps = p->eps;
  hb.head = L.enc_ext ? p->ext_head : L.head; hb.dhead = L.enc_ext ? p->ext_dhead : L.dhead;
  hb.scale_direct = L.enc_ext ? 1 : 0; hb.B = p->batch; hb.z_dim = p->z_dim; hb.coord_dim = p->coord_dim;
  hb.has_r = p->has_r; hb.has_t = p->has_t; hb.has_s = p->has_s;
  hb.tp0 = p->t_prior[0]; hb.tp1 = p->t_prior[1]; hb.sc_prior = p->sc_prior; hb.beta = p->beta;
  hb.ldh = L.enc_ext ? 0 : (int)plan_head_w(p);
  hb.w = p->row_w;
  PV_TRY(pv_head_bwd(hb, s));
  PV_TRY(encoder_bwd(p, L, nullptr, 0, s));
  return extra_outputs(p, L, L.dzc, lat_in, s);
}

// guide: encoder -> (z_loc, z_scale) -> z = z_loc + z_scale*eps, sampled-KL terms, transform parameters
// hzr != null (not the compact encoder): fc_latent of the spatial decoder may ride in the head launch — hzr->done says whether it did
// (kl: the KL sums of scalars[2], [3] left as partials in L.kl_part: pass them to pv_finish_scalars)
struct PvHzReq { const float* zin; int64_t ldz; int lat_in, H; const float* Wz; float* hz; bool done, kl; };
int guide_fwd(const pv_ivae_plan* p, const Layout& L, hipStream_t s, const PvFbPrep* prep = nullptr,
              float hz_scale = 0.0f, PvHzReq* hzr = nullptr) {
  if (L.enc_compact) {
    PvEncFwd e{};
    if (prep) e.prep = *prep;
    e.params = p->params; e.n_enc = p->n_enc; e.head = p->head;
    for (int i = 0; i < p->n_enc; ++i) { e.enc[i] = p->enc[i]; e.eact[i] = L.eact[i]; }
    e.x = p->x; e.ldx = p->n_pix;
    if (p->c_dim > 0) {
      if (!p->y) return PV_EINVAL;
      PV_TRY(pv_concat(p->x, p->n_pix, p->n_pix, p->y, p->c_dim, p->c_dim, L.xin, p->batch, s));
      e.x = L.xin; e.ldx = p->n_pix + p->c_dim;
    }
    e.eps = p->eps; e.y = p->y; e.head_out = L.head; e.z = L.z; e.z_scale = L.z_scale;
    e.z_loc_out = p->z_loc; e.z_scale_out = p->z_scale;
    e.tp = p->coord_dim > 0 ? L.tp : nullptr; e.zy = L.zy; e.kl_part = L.kl_part;
    e.beta = p->beta; e.beta_disc = p->beta_disc; e.K = (int)plan_K(p); e.alpha = L.alpha; e.sw = L.sw;
    e.w = p->row_w;
    if (p->coord_dim > 0) { e.hz = L.hz; e.Wz = p->params + p->fc_latent.w_off; e.H0 = p->fc_coord.out_dim; }
    e.hz_scale = hz_scale;
    e.flags = (p->flags & PV_PLAN_ENC_TWO_LAUNCH) ? nullptr : L.enc_flags;
    e.spin_limit = (p->flags & PV_PLAN_ENC_NO_WAIT) ? 0 : 256;
    e.B = p->batch; e.z_dim = p->z_dim; e.c_dim = p->c_dim; e.coord_dim = p->coord_dim;
    e.has_r = p->has_r; e.has_t = p->has_t; e.has_s = p->has_s;
    e.tp0 = p->t_prior[0]; e.tp1 = p->t_prior[1]; e.sc_prior = p->sc_prior;
    return pv_enc_fwd(e, s);
  }
  if (prep && !L.enc_conv) return PV_EINVAL;      // (stand-alone preparation on this path)
  // a conv encoder in front of the spatial decoder (hzr): the conv head's partial sums, this head and fc_latent in ONE launch of
  // ceil(B / 16) workgroups (PV_HEAD_MERGE=0: the four launches)
  static const int ab_merge = pv_exp_int("PV_HEAD_MERGE", 1);
  const bool blocks = ab_merge && hzr && L.enc_conv && !L.enc_ext && plan_K(p) == 0 && p->head.out_dim == (int)plan_head_w(p);
  PvHeadPart hp;
  if (!L.enc_ext) PV_TRY(encoder_fwd(p, L, s, prep, blocks ? &hp : nullptr));
  PvHead h{};
  if (hp.part) { h.ch_part = hp.part; h.ch_bias = hp.bias; h.ch_nseg = hp.nseg; h.ch_out = hp.out; h.head_w = L.head; }
  if (blocks) {
    h.kl_part = L.kl_part; hzr->kl = true;
    if (hzr->lat_in <= 16) {
      h.zin = hzr->zin; h.ldz = hzr->ldz; h.lat_in = hzr->lat_in; h.H = hzr->H; h.Wz = hzr->Wz; h.hz = hzr->hz;
      hzr->done = true;
    }
  }
  h.head = L.enc_ext ? p->ext_head : L.head; h.scale_direct = L.enc_ext ? 1 : 0; h.eps = p->eps; h.y = p->y; h.z = L.z; h.z_scale = L.z_scale;
  h.z_loc_out = p->z_loc; h.z_scale_out = p->z_scale;
  h.tp = p->coord_dim > 0 ? L.tp : nullptr; h.zy = L.zy; h.scalars = p->scalars;
  h.B = p->batch; h.z_dim = p->z_dim; h.c_dim = p->c_dim; h.coord_dim = p->coord_dim;
  h.has_r = p->has_r; h.has_t = p->has_t; h.has_s = p->has_s;
  h.tp0 = p->t_prior[0]; h.tp1 = p->t_prior[1]; h.sc_prior = p->sc_prior; h.beta = p->beta;
  h.w = p->row_w;
  h.ldh = L.enc_ext ? 0 : (int)plan_head_w(p);
  const int64_t K = plan_K(p);
  if (K > 0) h.zy = nullptr;                      // (written per decoder sample below)
  if (blocks) PV_TRY(pv_head_fwd_blocks(h, s));
  else PV_TRY(pv_head_fwd(h, s));
  if (K > 0) {
    const int coord = p->z_dim - p->latent_dim;
    const int n_content = p->coord_dim > 0 ? p->latent_dim : p->z_dim;
    (void)coord;
    PV_TRY(pv_jiv_expand(L.head, (int)plan_head_w(p), L.z, p->z_dim, n_content, p->coord_dim > 0 ? L.tp : nullptr, L.zy,
                         L.alpha, L.sw, p->scalars, p->beta_disc, p->batch, (int)K, s));
  }
  return 0;
}

// the plan-side conditions of the folded guide (PvEncFold): plain-bf16 fused decoder, the plain two-hidden-layer fc encoder, no
// conditioning vector / discrete latent / per-sample weights; the launch-side ones are pv_sdec_fused_fold_ok's
// (plan_guide_one_image: the architecture one workgroup can run a whole image's guide for — the fold in the decoder launch and the
//  per-image guide launch, pv_guide_img.hip)
static bool plan_guide_one_image(const pv_ivae_plan* p, const Layout& L);
static bool plan_guide_may_fold(const pv_ivae_plan* p, const Layout& L) {
  return p->fused == 3 && !(p->flags & PV_PLAN_NO_ENC_FOLD) && plan_guide_one_image(p, L);
}
static bool plan_guide_one_image(const pv_ivae_plan* p, const Layout& L) {
  const int64_t z = p->z_dim;
  return p->fused >= 2 && L.fused && L.enc_compact && !L.enc_ext && plan_K(p) == 0 &&
         p->c_dim == 0 && !p->row_w && !p->row_elbo && !p->dy && p->n_enc == 2 && p->enc[0].in_dim == p->n_pix &&
         p->enc[0].out_dim <= FD_H && p->enc[1].out_dim <= FD_H && p->head.out_dim == 2 * z && z <= 16 && plan_lat_in(p) <= 16 &&
         p->enc[0].w_off % 4 == 0 && p->enc[1].w_off % 4 == 0 && p->head.w_off % 4 == 0 && p->enc[1].in_dim % 4 == 0 &&
         p->head.in_dim % 4 == 0 && p->coord_dim > 0 &&
         // the folded prologue also reads the decoder's hidden matrices, the observations and the parameter base as float4
         // (the Python engine aligns every offset; a raw C-ABI caller need not) — otherwise the guide keeps its launch
         p->dec[0].w_off % 4 == 0 && p->dec[1].w_off % 4 == 0 && p->n_pix % 4 == 0 &&
         ((uintptr_t)p->params & 15) == 0 && ((uintptr_t)p->x & 15) == 0;
}

// loss_and_grads with the fused persistent spatial-decoder kernel (pv_sdec_fused.hip)
int loss_and_grads_fused(const pv_ivae_plan* p, const Layout& L, int want_grads, hipStream_t s,
                         const PvAdamFuse* adam = nullptr, bool* adam_done = nullptr) {
  const int64_t B = p->batch, N = p->n_pix, R = L.rows, z = p->z_dim;
  const int64_t K = plan_K(p), S = plan_S(p);        // jiVAE: S = K*B decoder samples, rows R = S*N
  const int64_t lat_in = plan_lat_in(p);
  const int H = FD_H;
  float* G = p->grads;
  const int coord = (int)(z - p->latent_dim);
  const bool cat_in = p->c_dim > 0 || K > 0;         // the decoder's latent input is a materialised concatenation
  const float* zin = cat_in ? L.zy : L.z + coord;
  const int64_t ldz = cat_in ? lat_in : z;
  if (p->fc_latent.in_dim != lat_in) return PV_EINVAL;
  if (K > 0 && !L.enc_compact) return PV_EINVAL;
  PvFused f{};
  f.x = p->x; f.grid = p->grid; f.tp = L.tp; f.hz = L.hz;
  f.Wc = p->params + p->fc_coord.w_off; f.bc = p->params + p->fc_coord.b_off;
  f.W1 = p->params + p->dec[0].w_off; f.b1 = p->params + p->dec[0].b_off;
  f.W2 = p->params + p->dec[1].w_off; f.b2 = p->params + p->dec[1].b_off;
  f.wo = p->params + p->out.w_off; f.bo = p->params + p->out.b_off;
  f.llrow = L.llrow; f.loc = p->loc; f.rowtp = L.f_rowtp; f.part_hz = L.f_part_hz; f.part = L.f_part;
  // (round 6) training launches of the 4-wave kernels (pv_sdec_fused_bf16.hip: the ones that write PV_REC_LANE_F32 records) hand
  // over per-slot row sums instead of rows.  NOT the 8-wave throughput kernel: it has no registers for five running sums, and
  // keeping them in LDS cost its in-order waves what the next launch saved (read-modify-write: +-0 on the step; ds_add_f32: +1.4 us
  // on the kernel — profiles/r06h_row_sums_ab.txt)
  static const int ab_row_sums = pv_exp_int("PV_ROW_SUMS", 1);      // (experiments build: 0 = per-row outputs from the 4-wave kernels too)
  f.part_rs = (want_grads && p->fused >= 2 && H == FD_H && ab_row_sums &&
               pv_sdec_fused_bf16_record_fmt(p->fused == 2, R / FD_UNIT, p->dec_kernel) == PV_REC_LANE_F32)
                  ? L.f_part_hz + S * L.f_kmax * H : nullptr;
  f.wimg = L.f_wimg; f.park = L.f_park;
  f.M = R; f.units = R / FD_UNIT; f.N = (int)N; f.cd = p->coord_dim; f.B = (int)S; f.lik = p->lik;
  f.sw = K > 0 ? L.sw : p->row_w; f.x_units = K > 0 ? B * N / FD_UNIT : 0;
  f.sigmoid_out = p->sigmoid_out; f.kmax = L.f_kmax; f.sig = p->decoder_sig; f.sel = p->dec_kernel;
  // fp16 builds of the fp32-class kernel: where the per-row exponent of dL/dlogit is centred (|dL/dlogit| <= 1 for the Bernoulli
  // likelihoods; ~ residual / sig^2 for the Gaussian): exact powers of two either way, only the representable RANGE moves
  f.dl_exp = 4;
  if (p->lik == PV_LIK_GAUSSIAN && p->decoder_sig > 0.0f) {
    int e2 = 0;
    (void)frexpf(p->decoder_sig * p->decoder_sig, &e2);      // sig^2 = m 2^e2, m in [0.5, 1)
    f.dl_exp = e2 < -40 ? -40 : (e2 > 40 ? 40 : e2);
  }
  PvHzReq hzr{zin, ldz, (int)lat_in, H, p->params + p->fc_latent.w_off, L.hz, false, false};
  // ---- the guide folded into the decoder launch (pv_sdec_fused.h PvEncFold): the plain-bf16 8-wave kernel, a workgroup's units a
  // whole number of images, the plain fc encoder of two hidden layers — BASELINE's headline config is exactly this.  No encoder
  // launch, no weight-image copy, no hand-off: the step is decoder launch -> latent backward + record sums -> small weight gradients.
  PvEncFold ef{};
  bool fold = false;
  auto fill_fold = [&]() {
    ef.params = p->params; ef.enc0 = p->enc[0]; ef.enc1 = p->enc[1]; ef.head = p->head;
    ef.x = p->x; ef.ldx = N; ef.eps = p->eps;
    ef.eact0 = L.eact[0]; ef.eact1 = L.eact[1]; ef.head_out = L.head;
    ef.z = L.z; ef.z_scale = L.z_scale; ef.z_loc_out = p->z_loc; ef.z_scale_out = p->z_scale;
    ef.tp = L.tp; ef.kl_part = L.kl_part; ef.hz = L.hz; ef.Wz = p->params + p->fc_latent.w_off;
    ef.lat_in = (int)lat_in; ef.z_dim = (int)z; ef.coord_dim = p->coord_dim;
    ef.has_r = p->has_r; ef.has_t = p->has_t; ef.has_s = p->has_s;
    ef.tp0 = p->t_prior[0]; ef.tp1 = p->t_prior[1]; ef.sc_prior = p->sc_prior; ef.beta = p->beta;
  };
  if (plan_guide_may_fold(p, L)) {       // (no cross-workgroup hand-off in it: fine under stream capture too)
    f.hz_scale = 2.8853900817779268f;
    fold = pv_sdec_fused_fold_ok(f, L.f_grid, p->fused == 2);
    if (fold) fill_fold();
    else f.hz_scale = 0.0f;
    // (the hosting launch owns whole images: it sums their per-row outputs itself in its epilogue — PvFused::part_rs, first slot)
    static const int ab_fold_rs = pv_exp_int("PV_FOLD_RS", 1);      // (experiments build: 0 = the latent backward reduces the rows)
    if (fold && want_grads && H == FD_H && f.llrow && ab_fold_rs) f.part_rs = L.f_part_hz + S * L.f_kmax * H;
    // ... and its dL/d(hz): the eight waves' partials summed next to the other column sums (PvFused::dhz_out)
    static const int ab_fold_dhz = pv_exp_int("PV_FOLD_DHZ", 1);
    if (fold && want_grads && H == FD_H && ab_fold_dhz) {
      f.dhz_out = L.dhz;
      if (ab_fold_dhz > 1 || ab_fold_dhz == 1) f.dzc_out = (ab_fold_dhz == 2) ? nullptr : L.dzc;      // (PV_FOLD_DHZ=2: dhz only, A/B)
    }
  }
  int kl_n = fold ? (int)B : L.kl_blocks;            // KL partial sums in L.kl_part: per sample when a workgroup runs an image's guide, else per 16-row block
  if (fold) {
    // (nothing to launch before the decoder kernel)
  } else if (p->fused >= 2 && L.enc_compact) {
    const PvFbPrep prep = pv_sdec_fused_bf16_prep_args(f, want_grads != 0, p->fused == 2);
    f.hz_scale = prep.scale;                          // the compact encoder's last launch writes scale * hz directly
    // (round 6) the guide as one workgroup per image (pv_guide_img.hip: no tile hand-offs, 11-13 us where the tiled one-launch
    // encoder takes 18-19) for the minibatch sizes where every image streaming the first-layer matrix from L2 is still cheaper
    // than that latency; the flags that are about the tiled encoder's launch form keep the tiled encoder
    fill_fold();
    const bool per_image = plan_guide_one_image(p, L) && pv_guide_img_ok(ef, (int)B) &&
                           !(p->flags & (PV_PLAN_ENC_TILED | PV_PLAN_ENC_TWO_LAUNCH | PV_PLAN_ENC_NO_WAIT));
    if (per_image) {
      PV_TRY(pv_guide_img_launch(ef, &prep, f.hz_scale, (int)B, s));
      kl_n = (int)B;
    } else {
      PV_TRY(guide_fwd(p, L, s, &prep, f.hz_scale));
    }
  } else {
    // conv encoder: the decoder's weight images ride in its weight-tiling launch; generic encoders: a launch of their own
    const bool prep_in_enc = p->fused >= 2 && L.enc_conv;
    const PvFbPrep prep = p->fused >= 2 ? pv_sdec_fused_bf16_prep_args(f, want_grads != 0, p->fused == 2) : PvFbPrep{};
    PV_TRY(guide_fwd(p, L, s, prep_in_enc ? &prep : nullptr, 0.0f, &hzr));
    if (p->fused >= 2 && !prep_in_enc) {
      PV_TRY(pv_sdec_fused_bf16_prep(f, want_grads != 0, p->fused == 2, s));
    } else if (p->fused >= 2) {
      // (done)
    } else if (want_grads) {
      hipError_t e = hipMemsetAsync(L.f_part_hz, 0, (size_t)(S * L.f_kmax * H) * sizeof(float), s);
      if (e != hipSuccess) return (int)e;
    }
  }
  if (!L.enc_compact && !hzr.done) {
    if (lat_in <= 16) PV_TRY(pv_smallk_linear(zin, ldz, p->params + p->fc_latent.w_off, L.hz, B, (int)lat_in, (int)H, s));
    else PV_TRY(linear_fwd(zin, ldz, p->params + p->fc_latent.w_off, nullptr, L.hz, nullptr, H, B, lat_in, H, PV_ACT_NONE,
                           L.scratch, L.scratch_bytes, s));
  }
  PvFusedOffsets o{p->dec[0].w_off, p->dec[0].b_off, p->dec[1].w_off, p->dec[1].b_off,
                   p->fc_coord.w_off, p->out.w_off, p->out.b_off};
  // per sample: ll_b, d(phi, scale, tx, ty), dL/d(hz), dL/d(z content), head backward -> L.dhead ; in the same
  // launch: the per-workgroup gradient records summed into the flat gradient
  PvLatentBwd lb{};
  lb.llrow = L.llrow; lb.rowtp = L.f_rowtp; lb.part_hz = L.f_part_hz; lb.Wz = p->params + p->fc_latent.w_off;
  // (round 6) the 4-wave kernels too, wherever every workgroup's units are exactly one sample (batch == grid, e.g. C2's 256 on 256 CUs)
  static const int ab_own = pv_exp_int("PV_OWN_SAMPLE", 1);
  if (ab_own && !f.dhz_out && f.part_rs && K == 0 && lat_in <= 16 && S == L.f_grid && f.units == S * (N / FD_UNIT) &&
      !p->dy && pv_sdec_fused_bf16_record_fmt(p->fused == 2, R / FD_UNIT, p->dec_kernel) == PV_REC_LANE_F32) {
    f.dhz_out = L.dhz; f.dzc_out = L.dzc; f.Wz = p->params + p->fc_latent.w_off; f.lat_in = (int)lat_in;
  }
  lb.part_rs = f.part_rs;
  lb.dhz_ready = f.dhz_out ? 1 : 0;
  lb.dzc_in = f.dzc_out;
  lb.llb = L.llb; lb.dhz = L.dhz; lb.M = R; lb.N = (int)N; lb.kmax = L.f_kmax; lb.H = H; lb.lat_in = (int)lat_in;
  PvHeadBwd& hb = lb.hb;
  hb.z = L.z; hb.z_scale = L.z_scale; hb.eps = p->eps;
  hb.head = L.enc_ext ? p->ext_head : L.head; hb.dhead = L.enc_ext ? p->ext_dhead : L.dhead;
  hb.scale_direct = L.enc_ext ? 1 : 0;
  hb.B = (int)B; hb.z_dim = (int)z; hb.coord_dim = p->coord_dim;
  hb.has_r = p->has_r; hb.has_t = p->has_t; hb.has_s = p->has_s;
  hb.tp0 = p->t_prior[0]; hb.tp1 = p->t_prior[1]; hb.sc_prior = p->sc_prior; hb.beta = p->beta;
  hb.ldh = L.enc_ext ? 0 : (int)plan_head_w(p);
  lb.K = (int)K; lb.alpha = L.alpha; lb.beta_disc = p->beta_disc;
  hb.w = p->row_w; lb.row_ll = p->row_elbo ? L.row_ll : nullptr; lb.dzc_out = p->dy ? L.dzc : nullptr;
  // compact encoder: every sample's dgrad chain runs in its latent_bwd workgroup (one dependent launch less)
  static const int chain_env = pv_exp_int("PV_CHAIN", 1) ? 1 : 0;          // PV_CHAIN=0: keep pv_enc_dgrad as its own launch (A/B timing, experiments build)
  const bool chain = chain_env && L.enc_compact && !L.enc_ext && 2 * z + K <= 256;
  if (chain) {
    lb.enc_n = p->n_enc; lb.enc_params = p->params; lb.enc_head = p->head;
    for (int i = 0; i < p->n_enc; ++i) { lb.enc_l[i] = p->enc[i]; lb.enc_act[i] = L.eact[i]; lb.enc_dp[i] = L.edp[i]; }
  }
  // (round 6, third cut) where every workgroup of the decoder launch owns one image — the launch that hosts the guide, or a 4-wave
  // launch at batch == grid — it also runs the image's latent backward and encoder chain in its epilogue (PvEncFold::chain), and the
  // step closes with ONE launch of workgroups that need
  // nothing from each other: record sums, small weight gradients, loss scalars (pv_elementwise.hip: pv_rec_wgrad_kernel).
  // Needs what that epilogue is written for (the plain iVAE step: no per-sample weights / extra outputs, a head of <= 16 outputs)
  // and — with the optimizer riding — records + tiles covering every parameter.
  static const int ab_tail = pv_exp_int("PV_FOLD_CHAIN", 1);
  const int rec_fmt = p->fused >= 2 ? pv_sdec_fused_bf16_record_fmt(p->fused == 2, R / FD_UNIT, p->dec_kernel) : PV_REC_ROWMAJOR;
  const PvGemm wz = wgrad_problem(L.dhz, H, zin, ldz, G + p->fc_latent.w_off, G + p->fc_coord.b_off, S, lat_in, H);
  PvGemm tail_probs[4];
  int tail_np = -1;
  // (f.dhz_out / dzc_out / part_rs together: the hosting 8-wave launch, or a 4-wave launch whose workgroups own one sample each)
  if (ab_tail && want_grads && chain && f.part_rs && f.dhz_out && f.dzc_out && K == 0 && H == FD_H && !p->row_w &&
      !p->row_elbo && !p->dy && p->head.out_dim <= 16 && p->n_enc == 2 && p->enc[0].out_dim == FD_H && p->enc[1].in_dim == FD_H &&
      p->enc[1].out_dim == FD_H && p->head.in_dim == FD_H) {      // (the epilogue's chain is written for two hidden layers of width 128)
    tail_np = compact_wgrad_problems(p, L, &wz, 1, tail_probs);
    if (tail_np > 0 && adam) {
      // every PARAMETER must be finalised by a record block or a tile (the launch has no Adam guests): the plan's layers, counted,
      // against what the records and the tiles cover.  What else the flat buffer holds — alignment padding, batch-norm statistics —
      // never has a gradient or a moment, and Adam leaves such an element where it is.
      auto lsz = [](const pv_layer& l) { return (int64_t)l.in_dim * l.out_dim + (l.b_off >= 0 ? l.out_dim : 0); };
      int64_t want = lsz(p->head) + lsz(p->fc_coord) + lsz(p->fc_latent) + lsz(p->out);
      for (int i = 0; i < p->n_enc; ++i) want += lsz(p->enc[i]);
      for (int i = 0; i < p->n_dec; ++i) want += lsz(p->dec[i]);
      int64_t cov = 2 * (int64_t)H * H + (int64_t)H * (3 + p->coord_dim) + 1;
      for (int i = 0; i < tail_np; ++i) cov += (int64_t)tail_probs[i].M * tail_probs[i].N + (tail_probs[i].rowsumA ? tail_probs[i].M : 0);
      if (cov != want || !adam_done || p->n_dec != 2) tail_np = -1;
    }
  }
  const bool own_chain = tail_np > 0;
  if (own_chain) {
    ef.chain = 1; ef.dhead = L.dhead; ef.ldh = (int)plan_head_w(p); ef.edp0 = L.edp[0]; ef.edp1 = L.edp[1]; ef.llb = L.llb;
    // ... and, in the hosting launch, the guide's first layer shared among the workgroups of a group (pv_sdec_fused_w8.hip, build 3):
    // its hand-off tags live in coop_flags, whose last word this step's closing launch increments
    static const int ab_coop = pv_exp_int("PV_COOP_L0", 0);          // (experiments build only; measured slower than every workgroup for itself)
    if (fold && ab_coop && L.f_grid <= 1024) { ef.coop = 1; ef.coop_flags = L.coop_flags; }
  }
  if (p->ev_start && p->ev_stop) (void)hipEventRecord((hipEvent_t)p->ev_start, s);
  if (p->fused >= 2) PV_TRY(pv_sdec_fused_bf16_launch(f, L.f_grid, want_grads != 0, p->fused == 2, s, fold ? &ef : nullptr,
                                                      (own_chain && !fold) ? &ef : nullptr));
  else PV_TRY(pv_sdec_fused_launch(f, L.f_grid, want_grads != 0, s));
  if (p->ev_start && p->ev_stop) (void)hipEventRecord((hipEvent_t)p->ev_stop, s);
  if (!want_grads && K > 0) {                        // llb[b] = sum_k alpha_bk ll_kb
    PvLatentBwd lf{};
    lf.llrow = L.llrow; lf.llb = L.llb; lf.M = R; lf.N = (int)N; lf.H = 0; lf.K = (int)K; lf.alpha = L.alpha;
    lf.hb.B = (int)B; lf.fwd_only = 1;
    PV_TRY(pv_latent_bwd(lf, s));
    return pv_finish_scalars(L.llb, (int)B, p->scalars, L.kl_part, L.kl_blocks, 1.0f, s);
  }
  if (!want_grads) {
    PV_TRY(pv_segsum(L.llrow, B, N, L.llb, s));
    PV_TRY(weigh_llb(p, L, s));
    PV_TRY(pv_finish_scalars(L.llb, (int)B, p->scalars, (L.enc_compact || hzr.kl) ? L.kl_part : nullptr, kl_n, 1.0f /* partials come scaled */, s));
    return extra_outputs(p, L, nullptr, lat_in, s);
  }
  // conv encoder with a side stream: the head's weight gradient forks off this launch (encoder_bwd)
  static const int ab_side = pv_exp_int("PV_HEAD_SIDE", 1);
  static const int ab_fin = pv_exp_int("PV_FIN_RIDE", 1);
  const bool head_side = ab_side && L.enc_conv && !L.enc_ext && pv_side_stream_for(s, p->flags) && !pv_convhead_wgrad_uses_ws() &&
                         pv_convhead_supported(L.cF, p->head.out_dim) && L.chead_wt;
  if (head_side) pv_fork_arm();
  if (own_chain) {
    const PvFinish fin{L.llb, (int)B, p->scalars, L.kl_part, kl_n, 1.0f /* scaled */};
    PV_TRY(pv_rec_wgrad(L.f_part, L.f_grid, G, o, p->coord_dim, rec_fmt, tail_probs, tail_np, adam, &fin, s,
                        (fold && ef.coop) ? L.coop_flags + L.f_grid : nullptr));
    if (adam) *adam_done = true;
    return extra_outputs(p, L, L.dzc, lat_in, s);
  }
  // (rec_fmt: the record format the decoder launch above wrote: pv_sdec_fused.h PV_REC_*)
  PV_TRY(pv_latent_bwd_reduce(lb, L.f_part, L.f_grid, G, o, p->coord_dim, s, rec_fmt));
  // the loss scalars ride in the encoder dgrad launch (compact encoder), in the last weight-gradient launch (conv encoder) or get their own
  PvFinish fin{L.llb, (int)B, p->scalars, (L.enc_compact || hzr.kl) ? L.kl_part : nullptr, kl_n, 1.0f /* scaled */};
  const bool fin_rides = L.enc_compact || (ab_fin && L.enc_conv && !L.enc_ext);
  if (!fin_rides) PV_TRY(pv_finish_scalars(fin.llb, fin.B, fin.scalars, fin.kl_part, fin.n_part, fin.beta, s));
  // fc_latent: dWz = dhz^T zin; its row sums are fc_coord's bias gradient (dbc = sum_b dhz[b])
  // (jiVAE: over the K*B decoder samples, zin = [z content | onehot(k)])
  PV_TRY(encoder_bwd(p, L, &wz, 1, s, fin_rides ? &fin : nullptr, adam, adam_done, chain, head_side));
  return extra_outputs(p, L, L.dzc, lat_in, s);
}

int loss_and_grads_layered(const pv_ivae_plan* p, const Layout& L, int want_grads, hipStream_t s) {
  const int64_t K = plan_K(p);
  const int64_t B = p->batch, N = p->n_pix, R = L.rows, z = p->z_dim;
  const int64_t S = plan_S(p);                       // decoder samples (jiVAE: K per input, ordered [k][b])
  const int64_t lat_in = plan_lat_in(p);
  float* G = p->grads;
  void* ws = L.scratch;
  const int64_t wsb = L.scratch_bytes;
  const int nd = p->n_dec;

  // ---------------- forward ----------------
  PV_TRY(guide_fwd(p, L, s));
  const bool sampled = K > 0 && p->class_onehot != nullptr;       // jiVAE without enumeration (plan->class_onehot)
  if (sampled) {
    if (p->coord_dim > 0) return PV_EINVAL;
    PV_TRY(pv_jiv_sampled_prep(L.alpha, p->class_onehot, L.sw, L.jfix, p->beta_disc, (int)B, (int)K, s));
  }
  const int coord = (int)(z - p->latent_dim);
  const bool cat_in = p->c_dim > 0 || K > 0;
  const float* zin = cat_in ? L.zy : (p->coord_dim > 0 ? L.z + coord : L.z);
  const int64_t ldz = cat_in ? lat_in : z;
  PV_TRY(decoder_hidden_fwd(p, L, zin, ldz, lat_in, s));

  float* cur = L.dbuf[0];      // dL/d(pre-activation) of the layer being processed
  float* oth = L.dbuf[1];
  const float* hlast = L.dact[nd - 1];
  const int Hl = p->dec[nd - 1].out_dim;
  if (p->coord_dim > 0) {
    PvOutLik o{};
    o.h = hlast; o.hpre = L.dpre_[nd - 1]; o.ldh = Hl; o.wo = p->params + p->out.w_off;
    o.bo = p->out.b_off >= 0 ? p->params + p->out.b_off : nullptr; o.x = p->x; o.loc = p->loc; o.llrow = L.llrow;
    o.dpre = want_grads ? cur : nullptr; o.part_dwo = L.part_dwo; o.part_dbo = L.part_dbo; o.M = R; o.H = Hl;
    o.lik = p->lik; o.sigmoid_out = p->sigmoid_out; o.act_last = p->dec[nd - 1].act; o.sig = p->decoder_sig;
    o.sw = K > 0 ? L.sw : p->row_w; o.N = (int)N; o.xmod = K > 0 ? B * N : 0;
    PV_TRY(pv_out_lik(o, s));
  } else {
    // oth <- dL/dlogits (R, N); jiVAE: one pass per enumerated class against the same observations, rows then
    // weighted by alpha (the decoder's gradients become the enumerated expectation)
    for (int64_t k = 0; k < (K > 0 ? K : 1); ++k)
      PV_TRY(pv_lik_elem(L.logits + k * B * N, p->x, B * N, p->lik, p->sigmoid_out, p->decoder_sig,
                         p->loc ? p->loc + k * B * N : nullptr, L.llrow + k * B * N, want_grads ? oth + k * B * N : nullptr, s));
    if (K > 0 && want_grads) PV_TRY(pv_scale_rows(oth, L.sw, R, N, s));
    if (K == 0 && p->row_w && want_grads) PV_TRY(pv_scale_rows(oth, p->row_w, R, N, s));
  }
  if (K > 0) {
    PV_TRY(pv_segsum(L.llrow, S, N, L.llkb, s));
    if (!want_grads && sampled)
      PV_TRY(pv_jiv_combine_sampled(L.llkb, L.alpha, p->class_onehot, L.llb, nullptr, 0, 0, nullptr, 0, (int)z, (int)B,
                                    (int)K, p->beta, p->beta_disc, 0, nullptr, nullptr, nullptr, s));
    else if (!want_grads) PV_TRY(pv_jiv_combine(L.llkb, L.alpha, L.llb, nullptr, 0, 0, nullptr, 0, (int)z, (int)B, (int)K,
                                           p->beta_disc, 0, s));
  } else {
    PV_TRY(pv_segsum(L.llrow, B, N, L.llb, s));
    PV_TRY(weigh_llb(p, L, s));
  }
  if (K > 0 && want_grads) {
    // (llb is formed by pv_jiv_combine at the end of the decoder backward; the scalars are finished there)
  } else
  PV_TRY(pv_finish_scalars(L.llb, (int)B, p->scalars, L.enc_compact ? L.kl_part : nullptr, L.kl_blocks, 1.0f /* partials come scaled */, s));
  if (!want_grads && sampled) PV_TRY(pv_jiv_sampled_fix(p->scalars, L.jfix, s));
  if (!want_grads) return extra_outputs(p, L, nullptr, lat_in, s);

  // ---------------- backward: decoder ----------------
  if (p->coord_dim > 0) {
    const int64_t ob = pv_out_lik_blocks(R);
    PV_TRY(pv_reduce_partials(L.part_dwo, (int)ob, Hl, G + p->out.w_off, Hl, s));
    if (p->out.b_off >= 0) PV_TRY(pv_reduce_partials(L.part_dbo, (int)ob, 1, G + p->out.b_off, 1, s));
  } else {
    // out layer of the vanilla decoder: logits = hlast Wout^T + bout
    PV_TRY(linear_wgrad(oth, N, hlast, Hl, G + p->out.w_off, p->out.b_off >= 0 ? G + p->out.b_off : nullptr, R, Hl, N,
                        ws, wsb, s));
    PV_TRY(linear_dgrad(oth, N, p->params + p->out.w_off, cur, Hl, hlast, L.dpre_[nd - 1], Hl, p->dec[nd - 1].act, R,
                        Hl, N, ws, wsb, s));
  }
  for (int i = nd - 1; i >= 0; --i) {
    const pv_layer& l = p->dec[i];
    const float* in; const float* inpre; int64_t ldin; int act_in;
    if (i > 0) { in = L.dact[i - 1]; inpre = L.dpre_[i - 1]; ldin = p->dec[i - 1].out_dim; act_in = p->dec[i - 1].act; }
    else if (p->coord_dim > 0) { in = L.h0; inpre = nullptr; ldin = p->fc_coord.out_dim; act_in = PV_ACT_TANH; }
    else { in = zin; inpre = nullptr; ldin = ldz; act_in = PV_ACT_NONE; }
    PV_TRY(linear_wgrad(cur, l.out_dim, in, ldin, G + l.w_off, l.b_off >= 0 ? G + l.b_off : nullptr, R, l.in_dim,
                        l.out_dim, ws, wsb, s));
    if (i > 0 || p->coord_dim > 0) {
      PV_TRY(linear_dgrad(cur, l.out_dim, p->params + l.w_off, oth, l.in_dim, in, inpre, ldin, act_in, R, l.in_dim,
                          l.out_dim, ws, wsb, s));
      float* t = cur; cur = oth; oth = t;
    } else {
      // vanilla decoder: dL/d(decoder latent input) (B, lat_in)
      PV_TRY(linear_dgrad(cur, l.out_dim, p->params + l.w_off, L.dzc, lat_in, nullptr, nullptr, 0, PV_ACT_NONE, R,
                          l.in_dim, l.out_dim, ws, wsb, s));
    }
  }
  if (p->coord_dim > 0) {
    // cur = dL/d(pre-tanh of coord_latent) (R, H0)
    const int H0 = p->fc_coord.out_dim;
    PvCoordLatBwd cb{};
    cb.dpre0 = cur; cb.grid = p->grid; cb.tp = L.tp; cb.Wc = p->params + p->fc_coord.w_off;
    cb.part_hz = L.part_hz; cb.part_wc = L.part_wc; cb.part_tp = L.part_tp;
    cb.N = (int)N; cb.cd = p->coord_dim; cb.H0 = H0; cb.rows_per_chunk = L.rows_per_chunk;
    PV_TRY(pv_coordlat_bwd(cb, L.nchunk, (int)S, s));
    const int np = (int)(S * L.nchunk);
    PV_TRY(pv_reduce_mid(L.part_hz, (int)S, L.nchunk, H0, L.dhz, s));
    PV_TRY(pv_reduce_partials(L.part_hz, np, H0, G + p->fc_coord.b_off, H0, s));
    PV_TRY(pv_reduce_partials(L.part_wc, np, (int64_t)H0 * p->coord_dim, G + p->fc_coord.w_off,
                              (int64_t)H0 * p->coord_dim, s));
    PV_TRY(pv_reduce_mid(L.part_tp, (int)S, L.nchunk, 4, L.dtp, s));
    // fc_latent: hz = zin Wz^T
    PV_TRY(linear_wgrad(L.dhz, H0, zin, ldz, G + p->fc_latent.w_off, nullptr, S, lat_in, H0, ws, wsb, s));
    PV_TRY(linear_dgrad(L.dhz, H0, p->params + p->fc_latent.w_off, L.dzc, lat_in, nullptr, nullptr, 0, PV_ACT_NONE, S,
                        lat_in, H0, ws, wsb, s));
  }

  if (K > 0) {
    // sum the K replicas' dL/dz, form ll_b and the class-logit gradients; then the loss scalars
    if (sampled)
      PV_TRY(pv_jiv_combine_sampled(L.llkb, L.alpha, p->class_onehot, L.llb, L.dzc, (int)lat_in, (int)(lat_in - K), L.dhead,
                                    (int)plan_head_w(p), (int)z, (int)B, (int)K, p->beta, p->beta_disc, 1, L.z, L.head,
                                    L.z_scale, s));
    else
    PV_TRY(pv_jiv_combine(L.llkb, L.alpha, L.llb, L.dzc, (int)lat_in, (int)(lat_in - K), L.dhead, (int)plan_head_w(p), (int)z,
                          (int)B, (int)K, p->beta_disc, 1, s, p->coord_dim > 0 ? L.dtp : nullptr));
    PV_TRY(pv_finish_scalars(L.llb, (int)B, p->scalars, L.enc_compact ? L.kl_part : nullptr, L.kl_blocks, 1.0f, s));
    if (sampled) PV_TRY(pv_jiv_sampled_fix(p->scalars, L.jfix, s));
  }
  return latent_encoder_bwd(p, L, lat_in, 4, 1, s);
}

}  // namespace

extern "C" int pv_version(void) { return PV_ABI_VERSION; }
extern "C" int pv_experiments_build(void) {
#ifdef PV_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

// the plan pv_ivae_decode works on; jiVAE.decode(z, y) (jivae.py:255-267) is one decoder row block
// per given (z, one-hot class) pair — the class vector is a conditioning input there, not an enumeration axis
static pv_ivae_plan decode_plan(const pv_ivae_plan* plan) {
  pv_ivae_plan lay = *plan;
  if (lay.discrete_dim > 0) {
    lay.c_dim += lay.discrete_dim;
    lay.head.out_dim -= lay.discrete_dim;
    lay.discrete_dim = 0;
  }
  return lay;
}

static pv_ivae_plan guide_plan(const pv_ivae_plan* plan);

// ---- forward-only decode on the fused spatial-decoder kernel (models/base.py:145-171) ----
// decode(z, angle, shift, scale) of an invariant model whose decoder the fused kernel is specialised for: the per-sample
// transform is the uniform (angle, scale, shift), hz = fc_latent(z), and the persistent kernel runs its forward half only
// and writes `loc` — no (B N) x 128 activation is materialised (the layered path needs 3 of them: 63 GB at B = 32768).
// Always at fp32-class precision (fused = 1: f32 matrix instructions; otherwise bf16 split precision), whatever the
// training precision: reconstructions are what parity is judged on.
struct DecodeLayout { float* tp; float* hz; float* wimg; float* xdummy; void* scratch; int64_t scratch_bytes, total; };
static bool decode_fused(const pv_ivae_plan* lay) { return lay->fused && lay->coord_dim > 0 && pv_sdec_fused_supported(lay); }
static void carve_decode(const pv_ivae_plan* lay, char* base, DecodeLayout& D) {
  Carver c{base, 0};
  const int64_t B = lay->batch, H0 = lay->fc_coord.out_dim, lat_in = plan_lat_in(lay);
  D.tp = c.take(B * 8);
  D.hz = c.take(B * H0);
  D.wimg = c.take(FB_WIMG_BYTES / (int64_t)sizeof(float));
  D.xdummy = c.take(FD_UNIT);
  D.scratch_bytes = pv_align_up(lat_in > 16 ? gemm_ws_need(B, H0, lat_in) : 0, 256);
  D.scratch = base ? (void*)(base + c.off) : nullptr;
  c.off += D.scratch_bytes;
  D.total = c.off;
}
static int decode_fused_run(const pv_ivae_plan* lay, const float* z, float angle, float shift_x, float shift_y, float scale,
                            float* loc, hipStream_t s) {
  DecodeLayout D;
  carve_decode(lay, (char*)lay->ws, D);
  if (lay->ws_bytes < D.total) return PV_EWS;
  const int64_t B = lay->batch, N = lay->n_pix, lat_in = plan_lat_in(lay);
  const int H = FD_H;
  if (lay->fc_latent.in_dim != lat_in) return PV_EINVAL;
  PV_TRY(pv_fill_tp(D.tp, (int)B, angle, scale, shift_x, shift_y, s));
  if (lat_in <= 16) PV_TRY(pv_smallk_linear(z, lat_in, lay->params + lay->fc_latent.w_off, D.hz, B, (int)lat_in, H, s));
  else PV_TRY(linear_fwd(z, lat_in, lay->params + lay->fc_latent.w_off, nullptr, D.hz, nullptr, H, B, lat_in, H, PV_ACT_NONE,
                         D.scratch, D.scratch_bytes, s));
  hipError_t e = hipMemsetAsync(D.xdummy, 0, FD_UNIT * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  PvFused f{};
  f.x = D.xdummy; f.x_units = 1;                       // (no observations: every unit reads the same 16 zeros)
  f.grid = lay->grid; f.tp = D.tp; f.hz = D.hz;
  f.Wc = lay->params + lay->fc_coord.w_off; f.bc = lay->params + lay->fc_coord.b_off;
  f.W1 = lay->params + lay->dec[0].w_off; f.b1 = lay->params + lay->dec[0].b_off;
  f.W2 = lay->params + lay->dec[1].w_off; f.b2 = lay->params + lay->dec[1].b_off;
  f.wo = lay->params + lay->out.w_off; f.bo = lay->params + lay->out.b_off;
  f.llrow = nullptr; f.loc = loc; f.wimg = D.wimg;
  f.M = B * N; f.units = f.M / FD_UNIT; f.N = (int)N; f.cd = lay->coord_dim; f.B = (int)B;
  f.lik = PV_LIK_GAUSSIAN; f.sigmoid_out = lay->sigmoid_out; f.sig = 1.0f;      // loc = sigmoid(a) or a
  f.sel = lay->dec_kernel;
  const int grid = pv_sdec_fused_grid(f.units);
  if (lay->fused == 1) return pv_sdec_fused_launch(f, grid, false, s);
  f.hz_scale = 0.0f;                                   // hz is written unscaled here; the 8-wave kernels scale it themselves
  PV_TRY(pv_sdec_fused_bf16_prep(f, false, true, s));  // weight images (pre-scaled or not: the launch below makes the same choice)
  return pv_sdec_fused_bf16_launch(f, grid, false, true, s);
}

// what: PV_WS_ALL = enough for every entry point at this batch; PV_WS_STEP / _ENCODE / _DECODE = for that call alone
// (the training step's fused layout is far smaller than the layered one pv_ivae_decode uses at the same batch)
extern "C" int64_t pv_ivae_workspace_bytes_for(const pv_ivae_plan* plan, int what) {
  if (what < PV_WS_ALL || what > PV_WS_DECODE) return PV_EINVAL;
  if (plan && plan->ext_decoder) {                       // only the guide half runs in the library
    const pv_ivae_plan q = guide_plan(plan);
    if (!valid_plan(&q)) return PV_EINVAL;
    Layout L;
    carve(&q, nullptr, L, what == PV_WS_ENCODE, what == PV_WS_ENCODE);
    return L.total;
  }
  if (!valid_plan(plan)) return PV_EINVAL;
  Layout L;
  int64_t total = 0;
  if (what == PV_WS_ALL || what == PV_WS_STEP) {
    carve(plan, nullptr, L);
    total = L.total;
  }
  if (what == PV_WS_ALL || what == PV_WS_ENCODE) {   // encode / decode always use the layered layout, B samples
    pv_ivae_plan q = *plan;
    q.fused = 0;
    carve(&q, nullptr, L, true, true);
    if (L.total > total) total = L.total;
  }
  if (what == PV_WS_ALL || what == PV_WS_DECODE) {
    pv_ivae_plan q = decode_plan(plan);
    if (decode_fused(&q)) {
      DecodeLayout D;
      carve_decode(&q, nullptr, D);
      if (D.total > total) total = D.total;
    } else {
      carve(&q, nullptr, L, true);
      if (L.total > total) total = L.total;
    }
  }
  return total;
}

extern "C" int64_t pv_ivae_workspace_bytes(const pv_ivae_plan* plan) { return pv_ivae_workspace_bytes_for(plan, PV_WS_ALL); }

extern "C" int pv_ivae_uses_fused(const pv_ivae_plan* plan) {
  return valid_plan(plan) && plan->fused && pv_sdec_fused_supported(plan) ? 1 : 0;
}

extern "C" int pv_ivae_guide_folds(const pv_ivae_plan* plan) {
  if (!valid_plan(plan) || !plan->fused || !pv_sdec_fused_supported(plan)) return 0;
  Layout L;
  carve(plan, nullptr, L);                             // (offsets only: nothing is dereferenced)
  if (!L.fused || !plan_guide_may_fold(plan, L)) return 0;
  PvFused f{};
  f.M = L.rows; f.units = L.rows / FD_UNIT; f.N = plan->n_pix; f.B = (int)plan_S(plan); f.sel = plan->dec_kernel;
  return pv_sdec_fused_fold_ok(f, L.f_grid, false) ? 1 : 0;
}

// test hook (not in include/; pv_convstack.h: conv_trace): offsets into plan->ws of the conv encoder's stored activations and
// max-pool winners after a pv_ivae_loss_and_grads call — tests/test_gpu_parity.py compares gradients with the float64 oracle
// under the HIP forward's own decisions and counts the decisions that differ
extern "C" int pv_debug_ivae_conv_trace(const pv_ivae_plan* plan, int64_t* out) {
  if (!valid_plan(plan) || !plan->ws || !out || plan->n_enc_ops <= 0) return PV_EINVAL;
  Layout L;
  carve(plan, (char*)plan->ws, L);
  if (!L.enc_conv || L.cF < 0 || plan->ws_bytes < L.total) return PV_EINVAL;
  pvcs::conv_trace(plan->enc_ops, plan->n_enc_ops, plan->enc_ndim, plan->batch, L.ces, L.cea, L.ccode, L.ccode2,
                   (const char*)plan->ws, out);
  return 0;
}

extern "C" int pv_ivae_loss_and_grads(const pv_ivae_plan* plan, int want_grads, void* stream) {
  PV_RANGE("pv_ivae_loss_and_grads");
  if (plan && plan->ext_decoder) return PV_EINVAL;      // (pv_ivae_guide / pv_ivae_guide_backward)
  if (!valid_plan(plan) || !plan->params || !plan->x || !plan->eps || !plan->scalars || !plan->ws) return PV_EINVAL;
  if (want_grads && !plan->grads) return PV_EINVAL;
  if (plan->ext_encoder && (!plan->ext_head || (want_grads && !plan->ext_dhead))) return PV_EINVAL;
  if (plan->coord_dim > 0 && !plan->grid) return PV_EINVAL;
  Layout L;
  carve(plan, (char*)plan->ws, L);
  if (plan->ws_bytes < L.total) return PV_EWS;
  hipStream_t s = (hipStream_t)stream;
  // the sampled-class objective (Trace_ELBO on a drawn class) exists for the vanilla decoder only — the reference's own model
  // cannot run it with invariances (models/jivae.py:181-189) — and only the layer-by-layer path implements it
  if (plan->class_onehot && (L.fused || plan->coord_dim > 0)) return PV_EINVAL;
  if (L.fused) return loss_and_grads_fused(plan, L, want_grads, s);
  return loss_and_grads_layered(plan, L, want_grads, s);
}

// ---- external decoder (plan->ext_decoder): the library's half of the step works on a copy of the plan whose decoder is
// a placeholder vanilla one (all of z is "content": no transform parameters, no fc_latent product)
static pv_ivae_plan guide_plan(const pv_ivae_plan* plan) {
  pv_ivae_plan q = *plan;
  q.fused = 0; q.coord_dim = 0; q.has_r = q.has_t = q.has_s = 0; q.latent_dim = q.z_dim;
  q.lik = PV_LIK_GAUSSIAN; q.sigmoid_out = 1;
  q.n_dec = 1;
  q.dec[0] = pv_layer{q.z_dim + q.c_dim, 16, PV_ACT_NONE, 0, 0, -1};
  q.out = pv_layer{16, q.n_pix, PV_ACT_NONE, 0, 0, -1};
  q.row_w = nullptr; q.row_elbo = nullptr; q.dy = nullptr;
  return q;
}

extern "C" int pv_ivae_guide(const pv_ivae_plan* plan, void* stream) {
  PV_RANGE("pv_ivae_guide");
  if (!plan || !plan->ext_decoder || plan->discrete_dim > 0 || plan->row_w || plan->row_elbo || plan->dy) return PV_EINVAL;
  const pv_ivae_plan q = guide_plan(plan);
  if (!valid_plan(&q) || !q.params || !q.x || !q.eps || !q.scalars || !q.ws || !q.ext_z) return PV_EINVAL;
  if (q.ext_encoder && !q.ext_head) return PV_EINVAL;
  Layout L;
  carve(&q, (char*)q.ws, L);
  if (q.ws_bytes < L.total) return PV_EWS;
  hipStream_t s = (hipStream_t)stream;
  PV_TRY(guide_fwd(&q, L, s));
  hipError_t e = hipMemcpyAsync(q.ext_z, L.z, (size_t)q.batch * q.z_dim * sizeof(float), hipMemcpyDeviceToDevice, s);
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int pv_ivae_guide_backward(const pv_ivae_plan* plan, int want_grads, void* stream) {
  PV_RANGE("pv_ivae_guide_backward");
  if (!plan || !plan->ext_decoder || plan->discrete_dim > 0) return PV_EINVAL;
  const pv_ivae_plan q = guide_plan(plan);
  if (!valid_plan(&q) || !q.params || !q.eps || !q.scalars || !q.ws || !q.ext_ll) return PV_EINVAL;
  if (want_grads && (!q.ext_dz || (!q.ext_encoder && !q.grads) || (q.ext_encoder && !q.ext_dhead))) return PV_EINVAL;
  Layout L;
  carve(&q, (char*)q.ws, L);
  if (q.ws_bytes < L.total) return PV_EWS;
  hipStream_t s = (hipStream_t)stream;
  // scalars[1] = sum log p(x|z) from the caller; [2], [3] from the guide (kernel partials on the compact encoder path)
  PV_TRY(pv_finish_scalars(q.ext_ll, 1, q.scalars, L.enc_compact ? L.kl_part : nullptr, L.kl_blocks, 1.0f, s));
  if (!want_grads) return 0;
  L.dzc = const_cast<float*>(q.ext_dz);                // d(-ll)/dz for every column of z
  return latent_encoder_bwd(&q, L, q.z_dim, 4, 1, s);
}

// SVI.step in one call.  On the fused-decoder path with the compact encoder or a conv encoder the Adam update rides in the last
// gradient launch (pv_wgrad.hip: every element is updated by whoever finalises its gradient; bit-identical to
// pv_ivae_loss_and_grads + pv_adam_step, one launch fewer); everywhere else it is exactly that pair of calls.
extern "C" int pv_ivae_step(const pv_ivae_plan* plan, void* stream) {
  PV_RANGE("pv_ivae_step");
  if (plan && !plan->ext_decoder && valid_plan(plan) && plan->params && plan->x && plan->eps && plan->scalars && plan->ws &&
      plan->grads && plan->adam_m && plan->adam_v && plan->adam_step >= 1 && plan->n_params > 0 &&
      !(plan->coord_dim > 0 && !plan->grid) && !plan->ext_encoder &&
      // extra outputs (dy, row_elbo) are computed AFTER the encoder backward from the first-layer weights: with Adam
      // riding in that launch they would see the updated weights — such plans take the two-call sequence below
      !plan->dy && !plan->row_elbo) {
    Layout L;
    carve(plan, (char*)plan->ws, L);
    if (plan->ws_bytes < L.total) return PV_EWS;
    if (L.fused && (L.enc_compact || L.enc_conv)) {  // (conv encoder: in the launch of fc_latent's weight gradient, the step's last)
      const double bc1 = 1.0 - pow((double)plan->adam_beta1, (double)plan->adam_step);
      const double bc2 = 1.0 - pow((double)plan->adam_beta2, (double)plan->adam_step);
      const PvAdamFuse ad{plan->params, plan->grads, plan->adam_m, plan->adam_v, plan->n_params, plan->adam_beta1,
                          plan->adam_beta2, plan->adam_eps, (float)((double)plan->lr / bc1), (float)sqrt(bc2)};
      bool done = false;
      PV_TRY(loss_and_grads_fused(plan, L, 1, (hipStream_t)stream, &ad, &done));
      if (done) return 0;
      return pv_adam_step(plan->params, plan->grads, plan->adam_m, plan->adam_v, plan->n_params, plan->lr,
                          plan->adam_beta1, plan->adam_beta2, plan->adam_eps, plan->adam_step, stream);
    }
  }
  PV_TRY(pv_ivae_loss_and_grads(plan, 1, stream));
  return pv_adam_step(plan->params, plan->grads, plan->adam_m, plan->adam_v, plan->n_params, plan->lr,
                      plan->adam_beta1, plan->adam_beta2, plan->adam_eps, plan->adam_step, stream);
}

extern "C" int pv_ivae_encode(const pv_ivae_plan* plan, float* z_loc, float* z_scale, void* stream) {
  PV_RANGE("pv_ivae_encode");
  if (!plan) return PV_EINVAL;
  pv_ivae_plan lay = plan->ext_decoder ? guide_plan(plan) : *plan;
  if (!valid_plan(&lay) || plan->ext_encoder || !plan->params || !plan->x || !plan->ws || !z_loc || !z_scale) return PV_EINVAL;
  lay.fused = 0;
  plan = &lay;
  Layout L;
  carve(plan, (char*)plan->ws, L, true, true);
  if (plan->ws_bytes < L.total) return PV_EWS;
  hipStream_t s = (hipStream_t)stream;
  PV_TRY(encoder_fwd(plan, L, s));
  // split the merged head: z_loc = head[:, :z], z_scale = softplus(head[:, z:])  (fc.py:59-60)
  PvHead h{};
  h.head = L.head; h.eps = L.z_scale /* unused values; any valid buffer */; h.z = L.z; h.z_scale = L.z_scale;
  h.z_loc_out = z_loc; h.z_scale_out = z_scale; h.tp = nullptr; h.zy = nullptr; h.scalars = (float*)L.dhead;
  h.B = plan->batch; h.z_dim = plan->z_dim; h.c_dim = 0; h.coord_dim = 0; h.beta = 0.0f;
  h.ldh = (int)plan_head_w(plan);
  PV_TRY(pv_head_fwd(h, s));
  if (plan->discrete_dim > 0 && plan->alpha)
    PV_TRY(pv_softmax_rows(L.head + 2 * plan->z_dim, plan_head_w(plan), plan->batch, plan->discrete_dim, plan->alpha, s));
  return 0;
}

extern "C" int pv_ivae_decode(const pv_ivae_plan* plan, const float* z, float angle, float shift_x, float shift_y,
                              float scale, float* loc, void* stream) {
  PV_RANGE("pv_ivae_decode");
  if (!valid_plan(plan) || !plan->params || !plan->ws || !z || !loc) return PV_EINVAL;
  if (plan->coord_dim > 0 && !plan->grid) return PV_EINVAL;
  pv_ivae_plan lay = decode_plan(plan);
  plan = &lay;
  if (decode_fused(plan)) return decode_fused_run(plan, z, angle, shift_x, shift_y, scale, loc, (hipStream_t)stream);
  lay.fused = 0;
  Layout L;
  carve(plan, (char*)plan->ws, L, true);
  if (plan->ws_bytes < L.total) return PV_EWS;
  hipStream_t s = (hipStream_t)stream;
  const int64_t B = plan->batch, N = plan->n_pix, R = L.rows;
  const int64_t lat_in = plan_lat_in(plan);
  if (plan->coord_dim > 0) PV_TRY(pv_fill_tp(L.tp, (int)B, angle, scale, shift_x, shift_y, s));
  PV_TRY(decoder_hidden_fwd(plan, L, z, lat_in, lat_in, s));
  const int nd = plan->n_dec;
  if (plan->coord_dim > 0) {
    PvOutLik o{};
    o.h = L.dact[nd - 1]; o.ldh = plan->dec[nd - 1].out_dim; o.wo = plan->params + plan->out.w_off;
    o.bo = plan->out.b_off >= 0 ? plan->params + plan->out.b_off : nullptr;
    o.x = L.llrow /* unused for loc */; o.loc = loc; o.llrow = nullptr; o.dpre = nullptr; o.M = R;
    o.H = plan->dec[nd - 1].out_dim; o.lik = PV_LIK_GAUSSIAN; o.sigmoid_out = plan->sigmoid_out; o.sig = 1.0f;
    o.act_last = plan->dec[nd - 1].act;
    return pv_out_lik(o, s);
  }
  return pv_lik_elem(L.logits, L.logits, B * N, PV_LIK_GAUSSIAN, plan->sigmoid_out, 1.0f, loc, nullptr, nullptr, s);
}

// ---- building blocks -------------------------------------------------------------------------
extern "C" int64_t pv_linear_workspace_bytes(int64_t M, int64_t K, int64_t N) {
  if (M < 0 || K <= 0 || N <= 0) return PV_EINVAL;
  int64_t need = gemm_ws_need(M, N, K);
  const int64_t a = gemm_ws_need(M, K, N), b = gemm_ws_need(N, K, M), c = pv_colsum_ws(M, (int)N);
  if (a > need) need = a;
  if (b > need) need = b;
  if (c > need) need = c;
  return pv_align_up(need, 256);
}

extern "C" int pv_linear_fwd(const float* x, int64_t ldx, const float* w, const float* b, float* y, float* pre,
                             int64_t ldy, int64_t M, int64_t K, int64_t N, int act, void* ws, int64_t ws_bytes,
                             void* stream) {
  if (!x || !w || !y || M < 0 || K <= 0 || N <= 0) return PV_EINVAL;
  return linear_fwd(x, ldx, w, b, y, pre, ldy, M, K, N, act, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int pv_linear_bwd(const float* dpre, int64_t lddp, const float* x, int64_t ldx, const float* w, float* dx,
                             int64_t lddx, const float* xact, const float* xpre, int64_t ldxa, int act_prev, float* dw,
                             float* db, int64_t M, int64_t K, int64_t N, void* ws, int64_t ws_bytes, void* stream) {
  if (!dpre || M < 0 || K <= 0 || N <= 0) return PV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dx) {
    if (!w) return PV_EINVAL;
    PV_TRY(linear_dgrad(dpre, lddp, w, dx, lddx, xact, xpre, ldxa, act_prev, M, K, N, ws, ws_bytes, s));
  }
  if (dw || db) {
    if (dw && !x) return PV_EINVAL;
    PV_TRY(linear_wgrad(dpre, lddp, x, ldx, dw, db, M, K, N, ws, ws_bytes, s));
  }
  return 0;
}
