// pv_convstack.h — a convolutional op sequence (pv_op[]: conv k3/k1 + activation, 2x max-pool, 2x nearest upsample) over
// channels-last activations: shapes, workspace needs, forward and backward.  Shared by the VED step (pv_ved.hip) and
// the iVAE step with a convolutional encoder (pv_plan.hip).  Every convolution is a GEMM (im2col for kernel 3, the
// activation itself for kernel 1) on pv_gemm.hip's f32-input MFMA kernel with bias + activation fused; backward
// re-creates each im2col instead of keeping it.
#pragma once
#include "pv_common.h"
#include "pv_linear.h"
#include "pv_conv.h"
#include "pv_fb_layout.h"
#include "pv_side.h"

namespace pvcs {

struct Shape { int H, W, C; int64_t elems(int64_t B) const { return B * H * W * C; } };

inline bool op_shape(const pv_op& o, int nd, const Shape& in, Shape& out) {
  out = in;
  if (o.kind == PV_OP_CONV) {
    if (o.cin != in.C || o.cout < 1 || (o.ksize != 1 && o.ksize != 3)) return false;
    out.C = o.cout;
  } else if (o.kind == PV_OP_MAXPOOL2) {
    out.H = in.H / 2; out.W = nd == 2 ? in.W / 2 : 1;
    if (out.H < 1 || out.W < 1) return false;
  } else if (o.kind == PV_OP_UPSAMPLE2) {
    out.H = in.H * 2; out.W = nd == 2 ? in.W * 2 : 1;
  } else if (o.kind == PV_OP_UPSAMPLE2_BILINEAR) {
    if (nd != 2) return false;
    out.H = in.H * 2; out.W = in.W * 2;
  } else if (o.kind == PV_OP_BATCHNORM) {
    if (o.cin != in.C || o.b_off < 0 || o.aux0_off < 0 || o.aux1_off < 0) return false;
  } else {
    return false;
  }
  return true;
}

inline int kk_of(const pv_op& o, int nd) { return o.ksize == 3 ? (nd == 2 ? 9 : 3) : 1; }

// scratch the stack's GEMMs and im2col need for B samples: running maxima
struct Needs { int64_t maxact = 0, maxcol = 0, scratch = 0, code_bytes = 0, code2_bytes = 0, wg_sum = 0; int bn_maxC = 0; };   // wg_sum: all the
                                                   // kernel-3 weight gradients' partials at once (their finishes are deferred)
#define PVCS_BN_SLOTS (2 * PV_MAX_OPS)        // per-op statistics slots: stack 0 (encoder) and stack 1 (decoder)
inline int64_t bn_floats(const Needs& n) { return (int64_t)PVCS_BN_SLOTS * 4 * n.bn_maxC; }
inline void upd(int64_t& m, int64_t v) { if (v > m) m = v; }

// the first block of a 2-D encoder — conv k3 from ONE channel, activation, 2x max-pool — runs as one forward and one
// backward kernel that never materialise the full-resolution activation (pv_conv_c1.hip)
inline bool c1pool_fusable(const pv_op* ops, int n, int nd, const Shape& s0) {
  return n >= 2 && ops[0].kind == PV_OP_CONV && ops[0].ksize == 3 && ops[1].kind == PV_OP_MAXPOOL2 && s0.C == 1 &&
         pv_c1_convpool_supported(ops[0].cin, ops[0].cout, nd, ops[0].act, s0.H, s0.W);
}

// a 2-D kernel-3 convolution the split-operand kernel takes, followed by the 2x max-pool, on even image sides: the pooling
// runs in the convolution's epilogue (pooled values + one winner byte each; the full-resolution activation is never
// written) and its backward from those bytes (pv_maxpool2_bwd_code), the convolution's activation derivative folded in
inline bool convpool_fusable(const pv_op* ops, int n, int nd, int i, const Shape& si) {
  return nd == 2 && i >= 1 && i + 1 < n && ops[i].kind == PV_OP_CONV && ops[i].ksize == 3 && ops[i + 1].kind == PV_OP_MAXPOOL2 &&
         (si.H & 1) == 0 && (si.W & 1) == 0 && (ops[i].cout & 3) == 0 && ops[i].act != PV_ACT_GELU &&
         pv_conv3_sp_supported(ops[i].cin, ops[i].cout, nd, ops[i].act) && !pv_exp_str("PV_NO_CONVPOOL");
}
inline int64_t code2_off(const pv_op* ops, int n, int nd, int B, const Shape* sh, int upto) {   // winner bytes before op `upto`
  int64_t off = 0;
  for (int i = 1; i < upto; ++i)
    if (convpool_fusable(ops, n, nd, i, sh[i])) off += pv_align_up((int64_t)B * (sh[i].H / 2) * (sh[i].W / 2) * ops[i].cout, 256);
  return off;
}

// test / measurement hook behind pv_debug_ivae_conv_trace / pv_debug_ved_conv_trace (not in include/): where a step leaves the
// DECISIONS of the encoder stack's forward — the activations its backward reads (channels-last (B, H, W, C) fp32: the sign of a
// leaky-ReLU / ReLU output is the sign of its input) and the max-pool winner bytes ((B, H/2, W/2, C); k = 2 dy + dx of the 2x2
// window's first maximum in scan order).  6 int64 per op i = 0 .. n-1, describing the op's OUTPUT a[i + 1]:
//   [0] byte offset of a[i + 1] from `base` (-1: never written — the convolution's max-pool runs in its epilogue),
//   [1..3] H, W, C of a[i + 1],  [4] byte offset of the recorded winners when op i is a max-pool (-1: not recorded),  [5] op kind
inline void conv_trace(const pv_op* ops, int n, int nd, int B, const Shape* sh, float* const* a, const unsigned char* code,
                       const unsigned char* code2, const char* base, int64_t* out) {
  const bool c1 = code && c1pool_fusable(ops, n, nd, sh[0]);
  for (int i = 0; i < n; ++i) {
    int64_t* o = out + 6 * i;
    const bool fused_next = (i == 0 && c1) || (code2 && convpool_fusable(ops, n, nd, i, sh[i]));
    o[0] = (fused_next || !a[i + 1]) ? -1 : (int64_t)(reinterpret_cast<const char*>(a[i + 1]) - base);
    o[1] = sh[i + 1].H; o[2] = sh[i + 1].W; o[3] = sh[i + 1].C;
    o[4] = -1; o[5] = ops[i].kind;
    if (ops[i].kind == PV_OP_MAXPOOL2 && i >= 1) {
      if (i == 1 && c1) o[4] = (int64_t)(reinterpret_cast<const char*>(code) - base);
      else if (code2 && convpool_fusable(ops, n, nd, i - 1, sh[i - 1]))
        o[4] = (int64_t)(reinterpret_cast<const char*>(code2 + code2_off(ops, n, nd, B, sh, i - 1)) - base);
    }
  }
}

// shapes s[0..n] from s[0]; accumulates workspace needs; false on an inconsistent sequence
inline bool stack_shapes(const pv_op* ops, int n, int nd, int64_t B, Shape* s, Needs& nd_) {
  upd(nd_.maxact, s[0].elems(B));
  if (c1pool_fusable(ops, n, nd, s[0])) {
    upd(nd_.code_bytes, B * (s[0].H / 2) * (s[0].W / 2) * ops[0].cout);
    upd(nd_.scratch, pv_c1_convpool_ws((int)B, s[0].H, s[0].W, ops[0].cout));
    nd_.wg_sum += pv_align_up(pv_c1_convpool_ws((int)B, s[0].H, s[0].W, ops[0].cout), 256);
  }
  for (int i = 0; i < n; ++i) {
    if (!op_shape(ops[i], nd, s[i], s[i + 1])) return false;
    upd(nd_.maxact, s[i + 1].elems(B));
    if (convpool_fusable(ops, n, nd, i, s[i])) upd(nd_.code2_bytes, code2_off(ops, n, nd, (int)B, s, i + 1));
    if (ops[i].kind == PV_OP_BATCHNORM) {
      if (ops[i].cin > nd_.bn_maxC) nd_.bn_maxC = ops[i].cin;
      upd(nd_.scratch, pv_bn_ws(B * s[i].H * s[i].W, ops[i].cin));
    }
    if (ops[i].kind == PV_OP_CONV) {
      const int64_t rows = B * s[i].H * s[i].W, K = (int64_t)ops[i].cin * kk_of(ops[i], nd), N = ops[i].cout;
      if (ops[i].ksize == 3) {
        upd(nd_.maxcol, N * K);                                      // flipped weights of the dgrad-as-convolution
        upd(nd_.maxcol, pv_conv3_direct_wt_floats(ops[i].cin, ops[i].cout, nd));   // tiled weights of the direct kernel
        if (nd == 2) upd(nd_.maxcol, (pv_conv3_sp_wt_bytes(ops[i].cin, ops[i].cout) + 3) / 4);   // ... of the split-operand kernel
      }
      upd(nd_.scratch, gemm_ws_need(rows, N, K));                    // forward
      upd(nd_.scratch, gemm_ws_need(N, K, rows));                    // wgrad
      if (ops[i].ksize == 3) upd(nd_.scratch, pv_conv3_wgrad_direct_ws((int)B, s[i].H, s[i].W, ops[i].cin, ops[i].cout, nd));
      if (ops[i].ksize == 3 && nd == 2) upd(nd_.scratch, pv_conv3_sp_wgrad_ws((int)B, s[i].H, s[i].W, ops[i].cin, ops[i].cout));
      if (ops[i].ksize == 3) upd(nd_.scratch, pv_conv3_wgrad_c1_ws((int)B, s[i].H, s[i].W, ops[i].cin, ops[i].cout, nd));
      if (ops[i].ksize == 3) {                                       // whichever weight-gradient kernel takes this layer
        int64_t w = pv_conv3_wgrad_direct_ws((int)B, s[i].H, s[i].W, ops[i].cin, ops[i].cout, nd);
        upd(w, pv_conv3_wgrad_c1_ws((int)B, s[i].H, s[i].W, ops[i].cin, ops[i].cout, nd));
        if (nd == 2) upd(w, pv_conv3_sp_wgrad_ws((int)B, s[i].H, s[i].W, ops[i].cin, ops[i].cout));
        if (nd == 1) upd(w, pv_conv3_1d_wgrad_lean_ws((int)B, s[i].H, ops[i].cin, ops[i].cout));
        upd(nd_.scratch, w);
        nd_.wg_sum += pv_align_up(w, 256);
      } else {
        upd(nd_.scratch, pv_k1_wgrad_ws(rows, ops[i].cin, ops[i].cout));
        nd_.wg_sum += pv_align_up(pv_k1_wgrad_ws(rows, ops[i].cin, ops[i].cout), 256);
      }
      upd(nd_.scratch, gemm_ws_need(rows, K, N));                    // dgrad, kernel 1
      upd(nd_.scratch, gemm_ws_need(rows, ops[i].cin, N * kk_of(ops[i], nd)));   // dgrad, kernel 3
    }
  }
  return true;
}

// The tiled / split weights of every kernel-3 convolution of a stack, both orientations (forward, input gradient), are
// written ONCE per step by one launch (pv_conv_wprep_table) instead of before each convolution: the weights change once
// per Adam step.  off[2 * slot + flip] = byte offset in the region, -1: that convolution does not use tiled weights.
struct WtPlan {
  int64_t off[2 * PV_MAX_OPS * 2];
  int64_t bytes = 0;
  WtPlan() { for (auto& o : off) o = -1; }
};

struct Scratch {
  float* col /* flipped-weight scratch (maxcol floats) */; void* ws; int64_t ws_bytes;
  float* bn = nullptr; int bn_maxC = 0; int bn_eval = 0;     // batch-norm statistics slots (bn_floats), mode
  int conv_bf16 = 0;                                         // conv mode of the plan (pv_*_plan.conv_bf16): 0 fp32-class (two fp16 pieces,
                                                             // exact scaling), 1 mixed (two rounded bf16 pieces, 3 products), 2 fp32-class
                                                             // for weights outside fp16's range (three bf16 pieces), 3 throughput (ONE
                                                             // fp16 piece, one product), 4 fp32-class with a cheaper BACKWARD (round 5:
                                                             // forward as 0 — three products: its outputs pick max-pool winners and
                                                             // leaky-ReLU signs, a coarser forward flips them by the thousand —; input
                                                             // gradient with dL/dy as ONE fp16 piece, two products; weight gradient one
                                                             // piece per operand, one product); see conv_mixed / sp_mode below
  const char* wt = nullptr; const WtPlan* wtp = nullptr;     // the step's tiled weights (null: tile per call into col)
  unsigned char* code = nullptr;                             // winners of the fused first block's max-pool (stack 0 only)
  PvFinishList* fin = nullptr;                               // weight-gradient finishes deferred to pv_wgrad_finish_all
  unsigned char* code2 = nullptr;                            // winners of the convolution + max-pool pairs fused further up (stack 0)
  void* ev_start = nullptr; void* ev_stop = nullptr;         // measurement: events around op `ev_op` of stack 0's forward
  int ev_op = -1;
  // two-stream steps (pv_side.h).  side != null: the split-operand weight gradients of stack_bwd are enqueued there (each
  // after the launches on the main stream that produce its g; the CALLER joins before the finish) — needs per-op gradient
  // buffers (stack_bwd's gown), since nothing on the main stream waits for them.  wt_join != null && *wt_join: the step's
  // weight tilings are still running on `side`; stack_fwd joins before the first op that reads them.
  hipStream_t side = nullptr;
  bool* wt_join = nullptr;
  bool fork_after = false;       // stack_bwd: the stack's LAST input-gradient launch carries a fork event (the caller calls pv_fork_to)
  bool* side_joined = nullptr;   // stack_bwd sets it when it has joined the side stream itself (after its flush of the reductions)
};
inline int join_tilings(const Scratch& sc, hipStream_t s) {
  if (sc.wt_join && *sc.wt_join) { *sc.wt_join = false; return pv_stream_after(s, sc.side); }
  return 0;
}
// recorded (batched) weight gradients are flushed onto the side stream every k1_chunk() problems, next to the rest of the
// input-gradient chain (PV_K1_CHUNK=n; 0: one launch after the chain)
inline int k1_chunk() {
  static const int v = pv_exp_int("PV_K1_CHUNK", 5) < 0 ? 0 : pv_exp_int("PV_K1_CHUNK", 5);
  return v;
}
inline bool k1_flush_due(const Scratch& sc) {
  return sc.side && sc.fin && sc.fin->k1b && k1_chunk() > 0 && sc.fin->k1b->n >= k1_chunk();
}
// a layer's weight + input gradient in one launch (pv_conv3_sp_pair) although a side stream is there (PV_SIDE_PAIR=1, A/B)
inline bool side_keeps_pairs() {
  static const int v = pv_exp_int("PV_SIDE_PAIR", 0) != 0 ? 1 : 0;
  return v == 1;
}
// the stack's heaviest split-operand kernel-3 convolution (most multiply-adds) and its algorithmic FLOPs; -1: none
inline int heaviest_conv(const pv_op* ops, int n, int nd, int B, const Shape* sh, double* flops) {
  int best = -1;
  double bf = 0.0;
  for (int i = 0; i < n; ++i) {
    const pv_op& o = ops[i];
    if (o.kind != PV_OP_CONV || o.ksize != 3 || !pv_conv3_sp_supported(o.cin, o.cout, nd, o.act)) continue;
    const double fl = 2.0 * B * sh[i].H * sh[i].W * (double)o.cin * o.cout * kk_of(o, nd);
    if (fl > bf) { bf = fl; best = i; }
  }
  if (flops) *flops = bf;
  return best;
}
inline const void* wt_ready(const Scratch& sc, int slot, int flip) {
  if (!sc.wt || !sc.wtp || sc.wtp->off[2 * slot + flip] < 0) return nullptr;
  return sc.wt + sc.wtp->off[2 * slot + flip];
}

// pv_conv3_direct's precision argument: 1 two bf16 pieces (mixed), 2 two fp16 pieces with exact scaling (fp32-class), 0 the
// f32-input MFMA (PV_SP_X6=1, or fewer than 32 input channels)
// the conv mode is a PLAN fact (ABI v14; rounds 2-3 kept the wide-weights switch in a process-wide setter):
inline bool conv_mixed(int cm) { return cm == 1 || cm == 3; }   // (3: the kernels without a one-piece form run their mixed one)
// the split-operand kernels' piece count for fp32-class work: 4 = two fp16 pieces, 3 = three bf16 pieces
inline int sp_fp32_mode(int cm) { return cm == 2 ? 3 : pv_conv3_sp_fp32_mode(); }
// ... and the `mode` argument of pv_conv3_sp / _wgrad / _pair / pv_conv3_direct's callers: 2 mixed, else the fp32-class one
inline int sp_mode(int cm) { return cm == 3 ? 1 : (cm == 1 ? 2 : sp_fp32_mode(cm)); }
// ... of the INPUT gradient and of the split-operand WEIGHT gradient.  Mode 4, round 6: the input gradient keeps BOTH operands
// split (three products, as mode 0) and only the weight gradient — whose sums run over every pixel of every sample and whose
// error stays in its own tensor — takes one piece per operand.  Round 5 also ran the input gradient with dL/dy as ONE fp16 piece
// (two products): under equal forward decisions (tests: masked_conv_check; profiles/r06a_conv_bwd_forms.txt) that form's error
// ACCUMULATES down the chain of input gradients — 0.85 ... 1.65e-4 on the first layers' tensors on three draws, over the 1e-4
// bar on three of six — where three products leave 0.6 ... 3.6e-5; the one-piece weight gradient adds 5 ... 7e-5 to its own
// tensor only (worst 0.78 of the bar).  C5 +4.7 %, C4 +1.8 % against round 5's form; mode 0 (three products everywhere) another
// +6.5 % / +4.2 %.
// (the input gradient with ONE piece per operand too — one product — was measured in round 5: the weights' rounding is systematic,
//  the conv-stack gradients land AT the end-to-end bar: gpurun_out/r05r; not adopted)
// (experiments build: PV_CONV_DG / PV_CONV_WG override mode 4's two choices — 4 = both operands split, three products — for the
//  error / time table of profiles/r06*_conv_bwd_forms.txt)
inline int sp_dg_mode(int cm) { static const int e = pv_exp_int("PV_CONV_DG", 0); return (cm == 4 && e) ? e : sp_mode(cm); }
inline int sp_wg_mode(int cm) { static const int e = pv_exp_int("PV_CONV_WG", 0); return cm == 4 ? (e ? e : 1) : sp_mode(cm); }
inline int direct_mode(const Scratch& sc) { return conv_mixed(sc.conv_bf16) ? 1 : (sp_fp32_mode(sc.conv_bf16) == 4 ? 2 : 0); }

// Mode 4's one-piece backward operands carry independent rounding errors of 2^-12 that average out of the sums a gradient is:
// it is taken only where EVERY split-operand kernel-3 convolution of the stack sums at least PV_CONV_W1_MIN_PIXELS pixels
// (batch x height x width at that layer) — the rule of the decoder kernel's fp16 builds (pv_sdec_fused_bf16.hip: 16 384 rows);
// smaller problems keep mode 0 (three products everywhere).
#define PV_CONV_W1_MIN_PIXELS 16384
inline int conv_mode_for(int cm, const pv_op* ops, int n, int nd, int64_t B, int in_c, const int* in_dim) {
  if (cm != 4) return cm;
  Shape s{in_dim[0], nd == 2 ? in_dim[1] : 1, in_c}, t;
  for (int i = 0; i < n; ++i) {
    if (ops[i].kind == PV_OP_CONV && ops[i].ksize == 3 && pv_conv3_sp_supported(ops[i].cin, ops[i].cout, nd, ops[i].act) &&
        B * s.H * s.W < PV_CONV_W1_MIN_PIXELS)
      return 0;
    if (!op_shape(ops[i], nd, s, t)) return 0;
    s = t;
  }
  return 4;
}

// which tiling (pv_conv_wprep_table kind) a kernel-3 convolution uses in an orientation; -1: none (GEMM fallback, k1)
inline int wt_kind(const pv_op& o, int nd, int flip, int conv_bf16) {
  if (o.kind != PV_OP_CONV || o.ksize != 3) return -1;
  const int C = flip ? o.cout : o.cin, N = flip ? o.cin : o.cout, act = flip ? PV_ACT_NONE : o.act;
  if (pv_conv3_sp_supported(C, N, nd, act)) return conv_bf16 == 3 ? 9 : (conv_bf16 == 1 ? 2 : (sp_fp32_mode(conv_bf16) == 4 ? 5 : 3));
  if (pv_conv3_direct_supported(C, N, nd, act)) return C % 32 == 0 ? (conv_mixed(conv_bf16) ? 1 : (sp_fp32_mode(conv_bf16) == 4 ? 6 : 0)) : 0;
  return -1;
}

inline void wt_layout(const pv_op* ops, int n, int nd, int stack_id, int conv_bf16, bool need_input_grad, WtPlan& w) {
  for (int i = 0; i < n; ++i)
    for (int flip = 0; flip < 2; ++flip) {
      if (flip && i == 0 && !need_input_grad) continue;
      const int kind = wt_kind(ops[i], nd, flip, conv_bf16);
      if (kind < 0) continue;
      w.off[2 * (stack_id * PV_MAX_OPS + i) + flip] = w.bytes;
      w.bytes += pv_align_up(pv_conv_wt_bytes(kind, ops[i].cout, ops[i].cin, nd), 256);
    }
}

// a conv head's weight entry for wt_prep (kind 4): w (out, C*S) -> dst (out, S*C)
inline PvWprepEntry head_entry(const float* w, float* dst, int out, int C, int64_t S) {
  PvWprepEntry E{};
  E.w = w; E.dst = reinterpret_cast<char*>(dst); E.Co = out; E.Ci = C; E.KK = (int)S; E.flip = 0; E.kind = 4;
  return E;
}

// one launch per 16 tilings: everything wt_layout placed for this stack (flip 1 entries only when with_dgrad)
// (wt_entries: append a stack's entries to e[]; wt_prep: one stack, one launch)
inline void wt_entries(const float* params, const pv_op* ops, int n, int nd, int stack_id, int conv_bf16, const WtPlan& w, char* base,
                       bool with_dgrad, PvWprepEntry* e, int& ne, const PvWprepEntry* extra = nullptr, int n_extra = 0) {
  for (int k = 0; k < n_extra && k < 4; ++k) e[ne++] = extra[k];     // (e.g. a conv head's re-indexed Linear weight)
  for (int i = 0; i < n; ++i)
    for (int flip = 0; flip < (with_dgrad ? 2 : 1); ++flip) {
      const int64_t off = w.off[2 * (stack_id * PV_MAX_OPS + i) + flip];
      if (off < 0) continue;
      PvWprepEntry& E = e[ne++];
      E.w = params + ops[i].w_off; E.dst = base + off; E.Co = ops[i].cout; E.Ci = ops[i].cin; E.KK = kk_of(ops[i], nd);
      E.flip = flip; E.kind = wt_kind(ops[i], nd, flip, conv_bf16); E.pad_ = 0; E.start = E.total = 0;
    }
}
inline int wt_prep(const float* params, const pv_op* ops, int n, int nd, int stack_id, int conv_bf16, const WtPlan& w, char* base,
                   bool with_dgrad, hipStream_t s, const PvWprepEntry* extra = nullptr, int n_extra = 0, const PvFbPrep* fb = nullptr) {
  PvWprepEntry e[2 * PV_MAX_OPS + 4];
  int ne = 0;
  wt_entries(params, ops, n, nd, stack_id, conv_bf16, w, base, with_dgrad, e, ne, extra, n_extra);
  return (ne || fb) ? pv_conv_wprep_table(e, ne, s, fb) : 0;
}
inline float* bn_slot(const Scratch& sc, int slot) { return sc.bn + (int64_t)slot * 4 * sc.bn_maxC; }

// one op forward: in (shape si) -> out
// kernel-1 convolutions on the lean register-fed kernels of pv_conv_k1.hip (PV_NO_K1=1: the LDS-tiled GEMMs, for A/B timing)
inline bool k1_lean() {
  static const int v = pv_exp_int("PV_NO_K1", 0) != 0 ? 0 : 1;
  return v == 1;
}

// kernel-3 1-D weight gradients on the kernel-1 family's register-fed kernel (PV_NO_K3LEAN=1: the LDS-tiled direct kernels)
inline bool k3_lean_1d(int nd) {
  static const int v = pv_exp_int("PV_NO_K3LEAN", 0) != 0 ? 0 : 1;
  return v == 1 && nd == 1 && k1_lean();
}

// mixed-precision leg: the register-fed (exact fp32) form too when the weight gradients are batched into one launch
// (PV_K3LEAN_MIXED=0: keep the bf16 tile kernel there)
inline bool k3_lean_mixed(const Scratch& sc) {
  static const int v = pv_exp_int("PV_K3LEAN_MIXED", 1) == 0 ? 0 : 1;
  return v == 1 && sc.fin && sc.fin->k1b;
}

// a kernel-1 convolution without activation followed by the 1-D nearest upsample (UpsampleBlock with the convolution first):
// one launch each way — the forward stores every row twice, the backward kernels read the sum of the two rows
// (PV_NO_K1UP=1: separate upsample launches)
inline bool k1up_fusable(const pv_op* ops, int n, int nd, int i) {
  static const int v = pv_exp_int("PV_NO_K1UP", 0) != 0 ? 0 : 1;
  return v == 1 && k1_lean() && nd == 1 && i >= 0 && i + 1 < n && ops[i].kind == PV_OP_CONV && ops[i].ksize == 1 &&
         ops[i].act == PV_ACT_NONE && ops[i + 1].kind == PV_OP_UPSAMPLE2;
}

inline int op_fwd(const float* params, const pv_op& o, int nd, int B, const float* in, const Shape& si, float* out,
                  const Scratch& sc, int slot, hipStream_t s) {
  if (o.kind == PV_OP_BATCHNORM) {
    float* P = const_cast<float*>(params);             // the running statistics live in the parameter buffer
    return pv_bn_fwd(in, out, (int64_t)B * si.H * si.W, si.C, params + o.w_off, params + o.b_off, P + o.aux0_off,
                     P + o.aux1_off, sc.bn_eval, 0.1f, 1e-5f, bn_slot(sc, slot), sc.ws, sc.ws_bytes, s);
  }
  if (o.kind == PV_OP_CONV) {
    const int64_t rows = (int64_t)B * si.H * si.W, K = (int64_t)o.cin * kk_of(o, nd);
    const float* bias = o.b_off >= 0 ? params + o.b_off : nullptr;
    if (o.ksize == 3) {
      if (pv_conv3_sp_supported(o.cin, o.cout, nd, o.act))      // 2-D, Cin % 32 == 0: exactly split operands on the bf16 cores
        return pv_conv3_sp(in, B, si.H, si.W, params + o.w_off, o.cout, o.cin, 0, bias, out, o.act, sc.col, s, nullptr, 0,
                           sp_mode(sc.conv_bf16), wt_ready(sc, slot, 0));
      if (pv_conv3_direct_supported(o.cin, o.cout, nd, o.act))
        return pv_conv3_direct(in, B, si.H, si.W, nd, params + o.w_off, o.cout, o.cin, 0, bias, out, o.act, sc.col, s, nullptr,
                               0, direct_mode(sc), wt_ready(sc, slot, 0));
      return conv3_fwd(in, B, si.H, si.W, si.C, nd, params + o.w_off, bias, out, o.cout, o.act, sc.ws, sc.ws_bytes, s);
    }
    if (k1_lean()) return pv_k1_fwd(in, rows, o.cin, params + o.w_off, bias, out, o.cout, o.act, s);
    return linear_fwd(in, K, params + o.w_off, bias, out, nullptr, o.cout, rows, K, o.cout, o.act, sc.ws, sc.ws_bytes, s);
  }
  if (o.kind == PV_OP_MAXPOOL2) return pv_maxpool2_fwd(in, out, B, si.H, si.W, si.C, nd, s);
  if (o.kind == PV_OP_UPSAMPLE2_BILINEAR) return pv_upsample2_bil_fwd(in, out, B, si.H, si.W, si.C, s);
  return pv_upsample2_fwd(in, out, B, si.H, si.W, si.C, nd, s);
}

// one op backward: g = dL/d(out) (post-activation for CONV; modified in place), writes parameter gradients and, when
// gin != null, dL/d(in)
// g_is_pre: g already is dL/d(pre-activation) (the consumer's backward applied this layer's activation derivative);
// fuse_act != NONE: the PRODUCER of `in` is a convolution with that activation — apply act'(in) to gin here (returns
// *fused = true when done) instead of a separate elementwise pass in the producer's backward
// does a kernel-3 convolution's backward run its input gradient on the split-operand kernel as a launch of its own (not the
// weight + input gradient pair launch)?  That launch can carry the un-pooling epilogue.
// (ONE place decides the launch form of a kernel-3 convolution's backward — stack_bwd plans the un-pooling epilogue with it, op_bwd
//  launches by it: ADVICE r4, the two used to carry hand-copied predicates)
struct ConvBwdForm { bool on_side, pair; };
inline ConvBwdForm conv_bwd_form(const pv_op& o, int nd, int B, const Shape& si, const Scratch& sc, bool want_gin) {
  const bool wg = pv_conv3_sp_wgrad_supported(si.C, o.cout, nd);
  ConvBwdForm f;
  // the split-operand weight gradient goes to the side stream when its partials land in the finish list (never in sc.ws,
  // which the main stream keeps using)
  f.on_side = sc.side && wg && pv_wgrad_defers(sc.fin, pv_conv3_sp_wgrad_ws(B, si.H, si.W, si.C, o.cout));
  // weight gradient + input gradient in one launch (pv_conv_sp.hip)
  f.pair = wg && want_gin && sc.fin && !conv_mixed(sc.conv_bf16) && sc.conv_bf16 != 4 && sp_fp32_mode(sc.conv_bf16) == 4 &&
           pv_conv3_sp_supported(o.cout, o.cin, nd, PV_ACT_NONE) && (!f.on_side || side_keeps_pairs());
  return f;
}
inline bool conv_bwd_sp_alone(const pv_op& o, int nd, int B, const Shape& si, const Scratch& sc) {
  if (o.kind != PV_OP_CONV || o.ksize != 3 || !pv_conv3_sp_supported(o.cout, o.cin, nd, PV_ACT_NONE)) return false;
  return !conv_bwd_form(o, nd, B, si, sc, true).pair;
}
inline int op_bwd(const float* params, float* grads, const pv_op& o, int nd, int B, const float* in, const Shape& si,
                  const float* out, float* g, float* gin, const Scratch& sc, int slot, hipStream_t s, bool g_is_pre = false,
                  int fuse_act = PV_ACT_NONE, bool* fused = nullptr, int g_up = 0, const unsigned char* up_code = nullptr,
                  bool* unpooled = nullptr) {
  // up_code != null (kernel-3 convolutions on the split-operand kernels, not in a pair launch): gin is the UN-POOLED input gradient
  // (B, 2H, 2W, cin) — the max-pool below this op rides in the launch's epilogue; *unpooled tells the caller whether it did
  if (unpooled) *unpooled = false;
  if (fused) *fused = false;
  if (fuse_act == PV_ACT_GELU) fuse_act = PV_ACT_NONE;
  if (o.kind == PV_OP_BATCHNORM)
    return pv_bn_bwd(in, g, gin, (int64_t)B * si.H * si.W, si.C, params + o.w_off, bn_slot(sc, slot), sc.bn_eval,
                     grads + o.w_off, grads + o.b_off, sc.ws, sc.ws_bytes, s);
  if (o.kind == PV_OP_CONV) {
    const int64_t rows = (int64_t)B * si.H * si.W, K = (int64_t)o.cin * kk_of(o, nd);
    if (!g_is_pre) PV_TRY(pv_act_bwd(g, out, rows * o.cout, o.act, s));         // g = dL/d(pre-activation)
    float* db = o.b_off >= 0 ? grads + o.b_off : nullptr;
    if (o.ksize == 3) {
      const ConvBwdForm form = conv_bwd_form(o, nd, B, si, sc, gin != nullptr);
      const bool on_side = form.on_side, pair = form.pair;
      if (pair) {                                       // weight gradient + input gradient: one launch (pv_conv_sp.hip)
        pv_conv3_sp_pair_begin();
        int rc = pv_conv3_sp_wgrad(g, in, B, si.H, si.W, si.C, grads + o.w_off, db, o.cout, sc.ws, sc.ws_bytes, s, 4, sc.fin);
        if (rc == 0) {
          if (fused && fuse_act != PV_ACT_NONE) *fused = true;
          rc = pv_conv3_sp(g, B, si.H, si.W, params + o.w_off, o.cout, o.cin, 1, nullptr, gin, PV_ACT_NONE, sc.col, s, in, fuse_act, 4,
                           wt_ready(sc, slot, 1));
        }
        const int rc2 = pv_conv3_sp_pair_flush(s);
        return rc ? rc : rc2;
      }
      if (pv_conv3_sp_wgrad_supported(si.C, o.cout, nd)) {
        if (on_side) PV_TRY(pv_fork_to(sc.side, s));                // g is complete: its producer's stop event, or a record on s
        PV_TRY(pv_conv3_sp_wgrad(g, in, B, si.H, si.W, si.C, grads + o.w_off, db, o.cout, sc.ws, sc.ws_bytes, on_side ? sc.side : s,
                                 sp_wg_mode(sc.conv_bf16), sc.fin));
      }
      else if (k3_lean_1d(nd) && (!conv_mixed(sc.conv_bf16) || k3_lean_mixed(sc)))   // (launch by launch the mixed leg's bf16 kernel is faster)
        PV_TRY(pv_conv3_1d_wgrad_lean(g, in, B, si.H, si.C, o.cout, grads + o.w_off, db, sc.ws, sc.ws_bytes, s, sc.fin));
      else if (conv_mixed(sc.conv_bf16) && si.C % 32 == 0 && pv_conv3_wgrad_direct_supported(si.C, o.cout, nd))
        PV_TRY(pv_conv3_wgrad_direct_bf16(g, in, B, si.H, si.W, si.C, nd, grads + o.w_off, db, o.cout, sc.ws, sc.ws_bytes, s, sc.fin));
      else if (pv_conv3_wgrad_direct_supported(si.C, o.cout, nd))
        PV_TRY(pv_conv3_wgrad_direct(g, in, B, si.H, si.W, si.C, nd, grads + o.w_off, db, o.cout, sc.ws, sc.ws_bytes, s, sc.fin));
      else if (pv_conv3_wgrad_c1_supported(si.C, o.cout, nd))
        PV_TRY(pv_conv3_wgrad_c1(g, in, B, si.H, si.W, nd, grads + o.w_off, db, o.cout, sc.ws, sc.ws_bytes, s, sc.fin));
      else
        PV_TRY(conv3_wgrad(g, in, B, si.H, si.W, si.C, nd, grads + o.w_off, db, o.cout, sc.ws, sc.ws_bytes, s));
      if (!gin) return 0;
      // dX = conv3(dpre; taps flipped, channel roles swapped) — same spatial size, C = cout -> cin
      if (pv_conv3_sp_supported(o.cout, o.cin, nd, PV_ACT_NONE)) {
        if (fused && fuse_act != PV_ACT_NONE) *fused = true;
        if (sc.side) pv_fork_arm();                        // the layer below may fork its weight gradient off this launch
        const bool up = up_code && unpooled && (o.cin & 3) == 0;
        if (up) *unpooled = true;
        return pv_conv3_sp(g, B, si.H, si.W, params + o.w_off, o.cout, o.cin, 1, nullptr, gin, PV_ACT_NONE, sc.col, s, in, fuse_act,
                           sp_dg_mode(sc.conv_bf16), wt_ready(sc, slot, 1), nullptr, nullptr, up ? up_code : nullptr);
      }
      if (pv_conv3_direct_supported(o.cout, o.cin, nd, PV_ACT_NONE)) {
        if (fused && fuse_act != PV_ACT_NONE) *fused = true;
        if (k1_flush_due(sc)) pv_fork_arm();               // stack_bwd forks the recorded weight gradients off this launch
        return pv_conv3_direct(g, B, si.H, si.W, nd, params + o.w_off, o.cout, o.cin, 1, nullptr, gin, PV_ACT_NONE, sc.col, s,
                               in, fuse_act, direct_mode(sc), wt_ready(sc, slot, 1));
      }
      PV_TRY(pv_conv_wflip(params + o.w_off, sc.col, o.cout, o.cin, kk_of(o, nd), s));
      return conv3_fwd(g, B, si.H, si.W, o.cout, nd, sc.col, nullptr, gin, o.cin, PV_ACT_NONE, sc.ws, sc.ws_bytes, s);
    }
    if (g_up && !k1_lean()) return PV_EINVAL;
    if (k1_lean()) PV_TRY(pv_k1_wgrad(g, in, rows, o.cin, o.cout, grads + o.w_off, db, sc.ws, sc.ws_bytes, s, sc.fin, g_up));
    else PV_TRY(linear_wgrad(g, o.cout, in, K, grads + o.w_off, db, rows, K, o.cout, sc.ws, sc.ws_bytes, s));
    if (!gin) return 0;
    // (the producing convolution's activation derivative rides in this GEMM's epilogue: act'(in), in = that layer's output)
    if (fused && fuse_act != PV_ACT_NONE) *fused = true;
    if (k1_lean() && k1_flush_due(sc)) pv_fork_arm();
    if (k1_lean()) return pv_k1_dgrad(g, rows, o.cout, params + o.w_off, gin, o.cin, fuse_act != PV_ACT_NONE ? in : nullptr, fuse_act, s, g_up);
    return linear_dgrad(g, o.cout, params + o.w_off, gin, K, fuse_act != PV_ACT_NONE ? in : nullptr, nullptr, K, fuse_act, rows, K,
                        o.cout, sc.ws, sc.ws_bytes, s);
  }
  if (!gin) return 0;
  if (o.kind == PV_OP_MAXPOOL2) {
    if (fused && fuse_act != PV_ACT_NONE) *fused = true;
    return pv_maxpool2_bwd(in, g, gin, B, si.H, si.W, si.C, nd, s, fuse_act);
  }
  if (o.kind == PV_OP_UPSAMPLE2_BILINEAR) return pv_upsample2_bil_bwd(g, gin, B, si.H, si.W, si.C, s);
  if (fused && fuse_act != PV_ACT_NONE) *fused = true;         // the producing convolution's activation backward rides along
  return pv_upsample2_bwd(g, gin, B, si.H, si.W, si.C, nd, s, in, fuse_act);
}

// whole stack forward: a[0] given, a[1..n] written
inline int stack_fwd(const float* params, const pv_op* ops, int n, int nd, int B, float* const* a, const Shape* sh,
                     const Scratch& sc, hipStream_t s, int stack_id = 0) {
  int i0 = 0;
  if (stack_id == 0 && sc.code && c1pool_fusable(ops, n, nd, sh[0])) {   // a[1] is never written
    PV_TRY(pv_c1_convpool_fwd(a[0], B, sh[0].H, sh[0].W, params + ops[0].w_off, ops[0].b_off >= 0 ? params + ops[0].b_off : nullptr,
                              ops[0].cout, ops[0].act, a[2], sc.code, s));
    i0 = 2;
  }
  PV_TRY(join_tilings(sc, s));                         // (the fused first block reads the raw weights)
  for (int i = i0; i < n; ++i) {
    const bool timed = stack_id == 0 && i == sc.ev_op && sc.ev_start && sc.ev_stop;
    struct EvGuard {                     // start event now, stop event when the op's launches are enqueued
      bool on; void* stop; hipStream_t s;
      ~EvGuard() { if (on) (void)hipEventRecord((hipEvent_t)stop, s); }
    } evg{timed, sc.ev_stop, s};
    if (timed) (void)hipEventRecord((hipEvent_t)sc.ev_start, s);
    if (stack_id == 0 && sc.code2 && convpool_fusable(ops, n, nd, i, sh[i])) {      // a[i + 1] is never written
      const pv_op& o = ops[i];
      PV_TRY(pv_conv3_sp(a[i], B, sh[i].H, sh[i].W, params + o.w_off, o.cout, o.cin, 0, o.b_off >= 0 ? params + o.b_off : nullptr,
                         a[i + 1], o.act, sc.col, s, nullptr, 0, sp_mode(sc.conv_bf16),
                         wt_ready(sc, stack_id * PV_MAX_OPS + i, 0), a[i + 2], sc.code2 + code2_off(ops, n, nd, B, sh, i)));
      ++i;
      continue;
    }
    if (k1up_fusable(ops, n, nd, i)) {                                               // a[i + 1] is never written
      const pv_op& o = ops[i];
      PV_TRY(pv_k1_fwd(a[i], (int64_t)B * sh[i].H * sh[i].W, o.cin, params + o.w_off, o.b_off >= 0 ? params + o.b_off : nullptr,
                       a[i + 2], o.cout, PV_ACT_NONE, s, 1));
      ++i;
      continue;
    }
    PV_TRY(op_fwd(params, ops[i], nd, B, a[i], sh[i], a[i + 1], sc, stack_id * PV_MAX_OPS + i, s));
  }
  return 0;
}

// whole stack backward: g = dL/d(a[n]) (clobbered); gbuf[2] ping-pong buffers; returns in *gout the buffer holding
// dL/d(a[0]) (null when need_input_grad is false: the first op then skips its dgrad)
inline int stack_bwd(const float* params, float* grads, const pv_op* ops, int n, int nd, int B, float* const* a,
                     const Shape* sh, float* g, float* const* gbuf, int& pp, bool need_input_grad, float** gout,
                     const Scratch& sc, hipStream_t s, int stack_id = 0, bool g_is_pre0 = false, float* const* gown = nullptr) {
  // gown != null: op i writes dL/d(a[i]) into its OWN buffer gown[i] instead of the ping-pong pair — every layer's gradient
  // then survives the whole backward, which is what recorded (batched) weight gradients need (PvFinishList::k1b)
  bool g_is_pre = g_is_pre0;                           // g already carries the last op's activation derivative
  Scratch scl = sc;
  if (!gown) scl.side = nullptr;                       // (ping-pong buffers are rewritten while a side-stream reader could be behind)
  const bool c1pool = stack_id == 0 && sc.code && !need_input_grad && c1pool_fusable(ops, n, nd, sh[0]);
  // the max-pool of a fused conv + pool pair un-pooled in the epilogue of the input-gradient launch ABOVE it (pv_conv_sp.hip up_code;
  // PV_NO_UNPOOL_FUSE=1: pv_maxpool2_bwd_code's own launch).  Own gradient buffers only (gown: the side-stream form).
  static const int unpool_fuse = pv_exp_int("PV_NO_UNPOOL_FUSE", 0) ? 0 : 1;
  int pool_done_at = -1;                               // index of a pool op whose backward the launch above has written
  for (int i = n - 1; i >= 0; --i) {
    if (i == pool_done_at) {                           // g = dL/d(pre-activation of the convolution below), already un-pooled
      g = gown[i];
      g_is_pre = true;
      continue;
    }
    // an armed fork event is the stop event of the launch that produced g: only a kernel-3 convolution that takes g as it
    // is (no activation pass in between) may hand it to its side-stream weight gradient
    if (!(scl.side && ops[i].kind == PV_OP_CONV && ops[i].ksize == 3 && g_is_pre)) pv_fork_disarm();
    if (c1pool && i == 1) {                            // g = dL/d(a[2]): the fused backward of ops 1 and 0
      // two streams: every weight gradient recorded so far is reduced on the side stream, next to this launch
      // (the main stream then waits for that launch's stop event: the join without a marker packet on the side stream)
      const bool flushed = scl.side && sc.fin && sc.fin->n > 0;
      if (flushed) { pv_fork_arm(); PV_TRY(pv_wgrad_finish_flush(sc.fin, scl.side)); }
      PV_TRY(pv_c1_convpool_bwd(g, a[2], sc.code, a[0], B, sh[0].H, sh[0].W, ops[0].cout, ops[0].act, grads + ops[0].w_off,
                                ops[0].b_off >= 0 ? grads + ops[0].b_off : nullptr, sc.ws, sc.ws_bytes, s, sc.fin));
      if (flushed) {
        PV_TRY(pv_fork_to(s, scl.side));
        if (sc.side_joined) *sc.side_joined = true;
      }
      g = nullptr;
      break;
    }
    if (stack_id == 0 && sc.code2 && i >= 2 && convpool_fusable(ops, n, nd, i - 1, sh[i - 1])) {
      // the max-pool of a fused pair: g = dL/d(a[i + 1]) -> dL/d(pre-activation of the convolution below) from the winner bytes
      float* gin2 = gown ? gown[i] : gbuf[pp];
      if (scl.side) pv_fork_arm();
      PV_TRY(pv_maxpool2_bwd_code(g, a[i + 1], sc.code2 + code2_off(ops, n, nd, B, sh, i - 1), gin2, B, sh[i + 1].H, sh[i + 1].W,
                                  sh[i + 1].C, ops[i - 1].act, s));
      g_is_pre = true;
      g = gin2; pp ^= 1;
      continue;
    }
    if (k1up_fusable(ops, n, nd, i - 1)) continue;     // the upsample of a fused pair: its backward rides in the convolution's
    const int g_up = k1up_fusable(ops, n, nd, i) ? 1 : 0;
    float* gin = (i > 0 || need_input_grad) ? (gown ? gown[i] : gbuf[pp]) : nullptr;
    // the layer below is a convolution with an activation: let this op's backward apply act'(a[i]) to gin
    int fuse_act = (i > 0 && gin && ops[i - 1].kind == PV_OP_CONV) ? ops[i - 1].act : PV_ACT_NONE;
    // ... or the max-pool of a fused conv + pool pair: act'(the pooled activation a[i]) and the un-pooling ride along, the
    // result is dL/d(pre-activation of the convolution below the pool) in the POOL's gradient buffer
    const unsigned char* up_code = nullptr;
    if (unpool_fuse && gown && scl.side && stack_id == 0 && sc.code2 && i >= 3 && ops[i].kind == PV_OP_CONV && ops[i].ksize == 3 &&
        ops[i - 2].act != PV_ACT_GELU && (ops[i].cin & 3) == 0 && conv_bwd_sp_alone(ops[i], nd, B, sh[i], scl) &&
        convpool_fusable(ops, n, nd, i - 2, sh[i - 2])) {
      up_code = sc.code2 + code2_off(ops, n, nd, B, sh, i - 2);
      fuse_act = ops[i - 2].act;
      gin = gown[i - 1];
    }
    bool fused = false, unpooled = false;
    if (i == 0 && sc.fork_after && gin) pv_fork_arm();
    PV_TRY(op_bwd(params, grads, ops[i], nd, B, a[i], sh[i], a[i + 1], g, gin, scl, stack_id * PV_MAX_OPS + i, s, g_is_pre,
                  fuse_act, &fused, g_up, up_code, &unpooled));
    if (up_code && !unpooled) return PV_EINVAL;        // (the launch form was chosen above: it must have taken the epilogue)
    if (unpooled) pool_done_at = i - 1;
    g_is_pre = fused;
    g = gin; pp ^= 1;
    if (i > 0 && k1_flush_due(scl)) {                  // (the last chunk is the caller's: it knows what else follows the chain)
      PV_TRY(pv_fork_to(scl.side, s));
      PV_TRY(pv_k1_wgrad_flush(sc.fin->k1b, scl.side));
      for (int k = 0; k < sc.fin->n; ++k) sc.fin->st[k] = scl.side;   // (whatever s wrote before the fork is complete for the side stream too)
    }
  }
  if (!sc.fork_after) pv_fork_disarm();
  if (gout) *gout = g;
  return 0;
}

// the weight gradients of a stack whose input gradients are already there (pv_dec1d_bwd): op i reads g = the gradient buffer of
// the op above it (gown[i + 1]; gown[i + 2] un-summed when the nearest upsample after a kernel-1 convolution rides along;
// g_out for the last op), every one of them dL/d(pre-activation) already
inline int stack_wgrads(const float* params, float* grads, const pv_op* ops, int n, int nd, int B, float* const* a, const Shape* sh,
                        float* g_out, float* const* gown, const Scratch& sc, hipStream_t s, int stack_id) {
  for (int i = n - 1; i >= 0; --i) {
    if (ops[i].kind != PV_OP_CONV) continue;
    const int g_up = k1up_fusable(ops, n, nd, i) ? 1 : 0;
    const int j = i + 1 + g_up;
    float* g = j >= n ? g_out : gown[j];
    PV_TRY(op_bwd(params, grads, ops[i], nd, B, a[i], sh[i], a[i + 1], g, nullptr, sc, stack_id * PV_MAX_OPS + i, s, true,
                  PV_ACT_NONE, nullptr, g_up));
  }
  return 0;
}

}  // namespace pvcs
