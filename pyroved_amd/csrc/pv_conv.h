// pv_conv.h — data-movement kernels of the convolutional nets (pv_conv.hip).  Channels-last activations
// [B][H][W][C]; 1-D data has W = 1 (nd = 1), 2-D nd = 2.
#pragma once
#include "pv_common.h"

// The split-order reductions that finish the weight gradients of a step, recorded while the backward runs and done by ONE
// launch at its end (pv_wgrad_finish_all) instead of one launch per layer.  A deferring weight gradient takes its partials'
// workspace from the list's region (it must survive until then); a null / full list means "finish now".
struct PvFinishEntry { const float* part; float* out; const float* part_b; float* out_b; int64_t n; int nsplit, nb, blk0, nblk; };
// a register-fed weight-gradient problem (pv_conv_k1.hip) and a batch of them recorded for one launch (pv_k1_wgrad_flush)
struct PvK1Wg {
  const float* g; const float* in; float* part; float* part_b;
  int64_t rows, chunk;
  int Ci, Co, mtiles, ntiles, nsplit, up;
  int taps, L;            // taps = 3: the kernel-3 1-D convolution's weight gradient, dW[m][n][t] = sum_p g[p][m] in[p + t - 1][n] within
                          // a sample of L positions (zero padding); taps = 1: kernel 1
  int nblk;               // workgroups of the problem
  int fat;                // batched launches: 1 = 32 x 32 output tiles with all taps in the workgroup (mtiles / ntiles count those);
                          // 2 = the wide form: (16 mj) x (16 nj) tiles fed by mj- / nj-float vector loads (round 5)
  int mj, nj;
};
#define PV_K1_BATCH 12
struct PvK1Batch { PvK1Wg e[PV_K1_BATCH]; int n; };
int pv_k1_wgrad_flush(PvK1Batch* b, hipStream_t s);
// k1b != null: register-fed weight gradients whose partials live in the list are RECORDED there instead of launched — their g
// and input buffers must stay untouched until pv_k1_wgrad_flush (before pv_wgrad_finish_all)
// st[k]: the stream entry k's partials are written on (two-stream steps, pv_side.h)
struct PvFinishList { PvFinishEntry e[16]; int n; char* base; int64_t off, cap; PvK1Batch* k1b; hipStream_t st[16]; };
// ws / ws_bytes for a weight gradient needing `need` bytes: a slice of the list's region (returns true: deferred) or the caller's
bool pv_wgrad_ws(PvFinishList* list, int64_t need, void*& ws, int64_t& ws_bytes);
// would pv_wgrad_ws(list, need, ...) defer?
inline bool pv_wgrad_defers(const PvFinishList* list, int64_t need) {
  return list && list->base && list->n < 16 && list->off + need <= list->cap;
}
int pv_wgrad_finish(PvFinishList* list, const float* part, int nsplit, int64_t n, float* out, const float* part_b, int nb,
                    float* out_b, hipStream_t s);
int pv_wgrad_finish_all(PvFinishList* list, hipStream_t s);
// the reductions recorded so far as one launch on `s` (which first waits for every other stream that writes their partials);
// the list stays open — later entries keep taking fresh slices of its region — and pv_wgrad_finish_all closes it
int pv_wgrad_finish_flush(PvFinishList* list, hipStream_t s);

int pv_maxpool2_fwd(const float* in, float* out, int B, int H, int W, int C, int nd, hipStream_t s);
// eg_act != NONE: din *= act'(in) (in = the pooled tensor = the producing conv's post-activation output)
int pv_maxpool2_bwd(const float* in, const float* dout, float* din, int B, int H, int W, int C, int nd, hipStream_t s,
                    int eg_act = 0);
int pv_upsample2_fwd(const float* in, float* out, int B, int H, int W, int C, int nd, hipStream_t s);
// eg_y / eg_act: optionally din *= act'(eg_y) (eg_y = the upsampled tensor, shaped like din)
int pv_upsample2_bwd(const float* dout, float* din, int B, int H, int W, int C, int nd, hipStream_t s, const float* eg_y = nullptr,
                     int eg_act = 0);
int pv_ncs_to_nsc(const float* in, float* out, int64_t B, int C, int64_t S, hipStream_t s);
int pv_nsc_to_ncs(const float* in, float* out, int64_t B, int C, int64_t S, hipStream_t s);
int pv_act_bwd(float* dy, const float* y, int64_t n, int act, hipStream_t s);
int pv_conv_wflip(const float* w, float* wt, int Cout, int Cin, int KK, hipStream_t s);
// direct kernel-3 convolution on the matrix cores (pv_conv_direct.hip); wt_scratch: pv_conv3_direct_wt_floats floats
bool pv_conv3_direct_supported(int C, int Cout, int nd, int act);
int64_t pv_conv3_direct_wt_floats(int C, int Cout, int nd);
bool pv_conv3_wgrad_direct_supported(int C, int Cout, int nd);
int64_t pv_conv3_wgrad_direct_ws(int B, int H, int W, int C, int Cout, int nd);
int pv_conv3_wgrad_direct(const float* dy, const float* in, int B, int H, int W, int C, int nd, float* dw, float* db, int Cout,
                          void* ws, int64_t ws_bytes, hipStream_t s, PvFinishList* defer = nullptr);
// one input channel (first encoder layer): a streaming reduction instead of a GEMM (pv_conv_direct.hip)
bool pv_conv3_wgrad_c1_supported(int C, int Cout, int nd);
int64_t pv_conv3_wgrad_c1_ws(int B, int H, int W, int C, int Cout, int nd);
int pv_conv3_wgrad_c1(const float* dy, const float* in, int B, int H, int W, int nd, float* dw, float* db, int Cout, void* ws,
                      int64_t ws_bytes, hipStream_t s, PvFinishList* defer = nullptr);
// eg_y / eg_act: optionally out *= act'(eg_y) (eg_y shaped like out): the producing layer's activation backward fused
// into the input-gradient form
int pv_conv3_wgrad_direct_bf16(const float* dy, const float* in, int B, int H, int W, int C, int nd, float* dw, float* db,
                               int Cout, void* ws, int64_t ws_bytes, hipStream_t s, PvFinishList* defer = nullptr);     // C % 32 == 0 (mixed precision)
int pv_conv3_direct(const float* in, int B, int H, int W, int nd, const float* w, int Co, int Ci, int flip, const float* bias,
                    float* out, int act, float* wt_scratch, hipStream_t s, const float* eg_y = nullptr, int eg_act = 0,
                    int use_bf16 = 0, const void* wt_ready = nullptr);
// 2-D kernel-3 convolution on the bf16 matrix cores with exactly split operands (pv_conv_sp.hip): ns = 3 fp32-class
// (six products), ns = 2 mixed precision (three); C % 32 == 0.  wt_scratch: pv_conv3_sp_wt_bytes bytes
bool pv_conv3_sp_supported(int C, int Cout, int nd, int act);
// the fp32-class form of the forward / input-gradient kernel: 4 = fp16 two-piece with exact scaling (default), 3 = bf16
// three-piece (PV_SP_X6=1)
int pv_conv3_sp_fp32_mode();
int64_t pv_conv3_sp_wt_bytes(int C, int Cout);
// a layer's weight gradient + input gradient as ONE launch: between begin and flush the fp16-mode pv_conv3_sp_wgrad (with a
// deferring finish list) and pv_conv3_sp calls record their kernel instead of launching it; flush launches what was recorded
void pv_conv3_sp_pair_begin();
int pv_conv3_sp_pair_flush(hipStream_t s);
// up_code != null: out is (B, 2H, 2W, N) — the result un-pooled by the winner bytes up_code (B, H, W, N) of a fused conv + max-pool
int pv_conv3_sp(const float* in, int B, int H, int W, const float* w, int Co, int Ci, int flip, const float* bias, float* out,
                int act, void* wt_scratch, hipStream_t s, const float* eg_y, int eg_act, int ns, const void* wt_ready = nullptr,
                float* pool_out = nullptr, unsigned char* pool_code = nullptr, const unsigned char* up_code = nullptr);
// pool_out / pool_code (forward form, even H and W): the 2x max-pool of the output (B, H/2, W/2, Co) and one byte per pooled value
// (which of the 2x2 positions won) are written INSTEAD of the full-resolution output; pv_maxpool2_bwd_code is its backward
int pv_maxpool2_bwd_code(const float* g, const float* y_pooled, const unsigned char* code, float* din, int B, int Hp, int Wp, int C,
                         int eg_act, hipStream_t s);
// all of a step's weight tilings in one launch (per 16 entries).  kind 0: pv_conv3_direct f32, 1: its bf16 two-piece form,
// 6: that kernel's fp16 two-piece form, 2 / 3: pv_conv3_sp with 2 / 3 bf16 pieces, 5: its fp16 two-piece form (ns = 4), 4: a conv head's Linear weight re-indexed channels-last (Co = out, Ci = C,
// KK = spatial size); dst sized by pv_conv_wt_bytes
struct PvWprepEntry { const float* w; char* dst; int Co, Ci, KK, flip, kind; int pad_; int64_t start, total; };
int64_t pv_conv_wt_bytes(int kind, int Co, int Ci, int nd);
struct PvFbPrep;
// fb != null: the spatial decoder's bf16 weight images (pv_fb_layout.h) are written by extra workgroups of the (first) launch
int pv_conv_wprep_table(PvWprepEntry* e, int n, hipStream_t s, const PvFbPrep* fb = nullptr);
bool pv_conv3_sp_wgrad_supported(int C, int Cout, int nd);
int64_t pv_conv3_sp_wgrad_ws(int B, int H, int W, int C, int Cout);
int pv_conv3_sp_wgrad(const float* dy, const float* in, int B, int H, int W, int C, float* dw, float* db, int Cout, void* ws,
                      int64_t ws_bytes, hipStream_t s, int ns, PvFinishList* defer = nullptr);
// first encoder block (conv k3 from one channel + activation + 2x max-pool) fused (pv_conv_c1.hip); code: one byte per
// pooled value
bool pv_c1_convpool_supported(int Cin, int Cout, int nd, int act, int H, int W);
int64_t pv_c1_convpool_ws(int B, int H, int W, int Cout);
int pv_c1_convpool_fwd(const float* x, int B, int H, int W, const float* w, const float* bias, int Cout, int act, float* out,
                       unsigned char* code, hipStream_t s);
int pv_c1_convpool_bwd(const float* g, const float* y, const unsigned char* code, const float* x, int B, int H, int W, int Cout,
                       int act, float* dw, float* db, void* ws, int64_t ws_bytes, hipStream_t s, PvFinishList* defer = nullptr);
// the Linear head over a channels-last feature map without transposes (pv_convhead.hip); wt: the weight re-indexed to
// [out][s*C + c] (pv_conv_wprep_table kind 4: Co = out, Ci = C, KK = S)
bool pv_convhead_supported(int64_t F, int out);
int64_t pv_convhead_ws(int B, int64_t F, int out);
int pv_convhead_fwd(const float* a, const float* wt, const float* bias, float* head, int B, int64_t F, int out, void* ws,
                    int64_t ws_bytes, hipStream_t s);
// the forward without its finish launch: *part (B, *nseg, out) partial sums in ws, to be added in segment order on top of the bias
// by the consumer (pv_dec1d.hip, per sample); PV_EINVAL when the matrix-core form does not apply
int pv_convhead_fwd_partials(const float* a, const float* wt, int B, int64_t F, int out, void* ws, int64_t ws_bytes, hipStream_t s,
                             const float** part, int* nseg);
int pv_convhead_bwd(const float* dhead, const float* wt, const float* y, int act, float* g, int B, int64_t F, int out,
                    hipStream_t s);
bool pv_convhead_wgrad_uses_ws();
int pv_convhead_wgrad(const float* dhead, const float* a, float* dw, float* db, int B, int S, int C, int out, void* ws,
                      int64_t ws_bytes, hipStream_t s);
// latent_to_features (Linear z -> C*S, viewed channels-first) producing / consuming channels-last maps directly; wt: kind 7 of
// pv_conv_wprep_table (Co = z_dim, Ci = C, KK = S): wt[k][s*C + c] = w[c*S + s][k].  The input gradient is pv_convhead_fwd(g, wt).
bool pv_l2f_supported(int64_t F, int zd, int C);
int pv_l2f_fwd(const float* z, const float* wt, const float* bias, float* a, int B, int S, int C, int zd, hipStream_t s);
int pv_l2f_wgrad(const float* g, const float* z, float* dw, float* db, int B, int S, int C, int zd, hipStream_t s);
// kernel-1 convolutions over channels-last maps (pv_conv_k1.hip): operands straight from L2 into f32 MFMAs, no LDS stages
// up = 1: the 1-D nearest 2x upsample that follows the convolution fused (forward: every output row stored twice; backward:
// g has 2 rows rows and g[p] stands for g[2p] + g[2p + 1])
int pv_k1_fwd(const float* in, int64_t rows, int Ci, const float* w, const float* bias, float* out, int Co, int act, hipStream_t s,
              int up = 0);
int pv_k1_dgrad(const float* g, int64_t rows, int Co, const float* w, float* gin, int Ci, const float* eg_y, int eg_act,
                hipStream_t s, int up = 0);
int64_t pv_k1_wgrad_ws(int64_t rows, int Ci, int Co);
int pv_k1_wgrad(const float* g, const float* in, int64_t rows, int Ci, int Co, float* dw, float* db, void* ws, int64_t ws_bytes,
                hipStream_t s, PvFinishList* defer = nullptr, int up = 0);
// kernel-3 1-D convolution weight gradient on the same register-fed kernel (dw (Co, Ci, 3))
int64_t pv_conv3_1d_wgrad_lean_ws(int B, int L, int Ci, int Co);
int pv_conv3_1d_wgrad_lean(const float* g, const float* in, int B, int L, int Ci, int Co, float* dw, float* db, void* ws,
                           int64_t ws_bytes, hipStream_t s, PvFinishList* defer = nullptr);
// y (B, N) = x (B, K; row stride ldx) w(N, K)^T for K <= 16
int pv_smallk_linear(const float* x, int64_t ldx, const float* w, float* y, int64_t B, int K, int N, hipStream_t s);
int pv_upsample2_bil_fwd(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);
int pv_upsample2_bil_bwd(const float* dout, float* din, int B, int H, int W, int C, hipStream_t s);
// nn.BatchNormNd over channels-last rows x[R][C]: stats[0..C) = mean, stats[C..2C) = 1/sqrt(var + eps) (kept for backward)
int64_t pv_bn_ws(int64_t R, int C);
int pv_bn_fwd(const float* x, float* y, int64_t R, int C, const float* gamma, const float* beta, float* rmean, float* rvar,
              int eval, float momentum, float eps, float* stats, void* ws, int64_t ws_bytes, hipStream_t s);
// dy -> dx (dx may be null), dgamma, dbeta; x = the layer's input, stats from the forward
int pv_bn_bwd(const float* x, const float* dy, float* dx, int64_t R, int C, const float* gamma, const float* stats, int eval,
              float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, hipStream_t s);
