// pv_kernels.h — argument blocks and host launchers of the non-GEMM kernels (pv_elementwise.hip).
#pragma once
#include "pv_common.h"
#include "pv_fb_layout.h"

struct PvHead {
  const float* head;     // (B, 2*z_dim): [mu | softplus input]  (fc11 | fc12 of fcEncoderNet, fc.py:59-60)
  const float* eps;      // (B, z_dim)
  const float* y;        // (B, c_dim) or null
  float* z;              // (B, z_dim)
  float* z_scale;        // (B, z_dim)
  float* z_loc_out;      // optional copies for the caller
  float* z_scale_out;
  float* tp;             // (B, 8): cos, sin, scale, tx, ty  (null when coord_dim == 0)
  float* zy;             // (B, latent + c_dim) decoder latent input when c_dim > 0, else null
  float* scalars;        // [.., .., beta*logp, beta*logq]
  int B, z_dim, c_dim, coord_dim, has_r, has_t, has_s;
  float tp0, tp1, sc_prior, beta;
  int ldh;               // row stride of head (0: 2*z_dim)
  int scale_direct;      // 1: the second half of head IS z_scale (external encoder), not its softplus input
  const float* w;        // (B) per-sample weights of the KL sums (plan->row_w) or null
  // pv_head_fwd_blocks only (a conv encoder's tail as ONE launch of ceil(B / 16) workgroups, 16 samples each):
  //   ch_part != null: head[b][j] = ch_bias[j] + sum_seg ch_part[b][seg][j] first (pv_convhead_fwd_partials; ldh == ch_out), into head_w
  //   hz != null: hz[b][j] = sum_k zin[b * ldz + k] Wz[j * lat_in + k] last (fc_latent of the spatial decoder, lat_in <= 16)
  //   kl_part: (blocks, 2) partial sums of beta log p(z), beta log q(z|x) instead of scalars[2], [3] (pv_finish_scalars sums them)
  const float* ch_part; const float* ch_bias; float* head_w; int ch_nseg, ch_out;
  const float* zin; const float* Wz; float* hz; int64_t ldz; int lat_in, H;
  float* kl_part;
};
int pv_head_fwd(const PvHead& h, hipStream_t s);
int pv_head_fwd_blocks(const PvHead& h, hipStream_t s);
int pv_fill_tp(float* tp, int B, float angle, float sc, float tx, float ty, hipStream_t s);
int pv_concat(const float* a, int64_t lda, int na, const float* y, int64_t ldy, int nb, float* out, int64_t B,
              hipStream_t s);

struct PvCoordLat {
  const float* grid;     // (N, cd)
  const float* tp;       // (B, 8)
  const float* Wc;       // (H0, cd)  decoder.coord_latent.fc_coord.weight
  const float* bc;       // (H0)
  const float* hz;       // (B, H0) = fc_latent(z)
  float* h0;             // (M, H0)
  int64_t M;
  int N, cd, H0;
};
int pv_coordlat_fwd(const PvCoordLat& p, hipStream_t s);

struct PvOutLik {
  const float* h;        // (M, H) last hidden activation
  const float* hpre;     // its pre-activation (GELU only) or null
  int64_t ldh;
  const float* wo;       // (H)  decoder.out.weight (1, H)
  const float* bo;       // (1)
  const float* x;        // (M) observations
  float* loc;            // (M) or null
  float* llrow;          // (M) or null
  float* dpre;           // (M, H) dL/d(pre-activation of the last hidden layer), or null (no grads)
  float* part_dwo;       // (blocks, H)
  float* part_dbo;       // (blocks)
  int64_t M;
  int H, lik, sigmoid_out, act_last;
  float sig;
  const float* sw;       // per-sample weights of the gradients (row / N indexes it) or null
  int N;                 // rows per sample (used with sw)
  int64_t xmod;          // > 0: observations are x[row % xmod] (jiVAE: the K decoder passes score the same B*N pixels)
};
int pv_out_lik(const PvOutLik& p, hipStream_t s);
int64_t pv_out_lik_blocks(int64_t M);
int pv_segsum(const float* v, int64_t nseg, int64_t N, float* out, hipStream_t s);
// scalars[1] = sum_b llb; if kl_part: scalars[2] = beta*sum kl_part[2i], scalars[3] = beta*sum kl_part[2i+1];
// scalars[0] = -(scalars[1] + scalars[2] - scalars[3])
int pv_finish_scalars(const float* llb, int B, float* scalars, const float* kl_part, int n_part, float beta,
                      hipStream_t s);

struct PvCoordLatBwd {
  const float* dpre0;    // (M, H0)
  const float* grid;
  const float* tp;
  const float* Wc;
  float* part_hz;        // (B*nchunk, H0)
  float* part_wc;        // (B*nchunk, H0, cd)
  float* part_tp;        // (B*nchunk, 4)
  int N, cd, H0, rows_per_chunk;
};
int pv_coordlat_bwd(const PvCoordLatBwd& p, int nchunk, int B, hipStream_t s);
int pv_reduce_mid(const float* part, int nb, int nc, int n, float* out, hipStream_t s);

struct PvHeadBwd {
  const float* dzc;      // (B, ldzc): dL/d(decoder latent input) (content [+ y] columns)
  int64_t ldzc;
  const float* dtp;      // dphi, dscale, dtx, dty of sample b at dtp[b*dtp_sb + c*dtp_sc] (null when coord_dim == 0)
  int dtp_sb, dtp_sc;
  const float* z;
  const float* z_scale;
  const float* eps;
  const float* head;     // (B, ldh)
  float* dhead;          // (B, ldh): [dL/dmu | dL/d(softplus input) | (jiVAE) dL/d(class logits)]
  int B, z_dim, coord_dim, has_r, has_t, has_s;
  float tp0, tp1, sc_prior, beta;
  int ldh;               // row stride of head / dhead (0: 2*z_dim)
  int scale_direct;      // 1: head's second half is z_scale itself: dhead's second half = dloss/dz_scale
  const float* w;        // (B) per-sample weights (plan->row_w): scale the KL terms' derivatives (the decoder's arrive
                         // weighted already) or null
};
int pv_head_bwd(const PvHeadBwd& h, hipStream_t s);

// head_bwd: dL/d(mu), dL/d(softplus input) from the decoder's dL/dz and the sampled-KL terms.
// dz_coord(c) returns d(phi), d(scale), d(tx), d(ty) of the sample for c = 0..3; dz_content(k) the gradient
// w.r.t. the k-th column of the decoder's latent input.
// (the two halves of the element function below, for callers that hold the sample's values in registers: pv_sdec_fused_w8.hip)
template <class FC, class FK>
__device__ __forceinline__ float pv_head_dz(const PvHeadBwd& h, int i, FC dz_coord, FK dz_content) {
  if (h.coord_dim == 0) return dz_content(i);
  int idx = 0;
  float dz = 0.0f;
  bool done = false;
  if (h.coord_dim == 1) {
    if (h.has_t) { if (i == 0) { dz = dz_coord(2) * h.tp0; done = true; } idx = 1; }
  } else {
    if (h.has_r) { if (i == idx) { dz = dz_coord(0); done = true; } idx += 1; }
    if (h.has_t) {
      if (i == idx) { dz = dz_coord(2) * h.tp0; done = true; }
      if (i == idx + 1) { dz = dz_coord(3) * h.tp1; done = true; }
      idx += 2;
    }
    if (h.has_s) { if (i == idx) { dz = dz_coord(1) * h.sc_prior; done = true; } idx += 1; }
  }
  if (!done) dz = dz_content(i - idx);
  return dz;
}
// z, sig, ep, sp: the sample's z, z_scale, eps and softplus input of coordinate i; bw = beta (times the sample's weight)
__device__ __forceinline__ void pv_head_bwd_math(float dz, float z, float sig, float ep, float sp, float bw, int scale_direct,
                                                 float& g, float& ds) {
  g = dz + bw * z;                                 // d(-ll - beta*log p(z))/dz
  const float dsig = g * ep - bw / sig;            // + beta * d(log q)/d(sigma) (total derivative)
  const float sgm = scale_direct ? 1.0f : (sp > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-sp)));   // softplus'
  ds = dsig * sgm;
}
template <class FC, class FK>
__device__ __forceinline__ void pv_head_bwd_elem(const PvHeadBwd& h, int b, int i, FC dz_coord, FK dz_content,
                                                 float* dh_copy = nullptr) {
  const float dz = pv_head_dz(h, i, dz_coord, dz_content);
  const int e = b * h.z_dim + i;
  const int ldh = h.ldh > 0 ? h.ldh : 2 * h.z_dim;
  float g, ds;
  pv_head_bwd_math(dz, h.z[e], h.z_scale[e], h.eps[e], h.head[(int64_t)b * ldh + h.z_dim + i], h.w ? h.beta * h.w[b] : h.beta,
                   h.scale_direct, g, ds);
  h.dhead[(int64_t)b * ldh + i] = g;
  h.dhead[(int64_t)b * ldh + h.z_dim + i] = ds;
  if (dh_copy) { dh_copy[i] = g; dh_copy[h.z_dim + i] = ds; }
}


struct PvLatentBwd {
  const float* llrow;    // (M)
  const float* rowtp;    // (4, M)
  const float* part_hz;  // (B*kmax, H)
  const float* part_rs;  // (B*kmax, PV_RS_W) or null: the decoder's per-slot sums of {ll, d(phi), d(scale), d(tx), d(ty)} (pv_sdec_fused.h) —
                         // then llrow / rowtp are not read
  const float* Wz;       // (H, lat_in) decoder.coord_latent.fc_latent.weight
  float* llb;            // (B)
  float* dhz;            // (B, H)
  int dhz_ready;         // 1: the decoder launch already wrote dhz (PvFused::dhz_out): part_hz is not read
  const float* dzc_in;   // not null (with dhz_ready, K == 0): ... and dL/dz content (B, lat_in) = dhz Wz (PvFused::dzc_out)
  int64_t M;
  int N, kmax, H, lat_in;
  PvHeadBwd hb;          // dzc / dtp fields unused (values stay in LDS)
  // jiVAE (K > 0): the decoder ran on K*B samples ordered [k][b] with its rows already weighted by alpha[b][k];
  // llb[b] = sum_k alpha_bk ll_kb, and the class logits get their gradient (softmax backward) in dhead[:, 2z..]
  int K;                 // 0: iVAE
  const float* alpha;    // (B, K)
  float beta_disc;
  int fwd_only;          // 1: only llb (evaluation)
  float* row_ll;         // optional (B): the unweighted ll_b (llb gets hb.w[b] * ll_b when hb.w is set)
  float* dzc_out;        // optional (B, lat_in): dL/d(decoder latent input) (content and y columns)
  // compact encoder (enc_n > 0): the sample's encoder dgrad chain follows in the same workgroup — it needs nothing from
  // other samples: edp[last] = (dhead Whead) * act'(eact[last]); edp[i-1] = (edp[i] W_i) * act'(eact[i-1])
  // (what pv_enc_dgrad does 16 samples per workgroup on the matrix cores; here one sample, plain FMAs, one launch less)
  int enc_n;
  const float* enc_params;
  pv_layer enc_l[PV_MAX_LAYERS];
  pv_layer enc_head;
  const float* enc_act[PV_MAX_LAYERS];
  float* enc_dp[PV_MAX_LAYERS];
};
int pv_latent_bwd(const PvLatentBwd& p, hipStream_t s);
// jiVAE without enumeration (plan->class_onehot): see pv_elementwise.hip
int pv_jiv_sampled_prep(const float* alpha, const float* onehot, float* sw, float* fix, float beta_disc, int B, int K,
                        hipStream_t s);
int pv_jiv_sampled_fix(float* scalars, const float* fix, hipStream_t s);
int pv_jiv_combine_sampled(const float* llkb, const float* alpha, const float* onehot, float* llb, float* dzc, int ld_dzc,
                           int n_content, float* dhead, int ldh, int z_dim, int B, int K, float beta, float beta_disc,
                           int want_grads, const float* z, const float* head, const float* z_scale, hipStream_t s);
int pv_softmax_rows(const float* logits, int64_t ld, int B, int K, float* out, hipStream_t s);
int pv_jiv_combine(const float* llkb, const float* alpha, float* llb, float* dzc, int ld_dzc, int n_content, float* dhead,
                   int ldh, int z_dim, int B, int K, float beta_disc, int want_grads, hipStream_t s, float* dtp = nullptr);
// jiVAE on the generic encoder path, after pv_head_fwd: alpha = softmax(head[:, 2z:]), sw[k*B + b] = alpha_bk, the
// discrete KL terms added to scalars[2], scalars[3], tp rows and the decoder's latent input zy = [z content | onehot(k)]
// for the K*B decoder samples ordered [k][b]
int pv_jiv_expand(const float* head, int ldh, const float* z, int z_dim, int n_content, float* tp, float* zy, float* alpha,
                  float* sw, float* scalars, float beta_disc, int B, int K, hipStream_t s);
int pv_scale_rows(float* v, const float* w, int64_t rows, int64_t N, hipStream_t s);
// out[b] = row_ll[b] + beta * sum_i (log p(z_bi) - log q(z_bi | x_b))   (mu = head[b*ldh + i])
int pv_row_elbo(const float* row_ll, const float* z, const float* head, const float* z_scale, int B, int z_dim, int ldh,
                float beta, float* out, hipStream_t s);
// dst[b][i] += src[b*lds + i], i < n
int pv_add_cols(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t B, int n, hipStream_t s);
struct PvFusedOffsets;
int pv_latent_bwd_reduce(const PvLatentBwd& p, const float* part, int grid, float* G, const PvFusedOffsets& o, int cd,
                         hipStream_t s, int rec_fmt = 0);
// the step's closing launch where the decoder launch already ran every sample's latent backward and encoder chain (PvEncFold::chain):
// record sums (with Adam on the decoder parameters they finalise), the small weight gradients (Adam in their epilogues) and the
// loss scalars — three kinds of workgroups that need nothing from each other (pv_elementwise.hip: pv_rec_wgrad_kernel)
int pv_rec_wgrad(const float* part, int grid, float* G, const PvFusedOffsets& o, int cd, int rec_fmt, const PvGemm* gs, int n,
                 const PvAdamFuse* adam, const PvFinishArgs* fin, hipStream_t s, unsigned* tick = nullptr);

// pv_lik_elem + pv_segsum in one launch (one workgroup per sample; same summation order)
int pv_lik_rows(const float* a, const float* x, int64_t B, int64_t per, int lik, int sigmoid_out, float sig, float* loc,
                float* dlda, float* llb, hipStream_t s);
int pv_lik_elem(const float* a, const float* x, int64_t M, int lik, int sigmoid_out, float sig, float* loc,
                float* llrow, float* dlda, hipStream_t s);

// ---- compact encoder kernels (pv_encoder.hip) ----
// deterministic block-wide sum for any blockDim.x that is a multiple of 64 (<= 1024); result valid in every thread
__device__ __forceinline__ float pv_block_sum(float v, float* sm /* >= 16 floats */) {
  v = pv_wave_sum(v);
  pv_lds_barrier();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  pv_lds_barrier();
  float t = 0.0f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
  return t;
}
// scalars[1] = sum_b ll_b ; [2], [3] from the encoder kernel's partials when given ; scalars[0] = -(ll + lp - lq)
__device__ __forceinline__ void pv_finish_scalars_block(const float* __restrict__ llb, int B, float* scalars,
                                                        const float* __restrict__ kl_part, int n_part, float beta,
                                                        float* sm /* >= 16 floats */) {
  float a = 0.0f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) a += llb[b];
  a = pv_block_sum(a, sm);
  float lp = 0.0f, lq = 0.0f;
  if (kl_part) {
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) { lp += kl_part[2 * i]; lq += kl_part[2 * i + 1]; }
    lp = pv_block_sum(lp, sm);
    lq = pv_block_sum(lq, sm);
  }
  if (threadIdx.x == 0) {
    if (kl_part) { scalars[2] = beta * lp; scalars[3] = beta * lq; }
    scalars[1] = a;
    scalars[0] = -(a + scalars[2] - scalars[3]);
  }
}

struct PvEncFwd {
  const float* params;
  pv_layer enc[PV_MAX_LAYERS];
  pv_layer head;
  int n_enc;
  const float* x; int64_t ldx;      // (B, ldx) encoder input (x or cat(x, y))
  const float* eps; const float* y;
  float* eact[PV_MAX_LAYERS];       // hidden activations (B, out_i)
  float* head_out;                  // (B, 2*z_dim)
  float* z; float* z_scale; float* z_loc_out; float* z_scale_out;
  float* tp; float* zy; float* kl_part;   // kl_part: (blocks, 2) partial sums of log p(z), log q(z|x)
  float* hz; const float* Wz; int H0;     // fc_latent (null hz: skip)
  float hz_scale;                   // hz is stored multiplied by this (0: unscaled); see PvFused.hz_scale
  int B, z_dim, c_dim, coord_dim, has_r, has_t, has_s;
  float tp0, tp1, sc_prior;
  float beta, beta_disc;            // the KL partials in kl_part are stored scaled by these
  // jiVAE (K > 0): head = [mu | softplus input | class logits]; alpha = softmax(logits); tp, zy, hz are written for
  // the K*B decoder samples ordered [k][b] (zy = [z content | onehot(k)]); sw[k*B + b] = alpha[b][k]
  int K; float* alpha; float* sw;
  const float* w;                   // (B) per-sample weights of the KL partial sums (plan->row_w) or null
  PvFbPrep prep;                    // hosted in the first-layer launch when prep.img is set (bf16x3 decoder path)
  unsigned* flags; unsigned gen;    // != null: one launch for both kernels; (row blocks x 8) words of scratch, any content (pv_encoder.hip)
  int spin_limit;                   // merged launch: polls of a tile flag before the consumer computes the tile itself
};
bool pv_enc_compact_supported(const pv_ivae_plan* p);
int pv_enc_fwd(const PvEncFwd& e, hipStream_t s);

struct PvEncDgrad {
  const float* params;
  pv_layer enc[PV_MAX_LAYERS];
  pv_layer head;
  int n_enc, B;
  const float* dhead;               // (B, 2*z_dim)
  const float* eact[PV_MAX_LAYERS];
  float* edp[PV_MAX_LAYERS];        // out: dL/d(pre-activation) of every hidden layer
  // hosted in one extra workgroup when fin_scalars is set: the step's loss scalars (pv_finish_scalars)
  const float* fin_llb; float* fin_scalars; const float* fin_kl_part; int fin_n_part; float fin_beta;
};
int pv_enc_dgrad(const PvEncDgrad& e, hipStream_t s);
