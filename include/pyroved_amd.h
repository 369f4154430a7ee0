/*
 * pyroved_amd.h — C ABI of libpyroved_amd.so: the MI355X (gfx950) implementation of
 * pyroVED's SVI training hot path (encoder -> reparameterise -> coordinate-grid
 * spatial decoder -> ELBO -> gradients -> Adam).
 *
 * The reference (ziatdinovmax/pyroVED) has no FFI of its own: its boundary is the
 * Python API.  Each entry point below replaces the torch/Pyro call sequence of
 * the reference function it cites (paths relative to /root/reference/pyroved).
 * The Python host side (pyroved_amd/_abi.py) binds them with ctypes; the stub a
 * maintainer would add to the reference itself is in INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous fp32 unless stated;
 *    weights use torch.nn.Linear layout (out_features, in_features), row-major;
 *  - the caller owns every buffer (including the workspace); the library never
 *    allocates device memory, never frees, never keeps a pointer past the call;
 *  - all work is enqueued asynchronously on `stream` (a hipStream_t passed as
 *    void*); no implicit device synchronisation; safe inside stream capture;
 *  - return value: 0 on success, otherwise a hipError_t (>0) or a PV_E* code (<0);
 *    nothing throws across the ABI.
 */
#ifndef PYROVED_AMD_H
#define PYROVED_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PV_ABI_VERSION 16

/* error codes (negative; positive values are hipError_t) */
#define PV_EINVAL   (-1)   /* bad argument / unsupported configuration */
#define PV_EWS      (-2)   /* workspace too small */
#define PV_ECOLL    (-3)   /* (v16) the collective library (RCCL) could not be loaded or returned an error */

/* activations: utils/nn.py:118-124 (get_activation) + the output sigmoid */
enum pv_act {
  PV_ACT_NONE = 0, PV_ACT_TANH = 1, PV_ACT_RELU = 2, PV_ACT_LRELU = 3,
  PV_ACT_SOFTPLUS = 4, PV_ACT_GELU = 5, PV_ACT_SIGMOID = 6
};

/* decoder likelihoods: utils/prob.py:25-29 (get_sampler) */
enum pv_lik { PV_LIK_BERNOULLI = 0, PV_LIK_GAUSSIAN = 1, PV_LIK_CBERNOULLI = 2 /* ContinuousBernoulli(probs) */ };

#define PV_MAX_LAYERS 8

/* One nn.Linear(+activation) of make_fc_layers (nets/fc.py:307-324).  Offsets are
 * in floats into the flat parameter buffer (and, identically, the flat gradient
 * and Adam-moment buffers).  b_off < 0: no bias. */
typedef struct pv_layer {
  int32_t in_dim;
  int32_t out_dim;
  int32_t act;
  int32_t _pad;
  int64_t w_off;
  int64_t b_off;
} pv_layer;

/* ---- convolutional op sequences (nets/conv.py FeatureExtractor / Upsampler), used by pv_ved_plan and by
 * pv_ivae_plan's optional convolutional encoder ---- */
#define PV_MAX_OPS 32

enum pv_op_kind {
  PV_OP_CONV = 1,        /* nn.ConvNd(cin, cout, ksize, 1, ksize/2) + activation `act` (weights in the torch
                            layout (cout, cin, *kernel) at w_off, bias at b_off)                        */
  PV_OP_MAXPOOL2 = 2,    /* nn.MaxPoolNd(2, 2)                                                          */
  PV_OP_UPSAMPLE2 = 3,   /* F.interpolate(scale_factor=2, mode="nearest")                               */
  PV_OP_UPSAMPLE2_BILINEAR = 4,  /* F.interpolate(scale_factor=2, mode="bilinear"), 2-D only (align_corners=False) */
  PV_OP_BATCHNORM = 5    /* nn.BatchNormNd(cin) (affine, momentum 0.1, eps 1e-5; nets/conv.py:185-186, 239-240): weight at
                            w_off, bias at b_off, running_mean / running_var at aux0_off / aux1_off — all in the flat
                            parameter buffer (the running statistics never receive a gradient, so Adam leaves them).
                            Batch statistics unless the plan's bn_eval is set                              */
};

typedef struct pv_op {
  int32_t kind;          /* enum pv_op_kind */
  int32_t cin, cout;     /* CONV only       */
  int32_t ksize;         /* CONV: 1 or 3    */
  int32_t act;           /* CONV: enum pv_act applied to the output */
  int32_t _pad;
  int64_t w_off, b_off;  /* CONV / BATCHNORM: offsets in floats into the flat parameter buffer */
  int64_t aux0_off, aux1_off;   /* BATCHNORM: running_mean, running_var */
} pv_op;

/* Everything one SVI step of models.iVAE needs (models/ivae.py:122-221,
 * models/base.py:47-119, trainers/svi.py:64-115).  Plain data: filled by the
 * caller, read by the library during the call only. */
typedef struct pv_ivae_plan {
  /* ---- problem ---- */
  int32_t batch;          /* B: samples in this (local) minibatch                         */
  int32_t n_pix;          /* N = prod(data_dim)                                           */
  int32_t coord_dim;      /* 0: vanilla VAE (fcDecoderNet); 1: 1-D grid; 2: 2-D grid      */
  int32_t z_dim;          /* latent_dim + coord (ivae.py:159)                             */
  int32_t latent_dim;     /* content latents (last entries of z)                          */
  int32_t c_dim;          /* class-conditioning width (0 = none)                          */
  int32_t has_r, has_t, has_s;   /* invariances (base.py:110-118: fixed order r,t,s)      */
  float   t_prior[2];     /* base.py:73-77                                                */
  float   sc_prior;       /* base.py:79-80                                                */
  float   beta;           /* KL scale_factor (ivae.py:175,214)                            */
  int32_t lik;            /* enum pv_lik                                                  */
  int32_t sigmoid_out;    /* sigmoid_d (ivae.py:153)                                      */
  float   decoder_sig;    /* Normal scale for the gaussian sampler (prob.py:28)           */
  int32_t fused;          /* spatial-decoder path when the architecture allows a fused persistent kernel:
                             0 layer-by-layer kernels; 1 fused, f32-input MFMA; 2 fused, bf16 split-
                             precision MFMA (x = hi + lo, three products, fp32 accumulate: fp32-class
                             results); 3 fused, plain bf16 operands for the two hidden layers' contractions
                             (one product, fp32 accumulate; everything else fp32) — mixed-precision training */
  int32_t discrete_dim;   /* models.jiVAE (models/jivae.py:109-220): K classes of the joint discrete latent,
                             enumerated exactly in the ELBO (TraceEnum_ELBO, trainers/svi.py:83-90); 0: iVAE.
                             Then `head` has out_dim 2*z_dim + K (fc13 appended, softmax -> alpha), fc_latent
                             in_dim latent_dim + K, and the decoder runs on K*B rows ordered [k][b]           */
  float   beta_disc;      /* scale_factor of the discrete KL term (jivae.py:161-165); `beta` scales the
                             continuous one                                                                  */
  /* ---- networks ---- */
  int32_t  n_enc;                  /* hidden layers of encoder_z.fc_layers (fc.py:44-45)  */
  int32_t  n_dec;                  /* hidden layers of decoder.fc_layers                  */
  pv_layer enc[PV_MAX_LAYERS];
  pv_layer head;                   /* fc11 and fc12 (fc.py:46-47,59-60) as ONE Linear of out_dim 2*z_dim:
                                      rows [0,z_dim) = fc11.weight, rows [z_dim,2*z_dim) = fc12.weight,
                                      bias = [fc11.bias | fc12.bias]; the caller lays the two tensors
                                      out adjacently in the flat buffer                      */
  pv_layer fc_coord, fc_latent;    /* coord_latent (fc.py:216-217); unused if coord_dim=0 */
  pv_layer dec[PV_MAX_LAYERS];
  pv_layer out;                    /* decoder.out (fc.py:186 / fc.py:140)                 */
  /* ---- optional convolutional encoder: iVAE.set_encoder(convEncoderNet(data_dim, latent_dim=z_dim))
   * (models/base.py:173-177, nets/conv.py:24-64).  n_enc_ops > 0: `enc` / n_enc are ignored, the encoder is this
   * op sequence over x viewed as (B, 1, *enc_in_dim), and `head` is features2latent.fc_latent (in_dim = C *
   * prod(spatial) in torch's flatten order, out_dim = 2*z_dim = [mu | softplus input]).  c_dim must be 0. ---- */
  int32_t  n_enc_ops;
  int32_t  enc_ndim;               /* 1 or 2                                              */
  int32_t  enc_in_dim[2];
  pv_op    enc_ops[PV_MAX_OPS];
  /* ---- caller-owned device buffers ---- */
  float*       params;    /* flat parameters, n_params floats                             */
  float*       grads;     /* flat gradients (written, not accumulated)                    */
  float*       adam_m;    /* flat first moments                                           */
  float*       adam_v;    /* flat second moments                                          */
  int64_t      n_params;
  const float* x;         /* (B, N) observations                                          */
  const float* y;         /* (B, c_dim) or NULL                                           */
  const float* eps;       /* (B, z_dim) standard-normal draws of Normal.rsample           */
  const float* grid;      /* (N, coord_dim) generate_grid(data_dim) (coord.py:21-44)      */
  void*        ws;        /* workspace, >= pv_ivae_workspace_bytes(plan) bytes            */
  int64_t      ws_bytes;
  float*       scalars;   /* out, 4 floats: loss, sum log p(x|z), beta*sum log p(z), beta*sum log q(z|x) */
  float*       z_loc;     /* out (B, z_dim), may be NULL                                  */
  float*       z_scale;   /* out (B, z_dim), may be NULL                                  */
  float*       loc;       /* out (B, N) decoder output, may be NULL ((K*B, N) for jiVAE)  */
  float*       alpha;     /* out (B, discrete_dim) class probabilities q(k|x), may be NULL */
  /* ---- external encoder: a user-supplied encoder_z (iVAE.set_encoder(any nn.Module), models/base.py:173-177) runs
   * in the caller's framework.  ext_encoder != 0: the library does not run an encoder; ext_head (B, 2*z_dim) holds
   * [z_loc | z_scale] = encoder_z(x) (scale already positive) and, with want_grads, ext_dhead (B, 2*z_dim) receives
   * [dloss/dz_loc | dloss/dz_scale] for the caller to back-propagate; `enc`, `enc_ops` and `head` are ignored and
   * only the decoder's parameters live in the flat buffers.  Not combined with discrete_dim / c_dim. ---- */
  const float* ext_head;
  float*       ext_dhead;
  int32_t      ext_encoder;
  int32_t      bn_eval;   /* batch-norm layers of the convolutional encoder use their running statistics (the module is
                             in eval() mode) instead of batch statistics                                            */
  /* ---- per-sample weights and extra outputs: what the semi-supervised models (models/ssivae.py, models/ss_reg_ivae.py)
   * need from the same step.  With e_b = log p(x_b|z_b,y_b) + beta (log p(z_b) - log q(z_b|x_b,y_b)):
   *   row_w    (B) or NULL: loss = -sum_b row_w[b] e_b (scalars and every gradient weighted accordingly) — the
   *            enumerated-label expectation of TraceEnum_ELBO with row_w = q(y|x) (ssivae.py:197-211);
   *   row_elbo (B) out or NULL: the unweighted e_b;
   *   dy       (B, c_dim) out or NULL (want_grads, c_dim > 0): dloss/dy — y is a reparameterised sample in
   *            ss_reg_iVAE.guide (ss_reg_ivae.py:196-199).
   * Not combined with discrete_dim, the convolutional or an external encoder. ---- */
  const float* row_w;
  float*       row_elbo;
  float*       dy;
  /* ---- external decoder: a user-supplied decoder (baseVAE.set_decoder(any nn.Module), models/base.py:179-183) runs in
   * the caller's framework together with the coordinate transform and the likelihood.  ext_decoder != 0: the step is
   * pv_ivae_guide (encoder, reparameterisation, sampled-KL terms; the sampled z (B, z_dim) lands in ext_z) -> the
   * caller's decoder forward / backward -> pv_ivae_guide_backward (ext_ll[0] = sum log p(x|z), ext_dz (B, z_dim) =
   * d(-sum log p(x|z))/dz: loss scalars, head and encoder backward).  `fc_coord`, `fc_latent`, `dec`, `out` are
   * ignored and only the encoder's parameters live in the flat buffers.  The workspace must stay untouched between the
   * two calls.  Not combined with discrete_dim or the row_w / row_elbo / dy fields. ---- */
  float*       ext_z;
  const float* ext_dz;
  const float* ext_ll;
  int32_t      ext_decoder;
  int32_t      conv_wide; /* (v14) convolutional encoder, fp32-class modes: a kernel-3 weight lies outside the range the two-piece
                             fp16 kernels are exact in (|w| < ~1000, not all < ~1e-6): three bf16 pieces instead (no range limit).
                             A fact of THIS model's weights — rounds 2-3 had a process-wide setter                              */
  /* ---- Adam (torch.optim.Adam defaults via pyro.optim.Adam, svi.py:79-81) ---- */
  float   lr, adam_beta1, adam_beta2, adam_eps;
  int32_t adam_step;      /* 1-based step count of THIS update                            */
  int32_t flags;          /* (v14) PV_PLAN_* bits below                                   */
  /* ---- optional instrumentation ---- */
  void*   ev_start;       /* hipEvent_t recorded on `stream` right before the dominant decoder  */
  void*   ev_stop;        /* kernel's launch and right after it (NULL: no recording)            */
  /* ---- jiVAE WITHOUT enumeration (ABI v10): SVItrainer's default enumerate_parallel=False (trainers/svi.py:66, 83-91)
   * runs Trace_ELBO on a class y_b ~ OneHotCategorical(alpha_b) DRAWN by the guide (models/jivae.py:213-220).
   * class_onehot (B, discrete_dim) one-hot rows or NULL (= exact enumeration, TraceEnum_ELBO).  When set:
   *   loss = -sum_b [ log p(x_b|z_b,y_b) + beta (log p(z_b) - log q(z_b)) + beta_disc (log(1/K) - log alpha_b[y_b]) ],
   *   decoder / encoder-through-z gradients as for a class-conditioned step on y_b, and the class logits receive the
   *   score-function gradient -log_r_b (onehot_b - alpha_b), log_r_b = the bracket above (Pyro: trace_elbo.py
   *   _compute_log_r + ScoreParts).  Vanilla decoder only (coord_dim == 0): with invariances the reference's model
   *   cannot broadcast its K-times repeated z against the drawn class (models/jivae.py:181-189) — PV_EINVAL.
   *   The caller draws y from `alpha` of pv_ivae_encode on the same x (the guide's order: eps first, then y). ---- */
  const float* class_onehot;
  /* ---- measurement (v12): hipEvent_t pair recorded on `stream` around the HEAVIEST kernel-3 convolution launch of the
   *   convolutional encoder's forward (most multiply-adds; NULL: no recording), and — out — that launch's algorithmic
   *   FLOPs 2 B H W Cin Cout 9 written to *conv_ev_flops when it is not NULL (host memory, written before the call returns).
   *   bench.py's roofline for the conv-encoder config; nothing in the reference corresponds. ---- */
  void*   conv_ev_start;
  void*   conv_ev_stop;
  double* conv_ev_flops;
  /* ---- (v15) which build of the fused decoder kernel runs: 0 = the library's choice by problem size (what every caller
   *   wants).  Non-zero values exist for parity tests and A/B timing of the other builds on the same inputs — a PLAN field:
   *   rounds 3-4 had process-wide debug setters.  fused == 3: 1 the 4-wave kernel, 2 the 8-wave kernel
   *   (pv_sdec_fused_w8.hip).  fused == 2: 1 the bf16 three-product kernel, 21 / 28 the fp16 builds H221 / H231
   *   (pv_sdec_fused_bf16.hip); forward-only launches ignore 21 / 28.  Further values name dropped variants that only
   *   the experiments build of the library contains (csrc/Makefile `experiments`); elsewhere they are PV_EINVAL. ---- */
  int32_t dec_kernel;
  int32_t reserved0;
} pv_ivae_plan;

/* Library / ABI version (PV_ABI_VERSION). */
int pv_version(void);
/* (v15) 1 when the library is the -DPV_EXPERIMENTS build (csrc/Makefile `experiments`: environment A/B switches and the
 * dropped kernel variants compiled in), 0 for the shipped one (no environment switch besides PV_ROCTX). */
int pv_experiments_build(void);

/* (v14) Per-plan switches that rounds 2-3 kept in process-wide setters (`flags` of pv_ivae_plan / pv_ved_plan /
 * pv_convnet_plan).  (v15) The library reads NO environment variable besides PV_ROCTX (roctx ranges) and keeps no mutable
 * process-wide switch: the v12 / v13 setters pv_conv_set_wide_weights / pv_set_side_stream are REMOVED (a binding that still
 * looks them up fails at load instead of silently doing nothing), the test hooks of rounds 3-4 are the plan fields below and
 * pv_ivae_plan.dec_kernel.
 *   PV_PLAN_ENC_TWO_LAUNCH  the compact fc encoder as two launches (first layer, then the rest) instead of one grid whose
 *                           second half waits on per-tile flags — e.g. for a caller that wants no in-launch hand-off;
 *   PV_PLAN_NO_SIDE_STREAM  steps with a convolutional encoder (VED, iVAE + convEncoderNet) keep every launch on the caller's
 *                           stream.  By default the weight gradients of the encoder's kernel-3 convolutions, the decoder's
 *                           batched weight gradients and their split-order reductions run on a second, low-priority stream the
 *                           library creates per device; the caller's stream waits for it before the entry point's last
 *                           launches (nothing is left running that the caller's stream does not wait for; bit-identical
 *                           either way).  Never while the stream is captured.
 * The numeric range of the convolution kernels is pv_ivae_plan.conv_wide / conv_bf16 == 2 of pv_ved_plan and
 * pv_convnet_plan: the default fp32-class form carries kernel-3 weights as two fp16 pieces of w * 64, valid for
 * 1e-6 < max|w| < 1023; the wide form uses three bf16 pieces (no range limit, same accuracy, ~1.5x the matrix time).  The
 * Python engines set it when a weight of THEIR model leaves the safe range (checked at bind time, on the first inference call
 * after a bind, and every 64 steps; Adam moves a weight by at most lr per step).
 * No reference counterpart: the reference's nn.Conv2d is plain fp32 on torch's current stream (nets/conv.py:24-60). */
#define PV_PLAN_ENC_TWO_LAUNCH 1
#define PV_PLAN_NO_SIDE_STREAM 2
/* (v15) PV_PLAN_ENC_NO_WAIT: the one-launch encoder's consumers do not poll the producers' flags at all and compute their
 *   first-layer tiles themselves (the bounded-poll fallback taken at once: bit-identical results, redundant work) — the
 *   parity test of that path, and a caller that wants no cross-workgroup hand-off but one launch.
 * PV_PLAN_NO_DEC1D (pv_ved_plan): the Conv1d decoder runs layer by layer instead of as one forward and one
 *   input-gradient launch (csrc/pv_dec1d.hip) — the launches that form replaced, kept as its parity reference.
 * PV_PLAN_NO_ENC_FOLD: keep the guide (fc encoder, sample, split, fc_latent) in its own launch even where the decoder launch could
 *   host it (round 5: at fused == 3, a batch that is a multiple of the decoder grid — one image per workgroup at batch 256 on 256
 *   CUs — every workgroup runs its images' guide in the decoder launch's prologue).  Results agree to fp32 rounding (another
 *   summation order in the encoder's matrix-vector products), not bit for bit. */
#define PV_PLAN_ENC_NO_WAIT    4
#define PV_PLAN_NO_DEC1D       8
#define PV_PLAN_NO_ENC_FOLD    16
/* (v16) PV_PLAN_ENC_TILED: keep the tiled one-launch fc encoder (first-layer tiles on the matrix cores + per-tile flags) for the
 *   guide of a training step.  By default a step of up to 512 samples with the plain two-hidden-layer fc encoder on a fused
 *   decoder path runs its guide as ONE launch with one workgroup per image (fp32 matrix-vector products straight from the
 *   L2-resident weights, no cross-workgroup hand-off: 11-13 us where the tiled form spends 18-19 us of dependent latency);
 *   PV_PLAN_ENC_TWO_LAUNCH / PV_PLAN_ENC_NO_WAIT, which are about the tiled form, imply it.  Results agree to fp32 rounding
 *   (another summation order), not bit for bit.  Matches models/ivae.py:204-221, nets/fc.py:51-61. */
#define PV_PLAN_ENC_TILED      32
/* (v15) PV_PLAN_CONV_X3 (pv_ivae_plan with a convolutional encoder; pv_ved_plan / pv_convnet_plan say it as conv_bf16 == 0): the
 *   fp32-class kernel-3 convolutions with BOTH operands as two fp16 pieces and three products per multiply-add in EVERY direction
 *   (rounds 2-4's form, 3e-7 per convolution vs float64).  Default since round 5 (conv_bf16 == 4): the FORWARD unchanged (its outputs
 *   decide max-pool winners and leaky-ReLU signs), the INPUT GRADIENT unchanged too (v16: round 5 ran it with dL/dy as ONE fp16
 *   piece, two products — an error that accumulates down the chain of input gradients to 1.6e-4 on the first layers' tensors under
 *   equal forward decisions), the WEIGHT GRADIENT with one piece per operand (one product): a rounding error that is independent
 *   from element to element, averages out of the sum over every pixel of every sample and stays in its own tensor (5 ... 8e-5;
 *   DESIGN.md section 4.3). */
#define PV_PLAN_CONV_X3        64

/* Bytes of workspace pv_ivae_* calls need for this plan (depends on batch, n_pix,
 * layer widths).  Returns < 0 on an unsupported plan. */
int64_t pv_ivae_workspace_bytes(const pv_ivae_plan* plan);
/* The same for one entry point only: the training step's layout is much smaller than the layered one pv_ivae_decode
 * uses at the same batch, so a trainer that never decodes `batch` samples at once need not pay for it. */
#define PV_WS_ALL    0
#define PV_WS_STEP   1   /* pv_ivae_loss_and_grads / pv_ivae_step / pv_ivae_guide(_backward) */
#define PV_WS_ENCODE 2   /* pv_ivae_encode */
#define PV_WS_DECODE 3   /* pv_ivae_decode */
int64_t pv_ivae_workspace_bytes_for(const pv_ivae_plan* plan, int what);

/* 1 if pv_ivae_loss_and_grads will run this plan on the fused persistent spatial-decoder
 * kernel, 0 if it will take the layer-by-layer path (plan->fused == 0 or an architecture the
 * fused kernel is not specialised for). */
int pv_ivae_uses_fused(const pv_ivae_plan* plan);

/* (v15) 1 if pv_ivae_loss_and_grads / pv_ivae_step will run this plan's guide (fc encoder, reparameterised sample, split,
 * fc_latent) INSIDE the decoder launch instead of as a launch of its own (see PV_PLAN_NO_ENC_FOLD), 0 otherwise. */
int pv_ivae_guide_folds(const pv_ivae_plan* plan);

/* Trace_ELBO.loss_and_grads for iVAE.guide + iVAE.model (models/ivae.py:165-221,
 * pyro's Trace_ELBO with one particle): writes plan->scalars and, when
 * want_grads != 0, d(loss)/d(params) into plan->grads (every entry overwritten).
 * Replaces: trainers/svi.py:107 `self.svi.step(x)` up to (not including) the
 * optimizer.  */
int pv_ivae_loss_and_grads(const pv_ivae_plan* plan, int want_grads, void* stream);

/* pyro.optim.Adam / torch.optim.Adam step over the flat buffers, then grads set
 * to zero (pyro.infer.util.zero_grads).  `n` floats starting at each pointer. */
int pv_adam_step(float* params, float* grads, float* m, float* v, int64_t n,
                 float lr, float beta1, float beta2, float eps, int32_t step, void* stream);

/* The same, plus (in the same launch) a copy of `n_scalars` floats from `scalars_src` to `scalars_dst` — the
 * data-parallel step: after the one all-reduce of [flat gradient | 4 loss scalars] (which the caller issues between
 * pv_ivae_loss_and_grads and this call) the optimizer update and the write of the reduced loss into the caller's
 * device-side loss history are one launch.  Replaces: the part of Pyro's SVI.step after loss_and_grads
 * (trainers/svi.py:107: optimizer call, zero_grads, `return loss`).  scalars_src may lie inside `grads` + n. */
int pv_adam_step_hist(float* params, float* grads, float* m, float* v, int64_t n,
                      float lr, float beta1, float beta2, float eps, int32_t step,
                      const float* scalars_src, float* scalars_dst, int32_t n_scalars, void* stream);

/* The two halves of the step around an external decoder (plan->ext_decoder, see the struct): the guide (iVAE.guide,
 * models/ivae.py:204-221, plus the prior / posterior log-densities of the sampled z) and, after the caller's decoder
 * forward + backward, the rest of the backward pass.  want_grads == 0 in the second call: loss scalars only. */
int pv_ivae_guide(const pv_ivae_plan* plan, void* stream);
int pv_ivae_guide_backward(const pv_ivae_plan* plan, int want_grads, void* stream);

/* loss_and_grads + adam in one call (single-GPU SVI.step, trainers/svi.py:107).  Same results as the two calls,
 * bit for bit; on the fused decoder path the update is applied inside the last gradient launch (one launch fewer),
 * so plan->grads holds the zeroed gradients of pyro's zero_grads afterwards.  Needs adam_m / adam_v / adam_step. */
int pv_ivae_step(const pv_ivae_plan* plan, void* stream);

/* ---- (v16) the data-parallel step with its collective INSIDE the library --------------------------------------------------
 * The reference has no distributed code.  What is sharded is trainers/svi.py:104-113 (`self.svi.step(x)` per
 * minibatch): the loss is a SUM over the data plate (models/ivae.py:177,215), so every replica computes its contiguous slice of
 * the global minibatch and the global gradient is ONE all-reduce(SUM) of [flat gradient | 4 loss scalars] (SURVEY section 8b,
 * "Collective boundary": "a direct ncclAllReduce through the same C layer"; section 8e).
 *
 * RCCL is not a link dependency.  pv_dist_load(path) resolves ncclAllReduce & co. in the RCCL shared object the CALLER's process
 * already holds (NULL / "": the loaded "librccl.so.1" / "librccl.so", else the default search path) — once per process; a second
 * call naming another library is PV_EINVAL.  The communicator (ncclComm_t, created by the caller with ncclCommInitRank of that
 * same library for the device of `stream`) is handed in as an opaque pointer; the library never creates, keeps or destroys one.
 * Every entry point only ENQUEUES on `stream` (no second stream, no event, no host synchronisation): capturable in a hipGraph. */
int pv_dist_load(const char* rccl_path);
/* the path pv_dist_load resolved ("" before a successful load) */
const char* pv_dist_library(void);
/* rank / size of a communicator (ncclCommUserRank / ncclCommCount) — lets a caller check what it hands in */
int pv_dist_comm_info(void* comm, int32_t* rank, int32_t* world);
/* in-place fp32 SUM over the communicator's ranks of buf[0, n) (ncclAllReduce(buf, buf, n, ncclFloat32, ncclSum)) on `stream` */
int pv_dist_allreduce_sum(void* comm, float* buf, int64_t n, void* stream);
/* SVI.step of one replica as ONE enqueue: pv_ivae_loss_and_grads on this rank's shard -> all-reduce(SUM) of
 * plan->grads[0, n_params + 4) -> pv_adam_step_hist (Adam, zero_grads, the REDUCED loss scalars into hist_dst — 4 floats, may be
 * NULL).  Needs plan->scalars == plan->grads + plan->n_params (the flat gradient buffer's trailing slots), the Adam fields as
 * for pv_ivae_step, no external encoder / decoder.  Every replica must hold the same parameters and call it with the same
 * adam_step; results equal the single-process step on the global minibatch up to the summation order of the shards' sums. */
int pv_ivae_dp_step(const pv_ivae_plan* plan, void* comm, float* hist_dst, void* stream);

/* baseVAE._encode inner call (models/base.py:131-135): encoder_z(x[,y]) ->
 * z_loc, z_scale (B, z_dim) each; jiVAE: also plan->alpha (B, discrete_dim) when not NULL. */
int pv_ivae_encode(const pv_ivae_plan* plan, float* z_loc, float* z_scale, void* stream);

/* baseVAE._decode inner call (models/base.py:153-170): decoder(grid', z) with the
 * grid transformed once by (angle, shift, scale); z is (B, latent_dim + c_dim).
 * loc: (B, N). */
int pv_ivae_decode(const pv_ivae_plan* plan, const float* z, float angle, float shift_x,
                   float shift_y, float scale, float* loc, void* stream);

/* ---- building blocks (also used directly by pyroved_amd.nets.*.forward) ---- */

/* Scratch bytes pv_linear_fwd / pv_linear_bwd need for a layer of these dimensions. */
int64_t pv_linear_workspace_bytes(int64_t M, int64_t K, int64_t N);

/* y[M,N] = act(x[M,K] @ w[N,K]^T + b[N])   (nn.Linear + activation; fc.py:321-323).
 * ldx / ldy: row strides in floats.  b may be NULL.  If pre != NULL the
 * pre-activation is stored there too (ldy stride) — needed by GELU's backward. */
int pv_linear_fwd(const float* x, int64_t ldx, const float* w, const float* b,
                  float* y, float* pre, int64_t ldy, int64_t M, int64_t K, int64_t N,
                  int act, void* ws, int64_t ws_bytes, void* stream);

/* Backward of the above given dpre[M,N] = dL/d(pre-activation):
 *   dx[M,K]  = (dpre @ w) * act'(xact)   (xact/xpre: output / pre-activation of the layer that
 *                                          produced x; act_prev = PV_ACT_NONE -> no factor)
 *   dw[N,K]  = dpre^T @ x,   db[N] = colsum(dpre)
 * dx, dw, db may each be NULL to skip. */
int pv_linear_bwd(const float* dpre, int64_t lddp, const float* x, int64_t ldx, const float* w,
                  float* dx, int64_t lddx, const float* xact, const float* xpre, int64_t ldxa,
                  int act_prev, float* dw, float* db, int64_t M, int64_t K, int64_t N,
                  void* ws, int64_t ws_bytes, void* stream);

/* utils/coord.py:47-88 transform_coordinates on a batch: out[b,n,:] =
 * rotate(grid[n], phi[b]) * scale[b] + shift[b,:].  phi / scale: (B) or NULL
 * (0 / 1); shift: (B, coord_dim) or NULL.  */
int pv_transform_coordinates(const float* grid, int64_t n_pix, int coord_dim, const float* phi,
                             const float* shift, const float* scale, int64_t batch, float* out,
                             void* stream);

/* ===================================================================================================
 * models.VED (models/ved.py:89-163): convolutional encoder -> z -> convolutional decoder, Bernoulli /
 * Gaussian likelihood of a target y (image-to-spectrum and the like).  The networks are described as
 * op sequences read off nets/conv.py's FeatureExtractor (conv.py:150-213) and Upsampler (conv.py:216-262).
 * Tensors cross the ABI in the reference's layout, (B, channels, *spatial) row-major; inside the library
 * activations are channels-last.  Scope: 1-D / 2-D data, kernel 3 (padding 1) and kernel 1 convolutions,
 * stride 1, 2x max-pooling, 2x nearest-neighbour (1-D, 2-D) or bilinear (2-D) upsampling, batch normalisation after the
 * activation (the reference's order).
 * =================================================================================================== */
typedef struct pv_ved_plan {
  int32_t batch;
  int32_t ndim_in, ndim_out;       /* 1 or 2                                                            */
  int32_t in_dim[2], out_dim[2];   /* spatial sizes (second entry unused for 1-D)                       */
  int32_t in_ch, out_ch;
  int32_t z_dim;
  float   beta;                    /* KL scale_factor (ved.py:133,155)                                  */
  int32_t lik;                     /* enum pv_lik                                                       */
  int32_t sigmoid_out;
  float   decoder_sig;
  int32_t n_enc_ops, n_dec_ops;
  pv_op   enc[PV_MAX_OPS];         /* encoder_z.feature_extractor.layers                                */
  pv_op   dec[PV_MAX_OPS];         /* decoder.upsampler.layers (an UpsampleBlock = UPSAMPLE2 + CONV k1)  */
  pv_layer head;                   /* features2latent.fc_latent: in = C*prod(spatial) in the torch flatten
                                      order (c, spatial), out = 2*z_dim = [mu | softplus input]         */
  pv_layer l2f;                    /* latent2features.fc: in = z_dim, out = dec_c0*prod(dec_dim0), same order */
  int32_t dec_c0, dec_dim0[2];
  int32_t bn_eval;                 /* batch-norm layers use their running statistics (module.eval(): VED.encode / decode /
                                      manifold2d switch to it and nothing switches back, models/ved.py:178,193,230)  */
  int32_t conv_bf16;               /* precision of the kernel-3 convolutions with a multiple of 32 input channels (matrix
                                      cores, split operands): 0 fp32-class — two fp16 pieces with exact power-of-two scaling,
                                      3 products (3e-7 vs float64); 1 mixed — two rounded bf16 pieces, 3 products (~2^-16 per
                                      product: gradients that are sums with heavy cancellation lose digits, measured 7e-3 on
                                      the first layer's weights); 2 (v14) fp32-class for weights outside fp16's range — three
                                      bf16 pieces; 3 (v14) throughput — ONE fp16 piece per operand (power-of-two scaled per
                                      staged tile), one product: a third of the matrix instructions, no split arithmetic; 4 (v15)
                                      fp32-class with the cheaper backward — forward as 0, input gradient 2 products (dL/dy one fp16
                                      piece), weight gradient 1 (see PV_PLAN_CONV_X3);
                                      gradients to ~1e-2, the ELBO to 1e-4 (SVItrainer(precision="bf16")).
                                      (pv_ivae_plan's convolutional encoder: fused == 3 selects 3, conv_wide 2)             */
  int32_t flags;                   /* (v14) PV_PLAN_NO_SIDE_STREAM                                                          */
  float*       params;
  float*       grads;
  float*       adam_m;
  float*       adam_v;
  int64_t      n_params;
  const float* x;         /* (B, in_ch, *in_dim)                                                        */
  const float* y;         /* (B, out_ch, *out_dim) target                                               */
  const float* eps;       /* (B, z_dim)                                                                 */
  void*        ws;
  int64_t      ws_bytes;
  float*       scalars;   /* out, 4 floats as in pv_ivae_plan                                           */
  float*       z_loc;     /* out (B, z_dim), may be NULL                                                */
  float*       z_scale;
  float*       loc;       /* out (B, out_ch, *out_dim) decoder output, may be NULL                      */
  void*        conv_ev_start;  /* (v12) measurement: as in pv_ivae_plan, for the encoder stack's heaviest k3 convolution */
  void*        conv_ev_stop;
  double*      conv_ev_flops;
} pv_ved_plan;

/* Workspace bytes for pv_ved_* calls with this plan; < 0: unsupported plan. */
int64_t pv_ved_workspace_bytes(const pv_ved_plan* plan);

/* Trace_ELBO.loss_and_grads for VED.guide + VED.model (models/ved.py:122-163): plan->scalars and, when
 * want_grads != 0, plan->grads (every entry overwritten).  Replaces trainers/svi.py:109 `self.svi.step(x, y)`
 * up to the optimizer (pv_adam_step). */
int pv_ved_loss_and_grads(const pv_ved_plan* plan, int want_grads, void* stream);

/* convEncoderNet.forward (nets/conv.py:56-64): z_loc, z_scale (B, z_dim). */
int pv_ved_encode(const pv_ved_plan* plan, float* z_loc, float* z_scale, void* stream);

/* convDecoderNet.forward (nets/conv.py:95-102): loc (B, out_ch, *out_dim) for z (B, z_dim). */
int pv_ved_decode(const pv_ved_plan* plan, const float* z, float* loc, void* stream);
/* (v16) the data-parallel VED step as one enqueue (see pv_ivae_dp_step): pv_ved_loss_and_grads -> all-reduce(SUM) of
 * plan->grads[0, n_params + 4) -> pv_adam_step_hist.  pv_ved_plan carries no optimizer fields: Adam's arrive as arguments. */
int pv_ved_dp_step(const pv_ved_plan* plan, void* comm, float lr, float beta1, float beta2, float eps,
                   int32_t adam_step, float* hist_dst, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * A stand-alone convolutional stack — nets/conv.py:150-262 FeatureExtractor.forward / Upsampler.forward outside a
 * model — as one forward and one backward call on the same op executor the VED / conv-encoder steps use.
 * Tensors are in the reference's (torch) layout: x (B, in_ch, *in_dim), out (B, C', *dims').  The forward keeps
 * every activation in the workspace; the backward must get the SAME plan / workspace, untouched in between. */
typedef struct pv_convnet_plan {
  int32_t batch, ndim;                 /* ndim 1 or 2                                                          */
  int32_t in_ch, in_dim[2];
  int32_t n_ops;
  int32_t bn_eval;                     /* batch norm on the running statistics (module.eval())                 */
  int32_t conv_bf16;                   /* 0: fp32-class x3, 1: mixed, 2: fp32-class wide, 3: one fp16 piece, 4: fp32-class, cheaper backward (as pv_ved_plan) */
  int32_t need_dx;                     /* the backward will be asked for dL/dx (set at forward time too)       */
  int32_t flags;                       /* (v15) PV_PLAN_NO_SIDE_STREAM: every launch on the caller's stream    */
  pv_op   ops[PV_MAX_OPS];
  const float* params;                 /* flat buffer the ops' offsets refer to (running statistics included)  */
  float*  grads;                       /* same layout (backward only)                                          */
  void*   ws; int64_t ws_bytes;
} pv_convnet_plan;

int64_t pv_convnet_workspace_bytes(const pv_convnet_plan* plan);
/* out_shape[0] = channels, [1..ndim] = spatial dims of the stack's output */
int pv_convnet_out_shape(const pv_convnet_plan* plan, int32_t* out_shape);
int pv_convnet_forward(const pv_convnet_plan* plan, const float* x, float* out, void* stream);
/* dout shaped like the forward's out; dx (may be null unless need_dx) shaped like x; parameter gradients in plan->grads */
int pv_convnet_backward(const pv_convnet_plan* plan, const float* x, const float* dout, float* dx, void* stream);

/* ===================================================================================================
 * Semi-supervised models (models/ssivae.py ssiVAE, models/ss_reg_ivae.py ss_reg_iVAE) trained by
 * trainers/auxsvi.py auxSVItrainer: every step is pv_ivae_loss_and_grads on encoder_z / decoder with the label
 * vector as the conditioning input y (observed; enumerated over the K classes with row_w = q(y|x) — the caller
 * lays the K*B (x, onehot(k)) rows out [k][b]; or sampled, with dy handed back), plus the label network
 * encoder_y below and the small objective kernels.  All parameters share one flat buffer (one pv_adam_step).
 * =================================================================================================== */
enum pv_mlp_out { PV_MLP_LINEAR = 0 /* fcRegressorNet, nets/fc.py:274-304 */, PV_MLP_SOFTMAX = 1 /* fcClassifierNet, fc.py:240-271 */ };
enum pv_ss_task { PV_SS_CLASSIFICATION = 0, PV_SS_REGRESSION = 1 };

typedef struct pv_mlp_plan {
  int32_t  batch, in_dim, n_layers, out_kind;
  pv_layer layers[PV_MAX_LAYERS];   /* make_fc_layers(in_dim, hidden_dim, activation)                     */
  pv_layer out;                     /* .out (act ignored)                                                 */
  float*       params;              /* flat parameters (offsets of `layers` / `out` index it)             */
  float*       grads;               /* flat gradients: the network's entries are overwritten by backward  */
  const float* x;                   /* (B, in_dim)                                                        */
  void*        ws;                  /* >= pv_mlp_workspace_bytes; holds the activations between forward and backward */
  int64_t      ws_bytes;
} pv_mlp_plan;

int64_t pv_mlp_workspace_bytes(const pv_mlp_plan* plan);
/* out (B, out.out_dim) = net(x): probabilities (PV_MLP_SOFTMAX) or the linear output. */
int pv_mlp_forward(const pv_mlp_plan* plan, float* out, void* stream);
/* Gradients of the network's parameters from dout = dloss/d(out) (B, out_dim), after pv_mlp_forward with the same
 * plan, x and workspace; `out` = that forward's result (needed for the softmax backward; may be NULL for linear). */
int pv_mlp_backward(const pv_mlp_plan* plan, const float* out, const float* dout, void* stream);

/* Enumerated-label ELBO (TraceEnum_ELBO over ssiVAE.guide's "y", auxsvi.py:73-77): alpha (B, K) = q(y|x),
 * row_elbo (K*B) ordered [k][b] from the weighted iVAE step.  loss_add[0] = sum_bk alpha (log alpha + log K)
 * (add to the step's scalars[0]); dalpha (B, K) = dloss/dalpha (may be NULL). */
int pv_ss_enum_terms(const float* alpha, const float* row_elbo, int64_t batch, int32_t n_classes, float* dalpha,
                     float* loss_add, void* stream);
/* model_aux (ssivae.py:215-228, ss_reg_ivae.py:221-234): loss[0] = -multiplier * sum_b log p(y_b | out_b) with
 * out = encoder_y(x) (class probabilities / regression means), dout = dloss/dout (may be NULL). */
int pv_ss_aux_loss(int32_t task, const float* out, const float* y, int64_t batch, int32_t dim, float multiplier,
                   float reg_sig, float* dout, float* loss, void* stream);
/* ss_reg_iVAE.guide's label sample ys = c + reg_sig * eps (ss_reg_ivae.py:196-199). */
int pv_ss_reg_sample(const float* c, const float* eps, int64_t batch, int32_t dim, float reg_sig, float* ys, void* stream);
/* The label's own ELBO terms.  Sampled label (c != NULL): loss_add[0] = -sum(log N(ys; 0, sig) - log N(ys; c, sig)),
 * dc = dy + ys / sig^2 with dy = plan->dy of the iVAE step.  Observed label (c == NULL; eps, dy, dc unused):
 * loss_add[0] = -sum log N(ys; 0, sig). */
int pv_ss_reg_terms(const float* c, const float* eps, const float* ys, const float* dy, int64_t batch, int32_t dim,
                    float reg_sig, float* dc, float* loss_add, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYROVED_AMD_H */
