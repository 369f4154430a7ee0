// pv_guide_img.hip — iVAE.guide (models/ivae.py:204-221) as ONE launch with one workgroup per image (round 6).
//
// fcEncoderNet.forward (nets/fc.py:51-61) for a minibatch of a few hundred images is 51 MFLOP: the tiled one-launch encoder
// (pv_encoder.hip: first-layer tiles on the matrix cores, consumers waiting on per-tile flags) spends 18-19 us on it, nearly all
// of it dependent latency — tile hand-offs, a K = 784 loop per tile, four dependent phases behind it.  Round 5 showed a cheaper
// form inside the throughput kernel's prologue (pv_sdec_fused_w8.hip, PvEncFold): a workgroup runs ONE image's whole guide itself —
// the layers as fp32 matrix-vector products straight from the L2-resident weights (pv_gemv16.h), the reparameterised sample with
// its sampled-KL terms (torch Normal.log_prob), _split_latent -> the transform parameters (models/base.py:97-119) and
// fc_latent(z) — with no cross-workgroup hand-off at all: 11-13 us, bound by the count of load instructions of the first layer.
// This file is that guide as a launch of its own, for every plan the fold cannot take (the fp32-class decoder kernels, batches
// that are not one image per decoder workgroup): image workgroups [0, B), then guest workgroups that write the decoder's weight
// images and clear its dL/d(hz) slots (pv_fb_layout.h: pv_fb_prep) as the tiled encoder's guests did.
// Everything the decoder and the backward launches read lands exactly where the tiled encoder put it (PvEncFold's fields); the
// values agree with it to fp32 rounding (another summation order), not bit for bit.
#include "pv_sdec_fused.h"
#include "pv_fb_layout.h"
#include "pv_gemv16.h"
#include "pv_kernels.h"

#define GI_WAVES 8
#define GI_THREADS (64 * GI_WAVES)
#define GI_P_WAVE (64 * 17)                           // floats of a wave's partial-sum transpose buffer (pv_gemv16.h)
#define GI_LOG_SQRT_2PI 0.91893853320467274178f

__global__ __launch_bounds__(GI_THREADS) void pv_guide_img_kernel(PvEncFold e, PvFbPrep prep, float hz_mul, int n_img, int has_prep) {
  __shared__ __attribute__((aligned(16))) float h1s[128];
  __shared__ __attribute__((aligned(16))) float h2s[128];
  __shared__ __attribute__((aligned(16))) float hds[64];
  __shared__ __attribute__((aligned(16))) float zs[32];
  __shared__ __attribute__((aligned(16))) float Pall[GI_WAVES * GI_P_WAVE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= n_img) {
    // guests: the decoder's weight images + the zero fill (each guest workgroup finds the normalising maxima for itself)
    if (has_prep) {
      const int64_t gb = (int64_t)blockIdx.x - n_img, ng = (int64_t)gridDim.x - n_img;
      pv_fb_prep(prep, gb * GI_THREADS + tid, ng * GI_THREADS, GI_WAVES, wave);
    }
    return;
  }
  const int64_t b = blockIdx.x;
  const int g = (int)blockIdx.x;
  float* P = Pall + wave * GI_P_WAVE;
  const int N = (int)e.ldx, zd = e.z_dim;
  const int r_ = lane & 15, q_ = lane >> 4;
  // the image: float4 columns lane + 64 c (c < 4) and w8_gemv16's packed last group
  f32x4 xr[4], xpk;
  {
    const int K4 = N >> 2;
    const float* xg = e.x + b * e.ldx;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k4 = lane + 64 * c;
      xr[c] = *reinterpret_cast<const f32x4*>(xg + 4 * (k4 < K4 ? k4 : 0));
    }
    const int kp = 64 * (((K4 + 63) >> 6) - 1) + (lane & 3);
    xpk = *reinterpret_cast<const f32x4*>(xg + 4 * (kp < K4 ? kp : 0));
  }
  // small operands of the later phases, requested up front (each would otherwise head its phase with an L2 / HBM round trip)
  const int jw = 16 * wave + r_;
  const float pb1 = (e.enc1.b_off >= 0 && jw < e.enc1.out_dim) ? e.params[e.enc1.b_off + jw] : 0.0f;
  const float pbh = (e.head.b_off >= 0 && jw < e.head.out_dim) ? e.params[e.head.b_off + jw] : 0.0f;
  const float pep = (wave == 0 && lane < zd) ? e.eps[b * zd + lane] : 0.0f;
  f32x4 wl1[8], wlh[8];
  float pwz[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (tid < FD_H) {
#pragma unroll
    for (int i = 0; i < 4; ++i) pwz[i] = i < e.lat_in ? e.Wz[(int64_t)tid * e.lat_in + i] : 0.0f;
  }
  {
    // (every workgroup reads the same first-layer matrix at the same time: which wave takes which 16 rows rotates with the
    //  workgroup index, so that the chip's requests spread over the L2 channels instead of marching through them in step)
    const int jr = 16 * ((wave + g) & (GI_WAVES - 1));
    const float v = w8_gemv16(e.params + e.enc0.w_off, N, e.enc0.out_dim, jr, xr, xpk, P, lane,
                              [&]() { w8_gemv16_k128_load(e.params + e.enc1.w_off, e.enc1.in_dim, e.enc1.out_dim, 16 * wave, lane, wl1); });
    if (16 * wave < e.head.out_dim)
      w8_gemv16_k128_load(e.params + e.head.w_off, e.head.in_dim, e.head.out_dim, 16 * wave, lane, wlh);
    const int j = jr + r_;
    if (q_ == 0 && j < e.enc0.out_dim) {
      const float y = pv_act_fwd2(v + (e.enc0.b_off >= 0 ? e.params[e.enc0.b_off + j] : 0.0f), e.enc0.act);
      h1s[j] = y;
      e.eact0[b * e.enc0.out_dim + j] = y;
    }
  }
  pv_lds_barrier();
  {
    const float v = w8_gemv16_k128(wl1, e.enc1.in_dim, h1s, P, lane);
    const int j = 16 * wave + r_;
    if (q_ == 0 && j < e.enc1.out_dim) {
      const float y = pv_act_fwd2(v + pb1, e.enc1.act);
      h2s[j] = y;
      e.eact1[b * e.enc1.out_dim + j] = y;
    }
  }
  pv_lds_barrier();
  if (16 * wave < e.head.out_dim) {                            // [mu | softplus input]: 16 rows per wave
    const float v = w8_gemv16_k128(wlh, e.head.in_dim, h2s, P, lane);
    const int j = 16 * wave + r_;
    if (q_ == 0 && j < e.head.out_dim) {
      const float y = v + pbh;
      hds[j] = y;
      e.head_out[b * e.head.out_dim + j] = y;
    }
  }
  pv_lds_barrier();
  if (wave == 0) {
    // z = mu + softplus(s) eps and the sampled-KL terms (torch Normal.log_prob), one lane per latent coordinate
    float lp = 0.0f, lq = 0.0f;
    float rc_ = 1.0f, rs_ = 0.0f;                  // cos / sin of the rotation, by the lane that holds phi (coordinate 0): one range
    if (lane < zd) {                                // reduction, next to the other lanes' log-density terms
      const float mu = hds[lane], sig = pv_softplus(hds[zd + lane]);
      const float z = mu + sig * pep;
      if (lane == 0 && e.coord_dim == 2 && e.has_r) sincosf(z, &rs_, &rc_);
      e.z[b * zd + lane] = z;
      e.z_scale[b * zd + lane] = sig;
      if (e.z_loc_out) e.z_loc_out[b * zd + lane] = mu;
      if (e.z_scale_out) e.z_scale_out[b * zd + lane] = sig;
      const float d = z - mu;
      lq = -(d * d) / (2.0f * (sig * sig)) - logf(sig) - GI_LOG_SQRT_2PI;
      lp = -(z * z) / 2.0f - GI_LOG_SQRT_2PI;
      zs[lane] = z;
    }
    lp = pv_wave_sum(lp);
    lq = pv_wave_sum(lq);
    if (lane == 0) {
      e.kl_part[2 * b] = e.beta * lp;
      e.kl_part[2 * b + 1] = e.beta * lq;
      // _split_latent -> the transform parameters (models/base.py:97-119; the t / s priors of models/ivae.py:187-191)
      int idx = 0;
      float c = 1.0f, sn = 0.0f, sc = 1.0f, tx = 0.0f, ty = 0.0f;
      if (e.coord_dim == 1) {
        if (e.has_t) { tx = zs[0] * e.tp0; idx = 1; }
      } else if (e.coord_dim == 2) {
        if (e.has_r) { ++idx; c = rc_; sn = rs_; }
        if (e.has_t) { tx = zs[idx] * e.tp0; ty = zs[idx + 1] * e.tp1; idx += 2; }
        if (e.has_s) { sc = 1.0f + e.sc_prior * zs[idx++]; }
      }
      float* t = e.tp + b * 8;
      t[0] = c; t[1] = sn; t[2] = sc; t[3] = tx; t[4] = ty;
    }
  }
  pv_lds_barrier();
  if (tid < FD_H) {
    // hz = fc_latent(z content) (nets/fc.py:217,230: no bias), times what the decoder kernel wants it multiplied by
    int coord = 0;
    if (e.coord_dim == 1) coord = e.has_t ? 1 : 0;
    else if (e.coord_dim == 2) coord = e.has_r + 2 * e.has_t + e.has_s;
    const float* wz = e.Wz + (int64_t)tid * e.lat_in;
    float v = 0.0f;
    for (int i = 0; i < e.lat_in; ++i) v += zs[coord + i] * (i < 4 ? pwz[i] : wz[i]);
    e.hz[b * FD_H + tid] = v * hz_mul;
  }
}

// batches the per-image form is taken for (pv_plan.hip): every image streams the first layer's matrix from L2 once more, so the
// tiled encoder wins back what its latency costs somewhere above a few images per CU
bool pv_guide_img_ok(const PvEncFold& e, int B) {
  return B >= 1 && B <= PV_GUIDE_IMG_MAX_BATCH && e.ldx % 4 == 0 && e.ldx <= 1024 && e.enc0.out_dim <= 128 && e.enc1.out_dim <= 128 &&
         e.enc1.in_dim % 4 == 0 && e.head.in_dim % 4 == 0 && e.head.out_dim <= 64 && e.z_dim <= 16 && e.lat_in <= 16;
}

int pv_guide_img_launch(const PvEncFold& e, const PvFbPrep* prep, float hz_mul, int B, hipStream_t s) {
  if (!pv_guide_img_ok(e, B)) return PV_EINVAL;
  // guests: 8 workgroups split the two 128 x 128 matrices' image rows, a few more the zero fill when it is long
  int guests = 0;
  if (prep) {
    guests = 8 + (int)(prep->nzero4 / (16 * GI_THREADS));
    if (guests > 64) guests = 64;
  }
  const PvFbPrep pz = prep ? *prep : PvFbPrep{};
  hipLaunchKernelGGL(pv_guide_img_kernel, dim3(B + guests), dim3(GI_THREADS), 0, s, e, pz, hz_mul == 0.0f ? 1.0f : hz_mul, B,
                     prep ? 1 : 0);
  PV_LAUNCH_CHECK();
  return 0;
}
