// pv_elementwise.hip — the non-GEMM kernels of the iVAE SVI step (all HBM-/latency-bound):
//   head_fwd      Normal.rsample + log q(z|x) + log p(z) + _split_latent      (ivae.py:179-189,217-221; base.py:97-119)
//   coordlat_fwd  transform_coordinates fused into coord_latent's K=2 layer   (coord.py:47-88; fc.py:220-237)
//   out_lik       decoder.out + sigmoid + likelihood log_prob + d/dlogit      (fc.py:196; prob.py:25-29)
//   coordlat_bwd  backward of coordlat_fwd (dWc, dhz, d(phi,shift,scale))
//   head_bwd      backward of head_fwd
//   adam          torch.optim.Adam single-tensor update + zero_grads
// Reductions are tree/ordered (no float atomics): results are bit-reproducible run to run.
#include "pv_common.h"
#include "pv_kernels.h"
#include "pv_wgrad_small.h"
#include "pv_side.h"
#include "pv_sdec_fused.h"

#define LOG_SQRT_2PI 0.91893853320467274178f
#define BERN_EPS 1.1920928955078125e-07f   // torch.finfo(float32).eps used by clamp_probs

// deterministic block-wide sum (blockDim.x == 256); result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* sm /* >= 4 floats */) {
  v = pv_wave_sum(v);
  pv_lds_barrier();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  pv_lds_barrier();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// ---------------------------------------------------------------------------------------------
// head_fwd: one workgroup; B*z_dim elements.
__global__ __launch_bounds__(256) void pv_head_fwd_kernel(PvHead h) {
  __shared__ float sm[4];
  const int total = h.B * h.z_dim;
  float lp = 0.0f, lq = 0.0f;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int b = e / h.z_dim, i = e % h.z_dim;
    const int ldh = h.ldh > 0 ? h.ldh : 2 * h.z_dim;
    const float mu = h.head[(int64_t)b * ldh + i];
    const float sp = h.head[(int64_t)b * ldh + h.z_dim + i];
    const float sig = h.scale_direct ? sp : pv_softplus(sp);
    const float ep = h.eps[e];
    const float z = mu + sig * ep;
    h.z[e] = z;
    h.z_scale[e] = sig;
    if (h.z_loc_out) h.z_loc_out[e] = mu;
    if (h.z_scale_out) h.z_scale_out[e] = sig;
    const float d = z - mu;
    const float wb = h.w ? h.w[b] : 1.0f;
    // torch.distributions.Normal.log_prob
    lq += wb * (-(d * d) / (2.0f * (sig * sig)) - logf(sig) - LOG_SQRT_2PI);
    lp += wb * (-(z * z) / 2.0f - LOG_SQRT_2PI);
  }
  lp = block_sum_256(lp, sm);
  lq = block_sum_256(lq, sm);
  if (threadIdx.x == 0) {
    h.scalars[2] = h.beta * lp;
    h.scalars[3] = h.beta * lq;
  }
  __syncthreads();
  // per-sample transform parameters (base.py:97-119, ivae.py:187-189) and decoder latent input
  for (int b = threadIdx.x; b < h.B; b += 256) {
    const float* zb = h.z + (int64_t)b * h.z_dim;
    int idx = 0;
    float c = 1.0f, s = 0.0f, sc = 1.0f, tx = 0.0f, ty = 0.0f;
    if (h.coord_dim == 1) {
      if (h.has_t) { tx = zb[0] * h.tp0; idx = 1; }
    } else if (h.coord_dim == 2) {
      if (h.has_r) { const float phi = zb[idx++]; c = cosf(phi); s = sinf(phi); }
      if (h.has_t) { tx = zb[idx] * h.tp0; ty = zb[idx + 1] * h.tp1; idx += 2; }
      if (h.has_s) { sc = 1.0f + h.sc_prior * zb[idx++]; }
    }
    if (h.tp) {
      float* t = h.tp + (int64_t)b * 8;
      t[0] = c; t[1] = s; t[2] = sc; t[3] = tx; t[4] = ty;
    }
    if (h.zy) {   // cat([z_content, y]) (ivae.py:194-195)
      const int L = h.z_dim - idx;
      float* o = h.zy + (int64_t)b * (L + h.c_dim);
      for (int i = 0; i < L; ++i) o[i] = zb[idx + i];
      for (int i = 0; i < h.c_dim; ++i) o[L + i] = h.y[(int64_t)b * h.c_dim + i];
    }
  }
}

int pv_head_fwd(const PvHead& h, hipStream_t s) {
  if (h.ch_part || h.hz || h.kl_part) return PV_EINVAL;   // (pv_head_fwd_blocks's fields)
  hipLaunchKernelGGL(pv_head_fwd_kernel, dim3(1), dim3(256), 0, s, h);
  PV_LAUNCH_CHECK();
  return 0;
}

// The same per-sample work 16 samples per workgroup, with what stands before and after it on a conv encoder's path in the same
// launch: the conv head's partial sums (pv_convhead_fwd_finish_kernel's order), then head_fwd, then fc_latent
// (pv_smallk_linear_kernel's sum).  The KL sums leave as per-workgroup partials (kl_part), like the compact encoder's.
#define HB_ROWS 16
// (round 5: the launch is a chain of dependent round trips — it ran 13 us for 128 samples.  The conv head's partial sums are requested
//  eight at a time (the same additions in the same order), the noise before the head exists, and the head / z values a later
//  phase of the same workgroup needs pass through LDS instead of being read back from global memory.)
__global__ __launch_bounds__(256) void pv_head_fwd_blocks_kernel(PvHead h) {
  __shared__ float sm[16];
  __shared__ float s_head[HB_ROWS * 64];            // [row][<= 64 head outputs] when the conv head is finished here (ch_out <= 64)
  __shared__ float s_z[HB_ROWS * 32];               // [row][<= 32 latent coordinates]
  const int b0 = (int)blockIdx.x * HB_ROWS, nb = min(HB_ROWS, h.B - b0), t = threadIdx.x;
  const int ldh = h.ldh > 0 ? h.ldh : 2 * h.z_dim;
  const bool lds_head = h.ch_part && h.ch_out <= 64 && h.head == h.head_w, lds_z = h.z_dim <= 32;
  // the first sample element's noise: requested in front of the partial sums' round trips
  float eps0 = 0.0f;
  if (t < nb * h.z_dim) eps0 = h.eps[(int64_t)b0 * h.z_dim + t];
  if (h.ch_part) {
    for (int e = t; e < nb * h.ch_out; e += 256) {
      const int r = e / h.ch_out, b = b0 + r, j = e % h.ch_out;
      float v = h.ch_bias ? h.ch_bias[j] : 0.0f;
      const float* pp = h.ch_part + (int64_t)b * h.ch_nseg * h.ch_out + j;
      int k = 0;
      for (; k + 8 <= h.ch_nseg; k += 8) {
        float p8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p8[u] = pp[(int64_t)(k + u) * h.ch_out];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += p8[u];
      }
      for (; k < h.ch_nseg; ++k) v += pp[(int64_t)k * h.ch_out];
      h.head_w[(int64_t)b * h.ch_out + j] = v;
      if (lds_head) s_head[r * 64 + j] = v;
    }
    __syncthreads();
  }
  float lp = 0.0f, lq = 0.0f;
  for (int el = t; el < nb * h.z_dim; el += 256) {
    const int r = el / h.z_dim, b = b0 + r, i = el % h.z_dim;
    const int64_t e = (int64_t)b * h.z_dim + i;
    const float mu = lds_head ? s_head[r * 64 + i] : h.head[(int64_t)b * ldh + i];
    const float sp = lds_head ? s_head[r * 64 + h.z_dim + i] : h.head[(int64_t)b * ldh + h.z_dim + i];
    const float sig = h.scale_direct ? sp : pv_softplus(sp);
    const float z = mu + sig * (el == t ? eps0 : h.eps[e]);
    if (lds_z) s_z[r * 32 + i] = z;
    h.z[e] = z;
    h.z_scale[e] = sig;
    if (h.z_loc_out) h.z_loc_out[e] = mu;
    if (h.z_scale_out) h.z_scale_out[e] = sig;
    const float d = z - mu;
    const float wb = h.w ? h.w[b] : 1.0f;
    lq += wb * (-(d * d) / (2.0f * (sig * sig)) - logf(sig) - LOG_SQRT_2PI);
    lp += wb * (-(z * z) / 2.0f - LOG_SQRT_2PI);
  }
  lp = pv_block_sum(lp, sm);
  lq = pv_block_sum(lq, sm);
  if (t == 0) { h.kl_part[2 * blockIdx.x] = h.beta * lp; h.kl_part[2 * blockIdx.x + 1] = h.beta * lq; }
  __syncthreads();
  if (t < nb) {
    const int b = b0 + t;
    const float* zb = lds_z ? s_z + t * 32 : h.z + (int64_t)b * h.z_dim;
    int idx = 0;
    float c = 1.0f, s = 0.0f, sc = 1.0f, tx = 0.0f, ty = 0.0f;
    if (h.coord_dim == 1) {
      if (h.has_t) { tx = zb[0] * h.tp0; idx = 1; }
    } else if (h.coord_dim == 2) {
      if (h.has_r) { const float phi = zb[idx++]; c = cosf(phi); s = sinf(phi); }
      if (h.has_t) { tx = zb[idx] * h.tp0; ty = zb[idx + 1] * h.tp1; idx += 2; }
      if (h.has_s) { sc = 1.0f + h.sc_prior * zb[idx++]; }
    }
    if (h.tp) {
      float* tpb = h.tp + (int64_t)b * 8;
      tpb[0] = c; tpb[1] = s; tpb[2] = sc; tpb[3] = tx; tpb[4] = ty;
    }
    if (h.zy) {
      const int L = h.z_dim - idx;
      float* o = h.zy + (int64_t)b * (L + h.c_dim);
      for (int i = 0; i < L; ++i) o[i] = zb[idx + i];
      for (int i = 0; i < h.c_dim; ++i) o[L + i] = h.y[(int64_t)b * h.c_dim + i];
    }
  }
  if (h.hz) {
    __syncthreads();
    // (the decoder's latent input is a slice of z where no class vector is concatenated: from LDS then)
    const int64_t zo = h.zin - h.z;
    const bool z_lds = lds_z && zo >= 0 && zo + h.lat_in <= h.z_dim && h.ldz == h.z_dim;
    for (int e = t; e < nb * h.H; e += 256) {
      const int r = e / h.H, b = b0 + r, j = e % h.H;
      float v = 0.0f;
      if (z_lds) {
        for (int k = 0; k < h.lat_in; ++k) v = fmaf(s_z[r * 32 + (int)zo + k], h.Wz[(int64_t)j * h.lat_in + k], v);
      } else {
        for (int k = 0; k < h.lat_in; ++k) v = fmaf(h.zin[(int64_t)b * h.ldz + k], h.Wz[(int64_t)j * h.lat_in + k], v);
      }
      h.hz[(int64_t)b * h.H + j] = v;
    }
  }
}

int pv_head_fwd_blocks(const PvHead& h, hipStream_t s) {
  if (!h.kl_part || h.B < 1 || (h.ch_part && (!h.head_w || h.ch_out != (h.ldh > 0 ? h.ldh : 2 * h.z_dim))) ||
      (h.hz && (h.lat_in < 1 || h.lat_in > 16))) return PV_EINVAL;
  hipLaunchKernelGGL(pv_head_fwd_blocks_kernel, dim3((unsigned)((h.B + HB_ROWS - 1) / HB_ROWS)), dim3(256), 0, s, h);
  PV_LAUNCH_CHECK();
  return 0;
}

// fills identical transform parameters for every sample (baseVAE._decode: angle/shift/scale kwargs)
__global__ void pv_fill_tp_kernel(float* tp, int B, float c, float s, float sc, float tx, float ty) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) {
    float* t = tp + (int64_t)b * 8;
    t[0] = c; t[1] = s; t[2] = sc; t[3] = tx; t[4] = ty;
  }
}

int pv_fill_tp(float* tp, int B, float angle, float sc, float tx, float ty, hipStream_t s) {
  hipLaunchKernelGGL(pv_fill_tp_kernel, dim3((B + 255) / 256), dim3(256), 0, s, tp, B, cosf(angle), sinf(angle), sc,
                     tx, ty);
  PV_LAUNCH_CHECK();
  return 0;
}

// out[b, :] = cat(a[b, :na], y[b, :nb])
__global__ void pv_concat_kernel(const float* a, int64_t lda, int na, const float* y, int64_t ldy, int nb, float* out,
                                 int64_t B) {
  const int w = na + nb;
  const int64_t total = B * w;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / w;
    const int i = (int)(e % w);
    out[e] = i < na ? a[b * lda + i] : y[b * ldy + (i - na)];
  }
}

int pv_concat(const float* a, int64_t lda, int na, const float* y, int64_t ldy, int nb, float* out, int64_t B,
              hipStream_t s) {
  const int64_t total = B * (na + nb);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) return 0;
  hipLaunchKernelGGL(pv_concat_kernel, dim3(blocks), dim3(256), 0, s, a, lda, na, y, ldy, nb, out, B);
  PV_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// transformed coordinate of flattened row (b, n):  x' = (grid[n] . R(phi_b)) * s_b + shift_b
__device__ __forceinline__ void pv_xprime(const float* __restrict__ grid, int cd, int n, const float* __restrict__ t,
                                          float& x0, float& x1, float& u0, float& u1) {
  if (cd == 2) {
    const float gx = grid[2 * n], gy = grid[2 * n + 1];
    u0 = gx * t[0] - gy * t[1];      // coord.py:71-74: [x y] @ [[c, s], [-s, c]]
    u1 = gx * t[1] + gy * t[0];
    x0 = u0 * t[2] + t[3];
    x1 = u1 * t[2] + t[4];
  } else {
    u0 = grid[n]; u1 = 0.0f;
    x0 = u0 + t[3]; x1 = 0.0f;      // coord.py:56-57: 1-D grids only translate
  }
}

#define CL_ROWS 32
__global__ __launch_bounds__(256) void pv_coordlat_fwd_kernel(PvCoordLat p) {
  __shared__ float xs[CL_ROWS][2];
  __shared__ int bs[CL_ROWS];
  const int64_t r0 = (int64_t)blockIdx.x * CL_ROWS;
  const int t = threadIdx.x;
  if (t < CL_ROWS) {
    const int64_t row = r0 + t;
    if (row < p.M) {
      const int b = (int)(row / p.N), n = (int)(row % p.N);
      float x0, x1, u0, u1;
      pv_xprime(p.grid, p.cd, n, p.tp + (int64_t)b * 8, x0, x1, u0, u1);
      xs[t][0] = x0; xs[t][1] = x1; bs[t] = b;
    }
  }
  __syncthreads();
  const int H = p.H0;
  for (int e = t; e < CL_ROWS * H; e += 256) {
    const int r = e / H, j = e % H;
    const int64_t row = r0 + r;
    if (row >= p.M) break;
    float v = p.Wc[j * p.cd] * xs[r][0];
    if (p.cd == 2) v += p.Wc[j * 2 + 1] * xs[r][1];
    v += p.bc[j];
    v += p.hz[(int64_t)bs[r] * H + j];
    p.h0[row * H + j] = tanhf(v);      // coord_latent's activation is hard-wired tanh (fc.py:218)
  }
}

int pv_coordlat_fwd(const PvCoordLat& p, hipStream_t s) {
  const int64_t blocks = (p.M + CL_ROWS - 1) / CL_ROWS;
  hipLaunchKernelGGL(pv_coordlat_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
  PV_LAUNCH_CHECK();
  return 0;
}

// standalone utils.transform_coordinates (coord.py:47-60)
__global__ void pv_transform_kernel(const float* grid, int64_t N, int cd, const float* phi, const float* shift,
                                    const float* scale, int64_t B, float* out) {
  const int64_t total = B * N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / N;
    const int n = (int)(e % N);
    if (cd == 1) {
      out[e] = grid[n] + (shift ? shift[b] : 0.0f);
    } else {
      const float ph = phi ? phi[b] : 0.0f, sc = scale ? scale[b] : 1.0f;
      const float c = cosf(ph), s = sinf(ph);
      const float gx = grid[2 * n], gy = grid[2 * n + 1];
      out[2 * e] = (gx * c - gy * s) * sc + (shift ? shift[2 * b] : 0.0f);
      out[2 * e + 1] = (gx * s + gy * c) * sc + (shift ? shift[2 * b + 1] : 0.0f);
    }
  }
}

extern "C" int pv_transform_coordinates(const float* grid, int64_t n_pix, int coord_dim, const float* phi,
                                        const float* shift, const float* scale, int64_t batch, float* out,
                                        void* stream) {
  if (coord_dim != 1 && coord_dim != 2) return PV_EINVAL;
  const int64_t total = batch * n_pix;
  if (total <= 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pv_transform_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grid, n_pix, coord_dim, phi,
                     shift, scale, batch, out);
  PV_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// out_lik: a = h . wo + bo ; likelihood ; dL/da ; dpre of the last hidden layer ; partial dwo/dbo
#define OL_ROWS 64
#define OL_MAXJ 8     // supports hidden widths up to 512
__global__ __launch_bounds__(256) void pv_out_lik_kernel(PvOutLik p) {
  __shared__ float red[4][64 * OL_MAXJ];
  __shared__ float redb[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = p.H;
  float wo[OL_MAXJ], acc[OL_MAXJ];
#pragma unroll
  for (int jj = 0; jj < OL_MAXJ; ++jj) {
    const int j = lane + 64 * jj;
    wo[jj] = j < H ? p.wo[j] : 0.0f;
    acc[jj] = 0.0f;
  }
  const float bo = p.bo ? p.bo[0] : 0.0f;
  float accb = 0.0f;
  const int64_t rbeg = (int64_t)blockIdx.x * OL_ROWS + wave * (OL_ROWS / 4);
  for (int i = 0; i < OL_ROWS / 4; ++i) {
    const int64_t row = rbeg + i;
    if (row >= p.M) break;
    const float* hr = p.h + row * p.ldh;
    float hv[OL_MAXJ];
    float dot = 0.0f;
#pragma unroll
    for (int jj = 0; jj < OL_MAXJ; ++jj) {
      const int j = lane + 64 * jj;
      hv[jj] = j < H ? hr[j] : 0.0f;
      dot += hv[jj] * wo[jj];
    }
    const float a = pv_wave_sum(dot) + bo;
    const float x = p.x[p.xmod > 0 ? row % p.xmod : row];
    float ll, dlda, locv;
    if (p.lik == PV_LIK_BERNOULLI) {
      // torch.distributions.Bernoulli(probs=sigmoid(a), validate_args=False).log_prob(x):
      //   probs -> clamp_probs -> logits = log(p) - log1p(-p) -> -BCEWithLogits(logits, x)
      const float pr = 1.0f / (1.0f + expf(-a));
      const float pc = fminf(fmaxf(pr, BERN_EPS), 1.0f - BERN_EPS);
      const float lg = logf(pc) - log1pf(-pc);
      ll = -(fmaxf(lg, 0.0f) - lg * x + log1pf(expf(-fabsf(lg))));
      const float mask = (pr >= BERN_EPS && pr <= 1.0f - BERN_EPS) ? 1.0f : 0.0f;   // clamp's gradient
      dlda = (1.0f / (1.0f + expf(-lg)) - x) * mask;
      locv = pr;
    } else if (p.lik == PV_LIK_CBERNOULLI) {
      pv_cbern(a, x, ll, dlda, locv);
    } else {
      const float pr = p.sigmoid_out ? 1.0f / (1.0f + expf(-a)) : a;
      const float d = x - pr;
      ll = -(d * d) / (2.0f * p.sig * p.sig) - logf(p.sig) - LOG_SQRT_2PI;
      dlda = -d / (p.sig * p.sig) * (p.sigmoid_out ? pr * (1.0f - pr) : 1.0f);
      locv = pr;
    }
    if (lane == 0) {
      if (p.llrow) p.llrow[row] = ll;
      if (p.loc) p.loc[row] = locv;
    }
    if (p.sw) dlda *= p.sw[row / p.N];
    if (p.dpre) {
      float* dr = p.dpre + row * p.ldh;
      const float* pr_ = p.hpre ? p.hpre + row * p.ldh : nullptr;
#pragma unroll
      for (int jj = 0; jj < OL_MAXJ; ++jj) {
        const int j = lane + 64 * jj;
        if (j < H) {
          dr[j] = dlda * wo[jj] * pv_act_grad2(hv[jj], pr_ ? pr_[j] : 0.0f, p.act_last);
          acc[jj] += dlda * hv[jj];
        }
      }
      accb += dlda;
    }
  }
  if (!p.dpre) return;
#pragma unroll
  for (int jj = 0; jj < OL_MAXJ; ++jj) red[wave][lane + 64 * jj] = acc[jj];
  if (lane == 0) redb[wave] = accb;
  __syncthreads();
  for (int j = threadIdx.x; j < H; j += 256)
    p.part_dwo[(int64_t)blockIdx.x * H + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
  if (threadIdx.x == 0) p.part_dbo[blockIdx.x] = (redb[0] + redb[1]) + (redb[2] + redb[3]);
}

int64_t pv_out_lik_blocks(int64_t M) { return (M + OL_ROWS - 1) / OL_ROWS; }

int pv_out_lik(const PvOutLik& p, hipStream_t s) {
  if (p.H > 64 * OL_MAXJ) return PV_EINVAL;
  hipLaunchKernelGGL(pv_out_lik_kernel, dim3((unsigned)pv_out_lik_blocks(p.M)), dim3(256), 0, s, p);
  PV_LAUNCH_CHECK();
  return 0;
}

// per-sample sums: out[b] = sum_n v[b*N + n]; then scalars (one workgroup per sample, tree-ordered)
__global__ __launch_bounds__(256) void pv_segsum_kernel(const float* __restrict__ v, int64_t N, float* __restrict__ out) {
  __shared__ float sm[4];
  const float* vb = v + (int64_t)blockIdx.x * N;
  float a = 0.0f;
  for (int64_t n = threadIdx.x; n < N; n += 256) a += vb[n];
  a = block_sum_256(a, sm);
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}

int pv_segsum(const float* v, int64_t nseg, int64_t N, float* out, hipStream_t s) {
  hipLaunchKernelGGL(pv_segsum_kernel, dim3((unsigned)nseg), dim3(256), 0, s, v, N, out);
  PV_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void pv_finish_scalars_kernel(const float* __restrict__ llb, int B, float* scalars,
                                                                const float* __restrict__ kl_part, int n_part,
                                                                float beta) {
  __shared__ float sm[16];
  pv_finish_scalars_block(llb, B, scalars, kl_part, n_part, beta, sm);
}

int pv_finish_scalars(const float* llb, int B, float* scalars, const float* kl_part, int n_part, float beta,
                      hipStream_t s) {
  hipLaunchKernelGGL(pv_finish_scalars_kernel, dim3(1), dim3(256), 0, s, llb, B, scalars, kl_part, n_part, beta);
  PV_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// coordlat_bwd: given dpre0[M,H0] (= dL/d(pre-tanh of coord_latent)), per (sample, row-chunk):
//   part_hz[b,c,j]   = sum_rows dpre0[row,j]                      -> dhz (and dbc)
//   part_wc[b,c,j,k] = sum_rows dpre0[row,j] * x'[row,k]          -> dWc
//   part_tp[b,c,0:4] = sum_rows (dphi, dscale, dtx, dty)          -> d(latent coordinates)
#define CB_MAXJ 8
__global__ __launch_bounds__(256) void pv_coordlat_bwd_kernel(PvCoordLatBwd p) {
  __shared__ float red[4][64 * CB_MAXJ * 3];
  __shared__ float redt[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, c = blockIdx.x, H = p.H0;
  const int n_beg = c * p.rows_per_chunk;
  const int n_end = min(p.N, n_beg + p.rows_per_chunk);
  const float* t = p.tp + (int64_t)b * 8;
  float w0[CB_MAXJ], w1[CB_MAXJ], sj[CB_MAXJ], sx[CB_MAXJ], sy[CB_MAXJ];
#pragma unroll
  for (int jj = 0; jj < CB_MAXJ; ++jj) {
    const int j = lane + 64 * jj;
    w0[jj] = j < H ? p.Wc[j * p.cd] : 0.0f;
    w1[jj] = (j < H && p.cd == 2) ? p.Wc[j * 2 + 1] : 0.0f;
    sj[jj] = sx[jj] = sy[jj] = 0.0f;
  }
  float dphi = 0.0f, dsc = 0.0f, dtx = 0.0f, dty = 0.0f;
  for (int n = n_beg + wave; n < n_end; n += 4) {
    const int64_t row = (int64_t)b * p.N + n;
    float x0, x1, u0, u1;
    pv_xprime(p.grid, p.cd, n, t, x0, x1, u0, u1);
    const float* dr = p.dpre0 + row * H;
    float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
    for (int jj = 0; jj < CB_MAXJ; ++jj) {
      const int j = lane + 64 * jj;
      const float v = j < H ? dr[j] : 0.0f;
      sj[jj] += v; sx[jj] += v * x0; sy[jj] += v * x1;
      d0 += v * w0[jj]; d1 += v * w1[jj];
    }
    d0 = pv_wave_sum(d0);
    d1 = pv_wave_sum(d1);
    // x'0 = s*u0 + tx, x'1 = s*u1 + ty ; du0/dphi = -u1, du1/dphi = u0
    dphi += t[2] * (d1 * u0 - d0 * u1);
    dsc += d0 * u0 + d1 * u1;
    dtx += d0; dty += d1;
  }
#pragma unroll
  for (int jj = 0; jj < CB_MAXJ; ++jj) {
    red[wave][(lane + 64 * jj) * 3 + 0] = sj[jj];
    red[wave][(lane + 64 * jj) * 3 + 1] = sx[jj];
    red[wave][(lane + 64 * jj) * 3 + 2] = sy[jj];
  }
  if (lane == 0) { redt[wave][0] = dphi; redt[wave][1] = dsc; redt[wave][2] = dtx; redt[wave][3] = dty; }
  __syncthreads();
  const int64_t pc = (int64_t)b * gridDim.x + c;
  for (int j = threadIdx.x; j < H; j += 256) {
    float v[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) v[q] = (red[0][j * 3 + q] + red[1][j * 3 + q]) + (red[2][j * 3 + q] + red[3][j * 3 + q]);
    p.part_hz[pc * H + j] = v[0];
    p.part_wc[(pc * H + j) * p.cd] = v[1];
    if (p.cd == 2) p.part_wc[(pc * H + j) * 2 + 1] = v[2];
  }
  if (threadIdx.x < 4)
    p.part_tp[pc * 4 + threadIdx.x] =
        (redt[0][threadIdx.x] + redt[1][threadIdx.x]) + (redt[2][threadIdx.x] + redt[3][threadIdx.x]);
}

int pv_coordlat_bwd(const PvCoordLatBwd& p, int nchunk, int B, hipStream_t s) {
  if (p.H0 > 64 * CB_MAXJ) return PV_EINVAL;
  hipLaunchKernelGGL(pv_coordlat_bwd_kernel, dim3(nchunk, B), dim3(256), 0, s, p);
  PV_LAUNCH_CHECK();
  return 0;
}

// out[b, i] = sum_c part[(b*nc + c)*n + i]  (ordered)
__global__ void pv_reduce_mid_kernel(const float* __restrict__ part, int nb, int nc, int n, float* __restrict__ out) {
  const int64_t total = (int64_t)nb * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / n;
    const int i = (int)(e % n);
    float v = 0.0f;
    for (int c = 0; c < nc; ++c) v += part[(b * nc + c) * n + i];
    out[e] = v;
  }
}

int pv_reduce_mid(const float* part, int nb, int nc, int n, float* out, hipStream_t s) {
  const int64_t total = (int64_t)nb * n;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) return 0;
  hipLaunchKernelGGL(pv_reduce_mid_kernel, dim3(blocks), dim3(256), 0, s, part, nb, nc, n, out);
  PV_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ void pv_head_bwd_kernel(PvHeadBwd h) {
  const int total = h.B * h.z_dim;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int b = e / h.z_dim, i = e % h.z_dim;
  pv_head_bwd_elem(
      h, b, i, [&](int c) { return h.dtp[(int64_t)b * h.dtp_sb + (int64_t)c * h.dtp_sc]; },
      [&](int k) { return h.dzc[(int64_t)b * h.ldzc + k]; });
}

int pv_head_bwd(const PvHeadBwd& h, hipStream_t s) {
  const int total = h.B * h.z_dim;
  hipLaunchKernelGGL(pv_head_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, h);
  PV_LAUNCH_CHECK();
  return 0;
}

// alpha[b][:] = softmax(logits[b][:])  (jfcEncoderNet.forward, nets/fc.py:106) for pv_ivae_encode
__global__ void pv_softmax_rows_kernel(const float* __restrict__ logits, int64_t ld, int B, int K, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* l = logits + (int64_t)b * ld;
  float mx = l[0];
  for (int k = 1; k < K; ++k) mx = fmaxf(mx, l[k]);
  float s = 0.0f;
  for (int k = 0; k < K; ++k) s += expf(l[k] - mx);
  for (int k = 0; k < K; ++k) out[(int64_t)b * K + k] = expf(l[k] - mx) / s;
}
int pv_softmax_rows(const float* logits, int64_t ld, int B, int K, float* out, hipStream_t s) {
  hipLaunchKernelGGL(pv_softmax_rows_kernel, dim3((B + 63) / 64), dim3(64), 0, s, logits, ld, B, K, out);
  PV_LAUNCH_CHECK();
  return 0;
}

// jiVAE with the vanilla decoder (layer-by-layer path): per input b, from the K enumerated decoder passes [k][b]:
//   llb[b] = sum_k alpha_bk ll_kb ; dzc[b][:] += sum_{k>=1} dzc[(k,b)][:] (first n_content columns; in place in block
//   k = 0) ; dhead[b][2z + k] = softmax backward of dloss/dalpha_bk = -(ll_kb - b1 log K - b1 log alpha_bk - b1)
__global__ void pv_jiv_combine_kernel(const float* __restrict__ llkb, const float* __restrict__ alpha, float* __restrict__ llb,
                                      float* __restrict__ dzc, int ld_dzc, int n_content, float* __restrict__ dhead, int ldh,
                                      int z_dim, int B, int K, float beta_disc, int want_grads, float* __restrict__ dtp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* al = alpha + (int64_t)b * K;
  float ll = 0.0f;
  for (int k = 0; k < K; ++k) ll += al[k] * llkb[(int64_t)k * B + b];
  llb[b] = ll;
  if (!want_grads) return;
  for (int i = 0; i < n_content; ++i) {
    float v = dzc[(int64_t)b * ld_dzc + i];
    for (int k = 1; k < K; ++k) v += dzc[((int64_t)k * B + b) * ld_dzc + i];
    dzc[(int64_t)b * ld_dzc + i] = v;
  }
  if (dtp) {                                            // transform-parameter gradients of the K passes (4 per sample)
    for (int c = 0; c < 4; ++c) {
      float v = dtp[(int64_t)b * 4 + c];
      for (int k = 1; k < K; ++k) v += dtp[((int64_t)k * B + b) * 4 + c];
      dtp[(int64_t)b * 4 + c] = v;
    }
  }
  const float lK = logf((float)K);
  float dot = 0.0f;
  for (int k = 0; k < K; ++k)
    dot += al[k] * -(llkb[(int64_t)k * B + b] - beta_disc * lK - beta_disc * logf(al[k]) - beta_disc);
  for (int k = 0; k < K; ++k) {
    const float da = -(llkb[(int64_t)k * B + b] - beta_disc * lK - beta_disc * logf(al[k]) - beta_disc);
    dhead[(int64_t)b * ldh + 2 * z_dim + k] = al[k] * (da - dot);
  }
}
int pv_jiv_combine(const float* llkb, const float* alpha, float* llb, float* dzc, int ld_dzc, int n_content, float* dhead,
                   int ldh, int z_dim, int B, int K, float beta_disc, int want_grads, hipStream_t s, float* dtp) {
  hipLaunchKernelGGL(pv_jiv_combine_kernel, dim3((B + 63) / 64), dim3(64), 0, s, llkb, alpha, llb, dzc, ld_dzc, n_content,
                     dhead, ldh, z_dim, B, K, beta_disc, want_grads, dtp);
  PV_LAUNCH_CHECK();
  return 0;
}

// ---- jiVAE without enumeration (plan->class_onehot): the decoder still runs on the K*B rows [k][b] of the enumerated
// path, weighted by onehot(y_b) instead of alpha_b (rows of the classes not drawn carry weight 0).
// prep: sw <- onehot; fix[0] <- beta_disc * sum_b (log alpha_b[y_b] - sum_k alpha_bk log alpha_bk): what the drawn class
// changes in the guide's discrete log-probability sum relative to the enumerated expectation the encoder kernels wrote
__global__ __launch_bounds__(256) void pv_jiv_sampled_prep_kernel(const float* __restrict__ alpha, const float* __restrict__ onehot,
                                                                  float* __restrict__ sw, float* __restrict__ fix, float beta_disc,
                                                                  int B, int K) {
  __shared__ float sm[4];
  float d = 0.0f;
  for (int b = threadIdx.x; b < B; b += 256) {
    float samp = 0.0f, ent = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float a = alpha[(int64_t)b * K + k], o = onehot[(int64_t)b * K + k], la = logf(a);
      sw[(int64_t)k * B + b] = o;
      samp += o * la;
      ent += a * la;
    }
    d += samp - ent;
  }
  d = block_sum_256(d, sm);
  if (threadIdx.x == 0) fix[0] = beta_disc * d;
}
int pv_jiv_sampled_prep(const float* alpha, const float* onehot, float* sw, float* fix, float beta_disc, int B, int K,
                        hipStream_t s) {
  hipLaunchKernelGGL(pv_jiv_sampled_prep_kernel, dim3(1), dim3(256), 0, s, alpha, onehot, sw, fix, beta_disc, B, K);
  PV_LAUNCH_CHECK();
  return 0;
}
// scalars after pv_finish_scalars: the guide's log-probability sum and the loss take the correction
__global__ void pv_jiv_sampled_fix_kernel(float* scalars, const float* fix) {
  if (threadIdx.x == 0) { scalars[3] += fix[0]; scalars[0] += fix[0]; }
}
int pv_jiv_sampled_fix(float* scalars, const float* fix, hipStream_t s) {
  hipLaunchKernelGGL(pv_jiv_sampled_fix_kernel, dim3(1), dim3(64), 0, s, scalars, fix);
  PV_LAUNCH_CHECK();
  return 0;
}
// combine for the drawn class: llb[b] = ll of the drawn pass; dzc summed over the passes (only the drawn one is
// non-zero); class logits: dhead[b][2z + k] = -log_r_b (onehot_bk - alpha_bk) with
// log_r_b = ll_b + b0 (log p(z_b) - log q(z_b)) + b1 (log(1/K) - log alpha_b[y_b])      (Trace_ELBO's score-function term)
__global__ void pv_jiv_combine_sampled_kernel(const float* __restrict__ llkb, const float* __restrict__ alpha,
                                              const float* __restrict__ onehot, float* __restrict__ llb,
                                              float* __restrict__ dzc, int ld_dzc, int n_content, float* __restrict__ dhead,
                                              int ldh, int z_dim, int B, int K, float beta, float beta_disc, int want_grads,
                                              const float* __restrict__ z, const float* __restrict__ head,
                                              const float* __restrict__ z_scale) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* al = alpha + (int64_t)b * K;
  const float* oh = onehot + (int64_t)b * K;
  float ll = 0.0f, lqd = 0.0f;
  for (int k = 0; k < K; ++k) { ll += oh[k] * llkb[(int64_t)k * B + b]; lqd += oh[k] * logf(al[k]); }
  llb[b] = ll;
  if (!want_grads) return;
  for (int i = 0; i < n_content; ++i) {
    float v = dzc[(int64_t)b * ld_dzc + i];
    for (int k = 1; k < K; ++k) v += dzc[((int64_t)k * B + b) * ld_dzc + i];
    dzc[(int64_t)b * ld_dzc + i] = v;
  }
  float kl = 0.0f;                                     // log p(z_b) - log q(z_b | x_b)
  for (int i = 0; i < z_dim; ++i) {
    const float zz = z[(int64_t)b * z_dim + i], mu = head[(int64_t)b * ldh + i], sig = z_scale[(int64_t)b * z_dim + i];
    const float d = zz - mu;
    const float lq = -(d * d) / (2.0f * (sig * sig)) - logf(sig) - LOG_SQRT_2PI;
    const float lp = -(zz * zz) / 2.0f - LOG_SQRT_2PI;
    kl += lp - lq;
  }
  const float log_r = ll + beta * kl + beta_disc * (-logf((float)K) - lqd);
  for (int k = 0; k < K; ++k) dhead[(int64_t)b * ldh + 2 * z_dim + k] = -log_r * (oh[k] - al[k]);
}
int pv_jiv_combine_sampled(const float* llkb, const float* alpha, const float* onehot, float* llb, float* dzc, int ld_dzc,
                           int n_content, float* dhead, int ldh, int z_dim, int B, int K, float beta, float beta_disc,
                           int want_grads, const float* z, const float* head, const float* z_scale, hipStream_t s) {
  hipLaunchKernelGGL(pv_jiv_combine_sampled_kernel, dim3((B + 63) / 64), dim3(64), 0, s, llkb, alpha, onehot, llb, dzc, ld_dzc,
                     n_content, dhead, ldh, z_dim, B, K, beta, beta_disc, want_grads, z, head, z_scale);
  PV_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void pv_jiv_expand_kernel(const float* __restrict__ head, int ldh, const float* __restrict__ z,
                                                            int z_dim, int n_content, float* __restrict__ tp,
                                                            float* __restrict__ zy, float* __restrict__ alpha,
                                                            float* __restrict__ sw, float* scalars, float beta_disc, int B, int K) {
  __shared__ float sm[4];
  float lqd = 0.0f, lpd = 0.0f;
  const int lat_in = n_content + K, coord = z_dim - n_content;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* lg = head + (int64_t)b * ldh + 2 * z_dim;
    float mx = lg[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, lg[k]);
    float se = 0.0f;
    for (int k = 0; k < K; ++k) se += expf(lg[k] - mx);
    const float lse = logf(se);
    for (int k = 0; k < K; ++k) {
      const float la = lg[k] - mx - lse, a = expf(la);
      alpha[(int64_t)b * K + k] = a;
      sw[(int64_t)k * B + b] = a;
      lqd += a * la;
      const int64_t srow = (int64_t)k * B + b;
      if (tp && k > 0)
        for (int c = 0; c < 8; ++c) tp[srow * 8 + c] = tp[(int64_t)b * 8 + c];
      float* o = zy + srow * lat_in;
      for (int i = 0; i < n_content; ++i) o[i] = z[(int64_t)b * z_dim + coord + i];
      for (int i = 0; i < K; ++i) o[n_content + i] = i == k ? 1.0f : 0.0f;
    }
    lpd += -logf((float)K);
  }
  lqd = block_sum_256(lqd, sm);
  lpd = block_sum_256(lpd, sm);
  if (threadIdx.x == 0) { scalars[2] += beta_disc * lpd; scalars[3] += beta_disc * lqd; }
}
int pv_jiv_expand(const float* head, int ldh, const float* z, int z_dim, int n_content, float* tp, float* zy, float* alpha,
                  float* sw, float* scalars, float beta_disc, int B, int K, hipStream_t s) {
  hipLaunchKernelGGL(pv_jiv_expand_kernel, dim3(1), dim3(256), 0, s, head, ldh, z, z_dim, n_content, tp, zy, alpha, sw, scalars,
                     beta_disc, B, K);
  PV_LAUNCH_CHECK();
  return 0;
}
// v[row][:] *= w[row]
__global__ void pv_scale_rows_kernel(float* __restrict__ v, const float* __restrict__ w, int64_t rows, int64_t N) {
  const int64_t total = rows * N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    v[e] *= w[e / N];
}
int pv_scale_rows(float* v, const float* w, int64_t rows, int64_t N, hipStream_t s) {
  int64_t blocks = (rows * N + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) return 0;
  hipLaunchKernelGGL(pv_scale_rows_kernel, dim3((int)blocks), dim3(256), 0, s, v, w, rows, N);
  PV_LAUNCH_CHECK();
  return 0;
}

__global__ void pv_row_elbo_kernel(const float* __restrict__ row_ll, const float* __restrict__ z, const float* __restrict__ head,
                                   const float* __restrict__ z_scale, int B, int zd, int ldh, float beta, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float kl = 0.0f;
  for (int i = 0; i < zd; ++i) {
    const float zz = z[(int64_t)b * zd + i], mu = head[(int64_t)b * ldh + i], sig = z_scale[(int64_t)b * zd + i];
    const float d = zz - mu;
    const float lq = -(d * d) / (2.0f * (sig * sig)) - logf(sig) - LOG_SQRT_2PI;
    const float lp = -(zz * zz) / 2.0f - LOG_SQRT_2PI;
    kl += lp - lq;
  }
  out[b] = row_ll[b] + beta * kl;
}
int pv_row_elbo(const float* row_ll, const float* z, const float* head, const float* z_scale, int B, int z_dim, int ldh,
                float beta, float* out, hipStream_t s) {
  hipLaunchKernelGGL(pv_row_elbo_kernel, dim3((B + 63) / 64), dim3(64), 0, s, row_ll, z, head, z_scale, B, z_dim, ldh, beta,
                     out);
  PV_LAUNCH_CHECK();
  return 0;
}
__global__ void pv_add_cols_kernel(float* __restrict__ dst, int64_t ldd, const float* __restrict__ src, int64_t lds, int64_t B,
                                   int n) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * n) return;
  const int64_t b = e / n;
  const int i = (int)(e % n);
  dst[b * ldd + i] += src[b * lds + i];
}
int pv_add_cols(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t B, int n, hipStream_t s) {
  hipLaunchKernelGGL(pv_add_cols_kernel, dim3((unsigned)((B * n + 255) / 256)), dim3(256), 0, s, dst, ldd, src, lds, B, n);
  PV_LAUNCH_CHECK();
  return 0;
}

// latent_bwd (fused decoder path): one workgroup per sample gathers everything that flows from the decoder
// kernel back into that sample's latent code: ll_b and d(phi, scale, tx, ty) (sums over the sample's N rows),
// dL/d(hz[b]) (sum of the workgroup partials), dL/d(z content) = dhz Wz, then head_bwd.  Fixed-order sums.
// phase-timing trace (profiling builds only: -DLB_TRACE): shader-clock stamps of sample 0's workgroup, thread 0
#ifdef LB_TRACE
__device__ long long lb_trace[32];
#define LB_STAMP(k) do { if (b == 0 && threadIdx.x == 0) lb_trace[(k)] = (long long)__builtin_readcyclecounter(); } while (0)
extern "C" int pv_debug_read_trace_lb(long long* out, int n) {
  if (n > 32) n = 32;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lb_trace), n * sizeof(long long));
}
#else
#define LB_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void pv_latent_bwd_block(const PvLatentBwd& p, int b) {
  __shared__ float sm[4];
  __shared__ float sh_dhz[512];
  __shared__ float sh_dzc[64];
  __shared__ float sh_tp[4];
  __shared__ float sh_ll[128];
  __shared__ float sm5[5][4];
  __shared__ float sh_dh[256];          // the sample's dhead row (2*z_dim + K <= 128 + 64 on the compact encoder path)
  __shared__ __attribute__((aligned(16))) float sh_e[2][128];        // encoder dgrad chain (layer widths <= 128)
  __shared__ float sh_p[2][128];
  const int t = threadIdx.x;
  // The plan struct sits in the kernarg segment and the compiler fetches its fields where they are first used: five or six DEPENDENT
  // scalar-load round trips (~700 cycles each, cold scalar cache) before the first row load was issued.  Asking for everything the
  // first phase and the chain's requests need in ONE place makes them one batch of s_loads behind one wait.
  // (the rest of the struct — one field per 64-byte line is enough — rides in the same batch: later fetches then hit the scalar cache)
  asm volatile("" :: "s"(p.llrow), "s"(p.rowtp), "s"(p.part_hz), "s"(p.M), "s"(p.N), "s"(p.kmax), "s"(p.H), "s"(p.K), "s"(p.hb.B),
               "s"(p.fwd_only), "s"(p.enc_n), "s"(p.enc_params), "s"(p.lat_in), "s"(p.dhz), "s"(p.hb.z), "s"(p.hb.dhead),
               "s"(p.alpha), "s"(p.enc_l[0].w_off), "s"(p.enc_l[2].w_off), "s"(p.enc_head.w_off), "s"(p.enc_act[0]),
               "s"(p.enc_act[4]), "s"(p.enc_dp[0]), "s"(p.enc_dp[4]));
  const int K = p.K > 0 ? p.K : 1, Bq = p.hb.B;
  // Operands of the encoder chain that depend on nothing computed here (its last layer's weight column, the head's, the
  // saved activations) are requested EARLY — right after the first pass's row loads have been issued (loads return in
  // order: ahead of them, the row sums would wait on 74 cold weight loads) — so that their memory latency (the weights
  // were last touched a decoder-kernel ago) hides under the reductions instead of heading two dependent phases.
  f32x4 wv4[16];
  float whd[16];
  float pf_act0 = 0.0f, pf_act1 = 0.0f;
  const int ck = t & 127, chalf = t >> 7;
  const bool chain = p.enc_n > 1 && !p.fwd_only;
  auto prefetch_chain = [&]() {
    const pv_layer l = p.enc_l[p.enc_n - 1], lp = p.enc_l[p.enc_n - 2], hd = p.enc_head;
    const int jh = l.out_dim >> 1, j0 = chalf * jh;      // widths are multiples of 16 (pv_enc_compact_supported)
    if (ck < l.in_dim) {
      // (rows past the half are never used — the contraction below stops at jh, a multiple of 8 — so their loads are CLAMPED to the
      //  last row instead of guarded: a guard per load compiled into 64 scalar branches, ~2.5 k cycles of this launch's critical path)
      const float* wc = p.enc_params + l.w_off + ck + (int64_t)j0 * l.in_dim;
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) wv4[u][i] = wc[(4 * u + i < jh ? 4 * u + i : jh - 1) * l.in_dim];
      if (chalf == 0) pf_act0 = p.enc_act[p.enc_n - 2][(int64_t)b * lp.out_dim + ck];
    }
    if (t < l.out_dim) {
      pf_act1 = p.enc_act[p.enc_n - 1][(int64_t)b * l.out_dim + t];
      const float* Wh = p.enc_params + hd.w_off + t;
#pragma unroll
      for (int o = 0; o < 16; ++o) whd[o] = Wh[(o < hd.out_dim ? o : hd.out_dim - 1) * hd.in_dim];   // (clamped: used below only for o < out_dim)
    }
  };
  float tp_acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  LB_STAMP(0);
  for (int j = t; j < p.H; j += 256) sh_dhz[j] = 0.0f;
  for (int k = 0; k < K; ++k) {
    const int64_t s = (int64_t)k * Bq + b;            // decoder sample (k, b)
    const int64_t r0 = s * p.N;
    float a[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (p.part_rs) {
      // (round 6) the decoder launch already summed its rows per (sample, workgroup, wave): add the sample's kmax slots in ascending
      // order — thread c < 5 takes quantity c, all its slot loads in flight at once — instead of loading and block-reducing 5 x N
      // rows (7.7 k of this workgroup's 21.7 k cycles: 3.2 k until the row loads were even issued; scripts/gpu_trace_lb.py)
      float v = 0.0f;
      if (t < 5) {
        const float* pr = p.part_rs + (s * p.kmax) * 8 + t;
        for (int kk0 = 0; kk0 < p.kmax; kk0 += 16) {
          float w[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) w[u] = pr[(int64_t)(kk0 + u < p.kmax ? kk0 + u : p.kmax - 1) * 8];
#pragma unroll
          for (int u = 0; u < 16; ++u) v += kk0 + u < p.kmax ? w[u] : 0.0f;
        }
      }
      LB_STAMP(8);
      if (k == 0 && chain) prefetch_chain();
      LB_STAMP(9);
      LB_STAMP(7);
      if (t < 5) sm5[t][0] = v;
      pv_lds_barrier();
#pragma unroll
      for (int c = 0; c < 5; ++c) a[c] = sm5[c][0];
      pv_lds_barrier();                                 // (sm5 is rewritten by the next class pass)
    } else {
    {
      // four rows per thread and array in flight at once.  The rows come in chunks of 1024 (thread t: rows t + 256 u of a chunk); every
      // chunk but the last is whole, the LAST one (the only one at 784 rows) is predicated and its loads are all issued — with the
      // chain's operand requests behind them — before anything is added: a thread with all four rows in range used to wait for them in
      // the loop above before the rest were even requested (~1.9 k cycles on the wave the block reduction waits for).  Same sums.
      int n = t;
      const int n_last = ((p.N - 1) >> 10) << 10;       // first row of the last chunk
      for (; n < n_last; n += 1024) {
        float v[4][5];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[u][0] = p.llrow[r0 + n + 256 * u];
#pragma unroll
          for (int c = 0; c < 4; ++c) v[u][1 + c] = p.fwd_only ? 0.0f : p.rowtp[(int64_t)c * p.M + r0 + n + 256 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < 5; ++c) a[c] += v[u][c];
      }
      float v[4][5];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = n + 256 * u < p.N;
        v[u][0] = ok ? p.llrow[r0 + n + 256 * u] : 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[u][1 + c] = (ok && !p.fwd_only) ? p.rowtp[(int64_t)c * p.M + r0 + n + 256 * u] : 0.0f;
      }
      LB_STAMP(8);
      if (k == 0 && chain) prefetch_chain();           // (behind the row loads in the memory pipeline)
      LB_STAMP(9);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 5; ++c) a[c] += v[u][c];
    }
    LB_STAMP(7);
    // the five row sums in ONE block reduction (fixed order: wave sums, then waves 0..3)
#pragma unroll
    for (int c = 0; c < 5; ++c) a[c] = pv_wave_sum(a[c]);
    pv_lds_barrier();
    if ((t & 63) == 0) {
#pragma unroll
      for (int c = 0; c < 5; ++c) sm5[c][t >> 6] = a[c];
    }
    pv_lds_barrier();
#pragma unroll
    for (int c = 0; c < 5; ++c) a[c] = (sm5[c][0] + sm5[c][1]) + (sm5[c][2] + sm5[c][3]);
    }
    if (t == 0) sh_ll[k] = a[0];
    LB_STAMP(1);
    if (p.fwd_only) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) tp_acc[c] += a[1 + c];
    for (int j = t; j < p.H; j += 256) {
      // ascending slot order (fixed); four independent chains keep the loads in flight
      // (the same four chains in the same order — slot kk < (kmax & ~3) to chain kk & 3, the rest to chain 0 — but 16 slots requested
      //  at once, past-the-end ones clamped and added as zeros: the rolled loops waited for memory once per group of four and once per
      //  leftover slot — three dependent round trips at the usual nine slots, ~2.4 k cycles of this launch's critical path)
      if (p.dhz_ready) { sh_dhz[j] += p.dhz[s * p.H + j]; continue; }      // (the hosting decoder launch summed its waves' partials itself)
      const float* ph = p.part_hz + (s * p.kmax) * p.H + j;
      float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
      const int k4 = p.kmax & ~3;
      for (int kk0 = 0; kk0 < p.kmax; kk0 += 16) {
        float w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = ph[(int64_t)(kk0 + u < p.kmax ? kk0 + u : p.kmax - 1) * p.H];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int kk = kk0 + u;
          const float x = kk < p.kmax ? w[u] : 0.0f;
          const bool tail = kk >= k4;
          if ((u & 3) == 0) v0 += x;
          else if ((u & 3) == 1) { v1 += tail ? 0.0f : x; v0 += tail ? x : 0.0f; }
          else if ((u & 3) == 2) { v2 += tail ? 0.0f : x; v0 += tail ? x : 0.0f; }
          else { v3 += tail ? 0.0f : x; v0 += tail ? x : 0.0f; }
        }
      }
      const float v = (v0 + v1) + (v2 + v3);
      p.dhz[s * p.H + j] = v;
      sh_dhz[j] += v;                                  // the same thread owns j in every pass
    }
  }
  LB_STAMP(2);
  pv_lds_barrier();
  if (t == 0) {
    float ll = sh_ll[0];
    if (p.K > 0) {
      ll = 0.0f;
      for (int k = 0; k < K; ++k) ll += p.alpha[(int64_t)b * K + k] * sh_ll[k];
    }
    if (p.row_ll) p.row_ll[b] = ll;
    p.llb[b] = p.hb.w ? p.hb.w[b] * ll : ll;
    for (int c = 0; c < 4; ++c) sh_tp[c] = tp_acc[c];
  }
  if (p.fwd_only) return;
  const int n_content = p.lat_in - (p.K > 0 ? p.K : 0);        // the one-hot class columns carry no gradient to z
  if (p.dzc_in) {                                     // (the hosting decoder launch contracted dL/d(hz) with Wz itself)
    if (t < n_content) sh_dzc[t] = p.dzc_in[(int64_t)b * p.lat_in + t];
  } else {
    for (int i = 0; i < n_content; ++i) {
      float v = 0.0f;
      for (int j = t; j < p.H; j += 256) v += sh_dhz[j] * p.Wz[j * p.lat_in + i];
      v = block_sum_256(v, sm);
      if (t == 0) sh_dzc[i] = v;
    }
  }
  pv_lds_barrier();
  LB_STAMP(3);
  if (p.dzc_out && t < n_content) p.dzc_out[(int64_t)b * p.lat_in + t] = sh_dzc[t];
  if (t < p.hb.z_dim)
    pv_head_bwd_elem(p.hb, b, t, [&](int c) { return sh_tp[c]; }, [&](int k) { return sh_dzc[k]; },
                     p.enc_n > 0 ? sh_dh : nullptr);
  if (p.K > 0 && t == 0) {
    // loss = -sum_k alpha_k (ll_k + b1 log(1/K) - b1 log alpha_k):  dloss/dalpha_k = -(ll_k - b1 log K - b1 log alpha_k - b1)
    // then softmax backward: dlogit_k = alpha_k (dalpha_k - sum_j alpha_j dalpha_j)
    const float b1 = p.beta_disc, lK = logf((float)K);
    const float* al = p.alpha + (int64_t)b * K;
    float dot = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float da = -(sh_ll[k] - b1 * lK - b1 * logf(al[k]) - b1);
      dot += al[k] * da;
    }
    const int ldh = p.hb.ldh > 0 ? p.hb.ldh : 2 * p.hb.z_dim;
    for (int k = 0; k < K; ++k) {
      const float da = -(sh_ll[k] - b1 * lK - b1 * logf(al[k]) - b1);
      p.hb.dhead[(int64_t)b * ldh + 2 * p.hb.z_dim + k] = al[k] * (da - dot);
      if (p.enc_n > 0) sh_dh[2 * p.hb.z_dim + k] = al[k] * (da - dot);
    }
  }
  LB_STAMP(4);
  if (p.enc_n <= 0) return;
  // ---- the sample's encoder dgrad chain (fixed summation order: ascending j) ----
  pv_lds_barrier();
  const int ne = p.enc_n;
  int cur = 0;
  {
    const pv_layer hd = p.enc_head, ll_ = p.enc_l[ne - 1];
    const float* Wh = p.enc_params + hd.w_off;
    for (int k = t; k < ll_.out_dim; k += 256) {
      float v = 0.0f;
      if (chain && k == t) {
#pragma unroll
        for (int o = 0; o < 16; ++o) v += o < hd.out_dim ? sh_dh[o] * whd[o] : 0.0f;
        for (int o = 16; o < hd.out_dim; ++o) v += sh_dh[o] * Wh[(int64_t)o * hd.in_dim + k];
      } else {
        for (int o = 0; o < hd.out_dim; ++o) v += sh_dh[o] * Wh[(int64_t)o * hd.in_dim + k];
      }
      v *= pv_act_grad2((chain && k == t) ? pf_act1 : p.enc_act[ne - 1][(int64_t)b * ll_.out_dim + k], 0.0f, ll_.act);
      p.enc_dp[ne - 1][(int64_t)b * ll_.out_dim + k] = v;
      sh_e[cur][k] = v;
    }
    pv_lds_barrier();
  }
  LB_STAMP(5);
  for (int li = ne - 1; li > 0; --li) {
    const pv_layer l = p.enc_l[li], lp = p.enc_l[li - 1];
    const float* W = p.enc_params + l.w_off;
    // thread (k = t & 127, half = t >> 7) sums its half of the j range (coalesced across k); halves combined in fixed
    // order.  Widths <= 128 (pv_enc_compact_supported).  The last layer's operands were requested at the top.
    const bool pre = li == ne - 1;
    const int k = ck, half = chalf;
    const int jh = l.out_dim >> 1, j0 = half * jh;
    float v = 0.0f;
    if (k < l.in_dim) {
      if (!pre) {
        const float* wc = W + k;
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) wv4[u][i] = (4 * u + i < jh) ? wc[(int64_t)(j0 + 4 * u + i) * l.in_dim] : 0.0f;
      }
      f32x4 acc4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (4 * u < jh) acc4 += *reinterpret_cast<const f32x4*>(&sh_e[cur][j0 + 4 * u]) * wv4[u];
      v = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    }
    sh_p[half][k] = v;
    pv_lds_barrier();
    if (half == 0 && k < l.in_dim) {
      float y = sh_p[0][k] + sh_p[1][k];
      const float hv = pre ? pf_act0 : p.enc_act[li - 1][(int64_t)b * lp.out_dim + k];
      y *= pv_act_grad2(hv, 0.0f, lp.act);
      p.enc_dp[li - 1][(int64_t)b * lp.out_dim + k] = y;
      sh_e[cur ^ 1][k] = y;
    }
    pv_lds_barrier();
    cur ^= 1;
  }
  LB_STAMP(6);
}

__global__ __launch_bounds__(256) void pv_latent_bwd_kernel(PvLatentBwd p) { pv_latent_bwd_block(p, blockIdx.x); }

// one launch for the two consumers of the fused decoder kernel's outputs, which do not depend on each other:
// workgroups [0, PV_FUSED_REDUCE_BLOCKS) sum the per-workgroup gradient records, the rest run latent_bwd
__global__ __launch_bounds__(256) void pv_latent_bwd_reduce_kernel(PvLatentBwd p, const float* __restrict__ part,
                                                                   int G_, float* __restrict__ Gr, PvFusedOffsets o,
                                                                   int cd, int fmt) {
  __shared__ f32x4 smr[4][64];
#ifndef LB_EXP
#define LB_EXP 0                 // timing experiments (wrong results): 1 no record sums, 2 no latent backward
#endif
  const int nred = pv_fused_reduce_blocks(fmt);
  if ((int)blockIdx.x < nred) { if (!(LB_EXP & 1)) pv_sdec_fused_reduce_block(part, G_, Gr, o, cd, 0, blockIdx.x, smr, fmt); }
  else if (!(LB_EXP & 2)) pv_latent_bwd_block(p, blockIdx.x - nred);
}

int pv_latent_bwd_reduce(const PvLatentBwd& p, const float* part, int grid, float* G, const PvFusedOffsets& o, int cd,
                         hipStream_t s, int fmt) {
  if (p.H > 512 || p.lat_in > 64 + (p.K > 0 ? p.K : 0) || p.hb.z_dim > 256 || p.K > 128) return PV_EINVAL;
  // (a conv encoder's head weight gradient forks off this launch onto the side stream: it carries the fork event when one is armed)
  PV_LAUNCH_FORK(pv_latent_bwd_reduce_kernel, dim3(pv_fused_reduce_blocks(fmt) + p.hb.B), dim3(256), 0, s, p, part, grid, G, o, cd, fmt);
  PV_LAUNCH_CHECK();
  return 0;
}

// ---- the step's closing launch when nothing per-sample is left to do (round 6, third cut) ------------------------------------------
// Where the decoder launch hosts the guide AND every sample's latent backward + encoder chain (one image per workgroup:
// pv_sdec_fused_w8.hip, PvEncFold::chain), what remains of the step is three kinds of workgroups that need nothing from each other:
//   blocks [0, nred)          : the record sums — with Adam applied to the decoder parameters they finalise;
//   blocks [nred, + tiles)    : the small weight gradients (encoder layers, head, fc_latent), Adam in their epilogues (pv_wgrad_small.h);
//   last block                : the loss scalars.
// One launch (9-10 us) where the step ran [record sums | latent backward + chain] (9.9 us) and then the weight gradients (6.8 us).
// (A first form kept the chain in this launch, the tiles waiting for an arrival counter behind agent-scope stores: 23.2 us against
//  the two launches' 17.1 — at two workgroups per CU the waiting tiles get no slot before the producers are done, and 990
//  workgroups take ~4 us just to dispatch: profiles/r06j_tail_launch.txt.)
__global__ __launch_bounds__(256) void pv_rec_wgrad_kernel(const float* __restrict__ part, int G_, float* __restrict__ Gr, PvFusedOffsets o,
                                                           int cd, int fmt, PvWgradSmall w, PvRecAdam ra, int rwx) {
  // (rwx: timing experiments of the experiments build — 1 no record sums, 2 no tiles: wrong results; 0 in the shipped build)
  __shared__ f32x4 smr[4][64];
  __shared__ float wpart[WG_WAVES][16][17];
  __shared__ float wrpart[WG_WAVES][16];
  const int nred = pv_fused_reduce_blocks(fmt), id = (int)blockIdx.x;
  if (id < nred) { if (!(rwx & 1)) pv_sdec_fused_reduce_block(part, G_, Gr, o, cd, 0, id, smr, fmt, (rwx & 4) ? PvRecAdam{} : ra); return; }
  if (rwx & 2) return;
  pv_wgrad_small_block(w, id - nred, (int)gridDim.x - nred, wpart, wrpart);
}

// adam: every parameter must be finalised by a record block or a tile (no Adam guests here: the caller checked the coverage)
int pv_rec_wgrad(const float* part, int grid, float* G, const PvFusedOffsets& o, int cd, int fmt, const PvGemm* gs, int n,
                 const PvAdamFuse* adam, const PvFinishArgs* fin, hipStream_t s, unsigned* tick) {
  PvWgradSmall w;
  int guests = 0;
  const int tiles = pv_wgrad_small_fill(w, gs, n, adam, fin, &guests);
  if (tiles < 0) return tiles;
  const int fin_blocks = (fin && fin->scalars) ? 1 : 0;
  if (tick && !fin_blocks) return PV_EINVAL;           // (the loss block carries the increment)
  w.tick = tick;
  PvRecAdam ra{};
  if (adam) { ra.a = *adam; ra.on = 1; }
  static const int rwx = pv_exp_int("PV_RW_EXP", 0);
  hipLaunchKernelGGL(pv_rec_wgrad_kernel, dim3(pv_fused_reduce_blocks(fmt) + tiles + fin_blocks), dim3(256), 0, s, part, grid, G, o, cd,
                     fmt, w, ra, rwx);
  PV_LAUNCH_CHECK();
  return 0;
}

int pv_latent_bwd(const PvLatentBwd& p, hipStream_t s) {
  if (p.H > 512 || p.lat_in > 64 + (p.K > 0 ? p.K : 0) || p.hb.z_dim > 256 || p.K > 128) return PV_EINVAL;
  hipLaunchKernelGGL(pv_latent_bwd_kernel, dim3(p.hb.B), dim3(256), 0, s, p);
  PV_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// torch.optim.Adam (single-tensor form, no amsgrad / weight decay) + zero_grads
__global__ void pv_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                               float* __restrict__ v, int64_t n, float b1, float b2, float eps, float step_size,
                               float bc2_sqrt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    pv_adam_update(p, g, m, v, i, g[i], b1, b2, eps, step_size, bc2_sqrt);
  }
}

__global__ void pv_adam_hist_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                    float* __restrict__ v, int64_t n, float b1, float b2, float eps, float step_size,
                                    float bc2_sqrt, const float* __restrict__ ssrc, float* __restrict__ sdst, int ns) {
  if (blockIdx.x == 0 && (int)threadIdx.x < ns) sdst[threadIdx.x] = ssrc[threadIdx.x];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    pv_adam_update(p, g, m, v, i, g[i], b1, b2, eps, step_size, bc2_sqrt);
  }
}

extern "C" int pv_adam_step_hist(float* params, float* grads, float* m, float* v, int64_t n, float lr, float beta1,
                                 float beta2, float eps, int32_t step, const float* scalars_src, float* scalars_dst,
                                 int32_t n_scalars, void* stream) {
  PV_RANGE("pv_adam_step_hist");
  if (step < 1 || n < 0 || n_scalars < 0 || n_scalars > 256 || (n_scalars > 0 && (!scalars_src || !scalars_dst)))
    return PV_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(pv_adam_hist_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n, beta1,
                     beta2, eps, (float)((double)lr / bc1), (float)sqrt(bc2), scalars_src, scalars_dst, (int)n_scalars);
  PV_LAUNCH_CHECK();
  return 0;
}

extern "C" int pv_adam_step(float* params, float* grads, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, int32_t step, void* stream) {
  PV_RANGE("pv_adam_step");
  if (n <= 0) return 0;
  if (step < 1) return PV_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pv_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n, beta1,
                     beta2, eps, step_size, bc2_sqrt);
  PV_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// elementwise likelihood for the vanilla fcDecoderNet path (fc.py:143-152): a[M] logits -> loc, ll, dL/da
__global__ void pv_lik_elem_kernel(const float* __restrict__ a, const float* __restrict__ x, int64_t M, int lik,
                                   int sigmoid_out, float sig, float* __restrict__ loc, float* __restrict__ llrow,
                                   float* __restrict__ dlda) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < M; e += (int64_t)gridDim.x * blockDim.x) {
    float ll, d, lv;
    pv_lik_one(a[e], x[e], lik, sigmoid_out, sig, ll, d, lv);
    if (loc) loc[e] = lv;
    if (llrow) llrow[e] = ll;
    if (dlda) dlda[e] = d;
  }
}

// the same per element plus the per-sample sum (pv_segsum's order: thread t adds elements t, t + 256, ...): one workgroup
// per sample, ll_b[b] = sum over the sample's `per` elements — pv_lik_elem + pv_segsum in one launch, bit-identical
__global__ __launch_bounds__(256) void pv_lik_rows_kernel(const float* __restrict__ a, const float* __restrict__ x, int64_t per,
                                                          int lik, int sigmoid_out, float sig, float* __restrict__ loc,
                                                          float* __restrict__ dlda, float* __restrict__ llb) {
  __shared__ float sm[4];
  const int64_t base = (int64_t)blockIdx.x * per;
  float acc = 0.0f;
  for (int64_t n = threadIdx.x; n < per; n += 256) {
    float ll, d, lv;
    pv_lik_one(a[base + n], x[base + n], lik, sigmoid_out, sig, ll, d, lv);
    if (loc) loc[base + n] = lv;
    if (dlda) dlda[base + n] = d;
    acc += ll;
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) llb[blockIdx.x] = acc;
}

int pv_lik_rows(const float* a, const float* x, int64_t B, int64_t per, int lik, int sigmoid_out, float sig, float* loc,
                float* dlda, float* llb, hipStream_t s) {
  if (B < 1) return 0;
  hipLaunchKernelGGL(pv_lik_rows_kernel, dim3((unsigned)B), dim3(256), 0, s, a, x, per, lik, sigmoid_out, sig, loc, dlda, llb);
  PV_LAUNCH_CHECK();
  return 0;
}

int pv_lik_elem(const float* a, const float* x, int64_t M, int lik, int sigmoid_out, float sig, float* loc,
                float* llrow, float* dlda, hipStream_t s) {
  int blocks = (int)((M + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) return 0;
  hipLaunchKernelGGL(pv_lik_elem_kernel, dim3(blocks), dim3(256), 0, s, a, x, M, lik, sigmoid_out, sig, loc, llrow,
                     dlda);
  PV_LAUNCH_CHECK();
  return 0;
}
