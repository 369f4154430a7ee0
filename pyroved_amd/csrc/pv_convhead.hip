// pv_convhead.hip — the fully connected head on top of a convolutional feature extractor (nets/conv.py:
// features2latent = flatten (C, spatial) + nn.Linear) without the layout changes.  Activations are channels-last here, the
// Linear's weight is indexed channels-first (f = c*S + s): instead of transposing the feature map (33 MB at C5's size)
// before the forward GEMM and transposing its gradient back, the WEIGHT (out x F, out <= 16) is re-indexed once per step
// (pv_conv_wprep_table kind 4: wt[j][s*C + c] = w[j][c*S + s]) and three streaming kernels do the rest:
//   forward   head[b][j] = bias[j] + sum_f a[b][f] wt[j][f]                         reads a once
//   backward  g[b][f]    = act'(y[b][f]) * sum_j dhead[b][j] wt[j][f]               reads y, writes g (the producing
//                          convolution's activation backward folded in)
//   wgrad     dw[j][c*S + s] = sum_b dhead[b][j] a[b][s*C + c],  db[j] = sum_b dhead[b][j]
#include "pv_common.h"
#include "pv_side.h"
#include "pv_conv.h"
#include <stdlib.h>

#define CH_MAXOUT 16

// workgroup (b, seg): the dot products of sample b over segment seg of f -> part[b][seg][j]
template <int OUT>
__global__ __launch_bounds__(256) void pv_convhead_fwd_kernel(const float* __restrict__ a, const float* __restrict__ wt,
                                                              float* __restrict__ part, int64_t F, int out, int nseg) {
  __shared__ float sm[4][CH_MAXOUT];
  const int b = blockIdx.x, seg = blockIdx.y, tid = threadIdx.x;
  const int64_t F4 = F / 4, i_lo = F4 * seg / nseg, i_hi = F4 * (seg + 1) / nseg;
  const f32x4* ab = reinterpret_cast<const f32x4*>(a + (int64_t)b * F);
  float acc[OUT];
#pragma unroll
  for (int j = 0; j < OUT; ++j) acc[j] = 0.0f;
  for (int64_t i = i_lo + tid; i < i_hi; i += 256) {
    const f32x4 v = ab[i];
#pragma unroll
    for (int j = 0; j < OUT; ++j)
      if (j < out) {
        const f32x4 w = reinterpret_cast<const f32x4*>(wt + (int64_t)j * F)[i];
        acc[j] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
      }
  }
#pragma unroll
  for (int j = 0; j < OUT; ++j) {
    float v = acc[j];
    v = pv_wave_sum(v);
    if ((tid & 63) == 0) sm[tid >> 6][j] = v;
  }
  __syncthreads();
  if (tid < out) part[((int64_t)b * nseg + seg) * out + tid] = sm[0][tid] + sm[1][tid] + sm[2][tid] + sm[3][tid];
}

// head[b][j] = bias[j] + sum_seg part[b][seg][j] (segment order)
__global__ void pv_convhead_fwd_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                              float* __restrict__ head, int B, int out, int nseg) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * out) return;
  const int b = e / out, j = e - b * out;
  float v = bias ? bias[j] : 0.0f;
  const float* pp = part + (int64_t)b * nseg * out + j;
  int k = 0;
  for (; k + 8 <= nseg; k += 8) {                      // eight requests in flight, the additions in segment order (as one by one)
    float p8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) p8[u] = pp[(int64_t)(k + u) * out];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += p8[u];
  }
  for (; k < nseg; ++k) v += pp[(int64_t)k * out];
  head[e] = v;
}

// g[b][f] = (sum_j dhead[b][j] wt[j][f]) * act'(y[b][f]).  A thread owns one float4 column i of f and walks CH_NB samples: its OUT
// weight float4 stay in registers (round 5: one sample per thread re-read the whole (out x F) matrix from L2 for every sample —
// 50 MB at the conv-encoder iVAE's batch 128 — behind a 64-bit division per thread: 17-21 us for 8 MB of useful traffic)
// (heads of <= 4 outputs keep the round-3 form below: at VED's batch 256 the faster kernel made the STEP 0.7-1.2 % slower on every
//  A/B — the encoder's first weight gradient forks off this launch and then shares more of the input-gradient chain's time)
template <int OUT>
__global__ __launch_bounds__(256) void pv_convhead_bwd1_kernel(const float* __restrict__ dhead, const float* __restrict__ wt,
                                                               const float* __restrict__ y, int act, float* __restrict__ g,
                                                               int B, int64_t F, int out) {
  const int64_t F4 = F / 4, total = (int64_t)B * F4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / F4, i = e - b * F4;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < OUT; ++j)
      if (j < out) v += dhead[b * out + j] * reinterpret_cast<const f32x4*>(wt + (int64_t)j * F)[i];
    if (act != PV_ACT_NONE) {
      const f32x4 yy = reinterpret_cast<const f32x4*>(y)[e];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] *= pv_act_grad(yy[k], 0.0f, act);
    }
    reinterpret_cast<f32x4*>(g)[e] = v;
  }
}
template <int OUT, int CH_NB>
__global__ __launch_bounds__(256) void pv_convhead_bwd_kernel(const float* __restrict__ dhead, const float* __restrict__ wt,
                                                              const float* __restrict__ y, int act, float* __restrict__ g,
                                                              int B, int64_t F, int out) {
  const int64_t F4 = F / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= F4) return;
  f32x4 w[OUT];
#pragma unroll
  for (int j = 0; j < OUT; ++j) w[j] = j < out ? reinterpret_cast<const f32x4*>(wt + (int64_t)j * F)[i] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const int b0 = (int)blockIdx.y * CH_NB;
  f32x4 yy[CH_NB];
  if (act != PV_ACT_NONE) {
#pragma unroll
    for (int k = 0; k < CH_NB; ++k)
      if (b0 + k < B) yy[k] = reinterpret_cast<const f32x4*>(y)[(int64_t)(b0 + k) * F4 + i];
  }
#pragma unroll
  for (int k = 0; k < CH_NB; ++k) {
    const int b = b0 + k;
    if (b >= B) break;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < OUT; ++j)
      if (j < out) v += dhead[(int64_t)b * out + j] * w[j];
    if (act != PV_ACT_NONE) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] *= pv_act_grad(yy[k][c], 0.0f, act);
    }
    reinterpret_cast<f32x4*>(g)[(int64_t)b * F4 + i] = v;
  }
}

// partial[split][j][f] (channels-last f) over the samples of the split
template <int OUT>
__global__ __launch_bounds__(256) void pv_convhead_wgrad_kernel(const float* __restrict__ dhead, const float* __restrict__ a,
                                                                float* __restrict__ part, float* __restrict__ part_b, int B,
                                                                int64_t F, int out, int nsplit) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;           // float4 index along f
  const int split = blockIdx.y;
  const int b_lo = (int)((int64_t)B * split / nsplit), b_hi = (int)((int64_t)B * (split + 1) / nsplit);
  if (blockIdx.x == 0 && part_b && (int)threadIdx.x < out) {           // db partial of this split (sample order)
    float v = 0.0f;
    for (int b = b_lo; b < b_hi; ++b) v += dhead[(int64_t)b * out + threadIdx.x];
    part_b[(int64_t)split * out + threadIdx.x] = v;
  }
  if (i >= F / 4) return;
  f32x4 acc[OUT];
#pragma unroll
  for (int j = 0; j < OUT; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int b = b_lo; b < b_hi; ++b) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a + (int64_t)b * F)[i];
#pragma unroll
    for (int j = 0; j < OUT; ++j)
      if (j < out) acc[j] += dhead[(int64_t)b * out + j] * v;
  }
#pragma unroll
  for (int j = 0; j < OUT; ++j)
    if (j < out) reinterpret_cast<f32x4*>(part + ((int64_t)split * out + j) * F)[i] = acc[j];
}

int pv_wgrad_finish_blocks(int64_t nw, int nb);
extern __global__ void pv_conv3_wgrad_finish_kernel(const float* __restrict__ part, int nsplit, int64_t n, float* __restrict__ out,
                                                    const float* __restrict__ part_b, int nb, float* __restrict__ out_b);

bool pv_convhead_supported(int64_t F, int out) { return out >= 1 && out <= CH_MAXOUT && F >= 4 && F % 4 == 0; }

static int ch_splits(int B) { const int n = B / 16; return n > 16 ? 16 : (n < 1 ? 1 : n); }      // >= 16 samples per split
static int ch_segs(int B, int64_t F) {               // forward: ~1024 workgroups, at least 1024 floats per segment
  int64_t n = (1024 + B - 1) / B, cap = F / 1024;
  if (n > cap) n = cap;
  return (int)(n < 1 ? 1 : (n > 16 ? 16 : n));
}
static int chm_segs(int B, int64_t F);
int64_t pv_convhead_ws(int B, int64_t F, int out) {
  // weight gradient: per-split partials (+ bias partials), then their sum in channels-last order before the transposition
  const int64_t wg = (int64_t)ch_splits(B) * out * (F + 1) + (int64_t)out * F, fw = (int64_t)B * ch_segs(B, F) * out;
  const int64_t fm = (int64_t)B * chm_segs(B, F) * out;                 // the MFMA forward's partials
  const int64_t m = wg > fw ? wg : fw;
  return (m > fm ? m : fm) * (int64_t)sizeof(float) + 256;
}

#define CH_DISPATCH(KERNEL, GRID, ...)                                                                       \
  do {                                                                                                       \
    if (out <= 4) hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(256), 0, s, __VA_ARGS__);                         \
    else if (out <= 8) hipLaunchKernelGGL(KERNEL<8>, GRID, dim3(256), 0, s, __VA_ARGS__);                    \
    else hipLaunchKernelGGL(KERNEL<16>, GRID, dim3(256), 0, s, __VA_ARGS__);                                 \
    PV_LAUNCH_CHECK();                                                                                       \
  } while (0)

int64_t pv_convhead_mfma_ws(int B, int64_t F, int out);
int pv_convhead_fwd_mfma(const float* a, const float* wt, const float* bias, float* head, int B, int64_t F, int out, void* ws,
                         int64_t ws_bytes, hipStream_t s);
int pv_convhead_wgrad_mfma(const float* dhead, const float* a, float* dw, float* db, int B, int S, int C, int out, hipStream_t s);
static bool ch_use_mfma() { static const bool on = !pv_exp_int("PV_CONVHEAD_STREAM", 0); return on; }

int pv_convhead_fwd(const float* a, const float* wt, const float* bias, float* head, int B, int64_t F, int out, void* ws,
                    int64_t ws_bytes, hipStream_t s) {
  if (!pv_convhead_supported(F, out)) return PV_EINVAL;
  if (ch_use_mfma() && F % 16 == 0 && ws_bytes >= pv_convhead_mfma_ws(B, F, out))
    return pv_convhead_fwd_mfma(a, wt, bias, head, B, F, out, ws, ws_bytes, s);
  if (ws_bytes < pv_convhead_ws(B, F, out)) return PV_EWS;
  const int nseg = ch_segs(B, F);
  float* part = reinterpret_cast<float*>(ws);
  CH_DISPATCH(pv_convhead_fwd_kernel, dim3((unsigned)B, (unsigned)nseg), a, wt, part, F, out, nseg);
  hipLaunchKernelGGL(pv_convhead_fwd_finish_kernel, dim3((unsigned)((B * out + 255) / 256)), dim3(256), 0, s, part, bias, head, B,
                     out, nseg);
  PV_LAUNCH_CHECK();
  return 0;
}

int pv_convhead_bwd(const float* dhead, const float* wt, const float* y, int act, float* g, int B, int64_t F, int out,
                    hipStream_t s) {
  if (!pv_convhead_supported(F, out) || act == PV_ACT_GELU) return PV_EINVAL;
  // (the encoder's last weight gradient waits for this launch on the side stream: it carries the fork event when one is armed)
  if (out <= 4) {
    int64_t nb = ((int64_t)B * (F / 4) + 255) / 256;
    if (nb > 16384) nb = 16384;
    PV_LAUNCH_FORK(pv_convhead_bwd1_kernel<4>, dim3((unsigned)nb), dim3(256), 0, s, dhead, wt, y, act, g, B, F, out);
    PV_LAUNCH_CHECK();
    return 0;
  }
  const int nbs = 4;                                   // samples per thread
  const dim3 grid((unsigned)((F / 4 + 255) / 256), (unsigned)((B + nbs - 1) / nbs));
  if (grid.y > 65535) {                                // (a batch beyond the grid's second dimension: the one-sample-per-thread form)
    int64_t nb = ((int64_t)B * (F / 4) + 255) / 256;
    if (nb > 16384) nb = 16384;
    if (out <= 8) PV_LAUNCH_FORK(pv_convhead_bwd1_kernel<8>, dim3((unsigned)nb), dim3(256), 0, s, dhead, wt, y, act, g, B, F, out);
    else PV_LAUNCH_FORK(pv_convhead_bwd1_kernel<16>, dim3((unsigned)nb), dim3(256), 0, s, dhead, wt, y, act, g, B, F, out);
    PV_LAUNCH_CHECK();
    return 0;
  }
  if (out <= 8) PV_LAUNCH_FORK((pv_convhead_bwd_kernel<8, 4>), grid, dim3(256), 0, s, dhead, wt, y, act, g, B, F, out);
  else PV_LAUNCH_FORK((pv_convhead_bwd_kernel<16, 4>), grid, dim3(256), 0, s, dhead, wt, y, act, g, B, F, out);
  PV_LAUNCH_CHECK();
  return 0;
}

// does pv_convhead_wgrad use the caller's workspace (the streaming form)?  Then it must stay on the stream that owns it.
bool pv_convhead_wgrad_uses_ws() { return !ch_use_mfma(); }

int pv_convhead_wgrad(const float* dhead, const float* a, float* dw, float* db, int B, int S, int C, int out, void* ws,
                      int64_t ws_bytes, hipStream_t s) {
  const int64_t F = (int64_t)S * C;
  if (!pv_convhead_supported(F, out)) return PV_EINVAL;
  if (ch_use_mfma()) return pv_convhead_wgrad_mfma(dhead, a, dw, db, B, S, C, out, s);
  if (ws_bytes < pv_convhead_ws(B, F, out)) return PV_EWS;
  const int ns = ch_splits(B);
  float* part = reinterpret_cast<float*>(ws);
  float* part_b = part + (int64_t)ns * out * F;
  float* sum = part_b + (int64_t)ns * out;           // [out][S][C]
  const dim3 grid((unsigned)((F / 4 + 255) / 256), (unsigned)ns);
  CH_DISPATCH(pv_convhead_wgrad_kernel, grid, dhead, a, part, db ? part_b : nullptr, B, F, out, ns);
  // partials summed in split order (channels-last), then [out][S][C] -> the Linear's [out][C*S] by the tiled transpose
  const int64_t nw = (int64_t)out * F;
  hipLaunchKernelGGL(pv_conv3_wgrad_finish_kernel, dim3(pv_wgrad_finish_blocks(nw, db ? out : 0)), dim3(256), 0, s, part, ns, nw, sum,
                     db ? part_b : nullptr, out, db);
  PV_LAUNCH_CHECK();
  return pv_nsc_to_ncs(sum, dw, out, C, S, s);
}

// test hooks (tests/test_gpu_conv_kernels.py): the fused first block and the conv head on caller-provided tensors
extern "C" int pv_debug_c1_convpool(int bwd, const float* x, int B, int H, int W, const float* w, const float* bias, int Cout,
                                    int act, float* y, unsigned char* code, const float* g, float* dw, float* db, void* ws,
                                    long long ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!bwd) return pv_c1_convpool_fwd(x, B, H, W, w, bias, Cout, act, y, code, s);
  return pv_c1_convpool_bwd(g, y, code, x, B, H, W, Cout, act, dw, db, ws, ws_bytes, s);
}
extern "C" long long pv_debug_c1_convpool_ws(int B, int H, int W, int Cout) { return pv_c1_convpool_ws(B, H, W, Cout); }

extern "C" long long pv_debug_convhead_ws(int B, long long F, int out) { return pv_convhead_ws(B, F, out); }
// what 0: wt = re-indexed w; 1: head = forward(a); 2: g = backward(dhead, y = a, act); 3: dw, db = wgrad(dhead, a)
extern "C" int pv_debug_convhead(int what, const float* w, float* wt, const float* bias, const float* a, float* head,
                                 const float* dhead, float* g, float* dw, float* db, int B, int S, int C, int out, int act, void* ws,
                                 long long ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int64_t F = (int64_t)S * C;
  if (what == 0) {
    PvWprepEntry e{};
    e.w = w; e.dst = reinterpret_cast<char*>(wt); e.Co = out; e.Ci = C; e.KK = S; e.kind = 4;
    return pv_conv_wprep_table(&e, 1, s);
  }
  if (what == 1) return pv_convhead_fwd(a, wt, bias, head, B, F, out, ws, ws_bytes, s);
  if (what == 2) return pv_convhead_bwd(dhead, wt, a, act, g, B, F, out, s);
  return pv_convhead_wgrad(dhead, a, dw, db, B, S, C, out, ws, ws_bytes, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// The mirror image at the decoder's entry — nets/conv.py latent_to_features: Linear(z_dim -> C0*S0) + view(C0, *dims) — with
// the same idea: the feature map it produces (and the gradient it receives) are channels-last, the Linear's weight
// W[f = c*S + s][k] is re-indexed once per step (pv_conv_wprep_table kind 7: wt[k][s*C + c] = W[c*S + s][k]; zd <= 16):
//   forward   a[b][f'] = bias[f(f')] + sum_k z[b][k] wt[k][f']          (elementwise over the feature map)
//   wgrad     dw[f][k] = sum_b g[b][f'] z[b][k],  db[f] = sum_b g[b][f']  (32 features x 8 sample slices per workgroup, slices meet in
//             LDS in slice order)
//   dgrad     dz[b][k] = sum_f' g[b][f'] wt[k][f']  = pv_convhead_fwd(a = g, wt, no bias)
// instead of three GEMMs with a contraction or an output of 2 and two transposes.
template <int ZD>
__global__ __launch_bounds__(256) void pv_l2f_fwd_kernel(const float* __restrict__ z, const float* __restrict__ wt,
                                                         const float* __restrict__ bias, float* __restrict__ a, int B, int S, int C,
                                                         int zd) {
  const int64_t F = (int64_t)S * C, F4 = F / 4, total = (int64_t)B * F4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / F4, i = e - b * F4;
    const int s = (int)((4 * i) / C), c = (int)(4 * i - (int64_t)s * C);
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = bias[(int64_t)(c + j) * S + s];
    }
#pragma unroll
    for (int k = 0; k < ZD; ++k)
      if (k < zd) v += z[b * zd + k] * reinterpret_cast<const f32x4*>(wt + (int64_t)k * F)[i];
    reinterpret_cast<f32x4*>(a)[e] = v;
  }
}

template <int ZD>
__global__ __launch_bounds__(256) void pv_l2f_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                           float* __restrict__ dw, float* __restrict__ db, int B, int S, int C,
                                                           int zd) {
  __shared__ float sm[8][32][ZD + 1];
  const int fl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t F = (int64_t)S * C, fp = (int64_t)blockIdx.x * 32 + fl;       // channels-last feature index
  float acc[ZD + 1];
#pragma unroll
  for (int k = 0; k <= ZD; ++k) acc[k] = 0.0f;
  if (fp < F)
    for (int b0 = sl; b0 < B; b0 += 64) {             // eight samples of the slice at a time: their loads go out together
      float gv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[j] = b0 + 8 * j < B ? g[(int64_t)(b0 + 8 * j) * F + fp] : 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int b = b0 + 8 * j < B ? b0 + 8 * j : 0;   // (gv = 0 beyond the batch)
        acc[ZD] += gv[j];
#pragma unroll
        for (int k = 0; k < ZD; ++k)
          if (k < zd) acc[k] += gv[j] * z[(int64_t)b * zd + k];
      }
    }
#pragma unroll
  for (int k = 0; k <= ZD; ++k) sm[sl][fl][k] = acc[k];
  __syncthreads();
  if (sl == 0 && fp < F) {
    const int s = (int)(fp / C), c = (int)(fp - (int64_t)s * C);
    const int64_t f = (int64_t)c * S + s;                                      // the Linear's row
#pragma unroll
    for (int k = 0; k <= ZD; ++k) {
      float t = 0.0f;
      for (int q = 0; q < 8; ++q) t += sm[q][fl][k];
      if (k == ZD) { if (db) db[f] = t; }
      else if (k < zd) dw[f * zd + k] = t;
    }
  }
}

bool pv_l2f_supported(int64_t F, int zd, int C) { return zd >= 1 && zd <= CH_MAXOUT && F >= 4 && C % 4 == 0; }

#define L2F_DISPATCH(KERNEL, GRID, ...)                                                                      \
  do {                                                                                                       \
    if (zd <= 2) hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(256), 0, s, __VA_ARGS__);                          \
    else if (zd <= 4) hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(256), 0, s, __VA_ARGS__);                     \
    else if (zd <= 8) hipLaunchKernelGGL(KERNEL<8>, GRID, dim3(256), 0, s, __VA_ARGS__);                     \
    else hipLaunchKernelGGL(KERNEL<16>, GRID, dim3(256), 0, s, __VA_ARGS__);                                 \
    PV_LAUNCH_CHECK();                                                                                       \
  } while (0)

int pv_l2f_fwd(const float* z, const float* wt, const float* bias, float* a, int B, int S, int C, int zd, hipStream_t s) {
  if (!pv_l2f_supported((int64_t)S * C, zd, C)) return PV_EINVAL;
  int64_t nb = ((int64_t)B * S * C / 4 + 255) / 256;
  if (nb > 8192) nb = 8192;
  L2F_DISPATCH(pv_l2f_fwd_kernel, dim3((unsigned)nb), z, wt, bias, a, B, S, C, zd);
  return 0;
}

int pv_l2f_wgrad(const float* g, const float* z, float* dw, float* db, int B, int S, int C, int zd, hipStream_t s) {
  if (!pv_l2f_supported((int64_t)S * C, zd, C)) return PV_EINVAL;
  const int64_t F = (int64_t)S * C;
  L2F_DISPATCH(pv_l2f_wgrad_kernel, dim3((unsigned)((F + 31) / 32)), g, z, dw, db, B, S, C, zd);
  return 0;
}

// test hook: what 0: wt = re-indexed w (kind 7); 1: a = forward(z); 2: dw, db = wgrad(g, z); 3: dz = dgrad(g)
extern "C" int pv_debug_l2f(int what, const float* w, float* wt, const float* bias, const float* z, float* a, const float* g,
                            float* dw, float* db, float* dz, int B, int S, int C, int zd, void* ws, long long ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (what == 0) {
    PvWprepEntry e{};
    e.w = w; e.dst = reinterpret_cast<char*>(wt); e.Co = zd; e.Ci = C; e.KK = S; e.kind = 7;
    return pv_conv_wprep_table(&e, 1, s);
  }
  if (what == 1) return pv_l2f_fwd(z, wt, bias, a, B, S, C, zd, s);
  if (what == 2) return pv_l2f_wgrad(g, z, dw, db, B, S, C, zd, s);
  return pv_convhead_fwd(g, wt, nullptr, dz, B, (int64_t)S * C, zd, ws, ws_bytes, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// The conv head's forward and weight gradient as skinny GEMMs on the f32-input matrix cores (out <= 16 rows): the streaming
// kernels above re-read the out x F weight once per sample segment (12x the feature-map traffic at out = 12) and their
// weight gradient needs per-split partials, a finish and a transpose.  v_mfma_f32_16x16x4_f32 is an exact fp32 FMA chain.
//   D[j][col] layout: lane (col = r, q), register i -> row j = 4 q + i.
#define CH_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// forward: workgroup (16-sample group, segment of f); wave w takes a quarter of the segment in steps of 16 features: one float4
// of the re-indexed weight (row j = r) and one of the feature map (sample r) per lane feed four MFMAs (k = 4 q + s).
__global__ __launch_bounds__(256) void pv_convhead_fwd_mfma_kernel(const float* __restrict__ a, const float* __restrict__ wt,
                                                                   float* __restrict__ part, int B, int64_t F, int out, int nseg) {
  __shared__ f32x4 red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int grp = blockIdx.x, seg = blockIdx.y;
  const int64_t steps = F / 16, s_lo = steps * seg / nseg, s_hi = steps * (seg + 1) / nseg;
  const int64_t w_lo = s_lo + (s_hi - s_lo) * wave / 4, w_hi = s_lo + (s_hi - s_lo) * (wave + 1) / 4;
  const int b = grp * 16 + r;
  const bool bok = b < B, jok = r < out;
  const f32x4* ap = reinterpret_cast<const f32x4*>(a + (int64_t)(bok ? b : 0) * F) + q;
  const f32x4* wp = reinterpret_cast<const f32x4*>(wt + (int64_t)(jok ? r : 0) * F) + q;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 acc = zero;
#pragma unroll 4
  for (int64_t st = w_lo; st < w_hi; ++st) {
    const f32x4 av = bok ? ap[st * 4] : zero;
    const f32x4 wv = jok ? wp[st * 4] : zero;
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = CH_MFMA(wv[s], av[s], acc);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    const f32x4 t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (bok) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * q + i < out) part[((int64_t)b * nseg + seg) * out + 4 * q + i] = t[i];
    }
  }
}

// weight gradient: a wave owns 64 channels-last features x all out rows and contracts over the samples, four at a time: one
// float4 of the feature map (sample b0 + q, features 4 r .. 4 r + 3) and one value of dhead (sample b0 + q, row r) per lane feed
// four MFMAs, one per feature of the float4.  Results go straight to the Linear's layout dw[j][c*S + s]; no partials.
__global__ __launch_bounds__(256) void pv_convhead_wgrad_mfma_kernel(const float* __restrict__ dhead, const float* __restrict__ a,
                                                                     float* __restrict__ dw, float* __restrict__ db, int B, int S,
                                                                     int C, int out) {
  // a workgroup owns 64 features; its four waves take a quarter of the samples each and meet in LDS in wave order
  __shared__ f32x4 red[3][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int64_t F = (int64_t)S * C, f0 = (int64_t)blockIdx.x * 64;
  if (blockIdx.x == gridDim.x - 1 && db && wave == 3) {  // db[j] = sum_b dhead[b][j]: lanes stride the samples, then a fixed tree
    for (int j = 0; j < out; ++j) {
      float v = 0.0f;
      for (int b = lane; b < B; b += 64) v += dhead[(int64_t)b * out + j];
      v = pv_wave_sum(v);
      if (lane == 0) db[j] = v;
    }
  }
  const bool fok = f0 + 4 * r + 3 < F, jok = r < out;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 acc[4] = {zero, zero, zero, zero};
  const int nb4 = (B + 3) / 4, s_lo = nb4 * wave / 4, s_hi = nb4 * (wave + 1) / 4;
#pragma unroll 4
  for (int st = s_lo; st < s_hi; ++st) {
    const int b = 4 * st + q;
    const bool bok = b < B;
    const f32x4 av = (bok && fok) ? *reinterpret_cast<const f32x4*>(a + (int64_t)b * F + f0 + 4 * r) : zero;
    const float dv = (bok && jok) ? dhead[(int64_t)b * out + r] : 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = CH_MFMA(dv, av[s], acc[s]);
  }
  if (wave > 0) {
#pragma unroll
    for (int s = 0; s < 4; ++s) red[wave - 1][s][lane] = acc[s];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const f32x4 t = (acc[s] + red[0][s][lane]) + (red[1][s][lane] + red[2][s][lane]);
    const int64_t fp = f0 + 4 * r + s;                 // channels-last feature of column r of accumulator s
    if (fp >= F) continue;
    const int sp = (int)(fp / C), c = (int)(fp - (int64_t)sp * C);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (4 * q + i < out) dw[(int64_t)(4 * q + i) * F + (int64_t)c * S + sp] = t[i];
  }
}

static int chm_segs(int B, int64_t F) {              // forward: ~512 workgroups of (16 samples) x (segment of >= 64 features)
  const int64_t groups = (B + 15) / 16;
  int64_t n = (512 + groups - 1) / groups, cap = F / 64;
  if (n > cap) n = cap;
  return (int)(n < 1 ? 1 : (n > 32 ? 32 : n));       // (the finish sums the segments serially per output)
}
int64_t pv_convhead_mfma_ws(int B, int64_t F, int out) { return (int64_t)B * chm_segs(B, F) * out * (int64_t)sizeof(float) + 256; }

int pv_convhead_fwd_mfma(const float* a, const float* wt, const float* bias, float* head, int B, int64_t F, int out, void* ws,
                         int64_t ws_bytes, hipStream_t s) {
  if (!pv_convhead_supported(F, out) || F % 16 != 0) return PV_EINVAL;
  if (ws_bytes < pv_convhead_mfma_ws(B, F, out)) return PV_EWS;
  const int nseg = chm_segs(B, F);
  float* part = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(pv_convhead_fwd_mfma_kernel, dim3((unsigned)((B + 15) / 16), (unsigned)nseg), dim3(256), 0, s, a, wt, part, B, F,
                     out, nseg);
  PV_LAUNCH_CHECK();
  hipLaunchKernelGGL(pv_convhead_fwd_finish_kernel, dim3((unsigned)((B * out + 255) / 256)), dim3(256), 0, s, part, bias, head, B,
                     out, nseg);
  PV_LAUNCH_CHECK();
  return 0;
}

// the matrix-core forward WITHOUT its finish launch: *part (B, *nseg, out) partial sums in ws; the consumer adds them in segment
// order on top of the bias (pv_dec1d.hip does, per sample).  PV_EINVAL when that form does not apply.
int pv_convhead_fwd_partials(const float* a, const float* wt, int B, int64_t F, int out, void* ws, int64_t ws_bytes, hipStream_t s,
                             const float** part, int* nseg) {
  if (!pv_convhead_supported(F, out) || !ch_use_mfma() || F % 16 != 0 || ws_bytes < pv_convhead_mfma_ws(B, F, out)) return PV_EINVAL;
  const int ns = chm_segs(B, F);
  float* pt = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(pv_convhead_fwd_mfma_kernel, dim3((unsigned)((B + 15) / 16), (unsigned)ns), dim3(256), 0, s, a, wt, pt, B, F, out, ns);
  PV_LAUNCH_CHECK();
  *part = pt; *nseg = ns;
  return 0;
}

int pv_convhead_wgrad_mfma(const float* dhead, const float* a, float* dw, float* db, int B, int S, int C, int out, hipStream_t s) {
  const int64_t F = (int64_t)S * C;
  if (!pv_convhead_supported(F, out)) return PV_EINVAL;
  hipLaunchKernelGGL(pv_convhead_wgrad_mfma_kernel, dim3((unsigned)((F + 63) / 64)), dim3(256), 0, s, dhead, a, dw, db, B, S, C, out);
  PV_LAUNCH_CHECK();
  return 0;
}

// y[b][j] = sum_k x[b*ldx + k] * w[j*K + k]  (K <= 16, no bias): a Linear with a contraction this short is not a GEMM —
// fc_latent (content latents -> 128) of the spatial decoder when the encoder is not the compact fc one
__global__ void pv_smallk_linear_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w, float* __restrict__ y,
                                        int64_t B, int K, int N) {
  const int64_t total = B * N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / N;
    const int j = (int)(e - b * N);
    float v = 0.0f;
    for (int k = 0; k < K; ++k) v = fmaf(x[b * ldx + k], w[(int64_t)j * K + k], v);
    y[e] = v;
  }
}
int pv_smallk_linear(const float* x, int64_t ldx, const float* w, float* y, int64_t B, int K, int N, hipStream_t s) {
  if (K < 1 || K > 16 || B < 1 || N < 1) return PV_EINVAL;
  int64_t nb = (B * N + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(pv_smallk_linear_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, ldx, w, y, B, K, N);
  PV_LAUNCH_CHECK();
  return 0;
}
