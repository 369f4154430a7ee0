// pv_conv_c1.hip — the first block of a 2-D convolutional encoder as ONE forward and ONE backward kernel:
//   y = maxpool2( act( conv3x3(x; one input channel) + bias ) )            (nets/conv.py: ConvBlock + MaxPool2d, 146-199)
// With one input channel the convolution is 9 multiply-adds per output: nothing for the matrix cores, and the op-by-op
// form is bound by the full-resolution activation it writes, re-reads for the pooling, re-reads and re-writes (as a
// gradient that is zero in three of four places) in the backward and re-reads for the weight gradient — five passes
// over B*H*W*Cout floats.  Here the full-resolution tensors never exist: the forward writes the pooled activation and a
// byte per pooled value saying which of the 2x2 positions won (strict >, scan order: the first maximum, like torch);
// the backward reads the pooled gradient, the pooled activation (the activation derivative is a function of the
// output) and that byte, and accumulates dW / db straight from the input image.
#include "pv_common.h"
#include "pv_conv.h"
#include <stdlib.h>

struct C1Pool {
  const float* x; const float* w; const float* bias; float* out; unsigned char* code;
  int B, H, W, C, act, Hp, Wp;
};

template <int ACT> __device__ __forceinline__ float c1_act(float v) {
  if (ACT == PV_ACT_TANH) return tanhf(v);
  if (ACT == PV_ACT_RELU) return v > 0.0f ? v : 0.0f;
  if (ACT == PV_ACT_LRELU) return v > 0.0f ? v : 0.01f * v;
  if (ACT == PV_ACT_SOFTPLUS) return pv_softplus(v);
  if (ACT == PV_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

// one thread per (pooled pixel, 4 CV channels): the 4x4 input window, 4 convolution outputs x 4 CV channels.  CV = 2 (channel
// counts that are multiples of 8): the window loads and their bounds arithmetic — a third of the thread's instructions — serve
// twice the outputs
template <int ACT, int CV>
__global__ __launch_bounds__(256) void pv_c1_convpool_fwd_kernel(C1Pool p) {
  __shared__ __attribute__((aligned(16))) float wl[10 * 64];          // [tap 0..8 | bias][C]
  for (int i = threadIdx.x; i < 10 * p.C; i += 256) {
    const int t = i / p.C, c = i - t * p.C;
    wl[t * p.C + c] = t < 9 ? p.w[c * 9 + t] : (p.bias ? p.bias[c] : 0.0f);
  }
  __syncthreads();
  const int CG = p.C / (4 * CV);
  const int64_t total = (int64_t)p.B * p.Hp * p.Wp * CG;
  // element -> (image, pooled line, pooled pixel, channel group): five 64-bit divisions by run-time values cost more than the
  // convolution of the element; below 2^31 elements they are 32-bit, and shifts where the divisor is a power of two (uniform branches)
  const bool small = total < (1ll << 31);
  const int shCG = (CG & (CG - 1)) == 0 ? 31 - __clz(CG) : -1, shW = (p.Wp & (p.Wp - 1)) == 0 ? 31 - __clz(p.Wp) : -1;
  const int shH = (p.Hp & (p.Hp - 1)) == 0 ? 31 - __clz(p.Hp) : -1;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int cg, px, py;
    int64_t w_, b;
    if (small) {
      const unsigned u = (unsigned)e;
      const unsigned w32 = shCG >= 0 ? u >> shCG : u / (unsigned)CG;
      cg = (int)(u - w32 * (unsigned)CG);
      const unsigned l32 = shW >= 0 ? w32 >> shW : w32 / (unsigned)p.Wp;
      px = (int)(w32 - l32 * (unsigned)p.Wp);
      const unsigned b32 = shH >= 0 ? l32 >> shH : l32 / (unsigned)p.Hp;
      py = (int)(l32 - b32 * (unsigned)p.Hp);
      w_ = w32; b = b32;
    } else {
      cg = (int)(e % CG);
      w_ = e / CG;
      px = (int)(w_ % p.Wp); py = (int)((w_ / p.Wp) % p.Hp);
      b = w_ / ((int64_t)p.Wp * p.Hp);
    }
    const float* xb = p.x + b * p.H * p.W;
    float xw[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int y = 2 * py - 1 + r, x = 2 * px - 1 + c;
        xw[r][c] = (y >= 0 && y < p.H && x >= 0 && x < p.W) ? xb[y * p.W + x] : 0.0f;
      }
#pragma unroll
    for (int h = 0; h < CV; ++h) {
      const int c0 = 4 * (CV * cg + h);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(wl + 9 * p.C + c0);
      f32x4 v[4] = {bv, bv, bv, bv};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + t * p.C + c0);
        const int ty = t / 3, tx = t - 3 * ty;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += wv * xw[(k >> 1) + ty][(k & 1) + tx];
      }
      f32x4 m;
      unsigned best = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float mi = c1_act<ACT>(v[0][i]);
        unsigned bi = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const float a = c1_act<ACT>(v[k][i]);
          if (a > mi) { mi = a; bi = k; }
        }
        m[i] = mi;
        best |= bi << (8 * i);
      }
      *reinterpret_cast<f32x4*>(p.out + w_ * p.C + c0) = m;
      *reinterpret_cast<unsigned*>(p.code + w_ * p.C + c0) = best;
    }
  }
}

bool pv_c1_convpool_supported(int Cin, int Cout, int nd, int act, int H, int W) {
  return Cin == 1 && nd == 2 && Cout >= 4 && Cout <= 64 && Cout % 4 == 0 && act != PV_ACT_GELU && H >= 2 && W >= 2 && W <= 3000;
}

int pv_c1_convpool_fwd(const float* x, int B, int H, int W, const float* w, const float* bias, int Cout, int act, float* out,
                       unsigned char* code, hipStream_t s) {
  if (!pv_c1_convpool_supported(1, Cout, 2, act, H, W)) return PV_EINVAL;
  C1Pool p{x, w, bias, out, code, B, H, W, Cout, act, H / 2, W / 2};
  static const int cv_env = pv_exp_int("PV_C1_CV", 2);            // PV_C1_CV=1|2|4: float4 channel groups per thread (A/B timing)
  const int cv = (cv_env == 4 && Cout % 16 == 0) ? 4 : (cv_env >= 2 && Cout % 8 == 0) ? 2 : 1;
  const int64_t total = (int64_t)B * p.Hp * p.Wp * (Cout / (4 * cv));
  int64_t nb = (total + 255) / 256;
  if (nb > 16384) nb = 16384;
  if (nb < 1) return 0;
  const dim3 grid((unsigned)nb);
#define C1_LAUNCH(A) { if (cv == 4) hipLaunchKernelGGL((pv_c1_convpool_fwd_kernel<A, 4>), grid, dim3(256), 0, s, p); \
                       else if (cv == 2) hipLaunchKernelGGL((pv_c1_convpool_fwd_kernel<A, 2>), grid, dim3(256), 0, s, p); \
                       else hipLaunchKernelGGL((pv_c1_convpool_fwd_kernel<A, 1>), grid, dim3(256), 0, s, p); }
  switch (act) {
    case PV_ACT_TANH: C1_LAUNCH(PV_ACT_TANH); break;
    case PV_ACT_RELU: C1_LAUNCH(PV_ACT_RELU); break;
    case PV_ACT_LRELU: C1_LAUNCH(PV_ACT_LRELU); break;
    case PV_ACT_SOFTPLUS: C1_LAUNCH(PV_ACT_SOFTPLUS); break;
    case PV_ACT_SIGMOID: C1_LAUNCH(PV_ACT_SIGMOID); break;
    default: C1_LAUNCH(PV_ACT_NONE); break;
  }
#undef C1_LAUNCH
  PV_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// backward: dpre = g * act'(y) at the winning position of each pooled value, zero elsewhere;
//   dW[co][tap] = sum dpre[pixel][co] * x[pixel + tap],  db[co] = sum dpre[pixel][co].
// A workgroup takes a range of pooled lines; its threads are (channel, line group): g / y / code of a pooled pixel are one
// coalesced load across the channel lanes, the 4x4 input window of the pooled pixel slides along the line in registers
// (the same for every channel lane).  Per-workgroup partials go through pv_conv3_wgrad_finish_kernel (fixed order).
struct C1PoolBwd {
  const float* g; const float* y; const unsigned char* code; const float* x; float* part; float* part_b;
  int B, H, W, C, act, Hp, Wp, CP, nsplit, lpw;
};

// A workgroup takes lpw consecutive pooled lines: the 4 input lines under each (with a zero halo) are staged in LDS in
// one go, then thread (channel, pixel group) walks the pooled pixels: g / y / code are one coalesced load across the
// channel lanes, the 3x3 input window under the winning position comes out of LDS at per-lane addresses.
__global__ __launch_bounds__(256) void pv_c1_convpool_bwd_kernel(C1PoolBwd p) {
  extern __shared__ float xs[];                       // [lpw][4][W + 2], column index = image column + 1
  __shared__ float sm[256][10];
  const int tid = threadIdx.x, co = tid % p.CP, rg = tid / p.CP, RG = 256 / p.CP;
  const int64_t lines = (int64_t)p.B * p.Hp;
  const int64_t l_lo = (int64_t)blockIdx.x * p.lpw;
  const int nl = (int)(lines - l_lo < p.lpw ? lines - l_lo : p.lpw);
  const int WS = p.W + 2;
  // (staged a row of W + 2 per wave and pass: the row's image / line are wave-uniform, no division per element)
  for (int row = tid >> 6; row < nl * 4; row += 4) {
    const int li = row >> 2, r = row & 3;
    const int64_t l = l_lo + li;
    const int py = (int)(l % p.Hp), y = 2 * py - 1 + r;
    const int64_t b = l / p.Hp;
    const bool yok = y >= 0 && y < p.H;
    const float* xrow = p.x + (b * p.H + (yok ? y : 0)) * p.W;
    for (int cc = tid & 63; cc < WS; cc += 64) {
      const int c = cc - 1;
      xs[row * WS + cc] = (yok && c >= 0 && c < p.W) ? xrow[c] : 0.0f;
    }
  }
  __syncthreads();
  float acc[9], accb = 0.0f;
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.0f;
  const bool cok = co < p.C;
  const int coc = cok ? co : p.C - 1;
  const float msk = cok ? 1.0f : 0.0f;
  // (round 5: a pixel column of ALL the workgroup's lines per pass — the 3 x 8 global loads of a pass are independent and in
  //  flight together; line by line the kernel waited for memory once per line: 42 -> ? us at batch 128)
  for (int px = rg; px < p.Wp; px += RG) {
    float gv[8], yv[8];
    int kv[8];
#pragma unroll
    for (int li = 0; li < 8; ++li) {
      const int lc = li < nl ? li : nl - 1;
      const int64_t o = ((l_lo + lc) * p.Wp + px) * p.C + coc;
      gv[li] = p.g[o]; yv[li] = p.y[o]; kv[li] = p.code[o];
    }
#pragma unroll
    for (int li = 0; li < 8; ++li) {
      if (li >= nl) break;
      const float dv = msk * gv[li] * pv_act_grad(yv[li], 0.0f, p.act);
      const int k = kv[li];
      // the 3x3 input window under the WINNING position (k >> 1, k & 1) of this pooled value: per-lane LDS addresses
      const float* xp = xs + li * 4 * WS + (k >> 1) * WS + 2 * px + (k & 1);
      accb += dv;
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] += dv * xp[(t / 3) * WS + t % 3];
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) sm[tid][t] = acc[t];
  sm[tid][9] = accb;
  __syncthreads();
  for (int o = tid; o < p.C * 10; o += 256) {        // (channel, tap) outputs: sum the pixel groups in group order
    const int c = o / 10, t = o % 10;
    float v = 0.0f;
    for (int g = 0; g < RG; ++g) v += sm[g * p.CP + c][t];
    if (t == 9) { if (p.part_b) p.part_b[(int64_t)blockIdx.x * p.C + c] = v; }
    else p.part[(int64_t)blockIdx.x * p.C * 9 + c * 9 + t] = v;
  }
}

int pv_wgrad_finish_blocks(int64_t nw, int nb);
extern __global__ void pv_conv3_wgrad_finish_kernel(const float* __restrict__ part, int nsplit, int64_t n, float* __restrict__ out,
                                                    const float* __restrict__ part_b, int nb, float* __restrict__ out_b);

static int c1p_lpw(int W) {                          // pooled lines per workgroup: up to 8, 48 KB of staged input lines
  int l = (48 * 1024) / (16 * (W + 2));
  return l > 8 ? 8 : (l < 1 ? 1 : l);
}
static int c1p_splits(int B, int Hp, int W) {
  const int64_t lines = (int64_t)B * Hp;
  const int lpw = c1p_lpw(W);
  return (int)((lines + lpw - 1) / lpw);
}
int64_t pv_c1_convpool_ws(int B, int H, int W, int Cout) {
  return (int64_t)c1p_splits(B, H / 2, W) * (int64_t)Cout * 10 * (int64_t)sizeof(float) + 256;
}

// g: dL/d(pooled output), y: the pooled output, code: from the forward.  dw (Cout, 1, 3, 3), db (Cout) or null.
int pv_c1_convpool_bwd(const float* g, const float* y, const unsigned char* code, const float* x, int B, int H, int W, int Cout,
                       int act, float* dw, float* db, void* ws, int64_t ws_bytes, hipStream_t s, PvFinishList* defer) {
  if (!pv_c1_convpool_supported(1, Cout, 2, act, H, W) || W > 3000) return PV_EINVAL;
  if (!pv_wgrad_ws(defer, pv_c1_convpool_ws(B, H, W, Cout), ws, ws_bytes)) defer = nullptr;
  if (ws_bytes < pv_c1_convpool_ws(B, H, W, Cout)) return PV_EWS;
  const int ns = c1p_splits(B, H / 2, W), lpw = c1p_lpw(W);
  int CP = 1;
  while (CP < Cout) CP *= 2;
  float* part = reinterpret_cast<float*>(ws);
  float* part_b = db ? part + (int64_t)ns * Cout * 9 : nullptr;
  C1PoolBwd p{g, y, code, x, part, part_b, B, H, W, Cout, act, H / 2, W / 2, CP, ns, lpw};
  hipLaunchKernelGGL(pv_c1_convpool_bwd_kernel, dim3(ns), dim3(256), (size_t)lpw * 16 * (W + 2), s, p);
  PV_LAUNCH_CHECK();
  return pv_wgrad_finish(defer, part, ns, (int64_t)Cout * 9, dw, part_b, Cout, db, s);
}
