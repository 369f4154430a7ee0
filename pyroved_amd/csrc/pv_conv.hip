// pv_conv.hip — the data-movement kernels around the convolutional nets of models.VED (nets/conv.py): everything a
// conv / pool / upsample stack needs besides its GEMMs (the convolutions themselves gather their patches inside
// pv_gemm.hip's tile loads).  Activations are channels-last, [B][spatial...][C].
// All of these are HBM-bound gathers (one thread per OUTPUT element, no atomics):
//   conv_wflip           (the flipped / channel-swapped weight of a convolution's dgrad)
//   maxpool2 fwd / bwd   (2x, stride 2; ties resolved as torch: first maximum in window scan order)
//   upsample2 fwd / bwd  (nearest)
//   ncs <-> nsc          ((B, C, S) <-> (B, S, C) transposes at the torch-layout boundaries)
//   act_bwd              (dY *= act'(Y) in place)
#include "pv_common.h"
#include "pv_side.h"
#include "pv_conv.h"

#define CONV_THREADS 256

static inline int conv_blocks(int64_t n) {
  int64_t b = (n + CONV_THREADS - 1) / CONV_THREADS;
  return (int)(b > 65535 * 16 ? 65535 * 16 : (b < 1 ? 1 : b));
}

// out[b][y][x][c] = max over the 2x2 (2-D) / 2 (1-D) window of in
__global__ void pv_maxpool2_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C,
                                       int nd) {
  const int Ho = H / 2, Wo = nd == 2 ? W / 2 : 1;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t pix = e / C;
    const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho);
    const int64_t b = pix / ((int64_t)Wo * Ho);
    float m = -INFINITY;
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < (nd == 2 ? 2 : 1); ++dx) {
        const float v = in[((b * H + 2 * y + dy) * W + (nd == 2 ? 2 * x + dx : 0)) * C + c];
        m = v > m ? v : m;
      }
    out[e] = m;
  }
}

// din[b][y][x][c] = dout of its window if in[...] is the window's FIRST maximum (torch's argmax rule), else 0
__global__ void pv_maxpool2_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                       float* __restrict__ din, int B, int H, int W, int C, int nd, int eg_act) {
  const int Ho = H / 2, Wo = nd == 2 ? W / 2 : 1;
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t pix = e / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    const int oy = y / 2, ox = nd == 2 ? x / 2 : 0;
    float g = 0.0f;
    if (oy < Ho && ox < Wo) {                        // (odd trailing rows / columns are not pooled)
      int best = -1;
      float m = -INFINITY;
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < (nd == 2 ? 2 : 1); ++dx) {
          const float v = in[((b * H + 2 * oy + dy) * W + (nd == 2 ? 2 * ox + dx : 0)) * C + c];
          if (v > m || best < 0) { m = v; best = dy * 2 + dx; }
        }
      const int mine = (y - 2 * oy) * 2 + (nd == 2 ? x - 2 * ox : 0);
      if (mine == best) g = dout[((b * Ho + oy) * Wo + ox) * C + c];
    }
    if (eg_act != PV_ACT_NONE) g *= pv_act_grad(in[e], 0.0f, eg_act);
    din[e] = g;
  }
}

// the same for even H (and W) and C % 4 == 0: one thread per (window, 4 channels) — reads the window once, 16-byte accesses
__global__ void pv_maxpool2_bwd4_kernel(const float* __restrict__ in, const float* __restrict__ dout, float* __restrict__ din,
                                        int B, int H, int W, int C, int nd, int eg_act) {
  const int Ho = H / 2, Wo = nd == 2 ? W / 2 : 1, C4 = C / 4, nx = nd == 2 ? 2 : 1;
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % C4);
    const int64_t w = e / C4;
    const int ox = (int)(w % Wo), oy = (int)((w / Wo) % Ho);
    const int64_t b = w / ((int64_t)Wo * Ho);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dout + w * C + 4 * c4);
    f32x4 v[4];
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < nx; ++dx)
        v[dy * 2 + dx] = *reinterpret_cast<const f32x4*>(in + ((b * H + 2 * oy + dy) * W + (nd == 2 ? 2 * ox + dx : 0)) * C + 4 * c4);
    int best[4] = {0, 0, 0, 0};
    f32x4 m = v[0];
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < nx; ++dx) {
        const int k = dy * 2 + dx;
        if (k == 0) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (v[k][i] > m[i]) { m[i] = v[k][i]; best[i] = k; }      // strict >: the FIRST maximum wins (torch)
      }
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < nx; ++dx) {
        const int k = dy * 2 + dx;
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          o[i] = best[i] == k ? (eg_act != PV_ACT_NONE ? g[i] * pv_act_grad(v[k][i], 0.0f, eg_act) : g[i]) : 0.0f;
        *reinterpret_cast<f32x4*>(din + ((b * H + 2 * oy + dy) * W + (nd == 2 ? 2 * ox + dx : 0)) * C + 4 * c4) = o;
      }
  }
}

// backward of a max-pool whose forward kept only the pooled values and the winners (pv_conv3_sp with pool_out): din (B, 2 Hp,
// 2 Wp, C) = g * act'(y_pooled) at the winning position of each window, zero elsewhere; one thread per (window, 4 channels)
__global__ void pv_maxpool2_bwd_code_kernel(const float* __restrict__ g, const float* __restrict__ yp, const unsigned char* __restrict__ code,
                                            float* __restrict__ din, int B, int Hp, int Wp, int C, int eg_act) {
  const int C4 = C / 4, W = 2 * Wp;
  const int64_t total = (int64_t)B * Hp * Wp * C4;
  // index arithmetic: three 64-bit divisions per 16 input bytes made this gather instruction-bound (2 TB/s); with power-of-two
  // extents and fewer than 2^29 windows (the default stacks) they are shifts and masks of a 32-bit index
  const bool pow2 = total < (1ll << 27) && (C4 & (C4 - 1)) == 0 && (Wp & (Wp - 1)) == 0 && (Hp & (Hp - 1)) == 0;
  const int lc = __builtin_ctz(C4), lw = __builtin_ctz(Wp), lh = __builtin_ctz(Hp);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    size_t gi, d0;                                    // element offsets: pooled tensors; window (0, 0) of din
    if (pow2) {
      const unsigned eu = (unsigned)e, wu = eu >> lc, c4 = eu & (unsigned)(C4 - 1);
      const unsigned ox = wu & (unsigned)(Wp - 1), oy = (wu >> lw) & (unsigned)(Hp - 1), b = wu >> (lw + lh);
      gi = 4u * eu;                                   // w * C + 4 * c4
      d0 = (((b * 2u * (unsigned)Hp + 2u * oy) * (unsigned)W) + 2u * ox) * (unsigned)C + 4u * c4;   // < 2^31 elements
    } else {
      const int c4 = (int)(e % C4);
      const int64_t w = e / C4;
      const int ox = (int)(w % Wp), oy = (int)((w / Wp) % Hp);
      const int64_t b = w / ((int64_t)Wp * Hp);
      gi = (size_t)(w * C + 4 * c4);
      d0 = (size_t)((((b * 2 * Hp + 2 * oy) * W) + 2 * ox) * C + 4 * c4);
    }
    f32x4 v = *reinterpret_cast<const f32x4*>(g + gi);
    if (eg_act != PV_ACT_NONE) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(yp + gi);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= pv_act_grad(y[i], 0.0f, eg_act);
    }
    const unsigned cd = *reinterpret_cast<const unsigned*>(code + gi);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = ((cd >> (8 * i)) & 3u) == (unsigned)k ? v[i] : 0.0f;
      *reinterpret_cast<f32x4*>(din + d0 + (size_t)((k >> 1) * W + (k & 1)) * C) = o;
    }
  }
}

// out[b][y][x][c] = in[b][y/2][x/2][c]
__global__ void pv_upsample2_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C,
                                        int nd) {
  const int Ho = 2 * H, Wo = nd == 2 ? 2 * W : 1;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t pix = e / C;
    const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho);
    const int64_t b = pix / ((int64_t)Wo * Ho);
    out[e] = in[((b * H + y / 2) * W + (nd == 2 ? x / 2 : 0)) * C + c];
  }
}

// din[b][y][x][c] = sum of the 2 (1-D) / 4 (2-D) output positions it was copied to
// (eg_act != NONE: times act'(eg_y[e]), eg_y = the upsampled tensor = the producing convolution's output)
__global__ void pv_upsample2_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int H, int W,
                                        int C, int nd, const float* __restrict__ eg_y, int eg_act) {
  const int Ho = 2 * H, Wo = nd == 2 ? 2 * W : 1;
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t pix = e / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    float v = 0.0f;
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < (nd == 2 ? 2 : 1); ++dx)
        v += dout[((b * Ho + 2 * y + dy) * Wo + (nd == 2 ? 2 * x + dx : 0)) * C + c];
    if (eg_y) v *= pv_act_grad(eg_y[e], 0.0f, eg_act);
    din[e] = v;
  }
}

// F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) along one axis of length n: output o reads
// inputs i0, i1 with weights (1 - l), l:  src = max((o + 0.5)/2 - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, n - 1)
__device__ __forceinline__ void bil_src(int o, int n, int& i0, int& i1, float& l) {
  float src = (o + 0.5f) * 0.5f - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  i0 = (int)src;
  i1 = i0 + 1 < n ? i0 + 1 : n - 1;
  l = src - (float)i0;
}

__global__ void pv_upsample2_bil_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                            int C) {
  const int Ho = 2 * H, Wo = 2 * W;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t pix = e / C;
    const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho);
    const int64_t b = pix / ((int64_t)Wo * Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    bil_src(y, H, y0, y1, ly);
    bil_src(x, W, x0, x1, lx);
    const float* p = in + b * H * W * C + c;
    const float v00 = p[((int64_t)y0 * W + x0) * C], v01 = p[((int64_t)y0 * W + x1) * C];
    const float v10 = p[((int64_t)y1 * W + x0) * C], v11 = p[((int64_t)y1 * W + x1) * C];
    out[e] = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
  }
}

// gather form of the transpose: every input pixel collects from the (up to 4 x 4) outputs whose stencil touches it
__global__ void pv_upsample2_bil_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int H, int W,
                                            int C) {
  const int Ho = 2 * H, Wo = 2 * W;
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t pix = e / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    float v = 0.0f;
    for (int oy = 2 * y - 2; oy <= 2 * y + 2; ++oy) {
      if (oy < 0 || oy >= Ho) continue;
      int y0, y1; float ly;
      bil_src(oy, H, y0, y1, ly);
      const float wy = (y0 == y ? 1.0f - ly : 0.0f) + (y1 == y ? ly : 0.0f);
      if (wy == 0.0f) continue;
      for (int ox = 2 * x - 2; ox <= 2 * x + 2; ++ox) {
        if (ox < 0 || ox >= Wo) continue;
        int x0, x1; float lx;
        bil_src(ox, W, x0, x1, lx);
        const float wx = (x0 == x ? 1.0f - lx : 0.0f) + (x1 == x ? lx : 0.0f);
        if (wx != 0.0f) v += wy * wx * dout[((b * Ho + oy) * Wo + ox) * C + c];
      }
    }
    din[e] = v;
  }
}

// out[b][s][c] = in[b][c][s]   (to_nsc)   /   out[b][c][s] = in[b][s][c]   (to_ncs): per sample a (rows x cols) ->
// (cols x rows) transpose, 32 x 32 tiles through LDS so that both the read and the write are coalesced
__global__ __launch_bounds__(256) void pv_ncs_nsc_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t B,
                                                         int C, int64_t S, int to_nsc) {
  __shared__ float tile[32][33];
  const int64_t rows = to_nsc ? C : S, cols = to_nsc ? S : C;       // in[b][rows][cols] -> out[b][cols][rows]
  const int64_t tr = (rows + 31) / 32, tc = (cols + 31) / 32, per = tr * tc, total = B * per;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8 threads
  for (int64_t t = blockIdx.x; t < total; t += gridDim.x) {
    const int64_t b = t / per, rem = t - b * per;
    const int64_t r0 = (rem / tc) * 32, c0 = (rem % tc) * 32;
    const float* ib = in + b * rows * cols;
    float* ob = out + b * rows * cols;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
      if (r < rows && c < cols) tile[ty + 8 * k][tx] = ib[r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
      if (r < rows && c < cols) ob[c * rows + r] = tile[tx][ty + 8 * k];
    }
    __syncthreads();
  }
}

__global__ void pv_act_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, int64_t n, int act) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    dy[e] *= pv_act_grad(y[e], 0.0f, act);
}

// dgrad of a kernel-3 convolution = the same convolution of dY with the taps flipped and the channel roles swapped:
//   wt[ci][co*KK + t] = w[co][ci*KK + (KK-1-t)]
__global__ void pv_conv_wflip_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int KK) {
  const int total = Cout * Cin * KK;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int t = e % KK, co = (e / KK) % Cout, ci = e / (KK * Cout);
    wt[e] = w[((int64_t)co * Cin + ci) * KK + (KK - 1 - t)];
  }
}

#define CONV_LAUNCH(kernel, n, ...)                                                                   \
  do {                                                                                                \
    if ((n) > 0) hipLaunchKernelGGL(kernel, dim3(conv_blocks(n)), dim3(CONV_THREADS), 0, s, __VA_ARGS__); \
    PV_LAUNCH_CHECK();                                                                                \
    return 0;                                                                                         \
  } while (0)

int pv_maxpool2_fwd(const float* in, float* out, int B, int H, int W, int C, int nd, hipStream_t s) {
  CONV_LAUNCH(pv_maxpool2_fwd_kernel, (int64_t)B * (H / 2) * (nd == 2 ? W / 2 : 1) * C, in, out, B, H, W, C, nd);
}
int pv_maxpool2_bwd(const float* in, const float* dout, float* din, int B, int H, int W, int C, int nd, hipStream_t s,
                    int eg_act) {
  if (eg_act == PV_ACT_GELU) return PV_EINVAL;
  if (C % 4 == 0 && H % 2 == 0 && (nd == 1 || W % 2 == 0))
    CONV_LAUNCH(pv_maxpool2_bwd4_kernel, (int64_t)B * (H / 2) * (nd == 2 ? W / 2 : 1) * (C / 4), in, dout, din, B, H, W, C, nd,
                eg_act);
  CONV_LAUNCH(pv_maxpool2_bwd_kernel, (int64_t)B * H * W * C, in, dout, din, B, H, W, C, nd, eg_act);
}
int pv_maxpool2_bwd_code(const float* g, const float* y_pooled, const unsigned char* code, float* din, int B, int Hp, int Wp, int C,
                         int eg_act, hipStream_t s) {
  if (C % 4 != 0 || eg_act == PV_ACT_GELU) return PV_EINVAL;
  const int64_t n = (int64_t)B * Hp * Wp * (C / 4);
  // (a side-stream weight gradient may wait for this launch: it carries the fork event when one is armed)
  if (n > 0) PV_LAUNCH_FORK(pv_maxpool2_bwd_code_kernel, dim3(conv_blocks(n)), dim3(CONV_THREADS), 0, s, g, y_pooled, code, din, B, Hp, Wp, C, eg_act);
  PV_LAUNCH_CHECK();
  return 0;
}
int pv_upsample2_fwd(const float* in, float* out, int B, int H, int W, int C, int nd, hipStream_t s) {
  CONV_LAUNCH(pv_upsample2_fwd_kernel, (int64_t)B * 2 * H * (nd == 2 ? 2 * W : 1) * C, in, out, B, H, W, C, nd);
}
int pv_upsample2_bwd(const float* dout, float* din, int B, int H, int W, int C, int nd, hipStream_t s, const float* eg_y,
                     int eg_act) {
  if (eg_act == PV_ACT_NONE || eg_act == PV_ACT_GELU) eg_y = nullptr;
  CONV_LAUNCH(pv_upsample2_bwd_kernel, (int64_t)B * H * W * C, dout, din, B, H, W, C, nd, eg_y, eg_act);
}
static int ncs_nsc(const float* in, float* out, int64_t B, int C, int64_t S, int to_nsc, hipStream_t s) {
  const int64_t tiles = B * ((C + 31) / 32) * ((S + 31) / 32);
  if (tiles < 1) return 0;
  hipLaunchKernelGGL(pv_ncs_nsc_kernel, dim3((unsigned)(tiles > 8192 ? 8192 : tiles)), dim3(256), 0, s, in, out, B, C, S, to_nsc);
  PV_LAUNCH_CHECK();
  return 0;
}
int pv_ncs_to_nsc(const float* in, float* out, int64_t B, int C, int64_t S, hipStream_t s) { return ncs_nsc(in, out, B, C, S, 1, s); }
int pv_nsc_to_ncs(const float* in, float* out, int64_t B, int C, int64_t S, hipStream_t s) { return ncs_nsc(in, out, B, C, S, 0, s); }
int pv_act_bwd(float* dy, const float* y, int64_t n, int act, hipStream_t s) {
  if (act == PV_ACT_NONE) return 0;
  if (act == PV_ACT_GELU) return PV_EINVAL;           // needs the pre-activation; not kept on this path
  CONV_LAUNCH(pv_act_bwd_kernel, n, dy, y, n, act);
}
int pv_conv_wflip(const float* w, float* wt, int Cout, int Cin, int KK, hipStream_t s) {
  CONV_LAUNCH(pv_conv_wflip_kernel, (int64_t)Cout * Cin * KK, w, wt, Cout, Cin, KK);
}
int pv_upsample2_bil_fwd(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
  CONV_LAUNCH(pv_upsample2_bil_fwd_kernel, (int64_t)B * 4 * H * W * C, in, out, B, H, W, C);
}
int pv_upsample2_bil_bwd(const float* dout, float* din, int B, int H, int W, int C, hipStream_t s) {
  CONV_LAUNCH(pv_upsample2_bil_bwd_kernel, (int64_t)B * H * W * C, dout, din, B, H, W, C);
}

// ---------------------------------------------------------------------------------------------------------------------
// nn.BatchNormNd (training mode: batch statistics, biased variance for the normalisation, unbiased for the running
// estimate; torch.nn.functional.batch_norm) over channels-last rows x[R][C].  Per-channel reductions are two-stage and
// deterministic: BN_BLOCKS row chunks x all channels -> partials -> fixed-order finish.
#define BN_BLOCKS 512
#define BN_MAXC 1024
// mode 0: sum x ; mode 1: sum (x - mean)^2 ; mode 2: a0 = sum dy, a1 = sum dy * xhat
template <int MODE>
__global__ __launch_bounds__(256) void pv_bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           int64_t R, int C, int cb /* pow2 >= min(C, 256) */,
                                                           const float* __restrict__ stats, float* __restrict__ part) {
  __shared__ float sm0[256], sm1[256];
  const int t = threadIdx.x, tc = t % cb, tr = t / cb, nrt = 256 / cb;
  const int64_t r0 = R * blockIdx.x / gridDim.x, r1 = R * (blockIdx.x + 1) / gridDim.x;
  for (int c0 = 0; c0 < C; c0 += cb) {
    const int c = c0 + tc;
    float a0 = 0.0f, a1 = 0.0f;
    if (c < C) {
      const float mean = MODE >= 1 ? stats[c] : 0.0f, inv = MODE == 2 ? stats[C + c] : 0.0f;
      for (int64_t r = r0 + tr; r < r1; r += nrt) {
        const float v = x[r * C + c];
        if (MODE == 0) a0 += v;
        else if (MODE == 1) { const float d = v - mean; a0 += d * d; }
        else { const float g = dy[r * C + c]; a0 += g; a1 += g * ((v - mean) * inv); }
      }
    }
    sm0[t] = a0; sm1[t] = a1;
    __syncthreads();
    if (tr == 0 && c < C) {
      float s0 = 0.0f, s1 = 0.0f;
      for (int k = 0; k < nrt; ++k) { s0 += sm0[k * cb + tc]; s1 += sm1[k * cb + tc]; }
      part[((int64_t)blockIdx.x * 2) * C + c] = s0;
      part[((int64_t)blockIdx.x * 2 + 1) * C + c] = s1;
    }
    __syncthreads();
  }
}

// step 0: mean = sum / R.  step 1: var -> invstd, running statistics.  step 2: dgamma, dbeta (sums kept in part[0..2C))
__global__ void pv_bn_finish_kernel(float* __restrict__ part, int nblk, int64_t R, int C, int step, float* __restrict__ stats,
                                    float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s0 = 0.0f, s1 = 0.0f;
  for (int b = 0; b < nblk; ++b) { s0 += part[((int64_t)b * 2) * C + c]; s1 += part[((int64_t)b * 2 + 1) * C + c]; }
  if (step == 0) {
    stats[c] = s0 / (float)R;
  } else if (step == 1) {
    const float var = s0 / (float)R;
    stats[C + c] = 1.0f / sqrtf(var + eps);
    if (rmean) {
      rmean[c] = (1.0f - momentum) * rmean[c] + momentum * stats[c];
      rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (R > 1 ? s0 / (float)(R - 1) : var);
    }
  } else {
    dbeta[c] = s0; dgamma[c] = s1;
    stats[2 * C + c] = s0; stats[3 * C + c] = s1;       // for the dx pass
  }
}

__global__ void pv_bn_running_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int C, float eps,
                                     float* __restrict__ stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { stats[c] = rmean[c]; stats[C + c] = 1.0f / sqrtf(rvar[c] + eps); }
}

__global__ void pv_bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ stats) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    y[e] = (x[e] - stats[c]) * stats[C + c] * gamma[c] + beta[c];
  }
}

// training: dx = gamma * invstd * (dy - sum(dy)/R - xhat * sum(dy*xhat)/R) ; eval: dx = gamma * invstd * dy
__global__ void pv_bn_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n,
                                int C, float invR, int eval, const float* __restrict__ gamma, const float* __restrict__ stats) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const float inv = stats[C + c];
    float g = dy[e];
    if (!eval) g = g - stats[2 * C + c] * invR - ((x[e] - stats[c]) * inv) * stats[3 * C + c] * invR;
    dx[e] = gamma[c] * inv * g;
  }
}

static int bn_cb(int C) { int cb = 1; while (cb < C && cb < 256) cb <<= 1; return cb; }
static int bn_blocks(int64_t R) { return (int)(R < BN_BLOCKS ? (R > 0 ? R : 1) : BN_BLOCKS); }

int64_t pv_bn_ws(int64_t R, int C) { return (int64_t)bn_blocks(R) * 2 * C * (int64_t)sizeof(float) + 256; }

int pv_bn_fwd(const float* x, float* y, int64_t R, int C, const float* gamma, const float* beta, float* rmean, float* rvar,
              int eval, float momentum, float eps, float* stats, void* ws, int64_t ws_bytes, hipStream_t s) {
  if (C < 1 || C > BN_MAXC || R < 1 || ws_bytes < pv_bn_ws(R, C)) return C > BN_MAXC ? PV_EINVAL : PV_EWS;
  float* part = reinterpret_cast<float*>(ws);
  const int nb = bn_blocks(R), cb = bn_cb(C), cg = (C + 63) / 64;
  if (eval) {
    hipLaunchKernelGGL(pv_bn_running_kernel, dim3(cg), dim3(64), 0, s, rmean, rvar, C, eps, stats);
  } else {
    hipLaunchKernelGGL(pv_bn_reduce_kernel<0>, dim3(nb), dim3(256), 0, s, x, (const float*)nullptr, R, C, cb, stats, part);
    hipLaunchKernelGGL(pv_bn_finish_kernel, dim3(cg), dim3(64), 0, s, part, nb, R, C, 0, stats, (float*)nullptr, (float*)nullptr,
                       momentum, eps, (float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(pv_bn_reduce_kernel<1>, dim3(nb), dim3(256), 0, s, x, (const float*)nullptr, R, C, cb, stats, part);
    hipLaunchKernelGGL(pv_bn_finish_kernel, dim3(cg), dim3(64), 0, s, part, nb, R, C, 1, stats, rmean, rvar, momentum, eps,
                       (float*)nullptr, (float*)nullptr);
  }
  PV_LAUNCH_CHECK();
  CONV_LAUNCH(pv_bn_apply_kernel, R * C, x, y, R * C, C, gamma, beta, stats);
}

int pv_bn_bwd(const float* x, const float* dy, float* dx, int64_t R, int C, const float* gamma, const float* stats_, int eval,
              float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, hipStream_t s) {
  if (C < 1 || C > BN_MAXC || R < 1 || ws_bytes < pv_bn_ws(R, C)) return C > BN_MAXC ? PV_EINVAL : PV_EWS;
  float* stats = const_cast<float*>(stats_);          // [2C, 4C): the two gradient sums, written here
  float* part = reinterpret_cast<float*>(ws);
  const int nb = bn_blocks(R), cb = bn_cb(C), cg = (C + 63) / 64;
  hipLaunchKernelGGL(pv_bn_reduce_kernel<2>, dim3(nb), dim3(256), 0, s, x, dy, R, C, cb, stats, part);
  hipLaunchKernelGGL(pv_bn_finish_kernel, dim3(cg), dim3(64), 0, s, part, nb, R, C, 2, stats, (float*)nullptr, (float*)nullptr,
                     0.0f, 0.0f, dgamma, dbeta);
  PV_LAUNCH_CHECK();
  if (!dx) return 0;
  CONV_LAUNCH(pv_bn_dx_kernel, R * C, x, dy, dx, R * C, C, 1.0f / (float)R, eval, gamma, stats);
}
