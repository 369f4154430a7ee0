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
                                       float* __restrict__ din, int B, int H, int W, int C, int nd) {
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
    din[e] = g;
  }
}

// the same for even H (and W) and C % 4 == 0: one thread per (window, 4 channels) — reads the window once, 16-byte accesses
__global__ void pv_maxpool2_bwd4_kernel(const float* __restrict__ in, const float* __restrict__ dout, float* __restrict__ din,
                                        int B, int H, int W, int C, int nd) {
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
        for (int i = 0; i < 4; ++i) o[i] = best[i] == k ? g[i] : 0.0f;
        *reinterpret_cast<f32x4*>(din + ((b * H + 2 * oy + dy) * W + (nd == 2 ? 2 * ox + dx : 0)) * C + 4 * c4) = o;
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
__global__ void pv_upsample2_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int H, int W,
                                        int C, int nd) {
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

// out[b][s][c] = in[b][c][s]   (to_nsc)   /   out[b][c][s] = in[b][s][c]   (to_ncs)
__global__ void pv_ncs_nsc_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t B, int C, int64_t S,
                                  int to_nsc) {
  const int64_t total = B * C * S;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    if (to_nsc) {
      const int c = (int)(e % C);
      const int64_t s = (e / C) % S, b = e / ((int64_t)C * S);
      out[e] = in[(b * C + c) * S + s];
    } else {
      const int64_t s = e % S;
      const int c = (int)((e / S) % C);
      const int64_t b = e / ((int64_t)C * S);
      out[e] = in[(b * S + s) * C + c];
    }
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
int pv_maxpool2_bwd(const float* in, const float* dout, float* din, int B, int H, int W, int C, int nd, hipStream_t s) {
  if (C % 4 == 0 && H % 2 == 0 && (nd == 1 || W % 2 == 0))
    CONV_LAUNCH(pv_maxpool2_bwd4_kernel, (int64_t)B * (H / 2) * (nd == 2 ? W / 2 : 1) * (C / 4), in, dout, din, B, H, W, C, nd);
  CONV_LAUNCH(pv_maxpool2_bwd_kernel, (int64_t)B * H * W * C, in, dout, din, B, H, W, C, nd);
}
int pv_upsample2_fwd(const float* in, float* out, int B, int H, int W, int C, int nd, hipStream_t s) {
  CONV_LAUNCH(pv_upsample2_fwd_kernel, (int64_t)B * 2 * H * (nd == 2 ? 2 * W : 1) * C, in, out, B, H, W, C, nd);
}
int pv_upsample2_bwd(const float* dout, float* din, int B, int H, int W, int C, int nd, hipStream_t s) {
  CONV_LAUNCH(pv_upsample2_bwd_kernel, (int64_t)B * H * W * C, dout, din, B, H, W, C, nd);
}
int pv_ncs_to_nsc(const float* in, float* out, int64_t B, int C, int64_t S, hipStream_t s) {
  CONV_LAUNCH(pv_ncs_nsc_kernel, B * C * S, in, out, B, C, S, 1);
}
int pv_nsc_to_ncs(const float* in, float* out, int64_t B, int C, int64_t S, hipStream_t s) {
  CONV_LAUNCH(pv_ncs_nsc_kernel, B * C * S, in, out, B, C, S, 0);
}
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
