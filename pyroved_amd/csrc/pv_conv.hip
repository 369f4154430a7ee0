// pv_conv.hip — the data-movement kernels around the convolutional nets of models.VED (nets/conv.py): everything a
// conv / pool / upsample stack needs besides its GEMMs.  Activations are channels-last, [B][spatial...][C], so a
// convolution is  Y[B*S, Cout] = act(col[B*S, Cin*k^d] W^T + b)  with the torch weight (Cout, Cin, *kernel) used as
// it lies in memory: the column index of `col` is ci*k^d + tap, taps in the kernel's own (ky, kx) order.
// All of these are HBM-bound gathers / scatters written as gathers (one thread per OUTPUT element, no atomics):
//   im2col / col2im      (kernel 3, stride 1, padding 1; 1-D and 2-D)
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

// col[(b, y, x)][ci*KK + t] = in[b][y + dy(t)][x + dx(t)][ci]   (zero outside); W = 1, KK = 3 for 1-D
__global__ void pv_im2col3_kernel(const float* __restrict__ in, float* __restrict__ col, int B, int H, int W, int C,
                                  int nd) {
  const int KK = nd == 2 ? 9 : 3;
  const int64_t total = (int64_t)B * H * W * C * KK;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(e % KK);
    const int ci = (int)((e / KK) % C);
    const int64_t row = e / ((int64_t)KK * C);
    const int x = (int)(row % W), y = (int)((row / W) % H);
    const int64_t b = row / ((int64_t)W * H);
    // 2-D: t = ky*3 + kx over (H, W); 1-D: the single spatial axis is H (W == 1), t = k
    const int yy = y + (nd == 2 ? t / 3 : t) - 1, xx = nd == 2 ? x + t % 3 - 1 : x;
    float v = 0.0f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = in[((b * H + yy) * W + xx) * C + ci];
    col[e] = v;
  }
}

// din[b][y][x][ci] = sum_t dcol[(b, y - dy(t), x - dx(t))][ci*KK + t]
__global__ void pv_col2im3_kernel(const float* __restrict__ dcol, float* __restrict__ din, int B, int H, int W, int C,
                                  int nd) {
  const int KK = nd == 2 ? 9 : 3;
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(e % C);
    const int64_t pix = e / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    float v = 0.0f;
    for (int t = 0; t < KK; ++t) {
      const int yy = y - ((nd == 2 ? t / 3 : t) - 1), xx = nd == 2 ? x - (t % 3 - 1) : x;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v += dcol[(((b * H + yy) * W + xx) * C + ci) * KK + t];
    }
    din[e] = v;
  }
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

#define CONV_LAUNCH(kernel, n, ...)                                                                   \
  do {                                                                                                \
    if ((n) > 0) hipLaunchKernelGGL(kernel, dim3(conv_blocks(n)), dim3(CONV_THREADS), 0, s, __VA_ARGS__); \
    PV_LAUNCH_CHECK();                                                                                \
    return 0;                                                                                         \
  } while (0)

int pv_im2col3(const float* in, float* col, int B, int H, int W, int C, int nd, hipStream_t s) {
  CONV_LAUNCH(pv_im2col3_kernel, (int64_t)B * H * W * C * (nd == 2 ? 9 : 3), in, col, B, H, W, C, nd);
}
int pv_col2im3(const float* dcol, float* din, int B, int H, int W, int C, int nd, hipStream_t s) {
  CONV_LAUNCH(pv_col2im3_kernel, (int64_t)B * H * W * C, dcol, din, B, H, W, C, nd);
}
int pv_maxpool2_fwd(const float* in, float* out, int B, int H, int W, int C, int nd, hipStream_t s) {
  CONV_LAUNCH(pv_maxpool2_fwd_kernel, (int64_t)B * (H / 2) * (nd == 2 ? W / 2 : 1) * C, in, out, B, H, W, C, nd);
}
int pv_maxpool2_bwd(const float* in, const float* dout, float* din, int B, int H, int W, int C, int nd, hipStream_t s) {
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
