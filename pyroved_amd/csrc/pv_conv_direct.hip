// pv_conv_direct.hip — kernel-3 (padding 1, stride 1) convolution over channels-last activations as a DIRECT
// convolution on the f32-input matrix cores: the forward of nn.ConvNd in nets/conv.py's FeatureExtractor / Upsampler
// stacks, and — with the taps flipped and the channel roles swapped — its input gradient.
//
// pv_gemm.hip's implicit-im2col form gathers every A element through (row -> b,y,x ; column -> ci,tap) index
// arithmetic and re-reads each input pixel for its 9 taps from L2.  Here a workgroup owns an 8x8 (2-D) / 64x1 (1-D)
// tile of output pixels x 64 output channels and walks the input channels in chunks of 16: the tile's input patch WITH
// ITS HALO (10x10 pixels x 16 channels) and the chunk's weights (taps x 64 x 16) are staged in LDS once and all 9 taps
// contract out of LDS.  D[co][pixel] = sum_{tap, ci} W[co][ci][tap] * in[pixel + tap][ci] (weights are the MFMA A
// operand, the patch the B operand), so a lane ends up with 4 consecutive output channels of one pixel: bias +
// activation + one 16-byte store.  Within a k-step s the MFMA's k slot kq stands for channel 4*kq + s of the chunk:
// both operands are then single ds_read_b128 out of the natural [row][16 channels] LDS layouts.
// The weights come pre-tiled ([co tile][chunk][tap][64][16], zero-padded) from pv_conv3_wprep, which also does the
// flip / role swap of the dgrad form, so staging them is a straight 16-byte copy.
#include "pv_common.h"
#include "pv_side.h"
#include "pv_conv.h"
#include <stdlib.h>

#define CD_TN 64                 // output channels per workgroup
#define CD_KC 16                 // input channels per stage
#define CD_PIX 64                // output pixels per workgroup
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct ConvD {
  const float* in; const float* wt; const float* bias; float* out;
  const float* eg_y; int eg_act;      // optional: out *= act'(eg_y) elementwise (the producing layer's activation backward)
  int B, H, W, Cin, Cout, nd, act, KK, tiles_x, tiles_y;
  int simple; float slope, gslope;    // piecewise-linear activations (none / relu / lrelu, both ways): one branch-free epilogue path
  int spt;                            // 1-D, bf16 / fp16 kernel: samples per 64-position tile (short signals share a tile, each with its own halo rows)
};

// epilogue of the tile kernels: four consecutive output channels co .. co + 3 of one pixel (orow = its output row).
// Piecewise-linear activations (none / relu / lrelu, forward and derivative) take an inline branch-free path; everything
// else goes through ONE out-of-line copy of the activation switch (inlined per value it was ~6 k of a kernel's 7.8 k
// instructions, 36 branches per value on the executed path).
__device__ __noinline__ void cd_store4_generic(const float* bias, const float* eg_y, float* orow, int co, int Cout, int act, int eg_act,
                                               f32x4 v) {
  for (int i = 0; i < 4; ++i)
    if (co + i < Cout) {
      float t = pv_act_fwd(v[i] + (bias ? bias[co + i] : 0.0f), act);
      if (eg_y) t *= pv_act_grad(eg_y[co + i], 0.0f, eg_act);
      orow[co + i] = t;
    }
}
__device__ __forceinline__ void cd_store4(const ConvD& p, float* orow, int co, f32x4 v) {
  if (p.simple && co + 3 < p.Cout && (p.Cout & 3) == 0) {
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.0f ? v[i] : v[i] * p.slope;
    if (p.eg_y) {
      const f32x4 yy = *reinterpret_cast<const f32x4*>(p.eg_y + (orow - p.out) + co);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= yy[i] > 0.0f ? 1.0f : p.gslope;
    }
    *reinterpret_cast<f32x4*>(orow + co) = v;
  } else {
    cd_store4_generic(p.bias, p.eg_y ? p.eg_y + (orow - p.out) : nullptr, orow, co, p.Cout, p.act, p.eg_act, v);
  }
}

// raw torch weight w[Co][Ci][KK] -> tiled logical matrix Wl[n][c][t]:
//   flip == 0 (forward):  Wl[n = co][c = ci][t] = w[co][ci][t]                (N = Co, C = Ci)
//   flip == 1 (dgrad):    Wl[n = ci][c = co][t] = w[co][ci][KK - 1 - t]       (N = Ci, C = Co)
// laid out [n tile][chunk][t][64][16], rows n >= N zero
__global__ void pv_conv3_wprep_kernel(const float* __restrict__ w, float* __restrict__ wt, int Co, int Ci, int KK, int flip) {
  const int N = flip ? Ci : Co, C = flip ? Co : Ci;
  const int nt = (N + CD_TN - 1) / CD_TN, nch = C / CD_KC;
  const int64_t total = (int64_t)nt * nch * KK * CD_TN * CD_KC;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int cl = (int)(e % CD_KC), nl = (int)((e / CD_KC) % CD_TN), t = (int)((e / (CD_KC * CD_TN)) % KK);
    const int ch = (int)((e / ((int64_t)CD_KC * CD_TN * KK)) % nch), tile = (int)(e / ((int64_t)CD_KC * CD_TN * KK * nch));
    const int n = tile * CD_TN + nl, c = ch * CD_KC + cl;
    float v = 0.0f;
    if (n < N) v = flip ? w[((int64_t)c * Ci + n) * KK + (KK - 1 - t)] : w[((int64_t)n * Ci + c) * KK + t];
    wt[e] = v;
  }
}

__global__ __launch_bounds__(256) void pv_conv3_direct_kernel(ConvD p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KK = p.KK, PW = p.nd == 2 ? 10 : 1, PH = p.nd == 2 ? 10 : CD_PIX + 2, NPIX = PH * PW;
  constexpr int TG = 3;                              // taps per weight stage: one kernel row (2-D) / all three (1-D)
  float* wl = smem;                                  // [TG][64][16]
  float* patch = smem + TG * CD_TN * CD_KC;          // [NPIX][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;           // pixel half, channel half of the 64 x 64 tile
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
  const int y0 = ty * (p.nd == 2 ? 8 : CD_PIX), x0 = tx * 8;
  const int cot = blockIdx.y;
  const int nch = p.Cin / CD_KC;
  const float* in_b = p.in + (int64_t)b * p.H * p.W * p.Cin;
  f32x4 acc[2][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // this lane's two B rows (output pixels) as patch indices of tap (0, 0)
  int pidx[2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int n = wm * 32 + pb * 16 + r;
    pidx[pb] = p.nd == 2 ? (n >> 3) * PW + (n & 7) : n;
  }
  // staging through registers, one stage ahead: the next chunk's patch pieces (<= 2 per thread) and the next tap group's
  // weights (3 per thread) are in flight while the current stage's MFMAs run — these layers are small (a 1-D decoder
  // layer is 0.4 GFLOP) and their time is the serial chain of stages, not FLOPs
  constexpr int PKD = 2;                              // ceil(100 * 4 / 256) (2-D), ceil(66 * 4 / 256) (1-D)
  int pg[PKD], pl_[PKD];                              // element offset of the piece in the image (-1: zero), LDS float offset
#pragma unroll
  for (int k = 0; k < PKD; ++k) {
    const int e = tid + 256 * k, pix = e >> 2, f4 = e & 3;
    const int py = p.nd == 2 ? pix / PW : pix, px = p.nd == 2 ? pix - py * PW : 0;
    const int y = y0 - 1 + py, x = p.nd == 2 ? x0 - 1 + px : 0;
    pl_[k] = e < NPIX * 4 ? pix * CD_KC + 4 * f4 : -1;
    pg[k] = (e < NPIX * 4 && y >= 0 && y < p.H && x >= 0 && x < p.W) ? (y * p.W + x) * p.Cin + 4 * f4 : -1;
  }
  f32x4 pv[PKD], wv[3];
  const int ngrp = KK / TG;                           // weight stages per chunk
#define CD_FETCH_P(CH)                                                                                               \
  _Pragma("unroll") for (int k = 0; k < PKD; ++k) {                                                                  \
    pv[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};                                                                           \
    if (pg[k] >= 0) pv[k] = *reinterpret_cast<const f32x4*>(in_b + pg[k] + (CH) * CD_KC);                            \
  }
#define CD_FETCH_W(STAGE)                                                                                            \
  {                                                                                                                  \
    const f32x4* src_ = reinterpret_cast<const f32x4*>(p.wt + ((int64_t)cot * nch * KK + (int64_t)(STAGE) * TG) * CD_TN * CD_KC); \
    _Pragma("unroll") for (int k = 0; k < 3; ++k) wv[k] = src_[tid + 256 * k];                                       \
  }
  CD_FETCH_P(0);
  CD_FETCH_W(0);
  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();                                 // the previous chunk's reads are done
#pragma unroll
    for (int k = 0; k < PKD; ++k)
      if (pl_[k] >= 0) *reinterpret_cast<f32x4*>(patch + pl_[k]) = pv[k];
    for (int g = 0; g < ngrp; ++g) {
    const int tg = g * TG;
    if (g > 0) __syncthreads();                      // the previous tap group's reads of the weights are done
#pragma unroll
    for (int k = 0; k < 3; ++k) reinterpret_cast<f32x4*>(wl)[tid + 256 * k] = wv[k];
    __syncthreads();
    {
      const int stage = ch * ngrp + g;
      if (stage + 1 < nch * ngrp) CD_FETCH_W(stage + 1);
      if (g + 1 == ngrp && ch + 1 < nch) CD_FETCH_P(ch + 1);
    }
#pragma unroll
    for (int tt = 0; tt < TG; ++tt) {
      const int tap = tg + tt;
      const int toff = p.nd == 2 ? (tap / 3) * PW + (tap % 3) : tap;
      f32x4 a[2], bb[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        a[cb] = *reinterpret_cast<const f32x4*>(wl + ((tt * CD_TN) + wn * 32 + cb * 16 + r) * CD_KC + 4 * q);
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
        bb[pb] = *reinterpret_cast<const f32x4*>(patch + (pidx[pb] + toff) * CD_KC + 4 * q);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = MFMA4(a[cb][s], bb[pb][s], acc[cb][pb]);
    }
    }
  }
  // C/D layout: lane (column = pixel r, q), reg i -> output channel 16*cb + 4q + i of the wave's half
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int n = wm * 32 + pb * 16 + r;
    const int y = p.nd == 2 ? y0 + (n >> 3) : y0 + n, x = p.nd == 2 ? x0 + (n & 7) : 0;
    if (y >= p.H || x >= p.W) continue;
    float* orow = p.out + (((int64_t)b * p.H + y) * p.W + x) * p.Cout;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int co = cot * CD_TN + wn * 32 + cb * 16 + 4 * q;
      cd_store4(p, orow, co, acc[cb][pb]);
    }
  }
}

// (bf16 split-precision forms: defined at the end of this file)
template <bool F16> __global__ void pv_conv3_wprep_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wt, int Co, int Ci, int KK, int flip);
template <bool F16, int NCH> __global__ __launch_bounds__(256, 2) void pv_conv3_direct_bf16_kernel(ConvD p);

bool pv_conv3_direct_supported(int C, int Cout, int nd, int act) {
  return C >= CD_KC && C % CD_KC == 0 && Cout >= 8 && (nd == 1 || nd == 2) && act != PV_ACT_GELU;
}

int64_t pv_conv3_direct_wt_floats(int C, int Cout, int nd) {
  const int KK = nd == 2 ? 9 : 3;
  const int64_t n = Cout > C ? Cout : C;            // either orientation (forward / dgrad) fits
  return ((n + CD_TN - 1) / CD_TN) * CD_TN * (int64_t)(Cout > C ? Cout : C) * KK + 64;
}

// w: raw torch weight (Co, Ci, KK).  flip == 0: out[.., Co] = act(conv(in[.., Ci]) + bias).
// flip == 1: out[.., Ci] = conv of in[.., Co] with the flipped / role-swapped weights (the input gradient).
int pv_conv3_direct(const float* in, int B, int H, int W, int nd, const float* w, int Co, int Ci, int flip, const float* bias,
                    float* out, int act, float* wt_scratch, hipStream_t s, const float* eg_y, int eg_act, int use_bf16,
                    const void* wt_ready) {
  const int KK = nd == 2 ? 9 : 3;
  const int N = flip ? Ci : Co, C = flip ? Co : Ci;
  if (!pv_conv3_direct_supported(C, N, nd, act)) return PV_EINVAL;
  const int nt = (N + CD_TN - 1) / CD_TN;
  const int64_t total = (int64_t)nt * (C / CD_KC) * KK * CD_TN * CD_KC;
  int pb = (int)((total + 255) / 256);
  if (pb > 2048) pb = 2048;
  const bool bf16 = use_bf16 && C % 32 == 0;
  if (wt_ready) {                                    // tiled once per step by pv_conv_wprep_table
    wt_scratch = const_cast<float*>(reinterpret_cast<const float*>(wt_ready));
  } else {
    if (bf16 && use_bf16 == 2)   // (the hi + lo 16-bit arrays take the same bytes as the fp32 tiling)
      hipLaunchKernelGGL(pv_conv3_wprep_bf16_kernel<true>, dim3(pb), dim3(256), 0, s, w, reinterpret_cast<__bf16*>(wt_scratch), Co,
                         Ci, KK, flip);
    else if (bf16)
      hipLaunchKernelGGL(pv_conv3_wprep_bf16_kernel<false>, dim3(pb), dim3(256), 0, s, w, reinterpret_cast<__bf16*>(wt_scratch), Co,
                         Ci, KK, flip);
    else
      hipLaunchKernelGGL(pv_conv3_wprep_kernel, dim3(pb), dim3(256), 0, s, w, wt_scratch, Co, Ci, KK, flip);
    PV_LAUNCH_CHECK();
  }
  ConvD p{};
  p.in = in; p.wt = wt_scratch; p.bias = bias; p.out = out;
  p.eg_y = (eg_y && eg_act != PV_ACT_NONE) ? eg_y : nullptr; p.eg_act = eg_act;
  p.B = B; p.H = H; p.W = W; p.Cin = C; p.Cout = N; p.nd = nd; p.act = act; p.KK = KK;
  p.tiles_x = nd == 2 ? (W + 7) / 8 : 1;
  p.tiles_y = nd == 2 ? (H + 7) / 8 : (H + CD_PIX - 1) / CD_PIX;
  {
    auto lin = [](int a_) { return a_ == PV_ACT_NONE || a_ == PV_ACT_RELU || a_ == PV_ACT_LRELU; };
    auto slope = [](int a_) { return a_ == PV_ACT_NONE ? 1.0f : a_ == PV_ACT_RELU ? 0.0f : 0.01f; };
    const int eg = p.eg_y ? p.eg_act : PV_ACT_NONE;
    p.simple = lin(p.act) && lin(eg);
    p.slope = slope(p.act); p.gslope = slope(eg);
  }
  const int npix = nd == 2 ? 100 : CD_PIX + 2;
  if (bf16) {
    // 1-D signals of <= 32 positions: floor(64 / H) samples per tile (at most 8: the patch then has <= 80 rows) — only when
    // one sample per tile would be more than two rounds of workgroups: a workgroup's time is its chain of K-loop stages
    // whatever the tile holds, so below that the emptier tiles cost nothing and packing only removes parallelism
    // (measured on VED C5, batch 256: +10 us per step with packing).  PV_PACK1D=0 / 1: never / always.
    static const int pack_raw = pv_exp_int("PV_PACK1D", -1);
    const int pack_env = pack_raw < 0 ? -1 : (pack_raw != 0 ? 1 : 0);
    p.spt = 1;
    if (nd == 1 && H >= 8 && H <= CD_PIX / 2 && (pack_env == 1 || (pack_env == -1 && (int64_t)B * nt > 1024))) p.spt = CD_PIX / H;
    const int npb = p.spt > 1 ? p.spt * (H + 2) : npix;
    const size_t ldsb = (size_t)(2 * 3 * CD_TN * 32 + 2 * npb * 32) * 2;
    const dim3 gridb((unsigned)(p.spt > 1 ? (B + p.spt - 1) / p.spt : p.tiles_x * p.tiles_y * B), (unsigned)nt);
    // PV_RES1D=1: the all-chunks-in-flight form (NCH > 0).  Off by default: measured +10 us per VED C5 step against the
    // streaming form once the epilogue's code bloat was gone — the stages were never memory-latency chains
    static const int res_env = pv_exp_int("PV_RES1D", 0) != 0 ? 1 : 0;
    const int nchr = (res_env && nd == 1 && C / 32 <= 4) ? C / 32 : 0;
#define CB_LAUNCH(F, N) PV_LAUNCH_FORK((pv_conv3_direct_bf16_kernel<F, N>), gridb, dim3(256), ldsb, s, p)   /* (carries an armed fork event: pv_side.h) */
    if (use_bf16 == 2) {
      if (nchr == 4) CB_LAUNCH(true, 4); else if (nchr == 3) CB_LAUNCH(true, 3); else if (nchr == 2) CB_LAUNCH(true, 2);
      else if (nchr == 1) CB_LAUNCH(true, 1); else CB_LAUNCH(true, 0);
    } else {
      if (nchr == 4) CB_LAUNCH(false, 4); else if (nchr == 3) CB_LAUNCH(false, 3); else if (nchr == 2) CB_LAUNCH(false, 2);
      else if (nchr == 1) CB_LAUNCH(false, 1); else CB_LAUNCH(false, 0);
    }
#undef CB_LAUNCH
    PV_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = (size_t)(3 * CD_TN * CD_KC + npix * CD_KC) * sizeof(float);
  PV_LAUNCH_FORK(pv_conv3_direct_kernel, dim3((unsigned)(p.tiles_x * p.tiles_y * B), (unsigned)nt), dim3(256), lds, s, p);
  PV_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolution, direct form:
//   dW[co][ci][tap] = sum_{b, pixel} dY[b, pixel][co] * in[b, pixel + tap][ci] ;   db[co] = sum dY[b, pixel][co]
// A workgroup owns (64 output channels) x (16 input channels) x all taps and walks a contiguous range of 8x8 / 64x1
// pixel tiles (split s of nsplit): per tile the dY tile (64 pixels x 64 channels) and the input patch with its halo
// (x 16 channels) are staged in LDS, the contraction over the tile's pixels runs on the f32-input matrix cores
// (A = dY^T, B = the patch shifted by the tap), and the 9 tap blocks of the wave's 16 output channels stay in
// accumulators across tiles.  Partial results per split are summed in split order afterwards (no atomics).
#define WG_LDY 68                // LDS row strides (floats): 2-way instead of 4-way bank conflicts on the scalar reads
#define WG_LDP 20

struct ConvWg {
  const float* dy; const float* in; float* part; float* part_b;
  int B, H, W, Cin, Cout, nd, KK, tiles_x, tiles_y, nsplit;
};

__global__ __launch_bounds__(256) void pv_conv3_wgrad_direct_kernel(ConvWg p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KK = p.KK, PW = p.nd == 2 ? 10 : 1, PH = p.nd == 2 ? 10 : CD_PIX + 2, NPIX = PH * PW;
  float* dyl = smem;                                 // [64 pixels][WG_LDY]
  float* patch = smem + CD_PIX * WG_LDY;             // [NPIX][WG_LDP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int split = blockIdx.x, ch = blockIdx.y, cot = blockIdx.z;
  const int64_t T = (int64_t)p.B * p.tiles_y * p.tiles_x;
  const int64_t t_lo = T * split / p.nsplit, t_hi = T * (split + 1) / p.nsplit;
  f32x4 acc[9], accb = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int64_t tt = t_lo; tt < t_hi; ++tt) {
    int t = (int)tt;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
    const int y0 = ty * (p.nd == 2 ? 8 : CD_PIX), x0 = tx * 8;
    __syncthreads();
    for (int e = tid; e < CD_PIX * 16; e += 256) {               // dY tile: pixel n, channels 4*c4 .. 4*c4+3
      const int n = e >> 4, c4 = e & 15;
      const int y = p.nd == 2 ? y0 + (n >> 3) : y0 + n, x = p.nd == 2 ? x0 + (n & 7) : 0;
      const int co = cot * CD_TN + 4 * c4;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (y < p.H && x < p.W) {
        const float* src = p.dy + (((int64_t)b * p.H + y) * p.W + x) * p.Cout + co;
        if (co + 3 < p.Cout && (p.Cout & 3) == 0) v = *reinterpret_cast<const f32x4*>(src);
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i) if (co + i < p.Cout) v[i] = src[i];
        }
      }
      *reinterpret_cast<f32x4*>(dyl + n * WG_LDY + 4 * c4) = v;
    }
    const float* in_b = p.in + (int64_t)b * p.H * p.W * p.Cin;
    for (int e = tid; e < NPIX * 4; e += 256) {
      const int pix = e >> 2, f4 = e & 3;
      const int py = p.nd == 2 ? pix / PW : pix, px = p.nd == 2 ? pix - py * PW : 0;
      const int y = y0 - 1 + py, x = p.nd == 2 ? x0 - 1 + px : 0;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (y >= 0 && y < p.H && x >= 0 && x < p.W)
        v = *reinterpret_cast<const f32x4*>(in_b + ((int64_t)y * p.W + x) * p.Cin + ch * CD_KC + 4 * f4);
      *reinterpret_cast<f32x4*>(patch + pix * WG_LDP + 4 * f4) = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {                                // 16 pixels: k slot kq of step s is pixel 16g + 4kq + s
      float a[4];
      int pbase[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int n = 16 * g + 4 * q + s;
        a[s] = dyl[n * WG_LDY + 16 * wave + r];
        pbase[s] = (p.nd == 2 ? (n >> 3) * PW + (n & 7) : n) * WG_LDP + r;
      }
      if (ch == 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) accb = MFMA4(a[s], 1.0f, accb);
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap < KK) {
          const int toff = (p.nd == 2 ? (tap / 3) * PW + (tap % 3) : tap) * WG_LDP;
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[tap] = MFMA4(a[s], patch[pbase[s] + toff], acc[tap]);
        }
      }
    }
  }
  // C/D layout: lane (column = ci r, q), reg i -> output channel 16*wave + 4q + i
  const int ci = ch * CD_KC + r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = cot * CD_TN + 16 * wave + 4 * q + i;
    if (co >= p.Cout) continue;
    float* dst = p.part + (((int64_t)split * p.Cout + co) * p.Cin + ci) * KK;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      if (tap < KK) dst[tap] = acc[tap][i];
    if (ch == 0 && r == 0 && p.part_b) p.part_b[(int64_t)split * p.Cout + co] = accb[i];
  }
}

// out[e] = sum_s part[s][e] (and out_b likewise), in a fixed order: a workgroup takes 32 outputs x 8 split slices
// (slice k sums splits k, k+8, ... ; the slices then meet in LDS in slice order) — hundreds of splits are a chain of
// dependent-latency loads otherwise
__device__ __forceinline__ void wgrad_finish_block(const float* __restrict__ part, int nsplit, int64_t n, float* __restrict__ out,
                                                   const float* __restrict__ part_b, int nb, float* __restrict__ out_b,
                                                   int64_t blk, float* sm) {
  // few outputs (a first layer's 9 * Cout): 8 outputs x 32 slices per workgroup, otherwise 32 x 8 (pv_wgrad_finish_blocks)
  const int og = n + (part_b ? nb : 0) <= 2048 ? 8 : 32, nsl = 256 / og;
  const int o = threadIdx.x % og, sl = threadIdx.x / og;
  const int64_t nblk = (n + og - 1) / og;
  const bool isb = blk >= nblk;
  const int64_t e = (isb ? blk - nblk : blk) * og + o, lim = isb ? nb : n;
  const float* src = isb ? part_b : part;
  float v = 0.0f;
  if (e < lim) {
    // four independent chains (a fixed order all the same) keep several loads in flight per thread
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
    int s = sl;
    for (; s + 3 * nsl < nsplit; s += 4 * nsl) {
      v0 += src[(int64_t)s * lim + e];
      v1 += src[(int64_t)(s + nsl) * lim + e];
      v2 += src[(int64_t)(s + 2 * nsl) * lim + e];
      v3 += src[(int64_t)(s + 3 * nsl) * lim + e];
    }
    for (; s < nsplit; s += nsl) v0 += src[(int64_t)s * lim + e];
    v = (v0 + v1) + (v2 + v3);
  }
  sm[sl * og + o] = v;
  __syncthreads();
  if (sl == 0 && e < lim) {
    float t = 0.0f;
    for (int k = 0; k < nsl; ++k) t += sm[k * og + o];
    (isb ? out_b : out)[e] = t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void pv_conv3_wgrad_finish_kernel(const float* __restrict__ part, int nsplit, int64_t n,
                                                                     float* __restrict__ out, const float* __restrict__ part_b,
                                                                     int nb, float* __restrict__ out_b) {
  __shared__ float sm[256];
  const int og = n + (part_b ? nb : 0) <= 2048 ? 8 : 32;
  const int64_t total = (n + og - 1) / og + (part_b ? (nb + og - 1) / og : 0);
  for (int64_t blk = blockIdx.x; blk < total; blk += gridDim.x) wgrad_finish_block(part, nsplit, n, out, part_b, nb, out_b, blk, sm);
}

// every recorded reduction of a step in one launch: workgroup b serves entry k with blk0[k] <= b < blk0[k] + nblk[k]
struct FinTab { PvFinishEntry e[16]; int n; };
__global__ __launch_bounds__(256) void pv_wgrad_finish_table_kernel(FinTab t) {
  __shared__ float sm[256];
  int k = 0;
  while (k + 1 < t.n && (int)blockIdx.x >= t.e[k + 1].blk0) ++k;
  const PvFinishEntry E = t.e[k];
  wgrad_finish_block(E.part, E.nsplit, E.n, E.out, E.part_b, E.nb, E.out_b, (int64_t)blockIdx.x - E.blk0, sm);
}

int pv_wgrad_finish_blocks(int64_t nw, int nb) {
  const int og = nw + nb <= 2048 ? 8 : 32;
  int64_t fb = (nw + og - 1) / og + (nb ? (nb + og - 1) / og : 0);
  return (int)(fb > 4096 ? 4096 : fb);
}

bool pv_wgrad_ws(PvFinishList* list, int64_t need, void*& ws, int64_t& ws_bytes) {
  if (!list || !list->base || list->n >= 16 || list->off + need > list->cap) return false;
  ws = list->base + list->off;
  ws_bytes = need;
  list->off += pv_align_up(need, 256);
  return true;
}

int pv_wgrad_finish(PvFinishList* list, const float* part, int nsplit, int64_t n, float* out, const float* part_b, int nb,
                    float* out_b, hipStream_t s) {
  if (!part_b || !out_b) { part_b = nullptr; out_b = nullptr; nb = 0; }
  if (list && list->n < 16) {
    const int og = n + nb <= 2048 ? 8 : 32;
    PvFinishEntry& E = list->e[list->n];
    E.part = part; E.out = out; E.part_b = part_b; E.out_b = out_b; E.n = n; E.nsplit = nsplit; E.nb = nb;
    E.nblk = (int)((n + og - 1) / og + (nb ? (nb + og - 1) / og : 0));
    E.blk0 = list->n ? list->e[list->n - 1].blk0 + list->e[list->n - 1].nblk : 0;
    list->st[list->n] = s;
    ++list->n;
    return 0;
  }
  hipLaunchKernelGGL(pv_conv3_wgrad_finish_kernel, dim3(pv_wgrad_finish_blocks(n, nb)), dim3(256), 0, s, part, nsplit, n, out, part_b,
                     nb, out_b);
  PV_LAUNCH_CHECK();
  return 0;
}

static int wgrad_finish_launch(PvFinishList* list, hipStream_t s) {
  if (!list || list->n == 0) return 0;
  FinTab t{};
  t.n = list->n;
  for (int k = 0; k < list->n; ++k) t.e[k] = list->e[k];
  const int total = list->e[list->n - 1].blk0 + list->e[list->n - 1].nblk;
  list->n = 0;
  if (total < 1) return 0;
  PV_LAUNCH_FORK(pv_wgrad_finish_table_kernel, dim3((unsigned)total), dim3(256), 0, s, t);   // (carries an armed join event: pv_side.h)
  PV_LAUNCH_CHECK();
  return 0;
}

int pv_wgrad_finish_all(PvFinishList* list, hipStream_t s) {
  if (!list) return 0;
  PV_TRY(wgrad_finish_launch(list, s));
  list->off = 0;
  return 0;
}

int pv_wgrad_finish_flush(PvFinishList* list, hipStream_t s) {
  if (!list || list->n == 0) return 0;
  for (int k = 0; k < list->n; ++k) {                 // one wait per other stream
    if (list->st[k] == s) continue;
    bool seen = false;
    for (int j = 0; j < k; ++j) seen = seen || list->st[j] == list->st[k];
    if (!seen) PV_TRY(pv_stream_after(s, list->st[k]));
  }
  return wgrad_finish_launch(list, s);
}

static int wgd_splits(int B, int H, int W, int C, int Cout, int nd) {
  const int tiles_x = nd == 2 ? (W + 7) / 8 : 1, tiles_y = nd == 2 ? (H + 7) / 8 : (H + CD_PIX - 1) / CD_PIX;
  const int64_t T = (int64_t)B * tiles_x * tiles_y;
  const int64_t owners = (int64_t)(C / CD_KC) * ((Cout + CD_TN - 1) / CD_TN);
  int64_t ns = (768 + owners - 1) / owners;                     // ~3 workgroups per CU in all
  if (ns > T) ns = T;
  if (ns < 1) ns = 1;
  return (int)ns;
}

bool pv_conv3_wgrad_direct_supported(int C, int Cout, int nd) {
  return C >= CD_KC && C % CD_KC == 0 && Cout >= 8 && (nd == 1 || nd == 2);
}

int64_t pv_conv3_wgrad_direct_ws(int B, int H, int W, int C, int Cout, int nd) {
  if (!pv_conv3_wgrad_direct_supported(C, Cout, nd)) return 0;
  const int KK = nd == 2 ? 9 : 3;
  return (int64_t)wgd_splits(B, H, W, C, Cout, nd) * ((int64_t)Cout * C * KK + Cout) * (int64_t)sizeof(float) + 256;
}

int pv_conv3_wgrad_direct(const float* dy, const float* in, int B, int H, int W, int C, int nd, float* dw, float* db, int Cout,
                          void* ws, int64_t ws_bytes, hipStream_t s, PvFinishList* defer) {
  if (!pv_conv3_wgrad_direct_supported(C, Cout, nd)) return PV_EINVAL;
  if (!pv_wgrad_ws(defer, pv_conv3_wgrad_direct_ws(B, H, W, C, Cout, nd), ws, ws_bytes)) defer = nullptr;
  if (ws_bytes < pv_conv3_wgrad_direct_ws(B, H, W, C, Cout, nd)) return PV_EWS;
  const int KK = nd == 2 ? 9 : 3;
  ConvWg p{};
  p.dy = dy; p.in = in; p.B = B; p.H = H; p.W = W; p.Cin = C; p.Cout = Cout; p.nd = nd; p.KK = KK;
  p.tiles_x = nd == 2 ? (W + 7) / 8 : 1;
  p.tiles_y = nd == 2 ? (H + 7) / 8 : (H + CD_PIX - 1) / CD_PIX;
  p.nsplit = wgd_splits(B, H, W, C, Cout, nd);
  const int64_t nw = (int64_t)Cout * C * KK;
  p.part = reinterpret_cast<float*>(ws);
  p.part_b = db ? p.part + (int64_t)p.nsplit * nw : nullptr;
  const int npix = nd == 2 ? 100 : CD_PIX + 2;
  const size_t lds = (size_t)(CD_PIX * WG_LDY + npix * WG_LDP) * sizeof(float);
  hipLaunchKernelGGL(pv_conv3_wgrad_direct_kernel, dim3((unsigned)p.nsplit, (unsigned)(C / CD_KC), (unsigned)((Cout + CD_TN - 1) / CD_TN)),
                     dim3(256), lds, s, p);
  PV_LAUNCH_CHECK();
  return pv_wgrad_finish(defer, p.part, p.nsplit, nw, dw, p.part_b, Cout, db, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of a kernel-3 convolution with ONE input channel (the first layer of every conv encoder:
// dW[co][tap] = sum_pixels dpre[pixel][co] * x[pixel + tap], db[co] = sum_pixels dpre[pixel][co]).  It is a pure
// reduction over B*H*W pixels with a handful of outputs: HBM-bound by the one read of dpre (4*Cout bytes per pixel),
// nothing for the matrix cores.  A workgroup takes a range of image lines; its threads are (channel, line group):
// the dpre row of a pixel is one coalesced load across the channel lanes, the 3x3 input window slides along the line
// in registers.  Per-workgroup partials go through the same fixed-order finish kernel as the direct wgrad.
#define C1_MAXCO 64
__global__ __launch_bounds__(256) void pv_conv3_wgrad_c1_kernel(const float* __restrict__ dy, const float* __restrict__ in,
                                                                int B, int H, int W, int Cout, int nd, int CP, int nsplit,
                                                                float* __restrict__ part, float* __restrict__ part_b) {
  __shared__ float sm[256][10];
  const int tid = threadIdx.x, co = tid % CP, rg = tid / CP, RG = 256 / CP;
  const int KK = nd == 2 ? 9 : 3;
  // 1-D data (H = length, W = 1) is one line per sample, taps along the line
  const int LW = nd == 2 ? W : H, LH = nd == 2 ? H : 1;
  const int64_t lines = (int64_t)B * LH;
  const int64_t l_lo = lines * blockIdx.x / nsplit, l_hi = lines * (blockIdx.x + 1) / nsplit;
  float acc[9], accb = 0.0f;
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.0f;
  const bool cok = co < Cout;
  for (int64_t l = l_lo + rg; l < l_hi; l += RG) {
    const int y = (int)(l % LH);
    const float* row1 = in + l * LW;                                   // the pixel's own line
    const bool up = nd == 2 && y > 0, dn = nd == 2 && y + 1 < LH;
    const float* row0 = row1 - LW;
    const float* row2 = row1 + LW;
    const float* dyl = dy + l * LW * (int64_t)Cout + co;
    // window columns x-1, x, x+1 of the three lines
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    a1 = up ? row0[0] : 0.0f; b1 = row1[0]; c1 = dn ? row2[0] : 0.0f;
    for (int x = 0; x < LW; ++x) {
      const bool nx = x + 1 < LW;
      a2 = (up && nx) ? row0[x + 1] : 0.0f;
      b2 = nx ? row1[x + 1] : 0.0f;
      c2 = (dn && nx) ? row2[x + 1] : 0.0f;
      const float dp = cok ? dyl[(int64_t)x * Cout] : 0.0f;
      accb += dp;
      if (nd == 2) {
        acc[0] += dp * a0; acc[1] += dp * a1; acc[2] += dp * a2;
        acc[3] += dp * b0; acc[4] += dp * b1; acc[5] += dp * b2;
        acc[6] += dp * c0; acc[7] += dp * c1; acc[8] += dp * c2;
      } else {
        acc[0] += dp * b0; acc[1] += dp * b1; acc[2] += dp * b2;
      }
      a0 = a1; a1 = a2; b0 = b1; b1 = b2; c0 = c1; c1 = c2;
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) sm[tid][t] = acc[t];
  sm[tid][9] = accb;
  __syncthreads();
  // (channel, tap) outputs: sum the line groups in group order
  for (int o = tid; o < Cout * (KK + 1); o += 256) {
    const int c = o / (KK + 1), t = o % (KK + 1);
    float v = 0.0f;
    for (int g = 0; g < RG; ++g) v += sm[g * CP + c][t == KK ? 9 : t];
    if (t == KK) { if (part_b) part_b[(int64_t)blockIdx.x * Cout + c] = v; }
    else part[(int64_t)blockIdx.x * Cout * KK + c * KK + t] = v;
  }
}

static int c1_splits(int B, int H, int W, int nd) {
  const int64_t lines = (int64_t)B * (nd == 2 ? H : 1);
  int64_t ns = 1024;
  if (ns > lines) ns = lines;
  return (int)(ns < 1 ? 1 : ns);
}
bool pv_conv3_wgrad_c1_supported(int C, int Cout, int nd) { return C == 1 && Cout >= 1 && Cout <= C1_MAXCO && (nd == 1 || nd == 2); }
int64_t pv_conv3_wgrad_c1_ws(int B, int H, int W, int C, int Cout, int nd) {
  if (!pv_conv3_wgrad_c1_supported(C, Cout, nd)) return 0;
  return (int64_t)c1_splits(B, H, W, nd) * ((int64_t)Cout * (nd == 2 ? 9 : 3) + Cout) * (int64_t)sizeof(float) + 256;
}
int pv_conv3_wgrad_c1(const float* dy, const float* in, int B, int H, int W, int nd, float* dw, float* db, int Cout, void* ws,
                      int64_t ws_bytes, hipStream_t s, PvFinishList* defer) {
  if (!pv_conv3_wgrad_c1_supported(1, Cout, nd)) return PV_EINVAL;
  if (!pv_wgrad_ws(defer, pv_conv3_wgrad_c1_ws(B, H, W, 1, Cout, nd), ws, ws_bytes)) defer = nullptr;
  if (ws_bytes < pv_conv3_wgrad_c1_ws(B, H, W, 1, Cout, nd)) return PV_EWS;
  const int KK = nd == 2 ? 9 : 3, ns = c1_splits(B, H, W, nd);
  int CP = 1;
  while (CP < Cout) CP *= 2;
  float* part = reinterpret_cast<float*>(ws);
  float* part_b = db ? part + (int64_t)ns * Cout * KK : nullptr;
  hipLaunchKernelGGL(pv_conv3_wgrad_c1_kernel, dim3(ns), dim3(256), 0, s, dy, in, B, H, W, Cout, nd, CP, ns, part, part_b);
  PV_LAUNCH_CHECK();
  const int64_t nw = (int64_t)Cout * KK;
  return pv_wgrad_finish(defer, part, ns, nw, dw, part_b, Cout, db, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same direct convolution on the bf16 matrix cores in split precision (x = hi + lo, three products, fp32
// accumulate) for layers with a multiple of 32 input channels — the mixed-precision mode (plan->conv_bf16): ~2^-16 per
// product is not enough for the 1e-4 gradient bar where a gradient is a sum with heavy cancellation (the first
// layer's weights: 7e-3), so the default stays on the f32-input MFMA kernel above.
// v_mfma_f32_16x16x32_bf16 contracts a whole 32-channel chunk of one tap per instruction, 3 instead of 8 MFMAs per
// 32 channels at half the cycles each.  Operands are split once: the weights by pv_conv3_wprep_bf16 (tiled
// [co tile][chunk][tap][64][32], a hi and a lo array), the patch while it is staged.  LDS holds the chunk's patch and
// the weights of one kernel row (3 taps) at a time: 37 KB, so several workgroups per CU overlap staging and MFMAs.
typedef __bf16 cbf8 __attribute__((ext_vector_type(8)));
typedef __bf16 cbf4 __attribute__((ext_vector_type(4)));
#define CB_KC 32
#define MFMA32B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// F16 (the fp32-class form of these kernels): two fp16 pieces of the value times 2^6, each staged patch chunk scaled by its own
// exact power of two — the scheme of pv_conv_sp.hip (see there), for the 1-D layers and the 2-D ones that kernel does not take
typedef _Float16 chf8 __attribute__((ext_vector_type(8)));
typedef __fp16 chp2 __attribute__((ext_vector_type(2)));
#define CB_WSHIFT 6
template <bool F16> __device__ __forceinline__ f32x4 cb_mma(const cbf8& a, const cbf8& b, const f32x4& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(chf8, a), __builtin_bit_cast(chf8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float cb_pow2(int k) {
  k = k < -126 ? -126 : (k > 127 ? 127 : k);
  return __uint_as_float((unsigned)(k + 127) << 23);
}

template <bool F16>
__global__ void pv_conv3_wprep_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wt, int Co, int Ci, int KK, int flip) {
  const int N = flip ? Ci : Co, C = flip ? Co : Ci;
  const int nt = (N + CD_TN - 1) / CD_TN, nch = C / CB_KC;
  const int64_t total = (int64_t)nt * nch * KK * CD_TN * CB_KC;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int cl = (int)(e % CB_KC), nl = (int)((e / CB_KC) % CD_TN), t = (int)((e / (CB_KC * CD_TN)) % KK);
    const int ch = (int)((e / ((int64_t)CB_KC * CD_TN * KK)) % nch), tile = (int)(e / ((int64_t)CB_KC * CD_TN * KK * nch));
    const int n = tile * CD_TN + nl, c = ch * CB_KC + cl;
    float v = 0.0f;
    if (n < N) v = flip ? w[((int64_t)c * Ci + n) * KK + (KK - 1 - t)] : w[((int64_t)n * Ci + c) * KK + t];
    if constexpr (F16) {
      const float t = v * (float)(1 << CB_WSHIFT);
      const _Float16 hi = (_Float16)t;
      reinterpret_cast<_Float16*>(wt)[e] = hi;
      reinterpret_cast<_Float16*>(wt)[total + e] = (_Float16)(t - (float)hi);
    } else {
      const __bf16 hi = (__bf16)v;
      wt[e] = hi;
      wt[total + e] = (__bf16)(v - (float)hi);
    }
  }
}

// NCH > 0 (1-D layers with 32 NCH input channels; experiment, PV_RES1D=1): EVERY chunk's patch pieces and weights are
// requested before the first stage (registers: 4 + 6 16-byte pieces per chunk and thread) instead of one stage ahead.
// Measured slower than the streaming form (see the launcher).
template <bool F16, int NCH>
__global__ __launch_bounds__(256, 2) void pv_conv3_direct_bf16_kernel(ConvD p) {
  extern __shared__ __attribute__((aligned(16))) char smb_[];
  __shared__ float smax[2][4];                        // F16: the waves' patch maxima of the chunk being staged
  // 1-D with p.spt > 1: the tile holds spt whole samples of H <= 32 positions; sample s of the tile owns patch rows
  // s (H + 2) ... s (H + 2) + H + 1 (its own zero halo), output pixel n = s H + l reads patch row s (H + 2) + l + tap
  const int SPT = p.nd == 2 ? 1 : p.spt, SEG = p.H + 2;
  const int KK = p.KK, PW = p.nd == 2 ? 10 : 1, PH = p.nd == 2 ? 10 : (SPT > 1 ? SPT * SEG : CD_PIX + 2), NPIX = PH * PW;
  const int TG = 3;                                  // taps per weight stage: one kernel row (2-D) / all three (1-D)
  __bf16* wh = reinterpret_cast<__bf16*>(smb_);                       // [TG][64][32]
  __bf16* wl = wh + TG * CD_TN * CB_KC;
  __bf16* ph = wl + TG * CD_TN * CB_KC;                               // [NPIX][32]
  __bf16* pl = ph + NPIX * CB_KC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int b = (t / p.tiles_y) * SPT;    // (first sample of the tile)
  const int y0 = ty * (p.nd == 2 ? 8 : CD_PIX), x0 = tx * 8;
  const int cot = blockIdx.y;
  const int nch = p.Cin / CB_KC;
  const int64_t wtot = (int64_t)gridDim.y * nch * KK * CD_TN * CB_KC;      // elements of the hi array
  const __bf16* wt = reinterpret_cast<const __bf16*>(p.wt);
  const float* in_b = p.in + (int64_t)b * p.H * p.W * p.Cin;
  f32x4 acc[2][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  int pidx[2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int n = wm * 32 + pb * 16 + r;
    pidx[pb] = p.nd == 2 ? (n >> 3) * PW + (n & 7) : (SPT > 1 ? (n < SPT * p.H ? n + 2 * (n / p.H) : 0) : n);
  }
  // staging through registers one stage ahead, as in pv_conv3_direct_kernel: the next chunk's patch pieces (<= 4 per
  // thread) and the next tap group's weights (3 + 3 16-byte pieces) are in flight under the current stage's MFMAs
  constexpr int PKB = 4;
  int pg[PKB], pls[PKB];
#pragma unroll
  for (int k = 0; k < PKB; ++k) {
    const int e = tid + 256 * k, pix = e >> 3, f4 = e & 7;
    const int py = p.nd == 2 ? pix / PW : pix, px = p.nd == 2 ? pix - py * PW : 0;
    int y = y0 - 1 + py, x = p.nd == 2 ? x0 - 1 + px : 0, sb = 0;
    if (SPT > 1) { sb = pix / SEG; y = pix - sb * SEG - 1; }
    pls[k] = e < NPIX * 8 ? pix * CB_KC + 4 * f4 : -1;
    pg[k] = (e < NPIX * 8 && b + sb < p.B && y >= 0 && y < p.H && x >= 0 && x < p.W) ? ((sb * p.H + y) * p.W + x) * p.Cin + 4 * f4 : -1;
  }
  f32x4 pv[PKB];
  int4 wvh0, wvh1, wvh2, wvl0, wvl1, wvl2;         // (an indexed register array here ends up in scratch)
  const int ngrp = KK / TG;
#define CB_FETCH_P(CH)                                                                                               \
  _Pragma("unroll") for (int k = 0; k < PKB; ++k) {                                                                  \
    pv[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};                                                                           \
    if (pg[k] >= 0) pv[k] = *reinterpret_cast<const f32x4*>(in_b + pg[k] + (CH) * CB_KC);                            \
  }
#define CB_FETCH_W(STAGE)                                                                                            \
  {                                                                                                                  \
    const int64_t src0_ = ((int64_t)cot * nch * KK + (int64_t)(STAGE) * TG) * CD_TN * CB_KC;                         \
    const int4* sh_ = reinterpret_cast<const int4*>(wt + src0_);                                                     \
    const int4* sl_ = reinterpret_cast<const int4*>(wt + wtot + src0_);                                              \
    wvh0 = sh_[tid]; wvh1 = sh_[tid + 256]; wvh2 = sh_[tid + 512];                                                   \
    wvl0 = sl_[tid]; wvl1 = sl_[tid + 256]; wvl2 = sl_[tid + 512];                                                   \
  }
  CB_FETCH_P(0);
  CB_FETCH_W(0);
  constexpr int NA = NCH > 1 ? NCH - 1 : 1;
  f32x4 pva[NA][PKB];
  int4 wva[NA][6];
  if constexpr (NCH > 1) {                             // chunks 1 .. NCH-1 (ngrp == 1 here: a stage is a chunk)
#pragma unroll
    for (int c = 1; c < NCH; ++c) {
#pragma unroll
      for (int k = 0; k < PKB; ++k) {
        pva[c - 1][k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (pg[k] >= 0) pva[c - 1][k] = *reinterpret_cast<const f32x4*>(in_b + pg[k] + c * CB_KC);
      }
      const int64_t src0_ = ((int64_t)cot * nch * KK + (int64_t)c * TG) * CD_TN * CB_KC;
      const int4* sh_ = reinterpret_cast<const int4*>(wt + src0_);
      const int4* sl_ = reinterpret_cast<const int4*>(wt + wtot + src0_);
#pragma unroll
      for (int j = 0; j < 3; ++j) { wva[c - 1][j] = sh_[tid + 256 * j]; wva[c - 1][3 + j] = sl_[tid + 256 * j]; }
    }
  }
  int E_cur = 0, E_min = 1 << 20;                     // F16: the patch scale 2^E of the current chunk, the smallest so far
  auto wave_max = [&](int slot) {
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < PKB; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) m = fmaxf(m, fabsf(pv[k][i]));
    m = pv_wave_max_nonneg(m);
    if (lane == 0) smax[slot][wave] = m;
    __threadfence_block();                            // (the write must have landed before the next barrier lets the others read:
                                                      //  hipcc 7.2 emits no s_waitcnt lgkmcnt between this store and the loop's s_barrier)
  };
  if constexpr (F16) wave_max(0);
#pragma unroll(NCH > 0 ? NCH : 1)
  for (int ch = 0; ch < (NCH > 0 ? NCH : nch); ++ch) {
    __syncthreads();                                 // the previous chunk's reads of the patch are done (F16: smax is in)
    float psc = 1.0f;
    if constexpr (F16) {
      const float m = fmaxf(fmaxf(smax[ch & 1][0], smax[ch & 1][1]), fmaxf(smax[ch & 1][2], smax[ch & 1][3]));
      const int e = (int)((__float_as_uint(m) >> 23) & 255);
      int E = E_cur;                                  // a (near-)zero chunk keeps the scale
      if (e >= 20) {
        E = 140 - e;                                  // max -> [2^13, 2^14)
        if (E_min != (1 << 20) && E > E_min + 30) E = E_min + 30;
        if (E < E_min) E_min = E;
      }
      if (ch > 0 && E != E_cur) {
        const float ratio = cb_pow2(E - E_cur);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[cb][pb] *= ratio;
      }
      E_cur = E;
      psc = cb_pow2(E);
    }
#pragma unroll
    for (int k = 0; k < PKB; ++k) {                  // patch: pixel, channels 4*f4 .. 4*f4+3 -> (hi, lo)
      cbf4 h4, l4;
      if constexpr (F16) {
        const f32x4 t = pv[k] * psc;
        const chp2 h01 = __builtin_amdgcn_cvt_pkrtz(t[0], t[1]), h23 = __builtin_amdgcn_cvt_pkrtz(t[2], t[3]);
        const chp2 l01 = __builtin_amdgcn_cvt_pkrtz(t[0] - (float)h01[0], t[1] - (float)h01[1]);
        const chp2 l23 = __builtin_amdgcn_cvt_pkrtz(t[2] - (float)h23[0], t[3] - (float)h23[1]);
        typedef unsigned int cu2 __attribute__((ext_vector_type(2)));
        h4 = __builtin_bit_cast(cbf4, cu2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)});
        l4 = __builtin_bit_cast(cbf4, cu2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)});
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const __bf16 hh = (__bf16)pv[k][i]; h4[i] = hh; l4[i] = (__bf16)(pv[k][i] - (float)hh); }
      }
      if (pls[k] >= 0) {
        *reinterpret_cast<cbf4*>(ph + pls[k]) = h4;
        *reinterpret_cast<cbf4*>(pl + pls[k]) = l4;
      }
    }
    for (int g = 0; g < ngrp; ++g) {
      const int tg = g * TG;
      if (g > 0) __syncthreads();                    // the previous tap group's reads of the weights are done
      {
        int4* dh_ = reinterpret_cast<int4*>(wh) + tid;
        int4* dl_ = reinterpret_cast<int4*>(wl) + tid;
        dh_[0] = wvh0; dh_[256] = wvh1; dh_[512] = wvh2;
        dl_[0] = wvl0; dl_[256] = wvl1; dl_[512] = wvl2;
      }
      __syncthreads();
      if constexpr (NCH == 0) {
        const int stage = ch * ngrp + g;
        if (stage + 1 < nch * ngrp) CB_FETCH_W(stage + 1);
        if (g + 1 == ngrp && ch + 1 < nch) CB_FETCH_P(ch + 1);
      }
#pragma unroll
      for (int tt = 0; tt < TG; ++tt) {
        const int tap = tg + tt;
        const int toff = p.nd == 2 ? (tap / 3) * PW + (tap % 3) : tap;
        cbf8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int o = ((tt * CD_TN) + wn * 32 + cb * 16 + r) * CB_KC + 8 * q;
          ah[cb] = *reinterpret_cast<const cbf8*>(wh + o);
          al[cb] = *reinterpret_cast<const cbf8*>(wl + o);
        }
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const int o = (pidx[pb] + toff) * CB_KC + 8 * q;
          bh[pb] = *reinterpret_cast<const cbf8*>(ph + o);
          bl[pb] = *reinterpret_cast<const cbf8*>(pl + o);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = cb_mma<F16>(ah[cb], bh[pb], acc[cb][pb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = cb_mma<F16>(ah[cb], bl[pb], acc[cb][pb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = cb_mma<F16>(al[cb], bh[pb], acc[cb][pb]);
      }
    }
    if constexpr (NCH > 1) {
      if (ch + 1 < NCH) {                              // (unrolled: static indices)
#pragma unroll
        for (int k = 0; k < PKB; ++k) pv[k] = pva[ch < NA ? ch : 0][k];
        const int c_ = ch < NA ? ch : 0;
        wvh0 = wva[c_][0]; wvh1 = wva[c_][1]; wvh2 = wva[c_][2]; wvl0 = wva[c_][3]; wvl1 = wva[c_][4]; wvl2 = wva[c_][5];
      }
    }
    if constexpr (F16) {
      if (ch + 1 < (NCH > 0 ? NCH : nch)) wave_max((ch + 1) & 1);       // (the next chunk's values arrived under the MFMAs)
    }
  }
  if constexpr (F16) {
    const float inv = cb_pow2(-(E_cur + CB_WSHIFT));
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) acc[cb][pb] *= inv;
  }
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int n = wm * 32 + pb * 16 + r;
    const int y = p.nd == 2 ? y0 + (n >> 3) : y0 + n, x = p.nd == 2 ? x0 + (n & 7) : 0;
    if (SPT > 1 ? (n >= SPT * p.H || b + n / p.H >= p.B) : (y >= p.H || x >= p.W)) continue;   // (packed: y runs over the tile's samples)
    float* orow = p.out + (((int64_t)b * p.H + y) * p.W + x) * p.Cout;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int co = cot * CD_TN + wn * 32 + cb * 16 + 4 * q;
      cd_store4(p, orow, co, acc[cb][pb]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The direct weight gradient on the bf16 matrix cores in split precision (mixed-precision mode, plan->conv_bf16), for
// layers with a multiple of 32 input channels.  Same ownership as pv_conv3_wgrad_direct_kernel (64 output channels x one
// 32-channel chunk x all taps per workgroup, a range of pixel tiles per split); the dY tile and the patch are staged as
// bf16 (hi, lo) in their natural [pixel][channel] layouts and BOTH MFMA operands — which need 8 consecutive PIXELS per
// lane — come out of LDS through the transposing read ds_read_b64_tr_b16: lane (r, q) gets rows 4q .. 4q+3 of column r.
typedef short cshort4 __attribute__((ext_vector_type(4)));
typedef short short8_cd __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) cshort4 lds_cshort4;
__device__ __forceinline__ cbf4 cw_tr(const __bf16* p) {
  const cshort4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_cshort4*)p);
  return __builtin_bit_cast(cbf4, v);
}
__device__ __forceinline__ cbf8 cw_cat(const cbf4& a, const cbf4& b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
#define CW_LDY 72                // dY rows: 64 channels + 8 (bf16 elements)
#define CW_LDP 40                // patch rows: 32 channels + 8

__global__ __launch_bounds__(256) void pv_conv3_wgrad_bf16_kernel(ConvWg p) {
  extern __shared__ __attribute__((aligned(16))) char smb_[];
  const int KK = p.KK, PW = p.nd == 2 ? 10 : 1, PH = p.nd == 2 ? 10 : CD_PIX + 2, NPIX = PH * PW;
  __bf16* yh = reinterpret_cast<__bf16*>(smb_);                       // [64][CW_LDY]
  __bf16* yl = yh + CD_PIX * CW_LDY;
  __bf16* ph = yl + CD_PIX * CW_LDY;                                  // [NPIX][CW_LDP]
  __bf16* pl = ph + NPIX * CW_LDP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int split = blockIdx.x, ch = blockIdx.y, cot = blockIdx.z;
  const int64_t T = (int64_t)p.B * p.tiles_y * p.tiles_x;
  const int64_t t_lo = T * split / p.nsplit, t_hi = T * (split + 1) / p.nsplit;
  f32x4 acc[9][2], accb = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < 9; ++t) { acc[t][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[t][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
  const short one = 0x3f80;
  const short8_cd ones_s = {one, one, one, one, one, one, one, one};
  const cbf8 ones = __builtin_bit_cast(cbf8, ones_s);
  for (int64_t tt = t_lo; tt < t_hi; ++tt) {
    int t = (int)tt;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
    const int y0 = ty * (p.nd == 2 ? 8 : CD_PIX), x0 = tx * 8;
    __syncthreads();
    for (int e = tid; e < CD_PIX * 16; e += 256) {               // dY tile: pixel n, channels 4*c4 .. 4*c4+3 -> (hi, lo)
      const int n = e >> 4, c4 = e & 15;
      const int y = p.nd == 2 ? y0 + (n >> 3) : y0 + n, x = p.nd == 2 ? x0 + (n & 7) : 0;
      const int co = cot * CD_TN + 4 * c4;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (y < p.H && x < p.W) {
        const float* src = p.dy + (((int64_t)b * p.H + y) * p.W + x) * p.Cout + co;
        if (co + 3 < p.Cout && (p.Cout & 3) == 0) v = *reinterpret_cast<const f32x4*>(src);
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i) if (co + i < p.Cout) v[i] = src[i];
        }
      }
      cbf4 h4, l4;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const __bf16 hh = (__bf16)v[i]; h4[i] = hh; l4[i] = (__bf16)(v[i] - (float)hh); }
      *reinterpret_cast<cbf4*>(yh + n * CW_LDY + 4 * c4) = h4;
      *reinterpret_cast<cbf4*>(yl + n * CW_LDY + 4 * c4) = l4;
    }
    const float* in_b = p.in + (int64_t)b * p.H * p.W * p.Cin;
    for (int e = tid; e < NPIX * 8; e += 256) {
      const int pix = e >> 3, f4 = e & 7;
      const int py = p.nd == 2 ? pix / PW : pix, px = p.nd == 2 ? pix - py * PW : 0;
      const int y = y0 - 1 + py, x = p.nd == 2 ? x0 - 1 + px : 0;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (y >= 0 && y < p.H && x >= 0 && x < p.W)
        v = *reinterpret_cast<const f32x4*>(in_b + ((int64_t)y * p.W + x) * p.Cin + ch * CB_KC + 4 * f4);
      cbf4 h4, l4;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const __bf16 hh = (__bf16)v[i]; h4[i] = hh; l4[i] = (__bf16)(v[i] - (float)hh); }
      *reinterpret_cast<cbf4*>(ph + pix * CW_LDP + 4 * f4) = h4;
      *reinterpret_cast<cbf4*>(pl + pix * CW_LDP + 4 * f4) = l4;
    }
    __syncthreads();
#pragma unroll 1
    for (int ks = 0; ks < 2; ++ks) {                             // 32 pixels: k slots 8q+i <-> pixel 4q+i, 8q+4+i <-> 16+4q+i
      const int pa = 32 * ks + 4 * q;                            // this lane group's first pixel of the first half
      const int arow = (pa + (r >> 2)) * CW_LDY + 16 * wave + 4 * (r & 3);
      const cbf8 ah = cw_cat(cw_tr(yh + arow), cw_tr(yh + arow + 16 * CW_LDY));
      const cbf8 al = cw_cat(cw_tr(yl + arow), cw_tr(yl + arow + 16 * CW_LDY));
      if (ch == 0) { accb = MFMA32B(ah, ones, accb); accb = MFMA32B(al, ones, accb); }
      // patch rows of the two halves' pixels (4 consecutive pixels of one image row each)
      const int p0 = p.nd == 2 ? (pa >> 3) * PW + (pa & 7) : pa;
      const int half = p.nd == 2 ? 2 * PW : 16;                  // 16 pixels further: two image rows down (2-D)
      const int brow = (p0 + (r >> 2)) * CW_LDP + 4 * (r & 3);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap < KK) {
          const int toff = (p.nd == 2 ? (tap / 3) * PW + (tap % 3) : tap) * CW_LDP;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const int o = brow + toff + 16 * cb;
            const cbf8 bh = cw_cat(cw_tr(ph + o), cw_tr(ph + o + half * CW_LDP));
            const cbf8 bl = cw_cat(cw_tr(pl + o), cw_tr(pl + o + half * CW_LDP));
            acc[tap][cb] = MFMA32B(ah, bh, acc[tap][cb]);
            acc[tap][cb] = MFMA32B(ah, bl, acc[tap][cb]);
            acc[tap][cb] = MFMA32B(al, bh, acc[tap][cb]);
          }
        }
      }
    }
  }
  // C/D layout: lane (column = ci r, q), reg i -> output channel 16*wave + 4q + i
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int ci = ch * CB_KC + 16 * cb + r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = cot * CD_TN + 16 * wave + 4 * q + i;
      if (co >= p.Cout) continue;
      float* dst = p.part + (((int64_t)split * p.Cout + co) * p.Cin + ci) * KK;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        if (tap < KK) dst[tap] = acc[tap][cb][i];
      if (cb == 0 && ch == 0 && r == 0 && p.part_b) p.part_b[(int64_t)split * p.Cout + co] = accb[i];
    }
  }
}

int pv_conv3_wgrad_direct_bf16(const float* dy, const float* in, int B, int H, int W, int C, int nd, float* dw, float* db,
                               int Cout, void* ws, int64_t ws_bytes, hipStream_t s, PvFinishList* defer) {
  if (!pv_conv3_wgrad_direct_supported(C, Cout, nd) || C % CB_KC != 0) return PV_EINVAL;
  if (!pv_wgrad_ws(defer, pv_conv3_wgrad_direct_ws(B, H, W, C, Cout, nd), ws, ws_bytes)) defer = nullptr;
  if (ws_bytes < pv_conv3_wgrad_direct_ws(B, H, W, C, Cout, nd)) return PV_EWS;
  const int KK = nd == 2 ? 9 : 3;
  ConvWg p{};
  p.dy = dy; p.in = in; p.B = B; p.H = H; p.W = W; p.Cin = C; p.Cout = Cout; p.nd = nd; p.KK = KK;
  p.tiles_x = nd == 2 ? (W + 7) / 8 : 1;
  p.tiles_y = nd == 2 ? (H + 7) / 8 : (H + CD_PIX - 1) / CD_PIX;
  p.nsplit = wgd_splits(B, H, W, C, Cout, nd);          // (sized for 16-channel chunks: half as many owners here — fine)
  const int64_t nw = (int64_t)Cout * C * KK;
  p.part = reinterpret_cast<float*>(ws);
  p.part_b = db ? p.part + (int64_t)p.nsplit * nw : nullptr;
  const int npix = nd == 2 ? 100 : CD_PIX + 2;
  const size_t lds = (size_t)(2 * CD_PIX * CW_LDY + 2 * npix * CW_LDP) * 2;
  hipLaunchKernelGGL(pv_conv3_wgrad_bf16_kernel, dim3((unsigned)p.nsplit, (unsigned)(C / CB_KC), (unsigned)((Cout + CD_TN - 1) / CD_TN)),
                     dim3(256), lds, s, p);
  PV_LAUNCH_CHECK();
  return pv_wgrad_finish(defer, p.part, p.nsplit, nw, dw, p.part_b, Cout, db, s);
}
