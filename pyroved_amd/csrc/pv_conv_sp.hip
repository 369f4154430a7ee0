// pv_conv_sp.hip — 2-D kernel-3 (padding 1, stride 1) convolution over channels-last fp32 activations on the bf16 matrix
// cores with EXACTLY SPLIT operands: the forward of nn.Conv2d in nets/conv.py's FeatureExtractor / Upsampler stacks
// (reference: pyroved/nets/conv.py:146-249) and — taps flipped, channel roles swapped — its input gradient.
//
// An fp32 value is the exact sum of three bf16 pieces (24 = 8 + 8 + 8 significant bits, split by truncation), so
//   a*b = a0b0 + (a0b1 + a1b0) + (a0b2 + a2b0 + a1b1) + O(2^-24 |a||b|)
// : six v_mfma_f32_16x16x32_bf16 (fp32 accumulate) give an fp32-class product at 6 x 16 cycles per 16x16x32 block where
// the f32-input MFMA (16x16x4, 32 cycles) needs 8 x 32 — 2.6x the matrix-core rate of pv_conv_direct.hip's f32 kernel at
// the same accuracy.  NS = 2 keeps two rounded pieces and three products (~2^-17 per product): the mixed-precision mode.
//
// Work decomposition (one workgroup = 4 waves): a 16x16 tile of output pixels x 64 output channels; wave w owns the
// 8x8 pixel quadrant (w>>1, w&1) x all 64 channels = 4x4 MFMA blocks (64 accumulator registers), so an A fragment
// (weights) serves 4 pixel blocks and a B fragment (patch) 4 channel blocks: 8*NS 16-byte LDS reads per 16*NPROD
// MFMAs.  Input channels are walked in chunks of 32 (= one MFMA k): the tile's patch with its halo (18x18 pixels x 32
// channels) is split into NS bf16 planes while it is staged; the chunk's weights come pre-split and pre-tiled
// ([co tile][chunk][tap][plane][64][32], rows >= Cout zero) from pv_conv3_sp_wprep and are staged TG taps at a time,
// the next stage's global loads in flight under the current stage's MFMAs.
// LDS rows are 64 bytes ([row][32 bf16]); ds_read_b128 is serviced in the 16-lane groups {q: r in 0-3,12-15} U
// {q^1: r in 4-11} (MI355X_MICROARCH.md, LDS), so the 16-byte slot of a row is XORed with 2*(bit 3 of the fragment row) —
// for the patch that is the parity of the patch line — which makes every fragment read conflict-free.
#include "pv_common.h"
#include "pv_conv.h"
#include <stdlib.h>

typedef __bf16 sbf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#ifndef SP_EXP
#define SP_EXP 0                 // timing experiments (wrong results): 1 no MFMAs, 2 no weight re-staging, 4 no patch re-staging
#endif
#if SP_EXP & 1
#define SP_MFMA(a, b, c) (c)
#else
#define SP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
#define SP_TN 64                 // output channels per workgroup
#define SP_KC 32                 // input channels per chunk
#define SP_T 16                  // output pixel tile edge
#define SP_PW 18                 // patch edge
#define SP_NPIX (SP_PW * SP_PW)
#define SP_PPLANE (SP_NPIX * 64) // bytes per patch plane
#define SP_WPLANE (SP_TN * 64)   // bytes per weight plane of one tap

struct ConvSp {
  const float* in; const char* wt; const float* bias; float* out;
  const float* eg_y; int eg_act;      // optional: out *= act'(eg_y) elementwise (the producing layer's activation backward)
  int B, H, W, Cin, Cout, act, tiles_x, tiles_y;
};

// (hi16(b) << 16) | hi16(a): two truncated bf16 out of two fp32 bit patterns
__device__ __forceinline__ unsigned sp_pk_hi(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float sp_trunc(float v) { return __uint_as_float(__float_as_uint(v) & 0xffff0000u); }

// four fp32 -> NS planes of four bf16 (two dwords each)
template <int NS> __device__ __forceinline__ void sp_split4(const f32x4& v, u32x2 (&pl)[NS]) {
  if constexpr (NS == 3) {
    f32x4 r1, r2;
#pragma unroll
    for (int i = 0; i < 4; ++i) r1[i] = v[i] - sp_trunc(v[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) r2[i] = r1[i] - sp_trunc(r1[i]);
    pl[0] = u32x2{sp_pk_hi(v[0], v[1]), sp_pk_hi(v[2], v[3])};
    pl[1] = u32x2{sp_pk_hi(r1[0], r1[1]), sp_pk_hi(r1[2], r1[3])};
    pl[2] = u32x2{sp_pk_hi(r2[0], r2[1]), sp_pk_hi(r2[2], r2[3])};
  } else {
    unsigned short h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __bf16 hh = (__bf16)v[i];
      const __bf16 ll = (__bf16)(v[i] - (float)hh);
      h[i] = __builtin_bit_cast(unsigned short, hh);
      l[i] = __builtin_bit_cast(unsigned short, ll);
    }
    pl[0] = u32x2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
    pl[1] = u32x2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
  }
}

template <int NS> __device__ __forceinline__ void sp_split1(float v, unsigned short (&pl)[NS]) {
  if constexpr (NS == 3) {
    const float r1 = v - sp_trunc(v), r2 = r1 - sp_trunc(r1);
    pl[0] = (unsigned short)(__float_as_uint(v) >> 16);
    pl[1] = (unsigned short)(__float_as_uint(r1) >> 16);
    pl[2] = (unsigned short)(__float_as_uint(r2) >> 16);
  } else {
    const __bf16 hh = (__bf16)v;
    const __bf16 ll = (__bf16)(v - (float)hh);
    pl[0] = __builtin_bit_cast(unsigned short, hh);
    pl[1] = __builtin_bit_cast(unsigned short, ll);
  }
}

// raw torch weight w[Co][Ci][9] -> tiled, split logical matrix Wl[n][c][t]:
//   flip == 0 (forward):  Wl[n = co][c = ci][t] = w[co][ci][t]                (N = Co, C = Ci)
//   flip == 1 (dgrad):    Wl[n = ci][c = co][t] = w[co][ci][8 - t]            (N = Ci, C = Co)
// laid out [n tile][chunk][t][plane][64 rows][32 channels, 16-byte slots swizzled], rows n >= N zero
template <int NS>
__global__ void pv_conv3_sp_wprep_kernel(const float* __restrict__ w, unsigned short* __restrict__ wt, int Co, int Ci, int flip) {
  const int N = flip ? Ci : Co, C = flip ? Co : Ci;
  const int nt = (N + SP_TN - 1) / SP_TN, nch = C / SP_KC;
  const int64_t total = (int64_t)nt * nch * 9 * SP_TN * SP_KC;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int cl = (int)(e % SP_KC), nl = (int)((e / SP_KC) % SP_TN), t = (int)((e / (SP_KC * SP_TN)) % 9);
    const int ch = (int)((e / ((int64_t)SP_KC * SP_TN * 9)) % nch), tile = (int)(e / ((int64_t)SP_KC * SP_TN * 9 * nch));
    const int n = tile * SP_TN + nl, c = ch * SP_KC + cl;
    float v = 0.0f;
    if (n < N) v = flip ? w[((int64_t)c * Ci + n) * 9 + (8 - t)] : w[((int64_t)n * Ci + c) * 9 + t];
    unsigned short pl[NS];
    sp_split1<NS>(v, pl);
    const int scl = ((((cl >> 3) ^ (2 * ((nl >> 3) & 1)))) << 3) | (cl & 7);
    const int64_t base = (((int64_t)tile * nch + ch) * 9 + t) * NS;
#pragma unroll
    for (int k = 0; k < NS; ++k) wt[((base + k) * SP_TN + nl) * SP_KC + scl] = pl[k];
  }
}

template <int NS, int NCB>
__global__ __launch_bounds__(256, 2) void pv_conv3_sp_kernel(ConvSp p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TG = NS == 3 ? 1 : 3;                 // taps per weight stage
  constexpr int NG = 9 / TG;
  constexpr int WREGS = TG * NS;                      // 16-byte pieces of a weight stage per thread
  char* patch = smem;                                 // [NS][324][64 B]
  char* wl = smem + NS * SP_PPLANE;                   // [TG][NS][64][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int wy = wave >> 1, wx = wave & 1;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
  const int y0 = ty * SP_T, x0 = tx * SP_T;
  const int cot = blockIdx.y;
  const int nch = p.Cin / SP_KC;
  const float* in_b = p.in + (int64_t)b * p.H * p.W * p.Cin;
  f32x4 acc[NCB][4];                                  // NCB: 16-channel blocks per workgroup (2 when Cout <= 32)
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // fragment addresses: weights row (16 cb + r), slot q ^ 2*(r>>3); patch pixel (8 wy + 2 pb + (r>>3), 8 wx + (r&7)) of tap
  // (0,0), slot q ^ 2*(line parity) — the parity flips for the middle kernel row
  const int aoff = r * 64 + ((q ^ (2 * (r >> 3))) * 16);
  int boff[4][2];
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    const int pix = (8 * wy + 2 * pb + (r >> 3)) * SP_PW + 8 * wx + (r & 7);
    boff[pb][0] = pix * 64 + ((q ^ (2 * (r >> 3))) * 16);
    boff[pb][1] = pix * 64 + ((q ^ (2 * ((r >> 3) ^ 1))) * 16);
  }
  const char* wsrc = p.wt + (int64_t)cot * nch * 9 * NS * SP_WPLANE;   // this co tile's stages, in (chunk, tap) order
  uint4 w0 = {}, w1 = {}, w2 = {}, w3 = {}, w4 = {}, w5 = {};   // (a register array here ends up in scratch)
#define SP_W_FETCH(stage)                                                                                            \
  {                                                                                                                  \
    const uint4* src_ = reinterpret_cast<const uint4*>(wsrc + (int64_t)(stage) * TG * NS * SP_WPLANE) + tid;         \
    w0 = src_[0]; w1 = src_[256];                                                                                    \
    if constexpr (WREGS > 2) w2 = src_[512];                                                                         \
    if constexpr (WREGS > 3) { w3 = src_[768]; w4 = src_[1024]; w5 = src_[1280]; }                                   \
  }
#define SP_W_STORE()                                                                                                 \
  {                                                                                                                  \
    uint4* dst_ = reinterpret_cast<uint4*>(wl) + tid;                                                                \
    dst_[0] = w0; dst_[256] = w1;                                                                                    \
    if constexpr (WREGS > 2) dst_[512] = w2;                                                                         \
    if constexpr (WREGS > 3) { dst_[768] = w3; dst_[1024] = w4; dst_[1280] = w5; }                                   \
  }
  static_assert(WREGS == 3 || WREGS == 6, "weight stage size");
  // patch staging: thread e = tid + 256 k takes pixel e >> 3, channels 4 (e & 7) .. +3 of the chunk; the fp32 values of
  // the NEXT chunk are fetched into registers under the current chunk's MFMAs and split / written at the chunk boundary
  constexpr int PK = (SP_NPIX * 8 + 255) / 256;       // 11
  int goff[PK];                                       // element offset of the thread's k-th piece in the image, -1 outside
#pragma unroll
  for (int k = 0; k < PK; ++k) {
    const int e = tid + 256 * k, pix = e >> 3, f4 = e & 7;
    const int py = pix / SP_PW, px = pix - py * SP_PW;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    goff[k] = (e < SP_NPIX * 8 && y >= 0 && y < p.H && x >= 0 && x < p.W) ? (y * p.W + x) * p.Cin + 4 * f4 : -1;
  }
  f32x4 pre[PK];
#pragma unroll
  for (int k = 0; k < PK; ++k) {
    pre[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (goff[k] >= 0) pre[k] = *reinterpret_cast<const f32x4*>(in_b + goff[k]);
  }
  SP_W_FETCH(0);
  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();                                  // the previous chunk's fragment reads are done
#pragma unroll
    for (int k = 0; k < ((SP_EXP & 4) && ch > 0 ? 0 : PK); ++k) {
      const int e = tid + 256 * k, pix = e >> 3, f4 = e & 7;
      const int py = pix / SP_PW;
      u32x2 pl[NS];
      sp_split4<NS>(pre[k], pl);
      const int o = pix * 64 + (((f4 >> 1) ^ (2 * (py & 1))) * 16) + (f4 & 1) * 8;
      if (k + 1 < PK || e < SP_NPIX * 8) {
#pragma unroll
        for (int j = 0; j < NS; ++j) *reinterpret_cast<u32x2*>(patch + j * SP_PPLANE + o) = pl[j];
      }
    }
    SP_W_STORE();
    __syncthreads();
    if (ch + 1 < nch) {
#pragma unroll
      for (int k = 0; k < PK; ++k)
        if (goff[k] >= 0) pre[k] = *reinterpret_cast<const f32x4*>(in_b + goff[k] + (ch + 1) * SP_KC);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int stage = ch * NG + g;
      if (stage + 1 < nch * NG) SP_W_FETCH(stage + 1);   // in flight under this stage's MFMAs
#pragma unroll
      for (int tt = 0; tt < TG; ++tt) {
        const int tap = g * TG + tt, dy = tap / 3, dx = tap - 3 * dy;
        sbf8 a[NCB][NS];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int k = 0; k < NS; ++k)
            a[cb][k] = *reinterpret_cast<const sbf8*>(wl + (tt * NS + k) * SP_WPLANE + cb * 1024 + aoff);
        // B fragments one pixel block ahead of the MFMAs that use them
        const int tofs = (dy * SP_PW + dx) * 64;
        sbf8 bcur[NS], bnxt[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) bcur[k] = *reinterpret_cast<const sbf8*>(patch + k * SP_PPLANE + boff[0][dy & 1] + tofs);
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
          if (pb + 1 < 4) {
#pragma unroll
            for (int k = 0; k < NS; ++k)
              bnxt[k] = *reinterpret_cast<const sbf8*>(patch + k * SP_PPLANE + boff[pb + 1 < 4 ? pb + 1 : 3][dy & 1] + tofs);
          }
          // products in ascending magnitude; the NCB accumulators of a product are independent
          if constexpr (NS == 3) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = SP_MFMA(a[cb][1], bcur[1], acc[cb][pb]);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = SP_MFMA(a[cb][2], bcur[0], acc[cb][pb]);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = SP_MFMA(a[cb][0], bcur[2], acc[cb][pb]);
          }
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = SP_MFMA(a[cb][1], bcur[0], acc[cb][pb]);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = SP_MFMA(a[cb][0], bcur[1], acc[cb][pb]);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = SP_MFMA(a[cb][0], bcur[0], acc[cb][pb]);
#pragma unroll
          for (int k = 0; k < NS; ++k) bcur[k] = bnxt[k];
        }
      }
      if (g + 1 < NG && !(SP_EXP & 2)) {
        __syncthreads();                              // this stage's weight reads are done
        SP_W_STORE();
        __syncthreads();
      }
    }
  }
  // C/D layout: lane (column = pixel r, q), reg i -> output channel 16*cb + 4q + i.  The activation is uniform: one
  // switch around tight loops over the 16*NCB values (a switch per value costs more than the convolution's MFMAs)
  const bool vec = (p.Cout & 3) == 0;
#define SP_EACH(EXPR)                                                                                               \
  {                                                                                                                  \
    _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb) _Pragma("unroll") for (int pb = 0; pb < 4; ++pb)              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
      float v = acc[cb][pb][i];                                                                                      \
      EXPR;                                                                                                          \
      acc[cb][pb][i] = v;                                                                                            \
    }                                                                                                                \
  }
  if (p.bias) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int co = cot * SP_TN + cb * 16 + 4 * q;
      f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
      if (vec && co + 3 < p.Cout) bv = *reinterpret_cast<const f32x4*>(p.bias + co);
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (co + i < p.Cout) bv[i] = p.bias[co + i];
      }
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) acc[cb][pb] += bv;
    }
  }
  switch (p.act) {
    case PV_ACT_TANH: SP_EACH(v = tanhf(v)); break;
    case PV_ACT_RELU: SP_EACH(v = v > 0.0f ? v : 0.0f); break;
    case PV_ACT_LRELU: SP_EACH(v = v > 0.0f ? v : 0.01f * v); break;
    case PV_ACT_SOFTPLUS: SP_EACH(v = pv_softplus(v)); break;
    case PV_ACT_SIGMOID: SP_EACH(v = 1.0f / (1.0f + expf(-v))); break;
    default: break;
  }
  int64_t ro[4];
  bool ok[4];
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    const int y = y0 + 8 * wy + 2 * pb + (r >> 3), x = x0 + 8 * wx + (r & 7);
    ok[pb] = y < p.H && x < p.W;
    ro[pb] = (((int64_t)b * p.H + y) * p.W + x) * p.Cout;
  }
  if (p.eg_y) {                                       // out *= act'(eg_y): the producing layer's activation backward
    f32x4 gy[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        const int co = cot * SP_TN + cb * 16 + 4 * q;
        gy[cb][pb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (ok[pb]) {
          if (vec && co + 3 < p.Cout) gy[cb][pb] = *reinterpret_cast<const f32x4*>(p.eg_y + ro[pb] + co);
          else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (co + i < p.Cout) gy[cb][pb][i] = p.eg_y[ro[pb] + co + i];
          }
        }
      }
#define SP_EACH_G(EXPR) SP_EACH(const float yv = gy[cb][pb][i]; EXPR)
    switch (p.eg_act) {
      case PV_ACT_TANH: SP_EACH_G(v *= 1.0f - yv * yv); break;
      case PV_ACT_RELU: SP_EACH_G(v = yv > 0.0f ? v : 0.0f); break;
      case PV_ACT_LRELU: SP_EACH_G(v = yv > 0.0f ? v : 0.01f * v); break;
      case PV_ACT_SOFTPLUS: SP_EACH_G(v *= 1.0f - expf(-yv)); break;
      case PV_ACT_SIGMOID: SP_EACH_G(v *= yv * (1.0f - yv)); break;
      default: break;
    }
  }
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    if (!ok[pb]) continue;
    float* orow = p.out + ro[pb];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int co = cot * SP_TN + cb * 16 + 4 * q;
      if (vec && co + 3 < p.Cout) *reinterpret_cast<f32x4*>(orow + co) = acc[cb][pb];
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (co + i < p.Cout) orow[co + i] = acc[cb][pb][i];
      }
    }
  }
}

bool pv_conv3_sp_supported(int C, int Cout, int nd, int act) {
  return nd == 2 && C >= SP_KC && C % SP_KC == 0 && Cout >= 8 && act != PV_ACT_GELU;
}

// bytes of the tiled, split weights (either orientation fits)
int64_t pv_conv3_sp_wt_bytes(int C, int Cout) {
  const int64_t n = Cout > C ? Cout : C;
  return ((n + SP_TN - 1) / SP_TN) * SP_TN * n * 9 * 3 * 2 + 256;
}

template <int NS>
static int conv3_sp_launch(const ConvSp& p, const float* w, int Co, int Ci, int flip, char* wt, int nt, int64_t total,
                           hipStream_t s) {
  int pb = (int)((total + 255) / 256);
  if (pb > 2048) pb = 2048;
  hipLaunchKernelGGL(pv_conv3_sp_wprep_kernel<NS>, dim3(pb), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wt), Co, Ci,
                     flip);
  PV_LAUNCH_CHECK();
  constexpr int TG = NS == 3 ? 1 : 3;
  static const int lds_pad = getenv("PV_SP_LDS_PAD") ? atoi(getenv("PV_SP_LDS_PAD")) : 0;   // (occupancy experiments)
  const size_t lds = (size_t)NS * SP_PPLANE + (size_t)TG * NS * SP_WPLANE + lds_pad;
  const dim3 grid((unsigned)(p.tiles_x * p.tiles_y * p.B), (unsigned)nt);
  if (p.Cout <= 32) hipLaunchKernelGGL((pv_conv3_sp_kernel<NS, 2>), grid, dim3(256), lds, s, p);
  else hipLaunchKernelGGL((pv_conv3_sp_kernel<NS, 4>), grid, dim3(256), lds, s, p);
  PV_LAUNCH_CHECK();
  return 0;
}

// w: raw torch weight (Co, Ci, 3, 3).  flip == 0: out[.., Co] = act(conv(in[.., Ci]) + bias).
// flip == 1: out[.., Ci] = conv of in[.., Co] with the flipped / role-swapped weights (the input gradient).
// ns = 3: fp32-class (six products); ns = 2: mixed precision (three products).  wt_scratch: pv_conv3_sp_wt_bytes bytes.
int pv_conv3_sp(const float* in, int B, int H, int W, const float* w, int Co, int Ci, int flip, const float* bias, float* out,
                int act, void* wt_scratch, hipStream_t s, const float* eg_y, int eg_act, int ns) {
  const int N = flip ? Ci : Co, C = flip ? Co : Ci;
  if (!pv_conv3_sp_supported(C, N, 2, act) || (ns != 2 && ns != 3)) return PV_EINVAL;
  const int nt = (N + SP_TN - 1) / SP_TN;
  const int64_t total = (int64_t)nt * (C / SP_KC) * 9 * SP_TN * SP_KC;
  ConvSp p{};
  p.in = in; p.wt = reinterpret_cast<const char*>(wt_scratch); p.bias = bias; p.out = out;
  p.eg_y = (eg_y && eg_act != PV_ACT_NONE) ? eg_y : nullptr; p.eg_act = eg_act;
  p.B = B; p.H = H; p.W = W; p.Cin = C; p.Cout = N; p.act = act;
  p.tiles_x = (W + SP_T - 1) / SP_T; p.tiles_y = (H + SP_T - 1) / SP_T;
  return ns == 3 ? conv3_sp_launch<3>(p, w, Co, Ci, flip, reinterpret_cast<char*>(wt_scratch), nt, total, s)
                 : conv3_sp_launch<2>(p, w, Co, Ci, flip, reinterpret_cast<char*>(wt_scratch), nt, total, s);
}

// test / measurement hook: one convolution call on caller-provided device tensors.
// mode 0: f32-input MFMA direct kernel, 1: its bf16 two-piece form, 2 / 3: this file's kernels with 2 / 3 pieces
extern "C" int pv_debug_conv3(int mode, const float* in, int B, int H, int W, int nd, const float* w, int Co, int Ci, int flip,
                              const float* bias, float* out, int act, void* wt_scratch, const float* eg_y, int eg_act,
                              void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode >= 2) return nd == 2 ? pv_conv3_sp(in, B, H, W, w, Co, Ci, flip, bias, out, act, wt_scratch, s, eg_y, eg_act, mode)
                                : PV_EINVAL;
  return pv_conv3_direct(in, B, H, W, nd, w, Co, Ci, flip, bias, out, act, reinterpret_cast<float*>(wt_scratch), s, eg_y, eg_act,
                         mode);
}

extern "C" long long pv_debug_conv3_wgrad_ws(int mode, int B, int H, int W, int C, int Cout, int nd) {
  (void)mode;
  return pv_conv3_wgrad_direct_ws(B, H, W, C, Cout, nd);
}

extern "C" int pv_debug_conv3_wgrad(int mode, const float* dy, const float* in, int B, int H, int W, int C, int nd, float* dw,
                                    float* db, int Cout, void* ws, long long ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode == 1) return pv_conv3_wgrad_direct_bf16(dy, in, B, H, W, C, nd, dw, db, Cout, ws, ws_bytes, s);
  if (mode == 0) return pv_conv3_wgrad_direct(dy, in, B, H, W, C, nd, dw, db, Cout, ws, ws_bytes, s);
  return PV_EINVAL;
}

// resident workgroups per CU the runtime predicts for the forward kernel (ns pieces, 4 channel blocks) at lds bytes
extern "C" int pv_debug_conv3_sp_occupancy(int ns, int lds) {
  int n = -1;
  hipError_t e = ns == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pv_conv3_sp_kernel<3, 4>, 256, (size_t)lds)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pv_conv3_sp_kernel<2, 4>, 256, (size_t)lds);
  return e == hipSuccess ? n : -(int)e;
}
