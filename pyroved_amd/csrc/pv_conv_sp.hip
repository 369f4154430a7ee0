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
#include "pv_side.h"
#include "pv_conv.h"
#include "pv_fb_layout.h"
#include <stdlib.h>

typedef __bf16 sbf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 shf8 __attribute__((ext_vector_type(8)));
typedef __fp16 shp2 __attribute__((ext_vector_type(2)));
#ifndef SP_EXP
#define SP_EXP 0                 // timing experiments (wrong results): 1 no MFMAs, 2 no weight re-staging, 4 no patch re-staging, 8 no operand split
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

// fp16 two-piece mode (F16): pieces are fp16 (11 + 11 significant bits: three products give ~2^-22 per multiply-add, fp32-class at
// the MFMA cost of the mixed mode).  fp16's narrow exponent is handled by exact power-of-two scaling: the weights are tiled
// times SP_F16_WSCALE, each 32-channel chunk of a tile's patch times 2^E with its maximum brought to [2^13, 2^14) — E never more
// than 30 above the smallest E of the tile so far, so the accumulator (rescaled by 2^(E - E_prev) between chunks) cannot
// overflow; a chunk that needs a larger E than that is 2^30 below what is already summed and its precision is moot.
#define SP_F16_WSCALE 64.0f
#define SP_F16_WSHIFT 6

struct ConvSp {
  const float* in; const char* wt; const float* bias; float* out;
  const float* eg_y; int eg_act;      // optional: out *= act'(eg_y) elementwise (the producing layer's activation backward)
  int B, H, W, Cin, Cout, act, tiles_x, tiles_y, halves;
  float* pool_out; unsigned char* pool_code;   // != null: write maxpool2(out) (B, H/2, W/2, Cout) and its winners instead of out
  // != null (input gradients whose result is dL/d(a max-pooled activation); round 4): the 2x un-pooling rides in the epilogue — out is
  // (B, 2H, 2W, Cout), a value (already times act'(eg_y), eg_y = the pooled activation) goes to its winner's position, zeros to the
  // other three: pv_maxpool2_bwd_code's launch, its read of this tensor and the write of it are gone
  const unsigned char* up_code;
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

// the second pieces of two values whose first pieces are the halves of h: fp16(t - h), one mixed-precision fma each (the f16 half of
// h and the f32 t go in as they are, the result is rounded once into its half of the pair) — instead of cvt, sub, sub, cvt_pk
__device__ __forceinline__ unsigned sp_lo_pair(unsigned h, float t0, float t1) {
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(t0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(t1));
  return l;
}
// four fp32 (times the exact scale sc) -> two planes of four fp16: a0 = rtz(v sc), a1 = fp16(v sc - a0) (exact difference, one rounding)
__device__ __forceinline__ void sp_split4_f16(const f32x4& v, float sc, u32x2 (&pl)[2]) {
#if SP_EXP & 8                   // (timing: operands that arrive split — no VALU work)
  pl[0] = u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])};
  pl[1] = u32x2{__float_as_uint(v[2]), __float_as_uint(v[3])};
  return;
#endif
  const f32x4 t = v * sc;
  const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(t[0], t[1]));
  const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(t[2], t[3]));
  pl[0] = u32x2{h01, h23};
  pl[1] = u32x2{sp_lo_pair(h01, t[0], t[1]), sp_lo_pair(h23, t[2], t[3])};
}
// one-piece forms (round 4, the throughput precision: ONE fp16 piece per operand, one product): round to nearest
typedef _Float16 shf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sp_split4_f16(const f32x4& v, float sc, u32x2 (&pl)[1]) {
  pl[0] = __builtin_bit_cast(u32x2, __builtin_convertvector(v * sc, shf4));
}
__device__ __forceinline__ void sp_split1_f16(float v, unsigned short (&pl)[1]) {
  pl[0] = __builtin_bit_cast(unsigned short, (_Float16)(v * SP_F16_WSCALE));
}
__device__ __forceinline__ void sp_split1_f16(float v, unsigned short (&pl)[2]) {
  const float t = v * SP_F16_WSCALE;
  const _Float16 h = (_Float16)t;
  const _Float16 l = (_Float16)(t - (float)h);
  pl[0] = __builtin_bit_cast(unsigned short, h);
  pl[1] = __builtin_bit_cast(unsigned short, l);
}
template <bool F16> __device__ __forceinline__ f32x4 sp_mma(const sbf8& a, const sbf8& b, const f32x4& c) {
#if SP_EXP & 1
  return c;
#else
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(shf8, a), __builtin_bit_cast(shf8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ float sp_pow2(int k) {          // 2^k, k clamped to the normal range
  k = k < -126 ? -126 : (k > 127 ? 127 : k);
  return __uint_as_float((unsigned)(k + 127) << 23);
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
template <int NS, bool F16 = false>
__device__ __forceinline__ void sp_wprep_elem(const float* __restrict__ w, unsigned short* __restrict__ wt, int Co, int Ci, int flip,
                                              int64_t e) {
  const int N = flip ? Ci : Co, C = flip ? Co : Ci;
  const int nch = C / SP_KC;
  {
    const int cl = (int)(e % SP_KC), nl = (int)((e / SP_KC) % SP_TN), t = (int)((e / (SP_KC * SP_TN)) % 9);
    const int ch = (int)((e / ((int64_t)SP_KC * SP_TN * 9)) % nch), tile = (int)(e / ((int64_t)SP_KC * SP_TN * 9 * nch));
    const int n = tile * SP_TN + nl, c = ch * SP_KC + cl;
    float v = 0.0f;
    if (n < N) v = flip ? w[((int64_t)c * Ci + n) * 9 + (8 - t)] : w[((int64_t)n * Ci + c) * 9 + t];
    unsigned short pl[NS];
    if constexpr (F16) sp_split1_f16(v, pl);
    else sp_split1<NS>(v, pl);
    const int scl = ((((cl >> 3) ^ (2 * ((nl >> 3) & 1)))) << 3) | (cl & 7);
    const int64_t base = (((int64_t)tile * nch + ch) * 9 + t) * NS;
#pragma unroll
    for (int k = 0; k < NS; ++k) wt[((base + k) * SP_TN + nl) * SP_KC + scl] = pl[k];
  }
}

template <int NS, bool F16 = false>
__global__ void pv_conv3_sp_wprep_kernel(const float* __restrict__ w, unsigned short* __restrict__ wt, int Co, int Ci, int flip) {
  const int N = flip ? Ci : Co, C = flip ? Co : Ci;
  const int64_t total = (int64_t)((N + SP_TN - 1) / SP_TN) * (C / SP_KC) * 9 * SP_TN * SP_KC;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    sp_wprep_elem<NS, F16>(w, wt, Co, Ci, flip, e);
}

// ---- every weight tiling of a step in one launch: entry k covers element indices [start, start + total) ----------------
#define WPREP_CAP 32
struct WprepTab { PvWprepEntry e[WPREP_CAP]; int n; int64_t total; PvFbPrep fb; int fb_blocks; };

__global__ void pv_conv_wprep_table_kernel(WprepTab t) {
  if (t.fb_blocks > 0) {                              // the last fb_blocks workgroups: the spatial decoder's weight images
    const int first = (int)gridDim.x - t.fb_blocks;
    if ((int)blockIdx.x >= first) {
      pv_fb_prep(t.fb, (int64_t)((int)blockIdx.x - first) * blockDim.x + threadIdx.x, (int64_t)t.fb_blocks * blockDim.x,
                 (int)(blockDim.x >> 6), (int)(threadIdx.x >> 6));
      return;
    }
  }
  const int nb = (int)gridDim.x - t.fb_blocks;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < t.total; idx += (int64_t)nb * blockDim.x) {
    int k = 0;
    while (k + 1 < t.n && idx >= t.e[k + 1].start) ++k;
    const PvWprepEntry E = t.e[k];
    const int64_t e = idx - E.start;
    const int Co = E.Co, Ci = E.Ci, KK = E.KK, flip = E.flip;
    if (E.kind == 8) {                                // pv_dec1d.hip; flip: taps reversed, channel roles swapped
      const int N = flip ? Ci : Co, C = flip ? Co : Ci;
      int c, n, tp;
      bool pad = false;
      if ((C & 15) == 0) {                            // MFMA fragment order: [16-row block][chunk = tap * C/16 + t][lane 16 q + r][4]
        const int kc = C >> 4, nck = KK * kc;         //   = (row 16 ob + r, column 16 t + 4 q + i): a wave's operand load is 1 KB contiguous;
        const int G = nck >= 12 ? 8 : 4;              //   chunks padded to groups of 4 / 8 with zeros (pv_dec1d.hip d1_group)
        const int nckp = (nck + G - 1) / G * G;
        const int i = (int)(e & 3), lane = (int)((e >> 2) & 63), ch = (int)((e >> 8) % nckp), ob = (int)((e >> 8) / nckp);
        pad = ch >= nck;
        tp = pad ? 0 : ch / kc;
        n = 16 * ob + (lane & 15); c = pad ? 0 : 16 * (ch - tp * kc) + 4 * (lane >> 4) + i;
      } else {                                        // a contraction narrower than 16 (plain loops): dst[tap][n][c]
        c = (int)(e % C); n = (int)((e / C) % N); tp = (int)(e / ((int64_t)C * N));
      }
      float v = 0.0f;
      if (n < N && !pad) v = flip ? E.w[((int64_t)c * Ci + n) * KK + (KK - 1 - tp)] : E.w[((int64_t)n * Ci + c) * KK + tp];
      reinterpret_cast<float*>(E.dst)[e] = v;
      continue;
    }
    if (E.kind == 7) {                                // latent2features: dst[k][s*C + c] = w[c*S + s][k]  (Co = z_dim, Ci = C, KK = S)
      const int64_t F = (int64_t)Ci * KK, k = e / F, f = e - k * F;
      const int sp = (int)(f / Ci), c = (int)(f - (int64_t)sp * Ci);
      reinterpret_cast<float*>(E.dst)[e] = E.w[((int64_t)c * KK + sp) * Co + k];
      continue;
    }
    if (E.kind == 4) {                                // conv head: dst[j][s*C + c] = w[j][c*S + s]  (Co = out, Ci = C, KK = S)
      const int64_t F = (int64_t)Ci * KK, j = e / F, f = e - j * F;
      const int sp = (int)(f / Ci), c = (int)(f - (int64_t)sp * Ci);
      reinterpret_cast<float*>(E.dst)[e] = E.w[j * F + (int64_t)c * KK + sp];
      continue;
    }
    if (E.kind == 5) {                                // fp16 two-piece tiling (weights times SP_F16_WSCALE)
      sp_wprep_elem<2, true>(E.w, reinterpret_cast<unsigned short*>(E.dst), Co, Ci, flip, e);
      continue;
    }
    if (E.kind == 9) {                                // fp16 one-piece tiling (the throughput precision)
      sp_wprep_elem<1, true>(E.w, reinterpret_cast<unsigned short*>(E.dst), Co, Ci, flip, e);
      continue;
    }
    if (E.kind >= 2 && E.kind != 6) {
      if (E.kind == 3) sp_wprep_elem<3>(E.w, reinterpret_cast<unsigned short*>(E.dst), Co, Ci, flip, e);
      else sp_wprep_elem<2>(E.w, reinterpret_cast<unsigned short*>(E.dst), Co, Ci, flip, e);
      continue;
    }
    // pv_conv_direct.hip's tilings: [n tile][chunk][tap][64][KC], KC = 16 fp32 (kind 0) / 32 bf16 hi + lo arrays (kind 1)
    const int N = flip ? Ci : Co, C = flip ? Co : Ci;
    const int KC = E.kind == 0 ? 16 : 32, nch = C / KC;                 // kind 6: as 1, fp16 pieces of the value times 2^6
    const int cl = (int)(e % KC), nl = (int)((e / KC) % 64), tp = (int)((e / (KC * 64)) % KK);
    const int ch = (int)((e / ((int64_t)KC * 64 * KK)) % nch), tile = (int)(e / ((int64_t)KC * 64 * KK * nch));
    const int n = tile * 64 + nl, c = ch * KC + cl;
    float v = 0.0f;
    if (n < N) v = flip ? E.w[((int64_t)c * Ci + n) * KK + (KK - 1 - tp)] : E.w[((int64_t)n * Ci + c) * KK + tp];
    if (E.kind == 0) reinterpret_cast<float*>(E.dst)[e] = v;
    else if (E.kind == 6) {
      const float tv = v * SP_F16_WSCALE;
      const _Float16 hi = (_Float16)tv;
      _Float16* d = reinterpret_cast<_Float16*>(E.dst);
      d[e] = hi;
      d[E.total + e] = (_Float16)(tv - (float)hi);
    } else {
      const __bf16 hi = (__bf16)v;
      __bf16* d = reinterpret_cast<__bf16*>(E.dst);
      d[e] = hi;
      d[E.total + e] = (__bf16)(v - (float)hi);
    }
  }
}

static int64_t wprep_elems(int kind, int Co, int Ci, int KK, int flip) {
  if (kind == 8) {                                   // (rows padded to 16 in fragment order)
    const int N = flip ? Ci : Co, C = flip ? Co : Ci;
    const int nck = KK * (C >> 4), G = nck >= 12 ? 8 : 4;
    return (C & 15) == 0 ? (int64_t)((N + 15) / 16) * ((nck + G - 1) / G * G) * 256 : (int64_t)N * C * KK;
  }
  if (kind == 4 || kind == 7) return (int64_t)Co * Ci * KK;
  if (kind == 5 || kind == 9) KK = 9;
  const int N = flip ? Ci : Co, C = flip ? Co : Ci;
  const int KC = kind == 0 ? 16 : 32;
  return (int64_t)((N + 63) / 64) * (C / KC) * KK * 64 * KC;
}

int64_t pv_conv_wt_bytes(int kind, int Co, int Ci, int nd) {
  if (kind == 4 || kind == 7 || kind == 8) return -1;   // (sized by the caller)
  if (kind >= 2 && kind != 6) return pv_conv3_sp_wt_bytes(Ci, Co);
  return pv_conv3_direct_wt_floats(Ci, Co, nd) * (int64_t)sizeof(float);
}

// fills start / total of the entries and launches (16 entries per launch)
int pv_conv_wprep_table(PvWprepEntry* e, int n, hipStream_t s, const PvFbPrep* fb) {
  if (n <= 0 && fb) {                                 // nothing to tile: the images alone
    WprepTab t{};
    t.fb = *fb;
    const int64_t work = fb->nzero4 > 128 * 32 ? fb->nzero4 : 128 * 32;
    t.fb_blocks = (int)((work + 255) / 256 > 256 ? 256 : (work + 255) / 256);
    hipLaunchKernelGGL(pv_conv_wprep_table_kernel, dim3(t.fb_blocks), dim3(256), 0, s, t);
    PV_LAUNCH_CHECK();
    return 0;
  }
  for (int lo = 0; lo < n; lo += WPREP_CAP) {
    WprepTab t{};
    if (fb && lo == 0) {
      t.fb = *fb;
      const int64_t work = fb->nzero4 > 128 * 32 ? fb->nzero4 : 128 * 32;
      t.fb_blocks = (int)((work + 255) / 256 > 256 ? 256 : (work + 255) / 256);
    }
    t.n = n - lo < WPREP_CAP ? n - lo : WPREP_CAP;
    int64_t acc = 0;
    for (int k = 0; k < t.n; ++k) {
      e[lo + k].total = wprep_elems(e[lo + k].kind, e[lo + k].Co, e[lo + k].Ci, e[lo + k].KK, e[lo + k].flip);
      e[lo + k].start = acc;
      acc += e[lo + k].total;
      t.e[k] = e[lo + k];
    }
    t.total = acc;
    int pb = (int)((acc + 255) / 256);
    if (pb > 4096) pb = 4096;
    if (pb < 1 && t.fb_blocks == 0) continue;
    if (pb < 1) pb = 1;
    hipLaunchKernelGGL(pv_conv_wprep_table_kernel, dim3(pb + t.fb_blocks), dim3(256), 0, s, t);
    PV_LAUNCH_CHECK();
  }
  return 0;
}

// phase-timing trace (profiling builds only: -DSP_TRACE, scripts/gpu_trace_sw.py fwd): shader-clock stamps of a middle workgroup
#ifdef SP_TRACE
__device__ long long sp_trace[16];
#define SP_STAMP(k) do { if (tr_on) { const long long c_ = (long long)__builtin_readcyclecounter(); tr_sum[(k)] += c_ - tr_prev; tr_prev = c_; } } while (0)
extern "C" int pv_debug_read_trace_sp(long long* out, int n) {
  if (n > 16) n = 16;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sp_trace), n * sizeof(long long));
}
#else
#define SP_STAMP(k) do { } while (0)
#endif
// NSA: pieces of the PATCH (the activations, or dL/d(output) in the input-gradient form); NS: pieces of the weights.  NSA == NS
// except in the round-5 form <2, .., true, 1>: the weights keep their two exact fp16 pieces, the patch is ONE fp16 piece
// (round to nearest, power-of-two scaled per staged chunk) — two products instead of three.  A weight's rounding error is
// systematic (the same perturbation at every pixel of every sample), an activation's is independent from element to element
// and averages out of every sum the step forms: the argument of the decoder kernel's fp16 builds (pv_sdec_fused_bf16.hip).
template <int NS, int NCB, bool F16 = false, int NSA = NS>
__device__ __forceinline__ void sp_conv_body(const ConvSp& p, const int bid_x, const int bid_y, char* smem) {
  static_assert(F16 ? (NS == 2 || NS == 1) : NS >= 2, "the fp16 modes have two pieces (fp32-class) or one (throughput)");
  static_assert(NSA == NS || (F16 && NS == 2 && NSA == 1), "patch pieces");
  __shared__ float smax[2][4];                        // F16: the waves' patch maxima of the chunk being staged
  constexpr int TG = NS == 3 ? 1 : 3;                 // taps per weight stage
  constexpr int NG = 9 / TG;
  constexpr int WREGS = TG * NS;                      // 16-byte pieces of a weight stage per thread
  char* patch = smem;                                 // [NSA][324][64 B]
  char* wl = smem + NSA * SP_PPLANE;                  // [TG][NS][64][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int wy = wave >> 1, wx = wave & 1;
#ifdef SP_TRACE
  const bool tr_on = threadIdx.x == 0 && bid_x == (int)(gridDim.x / 2) && bid_y == 0;
  long long tr_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_prev = (long long)__builtin_readcyclecounter();
  const long long tr_t0 = tr_prev;
#endif
  int t = bid_x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
  const int y0 = ty * SP_T, x0 = tx * SP_T;
  // p.halves == 2 (NCB = 2 only): a workgroup takes 32 of a 64-channel tile's output channels — twice the workgroups for
  // launches that would otherwise be a single round of workgroups all in the same phase
  const int cot = p.halves == 2 ? bid_y >> 1 : bid_y;
  const int hco = p.halves == 2 ? 32 * (bid_y & 1) : 0;          // first output channel of this workgroup within the tile
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
  int E_cur = 0, E_min = 1 << 20;                     // F16: the patch scale 2^E of the current chunk, the smallest so far
  auto wave_max = [&](int slot) {                     // max |pre| of this wave -> smax[slot][wave]
    if (SP_EXP & 8) return;
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < PK; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) m = fmaxf(m, fabsf(pre[k][i]));
    m = pv_wave_max_nonneg(m);
    if (lane == 0) smax[slot][wave] = m;
    __threadfence_block();                            // (landed before the next barrier: see pv_conv3_direct_bf16_kernel)
  };
  if constexpr (F16) wave_max(0);
  SP_STAMP(0);                                        // prologue: addresses, first patch + weight requests, wave max (= first data)
  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();                                  // the previous chunk's fragment reads are done (F16: smax is in)
    SP_STAMP(1);
    float psc = 1.0f;
    if constexpr (F16) {
      const float m = fmaxf(fmaxf(smax[ch & 1][0], smax[ch & 1][1]), fmaxf(smax[ch & 1][2], smax[ch & 1][3]));
      const int e = (int)((__float_as_uint(m) >> 23) & 255);
      int E = E_cur;                                  // a (near-)zero chunk keeps the scale and does not count for E_min
      if (e >= 20) {
        E = 140 - e;                                  // max -> [2^13, 2^14)
        if (E_min != (1 << 20) && E > E_min + 30) E = E_min + 30;
        if (E < E_min) E_min = E;
      }
      if (ch > 0 && E != E_cur) {
        const float ratio = sp_pow2(E - E_cur);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int pb = 0; pb < 4; ++pb) acc[cb][pb] *= ratio;
      }
      E_cur = E;
      psc = sp_pow2(E);
    }
#pragma unroll
    for (int k = 0; k < ((SP_EXP & 4) && ch > 0 ? 0 : PK); ++k) {
      const int e = tid + 256 * k, pix = e >> 3, f4 = e & 7;
      const int py = pix / SP_PW;
      u32x2 pl[NSA];
      if constexpr (F16) sp_split4_f16(pre[k], psc, pl);
      else sp_split4<NS>(pre[k], pl);
      const int o = pix * 64 + (((f4 >> 1) ^ (2 * (py & 1))) * 16) + (f4 & 1) * 8;
      if (k + 1 < PK || e < SP_NPIX * 8) {
#pragma unroll
        for (int j = 0; j < NSA; ++j) *reinterpret_cast<u32x2*>(patch + j * SP_PPLANE + o) = pl[j];
      }
    }
    SP_W_STORE();
    SP_STAMP(2);
    __syncthreads();
    SP_STAMP(3);
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
            a[cb][k] = *reinterpret_cast<const sbf8*>(wl + (tt * NS + k) * SP_WPLANE + hco * 64 + cb * 1024 + aoff);
        // B fragments one pixel block ahead of the MFMAs that use them
        const int tofs = (dy * SP_PW + dx) * 64;
        sbf8 bcur[NSA], bnxt[NSA];
#pragma unroll
        for (int k = 0; k < NSA; ++k) bcur[k] = *reinterpret_cast<const sbf8*>(patch + k * SP_PPLANE + boff[0][dy & 1] + tofs);
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
          if (pb + 1 < 4) {
#pragma unroll
            for (int k = 0; k < NSA; ++k)
              bnxt[k] = *reinterpret_cast<const sbf8*>(patch + k * SP_PPLANE + boff[pb + 1 < 4 ? pb + 1 : 3][dy & 1] + tofs);
          }
          // products in ascending magnitude; the NCB accumulators of a product are independent
          if constexpr (NS == 3) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = sp_mma<F16>(a[cb][1], bcur[1], acc[cb][pb]);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = sp_mma<F16>(a[cb][2], bcur[0], acc[cb][pb]);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = sp_mma<F16>(a[cb][0], bcur[2], acc[cb][pb]);
          }
          if constexpr (NS >= 2) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = sp_mma<F16>(a[cb][1], bcur[0], acc[cb][pb]);
            if constexpr (NSA >= 2) {
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = sp_mma<F16>(a[cb][0], bcur[1], acc[cb][pb]);
            }
          }
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = sp_mma<F16>(a[cb][0], bcur[0], acc[cb][pb]);
#pragma unroll
          for (int k = 0; k < NSA; ++k) bcur[k] = bnxt[k];
        }
      }
      if (g + 1 < NG && !(SP_EXP & 2)) {
        __syncthreads();                              // this stage's weight reads are done
        SP_W_STORE();
        __syncthreads();
      }
    }
    SP_STAMP(4);
    if constexpr (F16) {
      if (ch + 1 < nch) wave_max((ch + 1) & 1);       // (the next chunk's values arrived under the MFMAs)
    }
    SP_STAMP(5);
  }
  if constexpr (F16) {                                // back to true units
    const float inv = sp_pow2(-(E_cur + SP_F16_WSHIFT));
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) acc[cb][pb] *= inv;
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
      const int co = cot * SP_TN + hco + cb * 16 + 4 * q;
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
  if (p.pool_out) {
    // 2x max-pool fused (H, W even): a pixel block pb is the two image lines of one pooled line; the window of pooled pixel
    // (pb, j) is lanes r = 2j, 2j + 1, 2j + 8, 2j + 9 of each 16-lane group — partners by DPP (quad xor 1, row rotate 8).
    // Strict > in scan order: the first maximum wins, like torch; the winner's index goes to pool_code.
    const bool writer = (r & 9) == 0;                 // r in {0, 2, 4, 6}
    const int ppy = (y0 >> 1) + 4 * wy, ppx = (x0 >> 1) + 4 * wx + (r >> 1);
    const int Hp = p.H >> 1, Wp = p.W >> 1;
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      const bool okp = ppy + pb < Hp && ppx < Wp;
      const int64_t po = (((int64_t)b * Hp + ppy + pb) * Wp + ppx) * p.Cout;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int co = cot * SP_TN + hco + cb * 16 + 4 * q;
        f32x4 m;
        unsigned code = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v00 = acc[cb][pb][i];
          const int b00 = __float_as_int(v00);
          const float v01 = __int_as_float(__builtin_amdgcn_update_dpp(0, b00, 0xB1, 0xF, 0xF, false));     // lane r ^ 1
          const int b10 = __builtin_amdgcn_update_dpp(0, b00, 0x128, 0xF, 0xF, false);                       // lane (r + 8) % 16
          const float v10 = __int_as_float(b10);
          const float v11 = __int_as_float(__builtin_amdgcn_update_dpp(0, b10, 0xB1, 0xF, 0xF, false));
          float mi = v00;
          unsigned bi = 0;
          if (v01 > mi) { mi = v01; bi = 1; }
          if (v10 > mi) { mi = v10; bi = 2; }
          if (v11 > mi) { mi = v11; bi = 3; }
          m[i] = mi;
          code |= bi << (8 * i);
        }
        if (writer && okp) {
          if (vec && co + 3 < p.Cout) {
            *reinterpret_cast<f32x4*>(p.pool_out + po + co) = m;
            *reinterpret_cast<unsigned*>(p.pool_code + po + co) = code;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (co + i < p.Cout) { p.pool_out[po + co + i] = m[i]; p.pool_code[po + co + i] = (unsigned char)(code >> (8 * i)); }
          }
        }
      }
    }
    return;
  }
  unsigned upc[NCB][4];                               // (un-pooling epilogue: the winner bytes, requested ahead of the arithmetic)
  if (p.up_code) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        const int co = cot * SP_TN + hco + cb * 16 + 4 * q;
        upc[cb][pb] = (ok[pb] && co + 3 < p.Cout) ? *reinterpret_cast<const unsigned*>(p.up_code + ro[pb] + co) : 0u;
      }
  }
  if (p.eg_y) {                                       // out *= act'(eg_y): the producing layer's activation backward
    f32x4 gy[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        const int co = cot * SP_TN + hco + cb * 16 + 4 * q;
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
  if (p.up_code) {                                    // (host: Cout % 4 == 0)
    const int W2 = 2 * p.W;
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      if (!ok[pb]) continue;
      const int y = y0 + 8 * wy + 2 * pb + (r >> 3), x = x0 + 8 * wx + (r & 7);
      float* o00 = p.out + (((int64_t)b * 2 * p.H + 2 * y) * W2 + 2 * x) * p.Cout;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int co = cot * SP_TN + hco + cb * 16 + 4 * q;
        if (co + 3 >= p.Cout) continue;
        const unsigned code = upc[cb][pb];
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
          f32x4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = ((code >> (8 * i)) & 0xffu) == (unsigned)pos ? acc[cb][pb][i] : 0.0f;
          *reinterpret_cast<f32x4*>(o00 + (int64_t)((pos >> 1) * W2 + (pos & 1)) * p.Cout + co) = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    if (!ok[pb]) continue;
    float* orow = p.out + ro[pb];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int co = cot * SP_TN + hco + cb * 16 + 4 * q;
      if (vec && co + 3 < p.Cout) *reinterpret_cast<f32x4*>(orow + co) = acc[cb][pb];
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (co + i < p.Cout) orow[co + i] = acc[cb][pb][i];
      }
    }
  }
#ifdef SP_TRACE
  SP_STAMP(6);
  if (tr_on) {
    for (int k = 0; k < 7; ++k) sp_trace[k] = tr_sum[k];
    sp_trace[7] = (long long)nch;
    sp_trace[8] = (long long)__builtin_readcyclecounter() - tr_t0;
  }
#endif
}

template <int NS, int NCB, bool F16 = false, int NSA = NS>
__global__ __launch_bounds__(256, 2) void pv_conv3_sp_kernel(ConvSp p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  sp_conv_body<NS, NCB, F16, NSA>(p, (int)blockIdx.x, (int)blockIdx.y, smem);
}

// 4: two fp16 pieces with exact power-of-two scaling (default; needs |w| * 64 inside fp16's range, i.e. |w| < ~1023 and not
// all below ~1e-6), 3: three bf16 pieces (no range limit).  The process-wide default is 4 (PV_SP_X6=1 in the environment: 3);
// a plan whose weights leave the safe range asks for 3 itself (ABI v14: pv_ivae_plan.conv_wide, conv_bf16 == 2 of the other
// plans; pv_convstack.h sp_fp32_mode); the v12 / v13 setter is gone since v15
int pv_conv3_sp_fp32_mode() {
  static const int mode = pv_exp_int("PV_SP_X6", 0) ? 3 : 4;
  return mode;
}

bool pv_conv3_sp_supported(int C, int Cout, int nd, int act) {
  return nd == 2 && C >= SP_KC && C % SP_KC == 0 && Cout >= 8 && act != PV_ACT_GELU;
}

// bytes of the tiled, split weights (either orientation fits)
int64_t pv_conv3_sp_wt_bytes(int C, int Cout) {
  const int64_t n = Cout > C ? Cout : C;
  return ((n + SP_TN - 1) / SP_TN) * SP_TN * n * 9 * 3 * 2 + 256;
}

bool sp_pair_capture_fwd(const ConvSp& q, int ncb, dim3 grid, size_t lds);
template <int NS, bool F16 = false, int NSA = NS>
static int conv3_sp_launch(const ConvSp& p, const float* w, int Co, int Ci, int flip, char* wt, int nt, int64_t total,
                           hipStream_t s) {
  if (wt) {
    int pb = (int)((total + 255) / 256);
    if (pb > 2048) pb = 2048;
    hipLaunchKernelGGL((pv_conv3_sp_wprep_kernel<NS, F16>), dim3(pb), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wt), Co,
                       Ci, flip);
    PV_LAUNCH_CHECK();
  }
  constexpr int TG = NS == 3 ? 1 : (NS == 1 ? 3 : 3);
  static const int lds_pad = pv_exp_int("PV_SP_LDS_PAD", 0);   // (occupancy experiments)
  const size_t lds = (size_t)NSA * SP_PPLANE + (size_t)TG * NS * SP_WPLANE + lds_pad;
  const int64_t wgs = (int64_t)p.tiles_x * p.tiles_y * p.B * nt;
  // fewer workgroups than CUs: 32-channel halves fill the chip (measured: 51 -> 37 us on 128 workgroups); with more, the
  // doubled patch staging costs more than the extra round returns (PV_SP_HALVES overrides the limit)
  static const int split_lim = pv_exp_int("PV_SP_HALVES", 255);
  ConvSp q = p;
  q.halves = (p.Cout > 32 && wgs <= split_lim) ? 2 : 1;
  const dim3 grid((unsigned)(p.tiles_x * p.tiles_y * p.B), (unsigned)(nt * q.halves));
  if constexpr (F16 && NS == 2 && NSA == 2) {
    if (sp_pair_capture_fwd(q, (p.Cout <= 32 || q.halves == 2) ? 2 : 4, grid, lds)) return 0;      // (launched by pv_conv3_sp_pair_flush)
  }
  // (PV_LAUNCH_FORK: an input gradient whose result a side-stream weight gradient waits for carries the fork event)
  if (p.Cout <= 32 || q.halves == 2) PV_LAUNCH_FORK((pv_conv3_sp_kernel<NS, 2, F16, NSA>), grid, dim3(256), lds, s, q);
  else PV_LAUNCH_FORK((pv_conv3_sp_kernel<NS, 4, F16, NSA>), grid, dim3(256), lds, s, q);
  PV_LAUNCH_CHECK();
  return 0;
}

// w: raw torch weight (Co, Ci, 3, 3).  flip == 0: out[.., Co] = act(conv(in[.., Ci]) + bias).
// flip == 1: out[.., Ci] = conv of in[.., Co] with the flipped / role-swapped weights (the input gradient).
// ns = 3: fp32-class (six products); ns = 2: mixed precision (three products).  wt_scratch: pv_conv3_sp_wt_bytes bytes.
int pv_conv3_sp(const float* in, int B, int H, int W, const float* w, int Co, int Ci, int flip, const float* bias, float* out,
                int act, void* wt_scratch, hipStream_t s, const float* eg_y, int eg_act, int ns, const void* wt_ready, float* pool_out,
                unsigned char* pool_code, const unsigned char* up_code) {
  const int N = flip ? Ci : Co, C = flip ? Co : Ci;
  if (!pv_conv3_sp_supported(C, N, 2, act) || ns < 1 || ns > 5) return PV_EINVAL;   // 4: fp16 two-piece, 1: fp16 one-piece, 5: fp16 weights two / patch one
  const int nt = (N + SP_TN - 1) / SP_TN;
  const int64_t total = (int64_t)nt * (C / SP_KC) * 9 * SP_TN * SP_KC;
  ConvSp p{};
  p.in = in; p.wt = reinterpret_cast<const char*>(wt_ready ? wt_ready : wt_scratch); p.bias = bias; p.out = out;
  p.eg_y = (eg_y && eg_act != PV_ACT_NONE) ? eg_y : nullptr; p.eg_act = eg_act;
  p.B = B; p.H = H; p.W = W; p.Cin = C; p.Cout = N; p.act = act;
  p.tiles_x = (W + SP_T - 1) / SP_T; p.tiles_y = (H + SP_T - 1) / SP_T;
  if (up_code) {                                     // fused 2x un-pooling (pv_conv_sp.hip ConvSp): whole float4 channel groups only
    if (pool_out || (N & 3)) return PV_EINVAL;
    p.up_code = up_code;
  }
  if (pool_out) {                                    // fused 2x max-pool: forward form, even image sides
    if (flip || !pool_code || (H & 1) || (W & 1) || p.eg_y) return PV_EINVAL;
    p.pool_out = pool_out; p.pool_code = pool_code;
  }
  char* prep = wt_ready ? nullptr : reinterpret_cast<char*>(wt_scratch);    // null: tiled already (pv_conv_wprep_table)
  if (ns == 4) return conv3_sp_launch<2, true>(p, w, Co, Ci, flip, prep, nt, total, s);
  if (ns == 5) return conv3_sp_launch<2, true, 1>(p, w, Co, Ci, flip, prep, nt, total, s);
  if (ns == 1) return conv3_sp_launch<1, true>(p, w, Co, Ci, flip, prep, nt, total, s);
  return ns == 3 ? conv3_sp_launch<3>(p, w, Co, Ci, flip, prep, nt, total, s) : conv3_sp_launch<2>(p, w, Co, Ci, flip, prep, nt, total, s);
}

// test / measurement hook: one convolution call on caller-provided device tensors.
// mode 0: f32-input MFMA direct kernel, 1: its bf16 two-piece form, 5: its fp16 two-piece form, 2 / 3: this file's kernels with
// 2 / 3 bf16 pieces, 4: with two fp16 pieces, 7: with one fp16 piece
extern "C" int pv_debug_conv3(int mode, const float* in, int B, int H, int W, int nd, const float* w, int Co, int Ci, int flip,
                              const float* bias, float* out, int act, void* wt_scratch, const float* eg_y, int eg_act,
                              void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode == 5)      // the round-1 tile kernel (1-D and 2-D) on fp16 two-piece operands
    return pv_conv3_direct(in, B, H, W, nd, w, Co, Ci, flip, bias, out, act, reinterpret_cast<float*>(wt_scratch), s, eg_y, eg_act, 2);
  if (mode == 7) mode = 1 + 16;                      // (7: this file's kernels with ONE fp16 piece)
  if (mode == 8) mode = 5 + 16;                      // (8: weights two fp16 pieces, patch one)
  if (mode >= 2) return nd == 2 ? pv_conv3_sp(in, B, H, W, w, Co, Ci, flip, bias, out, act, wt_scratch, s, eg_y, eg_act, mode & 15)
                                : PV_EINVAL;
  return pv_conv3_direct(in, B, H, W, nd, w, Co, Ci, flip, bias, out, act, reinterpret_cast<float*>(wt_scratch), s, eg_y, eg_act,
                         mode);
}

extern "C" long long pv_debug_conv3_wgrad_ws(int mode, int B, int H, int W, int C, int Cout, int nd) {
  if (mode >= 2) return pv_conv3_sp_wgrad_ws(B, H, W, C, Cout);
  return pv_conv3_wgrad_direct_ws(B, H, W, C, Cout, nd);
}

extern "C" int pv_debug_conv3_wgrad(int mode, const float* dy, const float* in, int B, int H, int W, int C, int nd, float* dw,
                                    float* db, int Cout, void* ws, long long ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode == 1) return pv_conv3_wgrad_direct_bf16(dy, in, B, H, W, C, nd, dw, db, Cout, ws, ws_bytes, s);
  if (mode == 0) return pv_conv3_wgrad_direct(dy, in, B, H, W, C, nd, dw, db, Cout, ws, ws_bytes, s);
  return nd == 2 ? pv_conv3_sp_wgrad(dy, in, B, H, W, C, dw, db, Cout, ws, ws_bytes, s, mode == 7 ? 1 : mode) : PV_EINVAL;
}

// resident workgroups per CU the runtime predicts for the forward kernel (ns pieces, 4 channel blocks) at lds bytes
extern "C" int pv_debug_conv3_sp_occupancy(int ns, int lds) {
  int n = -1;
  hipError_t e = ns == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pv_conv3_sp_kernel<3, 4>, 256, (size_t)lds)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pv_conv3_sp_kernel<2, 4>, 256, (size_t)lds);
  return e == hipSuccess ? n : -(int)e;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolution with exactly split operands:
//   dW[co][ci][tap] = sum_{b, pixel} dY[b, pixel][co] * in[b, pixel + tap][ci] ;   db[co] = sum dY[b, pixel][co]
// A workgroup (4 waves) owns 64 output channels x 64 (or 32) input channels x all 9 taps and walks a contiguous range of
// 8x8 pixel tiles (split s of nsplit); wave (w&1, w>>1) owns 32 co x 32 (16) ci x 9 taps = up to 36 accumulator blocks
// that stay in registers across tiles.  The contraction runs over pixels (MFMA k = 32 pixels = 4 tile lines x 8): the dY
// tile and the input patch with its halo are split into NS bf16 planes while they are staged in their natural
// [pixel][channel] layouts, and both operands come out of LDS through the transposing read ds_read_b64_tr_b16 (a lane gets
// 4 consecutive pixels of one channel).  A lane's 8 k values are 8 consecutive pixels of one line, so the three taps
// dx = 0, 1, 2 of a kernel row are windows of ONE 12-pixel line read: dx = 0 and 2 are register sub-ranges, dx = 1 is
// four v_alignbit — a third of the patch reads of a per-tap formulation, which is what lets two workgroups per CU run
// under the LDS bandwidth.  (The fp16 forms, round 4, read the three windows separately after all: at their MFMA count the loop
// was bound by VALU issue, not by the LDS, and the register shifts were a quarter of its VALU work.)  Pixel rows are 160 bytes apart (64 channels + pad) with a per-line skew so that the two
// 16-lane halves of a transposing read (two adjacent lines) land on disjoint banks.
// Per-split partial results are summed in split order by pv_conv3_wgrad_finish_kernel (no atomics).
struct ConvWgSp {
  const float* dy; const float* in; float* part; float* part_b;
  int B, H, W, Cin, Cout, tiles_x, tiles_y, nsplit;
};

typedef short sshort4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) sshort4 lds_sshort4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 sw_tr(const char* p) {
  const sshort4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_sshort4*)p);
  return __builtin_bit_cast(u32x2, v);
}
__device__ __forceinline__ sbf8 sw_frag(unsigned a, unsigned b, unsigned c, unsigned d) {
  return __builtin_bit_cast(sbf8, u32x4{a, b, c, d});
}
// LDS geometry.  A transposing read's 16-lane group covers 4 consecutive pixels x 32 bytes; its two groups per 32-lane
// half are two adjacent lines.  Pixel pitches of 160 (64 channels) / 96 (32 channels) bytes put 4 consecutive pixels on
// disjoint 32-byte bank segments, and line pitches congruent to 128 mod 256 put the adjacent line on the other four.
#define SW_DYP 160                                 // dY tile: pixel pitch; line (8 pixels) pitch 1408
#define SW_DYL 1408
#define SW_DYPLANE (8 * SW_DYL)
template <int NCIB> struct SwGeo {
  static constexpr int PP = NCIB == 2 ? 160 : 96;    // patch pixel pitch
  static constexpr int RP = NCIB == 2 ? 1664 : 1152; // patch line (10 pixels) pitch
  static constexpr int PLANE = 10 * RP + 2 * PP;     // (+ the two never-used pixels a line window reads past the halo)
};

// F16: fp16 two-piece operands (see pv_conv3_sp_kernel) — each staged dY tile and patch tile is scaled by its own exact power
// of two (maximum to [2^13, 2^14)); the accumulators live in units of 2^(E_dy + E_patch) (the bias sums in 2^E_dy) and are
// rescaled when a tile changes them; neither exponent may exceed the smallest so far by more than 30 (overflow guard —
// a tile that would need more is 2^30 below what is already summed).
#ifndef SW_BUF
#define SW_BUF 1
#endif
// phase-timing trace (profiling builds only: -DSW_TRACE, scripts/gpu_trace_sw.py): shader-clock sums of the middle split's
// first owner, thread 0
#ifdef SW_TRACE
__device__ long long sw_trace[16];
#define SW_STAMP(k) do { if (tr_on) { const long long c_ = (long long)__builtin_readcyclecounter(); tr_sum[(k)] += c_ - tr_prev; tr_prev = c_; } } while (0)
extern "C" int pv_debug_read_trace_sw(long long* out, int n) {
  if (n > 16) n = 16;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sw_trace), n * sizeof(long long));
}
#else
#define SW_STAMP(k) do { } while (0)
#endif
template <int NS, int NCIB, bool F16 = false>
__device__ __forceinline__ void sp_wgrad_body(const ConvWgSp& p, const int bid_x, const int bid_y, const int bid_z, char* smem) {
  static_assert(F16 ? ((NS == 2 || NS == 1) && NCIB == 1) : NS >= 2, "the fp16 modes: two pieces or one, 32-channel workgroups (next-tile prefetch)");
  __shared__ float smx[2][2][4];                     // F16: [tile parity][dY | patch][wave] maxima of the tile being staged
  constexpr int PP = SwGeo<NCIB>::PP, RP = SwGeo<NCIB>::RP, PPLANE = SwGeo<NCIB>::PLANE;
  char* dyl = smem;                                  // [NS] planes
  char* patch = smem + NS * SW_DYPLANE;              // [NS] planes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int wco = wave & 1, wci = wave >> 1;
  const int split = bid_x, cit = bid_y, cot = bid_z;
  constexpr int CIW = 32 * NCIB;                     // input channels per workgroup
  const int64_t T = (int64_t)p.B * p.tiles_y * p.tiles_x;
  const int64_t t_lo = T * split / p.nsplit, t_hi = T * (split + 1) / p.nsplit;
  f32x4 acc[9][2][NCIB], accb[2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < NCIB; ++c) acc[t][a][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  accb[0] = accb[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const unsigned one2 = F16 ? 0x3c003c00u : 0x3f803f80u;                  // 1.0 twice, fp16 / bf16
  const sbf8 ones = sw_frag(one2, one2, one2, one2);
  const bool want_b = p.part_b && cit == 0 && wci == 0;
  // transposing-read addresses: lane r points at pixel (+ r>>2), channels 4 (r&3) .. +3 of the 16-channel block
  const int a_off = q * SW_DYL + (r >> 2) * SW_DYP + (r & 3) * 8 + wco * 64;
  int b_off[3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
    b_off[dy] = (q + dy) * RP + (r >> 2) * PP + (r & 3) * 8 + wci * (NCIB * 32);
  // staging: a thread's fp32 pieces of a tile — 4 of the dY tile (pixel n, channels 4 c4 .. +3), PK of the 10x10 halo window
  // of the input (CIW channels) — are fetched into registers, split into the NS planes and written to LDS.  With 32-channel
  // workgroups (PRE) the NEXT tile's pieces are fetched under the current tile's MFMAs.
  constexpr int PPT = CIW / 4;                       // float4 pieces per patch pixel
  constexpr int PK = (100 * PPT + 255) / 256;
  constexpr bool PRE = NCIB == 1;
  f32x4 vd[4], vp[PK];
  // per-thread constants of its pieces: position in the tile / halo window, LDS offset (the same for every tile)
  int dn_y[4], dn_x[4], d_lds[4], pp_r[PK], pp_c[PK], p_lds[PK];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = tid + 256 * k, n = e >> 4, c4 = e & 15;
    dn_y[k] = n >> 3; dn_x[k] = n & 7;
    d_lds[k] = (n >> 3) * SW_DYL + (n & 7) * SW_DYP + c4 * 8;
  }
#pragma unroll
  for (int k = 0; k < PK; ++k) {
    const int e = tid + 256 * k, pix = e / PPT, f = e - pix * PPT;
    const int row = pix / 10, col = pix - row * 10;
    pp_r[k] = e < 100 * PPT ? row : 1 << 20;         // (pieces past the window: never in the image)
    pp_c[k] = col;
    p_lds[k] = row * RP + col * PP + f * 8;
  }
  const int dco = cot * 64 + 4 * (tid & 15), pcf = cit * CIW + 4 * (tid % PPT);
  const bool dvec = dco + 3 < p.Cout && (p.Cout & 3) == 0;
  // the tile to fetch next, advanced incrementally (no divisions in the loop)
  int ftx, fty, fb;
  {
    int t = (int)t_lo;
    ftx = t % p.tiles_x; t /= p.tiles_x;
    fty = t % p.tiles_y; fb = t / p.tiles_y;
  }
  // Buffer-load form of the fetch (SW_BUF): one descriptor per tensor, a 32-bit byte offset per piece, pieces outside the image
  // pointed past the descriptor's range (they read 0) — no exec-mask branch, zero fill and 64-bit address chain per piece (the
  // loop issued ~330 VALU and ~210 SALU next to its 116 MFMA).  Tensors under 2 GiB and whole 64-channel dY blocks only.
  const bool use_buf = SW_BUF && (p.Cout & 63) == 0 && (int64_t)p.B * p.H * p.W * p.Cout < (1ll << 29) &&
                       (int64_t)p.B * p.H * p.W * p.Cin < (1ll << 29);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0,
                                         use_buf ? (int)((int64_t)p.B * p.H * p.W * p.Cout * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0,
                                         use_buf ? (int)((int64_t)p.B * p.H * p.W * p.Cin * 4) : 0, 0x00020000);
  // Exact tilings (H and W multiples of 8: every BASELINE shape): a piece's byte offset is the tile's base (scalar) plus a
  // per-thread constant, and a halo piece lies outside the image only when its tile touches that border — 5 class bits per piece
  // (top / bottom / left / right / no such piece) against the tile's border mask: no multiplies and no coordinate compares per
  // piece (they were 16 v_mul_lo / v_mad_u64 — quarter rate — and ~60 more VALU instructions per tile)
  const bool exact = use_buf && (p.H & 7) == 0 && (p.W & 7) == 0;
  int d_const[4], p_const[PK];
  unsigned long long p_cls = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) d_const[k] = ((dn_y[k] * p.W + dn_x[k]) * p.Cout + dco) * 4;
#pragma unroll
  for (int k = 0; k < PK; ++k) {
    const int ry = pp_r[k] - 1, rx = pp_c[k] - 1;
    p_const[k] = pp_r[k] < (1 << 20) ? ((ry * p.W + rx) * p.Cin + pcf) * 4 : 0;
    const unsigned c = pp_r[k] < (1 << 20) ? ((ry < 0 ? 1u : 0u) | (ry > 7 ? 2u : 0u) | (rx < 0 ? 4u : 0u) | (rx > 7 ? 8u : 0u)) : 16u;
    p_cls |= (unsigned long long)c << (5 * k);
  }
  static_assert(PK * 5 <= 64, "class bits of the patch pieces in one 64-bit word");
  auto fetch_exact = [&]() {
    const unsigned pix0 = ((unsigned)fb * (unsigned)p.H + (unsigned)(fty * 8)) * (unsigned)p.W + (unsigned)(ftx * 8);
    const unsigned base_d = pix0 * (unsigned)p.Cout * 4u, base_p = pix0 * (unsigned)p.Cin * 4u;
    const unsigned border = (fty == 0 ? 1u : 0u) | (fty == p.tiles_y - 1 ? 2u : 0u) | (ftx == 0 ? 4u : 0u) |
                            (ftx == p.tiles_x - 1 ? 8u : 0u) | 16u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(drs, base_d + (unsigned)d_const[k], 0, 0);
      vd[k] = __builtin_bit_cast(f32x4, v);
    }
#pragma unroll
    for (int k = 0; k < PK; ++k) {
      const bool out = ((unsigned)(p_cls >> (5 * k)) & border) != 0;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(irs, out ? 0x80000000u : base_p + (unsigned)p_const[k], 0, 0);
      vp[k] = __builtin_bit_cast(f32x4, v);
    }
    if (++ftx == p.tiles_x) { ftx = 0; if (++fty == p.tiles_y) { fty = 0; ++fb; } }
  };
  auto fetch_buf = [&]() {
    const int y0 = fty * 8, x0 = ftx * 8;
    const unsigned img = (unsigned)fb * (unsigned)(p.H * p.W);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned y = (unsigned)(y0 + dn_y[k]), x = (unsigned)(x0 + dn_x[k]);
      const unsigned off = ((img + y * (unsigned)p.W + x) * (unsigned)p.Cout + (unsigned)dco) * 4u;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(drs, (y < (unsigned)p.H && x < (unsigned)p.W) ? off : 0x80000000u, 0, 0);
      vd[k] = __builtin_bit_cast(f32x4, v);
    }
#pragma unroll
    for (int k = 0; k < PK; ++k) {
      const unsigned y = (unsigned)(y0 - 1 + pp_r[k]), x = (unsigned)(x0 - 1 + pp_c[k]);     // (negative: wraps past H / W)
      const unsigned off = ((img + y * (unsigned)p.W + x) * (unsigned)p.Cin + (unsigned)pcf) * 4u;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(irs, (y < (unsigned)p.H && x < (unsigned)p.W) ? off : 0x80000000u, 0, 0);
      vp[k] = __builtin_bit_cast(f32x4, v);
    }
    if (++ftx == p.tiles_x) { ftx = 0; if (++fty == p.tiles_y) { fty = 0; ++fb; } }
  };
  auto fetch_ptr = [&]() {
    const int y0 = fty * 8, x0 = ftx * 8;
    const float* dy_b = p.dy + (int64_t)fb * p.H * p.W * p.Cout + dco;
    const float* in_b = p.in + (int64_t)fb * p.H * p.W * p.Cin + pcf;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = y0 + dn_y[k], x = x0 + dn_x[k];
      vd[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (y < p.H && x < p.W) {
        const float* src = dy_b + ((int64_t)y * p.W + x) * p.Cout;
        if (dvec) vd[k] = *reinterpret_cast<const f32x4*>(src);
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i) if (dco + i < p.Cout) vd[k][i] = src[i];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < PK; ++k) {
      const int y = y0 - 1 + pp_r[k], x = x0 - 1 + pp_c[k];
      vp[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (y >= 0 && y < p.H && x >= 0 && x < p.W) vp[k] = *reinterpret_cast<const f32x4*>(in_b + ((int64_t)y * p.W + x) * p.Cin);
    }
    if (++ftx == p.tiles_x) { ftx = 0; if (++fty == p.tiles_y) { fty = 0; ++fb; } }
  };
  auto fetch = [&]() { if (exact) fetch_exact(); else if (use_buf) fetch_buf(); else fetch_ptr(); };
  if (PRE && t_lo < t_hi) fetch();
  int Ed = 0, Ep = 0, Ed_min = 1 << 20, Et_min = 1 << 20;           // F16: current exponents, smallest so far (dY, dY + patch)
  auto wave_max = [&](int slot) {                    // this wave's max |dY piece| and |patch piece| of the fetched tile
    if (SP_EXP & 8) return;
    float md = 0.0f, mp = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) md = fmaxf(md, fabsf(vd[k][i]));
#pragma unroll
    for (int k = 0; k < PK; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) mp = fmaxf(mp, fabsf(vp[k][i]));
    md = pv_wave_max_nonneg(md); mp = pv_wave_max_nonneg(mp);
    if (lane == 0) { smx[slot][0][wave] = md; smx[slot][1][wave] = mp; }
    __threadfence_block();                           // (landed before the next barrier: see pv_conv3_direct_bf16_kernel)
  };
  if constexpr (F16) { if (t_lo < t_hi) wave_max(0); }
  int par = 0;
#ifdef SW_TRACE
  const bool tr_on = tid == 0 && split == p.nsplit / 2 && cit == 0 && cot == 0;
  long long tr_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_prev = (long long)__builtin_readcyclecounter();
  const long long tr_t0 = tr_prev;
#endif
  for (int64_t tt = t_lo; tt < t_hi; ++tt) {
    __syncthreads();                                 // the previous tile's fragment reads are done (F16: smx is in)
    SW_STAMP(0);
    if (!PRE) fetch();
    float scd = 1.0f, scp = 1.0f;
    if constexpr (F16) {
      const float md = fmaxf(fmaxf(smx[par][0][0], smx[par][0][1]), fmaxf(smx[par][0][2], smx[par][0][3]));
      const float mp = fmaxf(fmaxf(smx[par][1][0], smx[par][1][1]), fmaxf(smx[par][1][2], smx[par][1][3]));
      const int ed = (int)((__float_as_uint(md) >> 23) & 255), ep = (int)((__float_as_uint(mp) >> 23) & 255);
      int nEd = Ed, nEp = Ep;                        // a (near-)zero tile keeps the scales and contributes nothing
      if (ed >= 20 && ep >= 20) {
        nEd = 140 - ed; nEp = 140 - ep;              // maxima -> [2^13, 2^14)
        if (Ed_min != (1 << 20) && nEd > Ed_min + 30) nEd = Ed_min + 30;
        if (nEd < Ed_min) Ed_min = nEd;
        if (Et_min != (1 << 20) && nEd + nEp > Et_min + 30) nEp = Et_min + 30 - nEd;
        if (nEd + nEp < Et_min) Et_min = nEd + nEp;
      }
      if (nEd + nEp != Ed + Ep) {
        const float ratio = sp_pow2(nEd + nEp - Ed - Ep);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < NCIB; ++c) acc[t][a][c] *= ratio;
      }
      if (nEd != Ed) {
        const float ratio = sp_pow2(nEd - Ed);
        accb[0] *= ratio; accb[1] *= ratio;
      }
      Ed = nEd; Ep = nEp;
      scd = sp_pow2(Ed); scp = sp_pow2(Ep);
      par ^= 1;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      u32x2 pl[NS];
      if constexpr (F16) sp_split4_f16(vd[k], scd, pl);
      else sp_split4<NS>(vd[k], pl);
#pragma unroll
      for (int j = 0; j < NS; ++j) *reinterpret_cast<u32x2*>(dyl + j * SW_DYPLANE + d_lds[k]) = pl[j];
    }
#pragma unroll
    for (int k = 0; k < PK; ++k) {
      u32x2 pl[NS];
      if constexpr (F16) sp_split4_f16(vp[k], scp, pl);
      else sp_split4<NS>(vp[k], pl);
      if (pp_r[k] < (1 << 20)) {
#pragma unroll
        for (int j = 0; j < NS; ++j) *reinterpret_cast<u32x2*>(patch + j * PPLANE + p_lds[k]) = pl[j];
      }
    }
    SW_STAMP(1);
    __syncthreads();
    SW_STAMP(2);
    if (PRE && tt + 1 < t_hi) fetch();
    SW_STAMP(3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                 // 32 pixels: lane group q <-> tile line 4 ks + q, k slot i <-> x = i
      sbf8 a[2][NS];
#pragma unroll
      for (int cob = 0; cob < 2; ++cob)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          const char* ap = dyl + j * SW_DYPLANE + a_off + ks * (4 * SW_DYL) + cob * 32;
          const u32x2 lo = sw_tr(ap), hi = sw_tr(ap + 4 * SW_DYP);
          a[cob][j] = sw_frag(lo[0], lo[1], hi[0], hi[1]);
        }
      if (want_b) {
#pragma unroll
        for (int cob = 0; cob < 2; ++cob)
#pragma unroll
          for (int j = 0; j < NS; ++j) accb[cob] = sp_mma<F16>(a[cob][j], ones, accb[cob]);
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
        for (int cib = 0; cib < NCIB; ++cib) {
          sbf8 bb[3][NS];
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            const char* bp = patch + j * PPLANE + b_off[dy] + ks * (4 * RP) + cib * 32;
            if constexpr (F16) {
              // fp16 forms (round 4): the three taps of a kernel row are three windows of the line read straight from LDS — a
              // transposing read starts at any pixel, and the pixel pitch keeps every start conflict-free — instead of one 12-pixel
              // window shifted in registers (4 v_alignbit + ~10 v_mov per plane and kernel row: a third of this loop's VALU work)
#pragma unroll
              for (int dx = 0; dx < 3; ++dx) {
                const u32x2 lo = sw_tr(bp + dx * PP), hi = sw_tr(bp + dx * PP + 4 * PP);
                bb[dx][j] = sw_frag(lo[0], lo[1], hi[0], hi[1]);
              }
            } else {
              // one 12-pixel line window per plane: dwords w0..w4 = pixels (0,1) .. (8,9) [(10,11) are never used]
              const u32x2 w01 = sw_tr(bp), w23 = sw_tr(bp + 4 * PP), w45 = sw_tr(bp + 8 * PP);
              bb[0][j] = sw_frag(w01[0], w01[1], w23[0], w23[1]);
              bb[1][j] = sw_frag(__builtin_amdgcn_alignbit(w01[1], w01[0], 16), __builtin_amdgcn_alignbit(w23[0], w01[1], 16),
                                 __builtin_amdgcn_alignbit(w23[1], w23[0], 16), __builtin_amdgcn_alignbit(w45[0], w23[1], 16));
              bb[2][j] = sw_frag(w01[1], w23[0], w23[1], w45[0]);
            }
          }
#define SW_PROD(KA, KB)                                                                                                 \
  _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) _Pragma("unroll") for (int cob = 0; cob < 2; ++cob)                   \
      acc[3 * dy + dx][cob][cib] = sp_mma<F16>(a[cob][KA], bb[dx][KB], acc[3 * dy + dx][cob][cib]);
          if constexpr (NS == 3) { SW_PROD(1, 1) SW_PROD(2, 0) SW_PROD(0, 2) }
          if constexpr (NS >= 2) { SW_PROD(1, 0) SW_PROD(0, 1) }
          SW_PROD(0, 0)
        }
      }
    }
    SW_STAMP(4);
    if constexpr (F16) {
      if (tt + 1 < t_hi) wave_max(par);              // (the next tile's pieces arrived under the MFMAs)
    }
    SW_STAMP(5);
  }
#ifdef SW_TRACE
  if (tr_on) {
    for (int k = 0; k < 6; ++k) sw_trace[k] = tr_sum[k];
    sw_trace[6] = (long long)(t_hi - t_lo);
    sw_trace[7] = (long long)__builtin_readcyclecounter() - tr_t0;
  }
#endif
  if constexpr (F16) {                               // back to true units
    const float inv = sp_pow2(-(Ed + Ep)), invb = sp_pow2(-Ed);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < NCIB; ++c) acc[t][a][c] *= inv;
    accb[0] *= invb; accb[1] *= invb;
  }
  // C/D layout: lane (column = ci r, q), reg i -> output channel 16*cob + 4q + i of the wave's 32
#pragma unroll
  for (int cob = 0; cob < 2; ++cob)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = cot * 64 + wco * 32 + cob * 16 + 4 * q + i;
      if (co >= p.Cout) continue;
#pragma unroll
      for (int cib = 0; cib < NCIB; ++cib) {
        const int ci = cit * CIW + wci * (NCIB * 16) + cib * 16 + r;
        float* dst = p.part + (((int64_t)split * p.Cout + co) * p.Cin + ci) * 9;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) dst[tap] = acc[tap][cob][cib][i];
      }
      if (want_b && r == 0) p.part_b[(int64_t)split * p.Cout + co] = accb[cob][i];
    }
}

template <int NS, int NCIB, bool F16 = false>
__global__ __launch_bounds__(256, 2) void pv_conv3_sp_wgrad_kernel(ConvWgSp p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  sp_wgrad_body<NS, NCIB, F16>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, smem);
}

// A layer's input gradient and weight gradient in ONE launch (fp16 two-piece forms): they are independent (both read dY),
// each is memory phase + MFMA phase in sequence with every workgroup of a launch in the same phase — interleaved on the
// same CUs (even workgroups: input gradient, odd: weight gradient, two resident per CU) one's MFMAs cover the other's
// fetches and stores.  Two streams gave -6...-19 % in isolation but lost it to their event barriers; one launch has none.
struct SpPairIdx { int nA, gAx, nB, gBx, gBy; };
template <int NCB>
__global__ __launch_bounds__(256, 2) void pv_conv3_sp_pair_kernel(ConvSp pa, ConvWgSp pb, SpPairIdx ix) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = (int)blockIdx.x, m = ix.nA < ix.nB ? ix.nA : ix.nB;
  bool isA; int idx;
  if (b < 2 * m) { isA = (b & 1) == 0; idx = b >> 1; }
  else { isA = ix.nA > ix.nB; idx = b - m; }
  if (isA) sp_conv_body<2, NCB, true>(pa, idx % ix.gAx, idx / ix.gAx, smem);
  else sp_wgrad_body<2, 1, true>(pb, idx % ix.gBx, (idx / ix.gBx) % ix.gBy, idx / (ix.gBx * ix.gBy), smem);
}

// capture of the two launches of a pair (pv_conv3_sp_pair_begin ... pv_conv3_sp_pair_flush, same host thread)
struct SpPairCap {
  bool on = false, haveA = false, haveB = false;
  ConvSp a; int ncb = 0; dim3 gridA; size_t ldsA = 0;
  ConvWgSp b; dim3 gridB; size_t ldsB = 0;
};
static thread_local SpPairCap g_pair;

bool sp_pair_capture_fwd(const ConvSp& q, int ncb, dim3 grid, size_t lds) {
  if (!g_pair.on || g_pair.haveA || q.pool_out) return false;
  g_pair.a = q; g_pair.ncb = ncb; g_pair.gridA = grid; g_pair.ldsA = lds; g_pair.haveA = true;
  return true;
}
void pv_conv3_sp_pair_begin() {
  static const int on = pv_exp_int("PV_NO_SPPAIR", 0) ? 0 : 1;
  g_pair = SpPairCap{};
  g_pair.on = on != 0;
}
int pv_conv3_sp_pair_flush(hipStream_t s) {
  SpPairCap c = g_pair;
  g_pair = SpPairCap{};
  // only while the two together are well under two rounds of workgroups (768: conv-encoder iVAE at batch 128 -12 us per step);
  // beyond that each launch fills the chip on its own and interleaving costs (VED at batch 256, every layer paired: +30 us).
  // PV_SPPAIR_MAX overrides.
  static const int64_t pair_max = pv_exp_ll("PV_SPPAIR_MAX", 768);
  const int64_t nab = (int64_t)c.gridA.x * c.gridA.y + (int64_t)c.gridB.x * c.gridB.y * c.gridB.z;
  if (c.haveA && c.haveB && nab > pair_max) {
    hipLaunchKernelGGL((pv_conv3_sp_wgrad_kernel<2, 1, true>), c.gridB, dim3(256), c.ldsB, s, c.b);
    PV_LAUNCH_CHECK();
    c.haveB = false;
  }
  if (c.haveA && c.haveB) {
    SpPairIdx ix{(int)(c.gridA.x * c.gridA.y), (int)c.gridA.x, (int)(c.gridB.x * c.gridB.y * c.gridB.z), (int)c.gridB.x, (int)c.gridB.y};
    const size_t lds = c.ldsA > c.ldsB ? c.ldsA : c.ldsB;
    const dim3 grid((unsigned)(ix.nA + ix.nB));
    if (c.ncb == 2) hipLaunchKernelGGL((pv_conv3_sp_pair_kernel<2>), grid, dim3(256), lds, s, c.a, c.b, ix);
    else hipLaunchKernelGGL((pv_conv3_sp_pair_kernel<4>), grid, dim3(256), lds, s, c.a, c.b, ix);
    PV_LAUNCH_CHECK();
  } else if (c.haveA) {
    if (c.ncb == 2) hipLaunchKernelGGL((pv_conv3_sp_kernel<2, 2, true>), c.gridA, dim3(256), c.ldsA, s, c.a);
    else hipLaunchKernelGGL((pv_conv3_sp_kernel<2, 4, true>), c.gridA, dim3(256), c.ldsA, s, c.a);
    PV_LAUNCH_CHECK();
  } else if (c.haveB) {
    hipLaunchKernelGGL((pv_conv3_sp_wgrad_kernel<2, 1, true>), c.gridB, dim3(256), c.ldsB, s, c.b);
    PV_LAUNCH_CHECK();
  }
  return 0;
}

int pv_wgrad_finish_blocks(int64_t nw, int nb);
extern __global__ void pv_conv3_wgrad_finish_kernel(const float* __restrict__ part, int nsplit, int64_t n, float* __restrict__ out,
                                                    const float* __restrict__ part_b, int nb, float* __restrict__ out_b);

static int sw_splits(int B, int H, int W, int C, int Cout, int nsp) {
  const int64_t T = (int64_t)B * ((W + 7) / 8) * ((H + 7) / 8);
  const int ciw = (nsp == 2 && C % 64 == 0) ? 64 : 32;           // input channels per workgroup (three planes: 32, for the LDS)
  const int64_t owners = (int64_t)(C / ciw) * ((Cout + 63) / 64);
  static const int wg_target = pv_exp_int("PV_SW_WGS", 512);
  // the one-piece form (conv mode 4's weight gradient) runs NEXT TO the input-gradient chain on the side stream: 320 workgroups
  // leave that chain more of the chip and write 5/8 of the partials (round 5, `gpurun_out/r05al`, `r05am`, two boxes: VED at batch
  // 256 0.7755 -> 0.7589 ms, conv-encoder iVAE at batch 128 0.7611 -> 0.7508; 256 / 288 / 352 / 384 / 448 in between or worse)
  static const int wg_target1 = pv_exp_int("PV_SW_WGS1", 320);
  int64_t ns = ((nsp == 1 ? wg_target1 : wg_target) + owners - 1) / owners;   // (512: two workgroups per CU in all)
  // ... but never fewer than 12 tiles per split: below that a split's fixed costs (its partial-sum block of the finish launch,
  // its prologue) outweigh the parallelism (round 5, `gpurun_out/r05ab3`: conv-encoder iVAE at batch 128 0.836 -> 0.826 ms
  // fp32-class, 0.562 -> 0.545 at the throughput precision; VED at batch 256 — 8+ tiles per split already — unchanged)
  static const int min_tiles = pv_exp_int("PV_SW_MINTILES", 12);
  if (min_tiles > 1 && ns > (T + min_tiles - 1) / min_tiles) ns = (T + min_tiles - 1) / min_tiles;
  if (ns > T) ns = T;
  return (int)(ns < 1 ? 1 : ns);
}

bool pv_conv3_sp_wgrad_supported(int C, int Cout, int nd) { return nd == 2 && C >= 32 && C % 32 == 0 && Cout >= 8; }

int64_t pv_conv3_sp_wgrad_ws(int B, int H, int W, int C, int Cout) {
  if (!pv_conv3_sp_wgrad_supported(C, Cout, 2)) return 0;
  const int n2 = sw_splits(B, H, W, C, Cout, 2), n3 = sw_splits(B, H, W, C, Cout, 3);       // either precision
  return (int64_t)(n2 > n3 ? n2 : n3) * ((int64_t)Cout * C * 9 + Cout) * (int64_t)sizeof(float) + 256;
}

int pv_conv3_sp_wgrad(const float* dy, const float* in, int B, int H, int W, int C, float* dw, float* db, int Cout, void* ws,
                      int64_t ws_bytes, hipStream_t s, int ns, PvFinishList* defer) {
  if (!pv_conv3_sp_wgrad_supported(C, Cout, 2) || ns < 1 || ns > 4) return PV_EINVAL;   // 4: fp16 two-piece, 1: fp16 one-piece
  if (!pv_wgrad_ws(defer, pv_conv3_sp_wgrad_ws(B, H, W, C, Cout), ws, ws_bytes)) defer = nullptr;
  if (ws_bytes < pv_conv3_sp_wgrad_ws(B, H, W, C, Cout)) return PV_EWS;
  ConvWgSp p{};
  p.dy = dy; p.in = in; p.B = B; p.H = H; p.W = W; p.Cin = C; p.Cout = Cout;
  p.tiles_x = (W + 7) / 8; p.tiles_y = (H + 7) / 8;
  p.nsplit = sw_splits(B, H, W, C, Cout, ns);
  const int64_t nw = (int64_t)Cout * C * 9;
  p.part = reinterpret_cast<float*>(ws);
  p.part_b = db ? p.part + (int64_t)p.nsplit * nw : nullptr;
  const bool wide = ns == 2 && C % 64 == 0;
  const dim3 grid((unsigned)p.nsplit, (unsigned)(wide ? C / 64 : C / 32), (unsigned)((Cout + 63) / 64));
  const size_t lds = (size_t)(ns == 4 ? 2 : ns) * (SW_DYPLANE + (wide ? SwGeo<2>::PLANE : SwGeo<1>::PLANE)) + (ns == 1 ? 0 : 0);
  if (ns == 4 && defer && g_pair.on && !g_pair.haveB) {          // (deferred reduction: the kernel may run later, in the pair launch)
    g_pair.b = p; g_pair.gridB = grid; g_pair.ldsB = lds; g_pair.haveB = true;
  } else if (ns == 4) {
    hipLaunchKernelGGL((pv_conv3_sp_wgrad_kernel<2, 1, true>), grid, dim3(256), lds, s, p);
  } else if (ns == 1) {
    hipLaunchKernelGGL((pv_conv3_sp_wgrad_kernel<1, 1, true>), grid, dim3(256), lds, s, p);
  } else if (ns == 3) {
    hipLaunchKernelGGL((pv_conv3_sp_wgrad_kernel<3, 1>), grid, dim3(256), lds, s, p);
  } else {
    if (wide) hipLaunchKernelGGL((pv_conv3_sp_wgrad_kernel<2, 2>), grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL((pv_conv3_sp_wgrad_kernel<2, 1>), grid, dim3(256), lds, s, p);
  }
  PV_LAUNCH_CHECK();
  return pv_wgrad_finish(defer, p.part, p.nsplit, nw, dw, p.part_b, Cout, db, s);
}

// test hook: convolution + activation + fused 2x max-pool (pooled values and winner bytes), and the pooling's backward
extern "C" int pv_debug_conv3_pool(int mode, const float* in, int B, int H, int W, const float* w, int Co, int Ci, const float* bias,
                                   int act, float* pooled, unsigned char* code, void* wt_scratch, const float* g, float* din,
                                   void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!g) return pv_conv3_sp(in, B, H, W, w, Co, Ci, 0, bias, pooled, act, wt_scratch, s, nullptr, 0, mode, nullptr, pooled, code);
  return pv_maxpool2_bwd_code(g, pooled, code, din, B, H / 2, W / 2, Co, act, s);
}
