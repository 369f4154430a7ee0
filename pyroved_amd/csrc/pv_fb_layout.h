// pv_fb_layout.h — the LDS / global layout of the bf16x3 kernel's weight images (pv_sdec_fused_bf16.hip) and the
// per-step preparation that writes them; shared with the kernel that hosts the preparation (pv_encoder.hip).
#pragma once
#include "pv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define LDB 128                  // bf16 elements per LDS row of the weight images: unpadded, 16-byte chunks XOR-swizzled
#define W_IMG (128 * LDB)        // elements of one weight image (hidden width 128)
#define IMG_BYTES (2 * W_IMG)    // 32,768

// x -> (hi, lo) bf16 with hi + lo = x to ~2^-17 relative
__device__ __forceinline__ void fb_split(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

// LDS weight images are stored with their columns permuted inside every block of 32: logical column
// k = 32m + 16h + 4q + i (h in {0,1}, q in 0..3, i in 0..3) sits at physical column 32m + 8q + 4h + i, so that the
// 8 k's a lane feeds to one v_mfma_f32_16x16x32_bf16 ({32m+4q+i} and {32m+16+4q+i}: the C/D layout of the
// producing layer) are CONTIGUOUS: the forward A operand is one ds_read_b128.  Groups of 4 consecutive logical
// columns stay contiguous, which is all the transposing dgrad read needs.
__device__ __forceinline__ int fb_pcol(int k) {
  return (k & ~31) | (((k >> 2) & 3) << 3) | (((k >> 4) & 1) << 2) | (k & 3);
}
// q-swapped form (PvFbPrep::qswap, round 6; the 4-wave kernel's images): the half a column block sits in is h ^ (q >> 1) — chunks
// q = 2, 3 hold [h = 1 | h = 0].  The forward's operand stays one ds_read_b128 (its lanes of groups q >= 2 feed the activation
// pieces in the same swapped order: fb_catq); the dgrad's transposing read of column block 2m + h now takes pieces q' = 0, 1 from
// half h and q' = 2, 3 from the other one, so its 32 lanes (8 rows x 4 pieces) cover 16 chunks x BOTH halves = all 64 banks
// once: conflict-free, where the plain form has every lane on the same half (2-way).
// ... and every row R of an image has its sixteen 16-byte chunks XOR-swizzled by fb_swz(R) = 4*(R&3) + SL[(R>>2)&3],
// SL = {0,2,3,1}: (a) the forward's ds_read_b128 (lane (r,q): row 16*ob + r, chunk 4m + q; serviced in the 16-lane
// groups {0-3,12-15,20-27}, ...) touches 16 distinct chunks per group = all 64 banks; (b) the dgrad's transposing
// 8-byte reads (32 lanes: 8 rows x 4 chunks, one half of each chunk) are 2-way while all lanes want the same half (the
// plain form; conflict-free in the q-swapped one).  Unpadded rows make an image exactly 32 KB.
__device__ __forceinline__ int fb_sl(int t) { return (0x78 >> (2 * t)) & 3; }
__device__ __forceinline__ int fb_swz(int R) { return 4 * (R & 3) + fb_sl((R >> 2) & 3); }
// element index of (row R, permuted column pc) in an image
__device__ __forceinline__ int fb_wel(int R, int pc) { return R * LDB + 8 * ((pc >> 3) ^ fb_swz(R)) + (pc & 7); }

// once per step: the hidden layers' weights as two-piece images (W1h W1l W2h W2l) and the zero fill of the
// dL/d(hz) partial-sum slots.  Thread t of T cooperating threads (any launch shape made of whole workgroups of <= 8 waves).
//   mode 0: bf16 hi / lo pieces of scale * W (rounded split, ~2^-17 relative);
//   mode 1: fp16 hi / lo pieces of s * W with s the power of two that brings max |W| into [1, 2) — fp16's narrow exponent
//           then costs nothing: hi + lo = s W to 2^-22 of an element, to 2^-25 of the largest for elements whose lo piece is
//           subnormal — and the three scales {s1, s2, s_o (of the output layer's weights)} written after the images
//           (FB_SCALE_OFF): pv_sdec_fused_bf16.hip's fp16 modes;
//   mode 2: as 1 for the images of C s W, C = 2 log2(e), with s = 1 while max |C W| lies in [2^-6, 2^10): pv_sdec_fused_w8h.hip.
#define FB_SCALE_OFF (4 * IMG_BYTES)   // byte offset of the fp32 scales {s1, s2, s_o, 0} behind the four images
struct PvFbPrep {
  const float* W1; const float* W2;   // (128, 128) fp32, nn.Linear layout
  void* img;                          // 4 * IMG_BYTES (+ 16 bytes of scales in mode 1)
  float* zero; int64_t nzero4;        // float4s to clear (0: none)
  float scale;                        // mode 0: images hold scale * W (0: unscaled) — pv_sdec_fused_w8.hip's 2 log2(e)
  int mode;                           // 0: bf16 pieces, 1: normalised fp16 pieces, 2: fp16 pieces of C s W
  const float* wo;                    // mode 1: decoder.out weights (128), for s_o
  int qswap;                          // the q-swapped column order (above)
};
// max |v| over float4s [lo4, hi4) of v, by one wave (every lane returns it); loads in independent batches of 16 per lane — a
// loop of single dependent loads paid an L2 round trip per iteration: +23 us on the launch that hosts the preparation
__device__ __forceinline__ float pv_fb_wave_absmax(const float* __restrict__ v, int lo4, int hi4, int lane) {
  float m = 0.0f;
  const f32x4* p = reinterpret_cast<const f32x4*>(v);
  for (int b0 = lo4; b0 < hi4; b0 += 64 * 16) {
    f32x4 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = b0 + 64 * k + lane;
      x[k] = i < hi4 ? p[i] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(x[k][0]), fabsf(x[k][1])), fmaxf(fabsf(x[k][2]), fabsf(x[k][3]))));
  }
  return pv_wave_max_nonneg(m);
}
// the power of two s with s * m in [1, 2) (1 for m = 0 or a non-finite m; exponent clamped to +-60)
__device__ __forceinline__ float pv_fb_norm_scale(float m) {
  if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f;
  int e = __builtin_amdgcn_frexp_expf(m);               // m = f * 2^e, f in [0.5, 1)
  e = e < -60 ? -60 : (e > 60 ? 60 : e);
  return __uint_as_float((unsigned)(127 + 1 - e) << 23);
}
__device__ __forceinline__ void fb_split_f16(float x, unsigned short& hi, unsigned short& lo) {
  const _Float16 h = (_Float16)x;
  const _Float16 l = (_Float16)(x - (float)h);
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}
// nw: the waves of the calling workgroup that take part (ALL of its live waves: mode 1 has a workgroup barrier), wv: this one
__device__ __forceinline__ void pv_fb_prep(const PvFbPrep& p, int64_t t, int64_t T, int nw, int wv) {
  __bf16* img = reinterpret_cast<__bf16*>(p.img);
  float s1 = p.scale != 0.0f ? p.scale : 1.0f, s2 = s1;
  if (p.mode >= 1) {
    // every WORKGROUP finds the maxima for itself, its waves a share of the two matrices each (128 KB of L2-resident reads per
    // workgroup, two rounds of loads per wave: the hosting launch must not wait for this), combined through LDS
    __shared__ float fbmax[2][8];
    const int lane = (int)(t & 63);
    const int per = (128 * 32 + nw - 1) / nw, lo4 = wv * per, hi4 = lo4 + per < 128 * 32 ? lo4 + per : 128 * 32;
    const float m1 = pv_fb_wave_absmax(p.W1, lo4, hi4, lane), m2 = pv_fb_wave_absmax(p.W2, lo4, hi4, lane);
    if (lane == 0) { fbmax[0][wv] = m1; fbmax[1][wv] = m2; }
    __syncthreads();
    float a1 = 0.0f, a2 = 0.0f;
    for (int w = 0; w < nw; ++w) { a1 = fmaxf(a1, fbmax[0][w]); a2 = fmaxf(a2, fbmax[1][w]); }
    if (p.mode == 2) {
      // images of C s W, C = 2 log2(e), with s = 1 wherever two fp16 pieces of C W are accurate as they stand (max |C W| in
      // [2^-6, 2^10): 2^-19 of the largest element at worst) — the kernel's tanh then needs no multiply — else normalised
      const float C_ = 2.8853900817779268f;
      const float c1 = C_ * a1, c2 = C_ * a2;
      s1 = (c1 >= 0.015625f && c1 < 1024.0f) ? 1.0f : pv_fb_norm_scale(c1);
      s2 = (c2 >= 0.015625f && c2 < 1024.0f) ? 1.0f : pv_fb_norm_scale(c2);
    } else {
      s1 = pv_fb_norm_scale(a1);
      s2 = pv_fb_norm_scale(a2);
    }
    if ((t >> 6) == 0) {                                 // (wave-uniform: the first wave of the first workgroup, all of its lanes)
      const float so = pv_fb_norm_scale(pv_fb_wave_absmax(p.wo, 0, 32, lane));
      if (lane == 0) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(p.img) + FB_SCALE_OFF) = f32x4{s1, s2, so, 0.0f};
    }
  }
  const float g1 = p.mode == 2 ? 2.8853900817779268f * s1 : s1, g2 = p.mode == 2 ? 2.8853900817779268f * s2 : s2;
  for (int64_t idx = t; idx < 128 * 32; idx += T) {
    const int row = (int)(idx >> 5), c4 = (int)(idx & 31);
    const f32x4 w1 = reinterpret_cast<const f32x4*>(p.W1)[idx] * g1;
    const f32x4 w2 = reinterpret_cast<const f32x4*>(p.W2)[idx] * g2;
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    us4 h1, l1, h2, l2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned short a, b;
      if (p.mode >= 1) {
        fb_split_f16(w1[i], a, b); h1[i] = a; l1[i] = b;
        fb_split_f16(w2[i], a, b); h2[i] = a; l2[i] = b;
      } else {
        __bf16 x, y;
        fb_split(w1[i], x, y); h1[i] = __builtin_bit_cast(unsigned short, x); l1[i] = __builtin_bit_cast(unsigned short, y);
        fb_split(w2[i], x, y); h2[i] = __builtin_bit_cast(unsigned short, x); l2[i] = __builtin_bit_cast(unsigned short, y);
      }
    }
    const int e = fb_wel(row, fb_pcol(4 * c4)) ^ (p.qswap ? 4 * ((c4 >> 1) & 1) : 0);
    *reinterpret_cast<us4*>(img + e) = h1;
    *reinterpret_cast<us4*>(img + W_IMG + e) = l1;
    *reinterpret_cast<us4*>(img + 2 * W_IMG + e) = h2;
    *reinterpret_cast<us4*>(img + 3 * W_IMG + e) = l2;
  }
  for (int64_t idx = t; idx < p.nzero4; idx += T) reinterpret_cast<f32x4*>(p.zero)[idx] = f32x4{0, 0, 0, 0};
}
