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
// ... and every row R of an image has its sixteen 16-byte chunks XOR-swizzled by fb_swz(R) = 4*(R&3) + SL[(R>>2)&3],
// SL = {0,2,3,1}: (a) the forward's ds_read_b128 (lane (r,q): row 16*ob + r, chunk 4m + q; serviced in the 16-lane
// groups {0-3,12-15,20-27}, ...) touches 16 distinct chunks per group = all 64 banks; (b) the dgrad's transposing
// 8-byte reads (32 lanes: 8 rows x 4 chunks, one half of each chunk) are 2-way, the minimum while all lanes want
// the same half.  Unpadded rows make an image exactly 32 KB.
__device__ __forceinline__ int fb_sl(int t) { return (0x78 >> (2 * t)) & 3; }
__device__ __forceinline__ int fb_swz(int R) { return 4 * (R & 3) + fb_sl((R >> 2) & 3); }
// element index of (row R, permuted column pc) in an image
__device__ __forceinline__ int fb_wel(int R, int pc) { return R * LDB + 8 * ((pc >> 3) ^ fb_swz(R)) + (pc & 7); }

// once per step: the hidden layers' weights as bf16 hi / lo images (W1h W1l W2h W2l) and the zero fill of the
// dL/d(hz) partial-sum slots.  Thread t of T cooperating threads (any launch shape).
struct PvFbPrep {
  const float* W1; const float* W2;   // (128, 128) fp32, nn.Linear layout
  void* img;                          // 4 * IMG_BYTES
  float* zero; int64_t nzero4;        // float4s to clear (0: none)
  float scale;                        // images hold scale * W (0: unscaled) — pv_sdec_fused_w8.hip's 2 log2(e)
};
__device__ __forceinline__ void pv_fb_prep(const PvFbPrep& p, int64_t t, int64_t T) {
  __bf16* img = reinterpret_cast<__bf16*>(p.img);
  for (int64_t idx = t; idx < 128 * 32; idx += T) {
    const int row = (int)(idx >> 5), c4 = (int)(idx & 31);
    const float sc = p.scale != 0.0f ? p.scale : 1.0f;
    const f32x4 w1 = reinterpret_cast<const f32x4*>(p.W1)[idx] * sc;
    const f32x4 w2 = reinterpret_cast<const f32x4*>(p.W2)[idx] * sc;
    bf16x4 h1, l1, h2, l2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __bf16 a, b;
      fb_split(w1[i], a, b); h1[i] = a; l1[i] = b;
      fb_split(w2[i], a, b); h2[i] = a; l2[i] = b;
    }
    const int e = fb_wel(row, fb_pcol(4 * c4));
    *reinterpret_cast<bf16x4*>(img + e) = h1;
    *reinterpret_cast<bf16x4*>(img + W_IMG + e) = l1;
    *reinterpret_cast<bf16x4*>(img + 2 * W_IMG + e) = h2;
    *reinterpret_cast<bf16x4*>(img + 3 * W_IMG + e) = l2;
  }
  for (int64_t idx = t; idx < p.nzero4; idx += T) reinterpret_cast<f32x4*>(p.zero)[idx] = f32x4{0, 0, 0, 0};
}
