// pv_gemv16.h — y = W x for ONE image by the waves of a workgroup, straight from L2-resident fp32 weights: the matrix-vector products
// of the guide (nets/fc.py:51-61 fcEncoderNet.forward for one sample) wherever a workgroup runs a whole image's guide itself — the
// decoder launch's prologue (pv_sdec_fused_w8.hip, PvEncFold) and the per-image guide launch (pv_guide_img.hip).
#pragma once
#include "pv_common.h"

// rows j0 .. j0 + 15 (clamped to nrows - 1) of y = W x for ONE wave: W row-major (nrows, K) fp32 in global memory (L2-resident:
// every workgroup reads the same matrix), K % 4 == 0, K <= 1024, x in registers (the image itself, requested at kernel entry:
// it comes from HBM, the weights from L2 — no staging, no barrier in front of the first layer).  Lane l takes the float4 columns l, l + 64, ... of every row (coalesced 1 KB per row and
// instruction, all of a pass's 34 loads independent), the 64 x 16 partial sums are transposed through LDS (P: 64 x 17 floats
// of this wave) and each lane returns y[j0 + (lane & 15)] — plain fp32 fused multiply-adds in a fixed order.
__device__ __forceinline__ f32x4 w8_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ float w8_dot4(const f32x4& w, const f32x4& x, float acc) {
  return fmaf(w[3], x[3], fmaf(w[2], x[2], fmaf(w[1], x[1], fmaf(w[0], x[0], acc))));
}
template <class HOOK>
__device__ __forceinline__ float w8_gemv16(const float* __restrict__ W, int K, int nrows, int j0, const f32x4 (&xr)[4],
                                           const f32x4& xpk, float* __restrict__ P, int lane, HOOK after_last_loads) {
  // The memory pipeline of a CU takes one 16-byte-per-lane load instruction per 16 cycles whatever it hits: the layer is bound
  // by its count of load instructions.  Column groups of 64 float4 that do not exist are skipped (wave-uniform), and a last
  // group of <= 4 columns (28 x 28: 196 = 3 * 64 + 4) is ONE instruction for all 16 rows — lane l takes row l >> 2, column
  // l & 3 of the group — instead of sixteen that serve four lanes each: 49 instructions per wave instead of 64.
  const int K4 = K >> 2;
  const int ng = (K4 + 63) >> 6;                      // column groups (<= 4)
  const int rem = K4 - 64 * (ng - 1);                 // columns of the last group
  const bool packed = ng >= 2 && rem <= 4;
  const int ngn = packed ? ng - 1 : ng;               // groups read the plain way
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  float pk = 0.0f;
#pragma unroll
  for (int pi = 0; pi < 2; ++pi) {                    // two column groups per pass
    if (2 * pi >= ngn) break;
    const int ka = 128 * pi + lane, kb = ka + 64;
    const bool hasb = 2 * pi + 1 < ngn;               // (wave-uniform)
    const bool oka = ka < K4, okb = hasb && kb < K4;
    f32x4 xa = xr[2 * pi], xb = xr[2 * pi + 1];
    f32x4 wa[16], wb[16], wp = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = j0 + i < nrows ? j0 + i : nrows - 1;
      const float* wr = W + (int64_t)row * K;
      wa[i] = w8_ld4(wr + 4 * (oka ? ka : 0));
      if (hasb) wb[i] = w8_ld4(wr + 4 * (okb ? kb : 0));
    }
    const bool last = 2 * pi + 2 >= ngn;
    if (last && packed) {
      const int row = j0 + (lane >> 2) < nrows ? j0 + (lane >> 2) : nrows - 1;
      wp = w8_ld4(W + (int64_t)row * K + 4 * (64 * (ng - 1) + ((lane & 3) < rem ? (lane & 3) : 0)));
    }
    if (last) after_last_loads();
    if (!oka) xa = f32x4{0.0f, 0.0f, 0.0f, 0.0f};    // (a column that does not exist contributes w * 0)
    if (!okb) xb = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[i] = w8_dot4(wa[i], xa, acc[i]);
      if (hasb) acc[i] = w8_dot4(wb[i], xb, acc[i]);
    }
    if (last && packed && (lane & 3) < rem) pk = w8_dot4(wp, xpk, 0.0f);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) P[lane * 17 + i] = acc[i];
  if (packed) P[lane * 17 + (lane >> 2)] += pk;       // (the packed group's product belongs to row lane >> 2)
  const int r = lane & 15, q = lane >> 4;
  float v = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) v += P[(16 * q + i) * 17 + r];
  return pv_sum_rows(v);
}
// the same for K <= 128 (the hidden layers and the head: x in LDS): TWO rows per load instruction — lanes 0-31 take row
// j0 + 2i, lanes 32-63 row j0 + 2i + 1 — so 8 loads cover the 16 rows with every lane busy
// (the weights are requested by w8_gemv16_k128_load long before x exists — behind the first layer's loads in the memory queue —
//  so that the later layers start with their operands in registers instead of an L2 round trip each)
__device__ __forceinline__ void w8_gemv16_k128_load(const float* __restrict__ W, int K, int nrows, int j0, int lane, f32x4 (&wv)[8]) {
  const int K4 = K >> 2, c = lane & 31, half = lane >> 5;
  const bool ok = c < K4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = j0 + 2 * i + half < nrows ? j0 + 2 * i + half : nrows - 1;
    wv[i] = w8_ld4(W + (int64_t)row * K + 4 * (ok ? c : 0));
  }
}
__device__ __forceinline__ float w8_gemv16_k128(const f32x4 (&wv)[8], int K, const float* __restrict__ xs, float* __restrict__ P, int lane) {
  const int K4 = K >> 2, c = lane & 31;
  const bool ok = c < K4;
  f32x4 xv = w8_ld4(xs + 4 * (ok ? c : 0));
  if (!ok) xv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < 8; ++i) P[lane * 9 + i] = w8_dot4(wv[i], xv, 0.0f);
  // lane (r, q): row j0 + r = j0 + 2 (r >> 1) + (r & 1): the 32 lanes of half r & 1, eight of them per q
  const int r = lane & 15, q = lane >> 4;
  float v = 0.0f;
#pragma unroll
  for (int t = 0; t < 8; ++t) v += P[(32 * (r & 1) + 8 * q + t) * 9 + (r >> 1)];
  return pv_sum_rows(v);
}

