// pv_sdec_fused_w8x3.hip — the fused persistent spatial-decoder forward+backward kernel in SPLIT precision (bf16 hi + lo,
// three matrix instructions per product: the fp32-class default path, plan.fused = 2) re-cut for TWO waves per SIMD.
// (round 3; the 4-wave form is pv_sdec_fused_bf16.hip <X3 = true>, the plain-bf16 8-wave form pv_sdec_fused_w8.hip)
//
// What is taken from the plain-bf16 8-wave kernel: 512-thread workgroups, one per CU, a wave carries one 16-row unit and
// owns one 32 x 64 block of dW1 / dW2 in accumulators; every small contraction on the matrix cores (coordinate layer,
// its row-local input gradient, the wave-local column sums); weight images, biases and the coordinate layer's operands
// pre-scaled by c = 2 log2(e) so tanh is exp2 -> +1 -> rcp -> fma; elementwise phases written as stages; a branch-free
// tile body (waves without a unit run on a valid unit with dL/dlogit = 0).
// What split precision changes:
//   * W1, W2 as hi and lo images are 128 KB of the CU's 160 KB; a 128-row tile's (dpre, h) staging in hi and lo is
//     147 KB.  The staging area therefore takes HALF a tile (64 rows, 72 KB) and lies OVER a weight image that is dead
//     at that moment, as in the 4-wave kernel: layer 2's exchange over W1 (+ an 8 KB gap), layer 1's over (gap +) W2;
//     the overwritten images come back by LDS-DMA under the dgrad of layer 2 / the next tile's forward of layer 1.
//     A layer's weight gradient is: [all waves: wave-local column sums in the dead region] barrier [waves 0-3 stage]
//     barrier [all consume 64 rows] barrier [waves 4-7 stage] barrier [all consume] barrier.
//   * the layer-0 activation h0 is needed again only at the very end of the tile (1 - h0^2 and the layer-1 weight
//     gradient); it does not fit next to the 104 accumulator registers through the dgrad of layer 2, so the wave parks
//     its hi/lo pieces (8 KB) in a private global slot that it rewrites every tile — L2-resident (2 MB per XCD), two
//     8-instruction transfers per tile — and takes them back under the dgrad of layer 1.
//   * saved activations are hi + lo, so 1 - h^2 is formed at fp32 precision; the column sums contract hi and lo pieces
//     (t_hi x [1 | x0h | x1h | dlh | dll | x0l | x1l] + t_lo x [1 | x0h | x1h | dlh]).
// NW (waves per workgroup) is a template parameter (round 3, second step):
//   NW = 8: the form described above.  Its forward-only launches (decode, evaluate) are the fastest (136 registers, two
//           waves per SIMD hide every latency); its training launches lose to the register limit (256: spills, parking,
//           dgrad in halves) and to 12 barriers per tile — 271 us at batch 256 against the old 4-wave kernel's 188.
//   NW = 4: the same source with one wave per SIMD and the 512-register budget: a tile is 64 rows = ONE staged exchange
//           per layer (5 barriers per tile), a wave owns a 64 x 64 block of dW1 / dW2, h0 stays in registers, nothing
//           spills.  This is the training kernel of the fp32-class path: every VALU->MFMA move of the plain-bf16 8-wave
//           kernel (coordinate layer, column sums, row-local input gradient, multiply-free tanh) without its register bill.
// Layout, row -> lane mapping, weight images, staging swizzles and the per-workgroup gradient record are those of
// pv_sdec_fused_bf16.hip / pv_sdec_fused_w8.hip (pv_fb_layout.h), so the rest of the step is unchanged.
#include "pv_sdec_fused.h"
#include "pv_fb_layout.h"
#include <stdlib.h>

typedef short short4_ __attribute__((ext_vector_type(4)));
typedef short short8_ __attribute__((ext_vector_type(8)));
typedef unsigned uint4_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4_ lds_short4;

#define X3_WAVES 8                         // the LDS map is laid out for the 8-wave form (NW = 4 leaves slots unused)
#define X3_ROWS (X3_WAVES * FD_UNIT)       // 128 rows per tile
#define X3_HALF 64                         // rows per staged half
#define X3_THREADS (64 * X3_WAVES)
#define LDS2 144                           // staging rows: 72 dwords -> conflict-free 4x16 transposing reads
#define X3_ARR (X3_HALF * LDS2)            // elements of one staging array (64 rows)
#define X3_ARR_BYTES (2 * X3_ARR)          // 18,432
#define X3_GAP_BYTES (4 * X3_ARR_BYTES - 2 * IMG_BYTES)   // 8,192
#define XO_R1 0                            // W1 hi | W1 lo
#define XO_GAP (2 * IMG_BYTES)
#define XO_R2 (XO_GAP + X3_GAP_BYTES)      // W2 hi | W2 lo
#define XO_ST2 XO_R1                       // layer 2's exchange: over W1 + gap
#define XO_ST1 XO_GAP                      // layer 1's exchange: over gap + W2
#define XO_VEC (XO_R2 + 2 * IMG_BYTES)     // fp32: wo[128], c*b1[128], c*b2[128]
#define XO_ATAB (XO_VEC + 3 * FD_H * 4)    // coordinate layer, A operands: 8 blocks x 64 lanes x bf16x4
#define XO_TTAB (XO_ATAB + 8 * 64 * 8)     // row-local dgrad, A operands: 4 k-blocks x 64 lanes x bf16x8
#define XO_INFO (XO_TTAB + 4 * 64 * 16)    // per row of the tile: x0[128], x1[128], dlda[128]
#define XO_RED (XO_INFO + 3 * X3_ROWS * 4)
#define XO_CHZ (XO_RED + 256)              // next tile's per-unit inputs by LDS-DMA: hz[b] (128 floats) per wave
#define XO_CTP (XO_CHZ + X3_WAVES * FD_H * 4)
#define XO_CGR (XO_CTP + X3_WAVES * 256)
#define X3_LDS_BYTES (XO_CGR + X3_WAVES * 256)
static_assert(X3_GAP_BYTES >= 0, "staging overlays");
static_assert(X3_LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(2 * IMG_BYTES % (X3_WAVES * 1024) == 0, "image load: whole 1 KB LDS-DMA pieces per wave");
static_assert(2 * X3_ROWS * LDS2 * 2 <= 4 * X3_ARR_BYTES, "column-sum scratch (hi and lo, 16 rows per wave) fits a region");
// per wave and tile: the parked h0 (8 blocks x 64 lanes x 16 B)
#define X3_PARK_BYTES_PER_WAVE (8 * 64 * 16)

#define X3_C 2.8853900817779268f           // 2 log2(e): tanh(x) = 1 - 2 / (exp2(C x) + 1)
#define X3_RC (1.0f / X3_C)
#define X3_RC2 (X3_RC * X3_RC)
#define LOG_SQRT_2PI 0.91893853320467274178f
#define BERN_EPS 1.1920928955078125e-07f
#define X3_FENCE() __builtin_amdgcn_sched_barrier(0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 x3_mfma16(const bf16x4& a, const bf16x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4_, a), __builtin_bit_cast(short4_, b), c, 0, 0, 0);
}
__device__ __forceinline__ float x3_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float x3_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float x3_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ bf16x8 x3_cat(const bf16x4& a, const bf16x4& b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ int x3_opaque0() { int z = 0; asm volatile("" : "+v"(z)); return z; }
__device__ __forceinline__ bf16x4 x3_tr(const __bf16* p) {
  const short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)p);
  return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ bf16x4 x3_zero4() { const short4_ z = {0, 0, 0, 0}; return __builtin_bit_cast(bf16x4, z); }
// LDS-DMA (see pv_sdec_fused_bf16.hip: not in hipcc's waitcnt bookkeeping; drain explicitly)
__device__ __forceinline__ void x3_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void x3_glds4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void x3_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void x3_wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }
__device__ __forceinline__ float x3_sum_q(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// one layer's hi + lo images (64 KB) from their global copy: 8 one-KB pieces per wave
template <int NW>
__device__ __forceinline__ void x3_reload(const char* __restrict__ gimg, unsigned lds_dst, int wave, int lane) {
  constexpr int PIECES = 2 * IMG_BYTES / (NW * 1024);
#pragma unroll
  for (int c = 0; c < PIECES; ++c) {
    const int off = (wave * PIECES + c) * 1024;
    x3_glds16(gimg + off + lane * 16, lds_dst + off);
  }
}

// lane offsets (elements) of the weight reads (pv_sdec_fused_w8.hip: W8Addr)
struct X3Addr { int fb, fx[4], db, dx[4]; };
__device__ __forceinline__ X3Addr x3_addr(int r, int q) {
  X3Addr a;
  a.fb = r * LDB + 8 * (q ^ fb_sl(r >> 2));
  a.db = (4 * q + (r >> 2)) * LDB + 8 * ((r & 3) ^ fb_sl(q));
#pragma unroll
  for (int m = 0; m < 4; ++m) { a.fx[m] = 32 * (m ^ (r & 3)); a.dx[m] = 32 * (m ^ (r >> 2)); }
  return a;
}

#ifndef X3_PF4
#define X3_PF4 2                    // operand prefetch distance (groups of 6 MFMAs) of the layer loops with 4 waves: one wave
#endif                              // per SIMD has nobody else to cover the LDS latency, and 512 registers to spend on it
#ifndef X3_FOLD
#define X3_FOLD 1                   // 4 waves: a layer's elementwise epilogue issued inside the consuming layer's k-loop
#endif
#ifndef X3_WAD_LOCAL
#define X3_WAD_LOCAL 1              // 1: weight-read lane offsets recomputed per layer call instead of living in 10 registers
#endif
// forward layer of the wave's unit: out = bias + W in, all pre-scaled by C; three products per block.
// PF: operand prefetch distance in groups (0: loaded where they are used)
// epi(g): independent elementwise work issued with group g's matrix instructions (see x3_tanh_split_chunk)
struct X3NoEpi { __device__ __forceinline__ void operator()(int) const {} };
template <int PF, class Epi = X3NoEpi>
__device__ __forceinline__ void x3_layer_fwd(const __bf16* __restrict__ Wh, const float* __restrict__ bs,
                                             const bf16x4 (&ih)[8], const bf16x4 (&il)[8], f32x4 (&out)[8],
                                             const X3Addr& ad, int q, Epi epi = Epi()) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) out[ob] = *reinterpret_cast<const f32x4*>(bs + 16 * ob + 4 * q);
  const __bf16* ah = Wh + ad.fb;
  const __bf16* al = ah + W_IMG;
  const int (&xm)[4] = ad.fx;
  bf16x8 wh[PF + 1][2], wl[PF + 1][2];
  auto load = [&](int g, bf16x8 (&h)[2], bf16x8 (&l)[2]) {
    const int m = g >> 2, op = (g & 3) * 2;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int off = 16 * (op + o) * LDB + xm[m];
      h[o] = *reinterpret_cast<const bf16x8*>(ah + off);
      l[o] = *reinterpret_cast<const bf16x8*>(al + off);
    }
  };
#pragma unroll
  for (int g = 0; g < PF; ++g) load(g, wh[g], wl[g]);
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int m = g >> 2, op = (g & 3) * 2;
    const int cur = g % (PF + 1);
    if (g + PF < 16) load(g + PF, wh[(g + PF) % (PF + 1)], wl[(g + PF) % (PF + 1)]);
    X3_FENCE();
    const bf16x8 bh = x3_cat(ih[2 * m], ih[2 * m + 1]), bl = x3_cat(il[2 * m], il[2 * m + 1]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[op + o] = MFMA32(wh[cur][o], bh, out[op + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[op + o] = MFMA32(wh[cur][o], bl, out[op + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[op + o] = MFMA32(wl[cur][o], bh, out[op + o]);
    epi(g);
    X3_FENCE();
  }
}

// dgrad of the wave's unit, output blocks 4*hf .. 4*hf+3: out[k] = sum_j (C W)[j][k] dp[j]; A = W^T via the transposing
// LDS read.  In two halves because the epilogue of a half (1 - h^2, split) ends the life of that half of the saved
// activation before the other half's accumulators exist: 16 registers less at the kernel's tightest point.
// PF: operand prefetch distance in groups (0 with 8 waves: the 256-register limit)
template <int PF>
__device__ __forceinline__ void x3_layer_dgrad_half(const __bf16* __restrict__ Wh, const bf16x4 (&ih)[8],
                                                    const bf16x4 (&il)[8], f32x4 (&out)[4], int hf, const X3Addr& ad) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) out[kb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const __bf16* ah = Wh + ad.db;
  const __bf16* al = ah + W_IMG;
  const int (&xk)[4] = ad.dx;
  bf16x8 wh[PF + 1][2], wl[PF + 1][2];
  auto load = [&](int g, bf16x8 (&h)[2], bf16x8 (&l)[2]) {
    const int m = g >> 1, kp = 4 * hf + (g & 1) * 2;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int off = 32 * m * LDB + xk[(kp + o) >> 1] + 4 * ((kp + o) & 1);
      h[o] = x3_cat(x3_tr(ah + off), x3_tr(ah + off + 16 * LDB));
      l[o] = x3_cat(x3_tr(al + off), x3_tr(al + off + 16 * LDB));
    }
  };
#pragma unroll
  for (int g = 0; g < PF; ++g) load(g, wh[g], wl[g]);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int m = g >> 1, kq = (g & 1) * 2;
    const int cur = g % (PF + 1);
    if (g + PF < 8) load(g + PF, wh[(g + PF) % (PF + 1)], wl[(g + PF) % (PF + 1)]);
    X3_FENCE();
    const bf16x8 bh = x3_cat(ih[2 * m], ih[2 * m + 1]), bl = x3_cat(il[2 * m], il[2 * m + 1]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[kq + o] = MFMA32(wh[cur][o], bh, out[kq + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[kq + o] = MFMA32(wh[cur][o], bl, out[kq + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[kq + o] = MFMA32(wl[cur][o], bh, out[kq + o]);
    X3_FENCE();
  }
}

// dgrad of the wave's unit, all 8 output blocks (the 4-wave form: registers are no concern), with an epilogue functor
template <int PF, class Epi = X3NoEpi>
__device__ __forceinline__ void x3_layer_dgrad(const __bf16* __restrict__ Wh, const bf16x4 (&ih)[8], const bf16x4 (&il)[8],
                                               f32x4 (&out)[8], const X3Addr& ad, Epi epi = Epi()) {
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) out[kb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const __bf16* ah = Wh + ad.db;
  const __bf16* al = ah + W_IMG;
  const int (&xk)[4] = ad.dx;
  bf16x8 wh[PF + 1][2], wl[PF + 1][2];
  auto load = [&](int g, bf16x8 (&h)[2], bf16x8 (&l)[2]) {
    const int m = g >> 2, kp = (g & 3) * 2;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int off = 32 * m * LDB + xk[(kp + o) >> 1] + 4 * ((kp + o) & 1);
      h[o] = x3_cat(x3_tr(ah + off), x3_tr(ah + off + 16 * LDB));
      l[o] = x3_cat(x3_tr(al + off), x3_tr(al + off + 16 * LDB));
    }
  };
#pragma unroll
  for (int g = 0; g < PF; ++g) load(g, wh[g], wl[g]);
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int m = g >> 2, kp = (g & 3) * 2;
    const int cur = g % (PF + 1);
    if (g + PF < 16) load(g + PF, wh[(g + PF) % (PF + 1)], wl[(g + PF) % (PF + 1)]);
    X3_FENCE();
    const bf16x8 bh = x3_cat(ih[2 * m], ih[2 * m + 1]), bl = x3_cat(il[2 * m], il[2 * m + 1]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[kp + o] = MFMA32(wh[cur][o], bh, out[kp + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[kp + o] = MFMA32(wh[cur][o], bl, out[kp + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[kp + o] = MFMA32(wl[cur][o], bh, out[kp + o]);
    epi(g);
    X3_FENCE();
  }
}

// Elementwise epilogues in CHUNKS of two values (chunk c of 16: block c >> 1, elements 2 (c & 1), 2 (c & 1) + 1), to be
// issued between the matrix instructions of the CONSUMING layer's k-loop (4-wave form): one wave per SIMD pays 6.5 cycles
// per VALU instruction in a pure elementwise phase but ~4 beside MFMAs, which themselves hide ~8 cycles of it each
// (scripts/ubench/mfma_overlap2.hip).  k-block m of the consumer needs blocks 2m, 2m+1 = chunks 4m .. 4m+3.
__device__ __forceinline__ void x3_split2(float t0, float t1, bf16x4& h, bf16x4& l, int e0) {
  const __bf16 h0 = (__bf16)t0, h1 = (__bf16)t1;
  h[e0] = h0; h[e0 + 1] = h1;
  l[e0] = (__bf16)(t0 - (float)h0); l[e0 + 1] = (__bf16)(t1 - (float)h1);
}
// tanh given C*x, then (hi, lo)
__device__ __forceinline__ void x3_tanh_split_chunk(const f32x4 (&v)[8], bf16x4 (&h)[8], bf16x4 (&l)[8], int c) {
  const int jb = c >> 1, e0 = 2 * (c & 1);
  float t0 = __builtin_amdgcn_exp2f(v[jb][e0]), t1 = __builtin_amdgcn_exp2f(v[jb][e0 + 1]);
  t0 = __builtin_amdgcn_rcpf(t0 + 1.0f); t1 = __builtin_amdgcn_rcpf(t1 + 1.0f);
  t0 = 1.0f - 2.0f * t0; t1 = 1.0f - 2.0f * t1;
  x3_split2(t0, t1, h[jb], l[jb], e0);
}
// d * (1 - (hh + hl)^2), then (hi, lo)
__device__ __forceinline__ void x3_dtanh_split_chunk(const f32x4 (&d)[8], const bf16x4 (&hh)[8], const bf16x4 (&hl)[8],
                                                     bf16x4 (&oh)[8], bf16x4 (&ol)[8], int c) {
  const int jb = c >> 1, e0 = 2 * (c & 1);
  const float a0 = (float)hh[jb][e0] + (float)hl[jb][e0], a1 = (float)hh[jb][e0 + 1] + (float)hl[jb][e0 + 1];
  const float t0 = d[jb][e0] * (1.0f - a0 * a0), t1 = d[jb][e0 + 1] * (1.0f - a1 * a1);
  x3_split2(t0, t1, oh[jb], ol[jb], e0);
}

// tanh of x given C*x, in place, written as stages (pv_sdec_fused_w8.hip)
__device__ __forceinline__ void x3_tanh8(f32x4 (&v)[8]) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[jb][i] = __builtin_amdgcn_exp2f(v[jb][i]);
  X3_FENCE();
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) v[jb] = v[jb] + 1.0f;
  X3_FENCE();
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[jb][i] = __builtin_amdgcn_rcpf(v[jb][i]);
  X3_FENCE();
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) v[jb] = 1.0f - 2.0f * v[jb];
  X3_FENCE();
}
__device__ __forceinline__ f32x4 x3_f32_of(const bf16x4& h) {
  typedef unsigned uint2_ __attribute__((ext_vector_type(2)));
  const uint2_ u = __builtin_bit_cast(uint2_, h);
  f32x4 f;
  f[0] = __builtin_bit_cast(float, u[0] << 16);
  f[1] = __builtin_bit_cast(float, u[0] & 0xffff0000u);
  f[2] = __builtin_bit_cast(float, u[1] << 16);
  f[3] = __builtin_bit_cast(float, u[1] & 0xffff0000u);
  return f;
}
// v -> (hi, lo) per C/D block, staged: all hi converts, then the residuals, then the lo converts
__device__ __forceinline__ void x3_split8(const f32x4 (&v)[8], bf16x4 (&h)[8], bf16x4 (&l)[8]) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 t[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int i = 0; i < 4; ++i) h[4 * half + jb][i] = (__bf16)v[4 * half + jb][i];
    X3_FENCE();
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) t[jb] = v[4 * half + jb] - x3_f32_of(h[4 * half + jb]);
    X3_FENCE();
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int i = 0; i < 4; ++i) l[4 * half + jb][i] = (__bf16)t[jb][i];
    X3_FENCE();
  }
}
// four blocks: d *= 1 - h^2 with h = hi + lo, then d -> (hi, lo)
__device__ __forceinline__ void x3_dtanh_split4(f32x4 (&d)[4], const bf16x4* hh, const bf16x4* hl, bf16x4* oh, bf16x4* ol) {
  f32x4 t[4];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) t[jb] = x3_f32_of(hh[jb]) + x3_f32_of(hl[jb]);
  X3_FENCE();
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) t[jb] = 1.0f - t[jb] * t[jb];
  X3_FENCE();
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) d[jb] = d[jb] * t[jb];
  X3_FENCE();
#pragma unroll
  for (int jb = 0; jb < 4; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) oh[jb][i] = (__bf16)d[jb][i];
  X3_FENCE();
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) t[jb] = d[jb] - x3_f32_of(oh[jb]);
  X3_FENCE();
#pragma unroll
  for (int jb = 0; jb < 4; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) ol[jb][i] = (__bf16)t[jb][i];
  X3_FENCE();
}
// d *= 1 - h^2 with h = hi + lo, four blocks at a time
__device__ __forceinline__ void x3_mul_dtanh(f32x4 (&d)[8], const bf16x4 (&hh)[8], const bf16x4 (&hl)[8]) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 t[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) t[jb] = x3_f32_of(hh[4 * half + jb]) + x3_f32_of(hl[4 * half + jb]);
    X3_FENCE();
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) t[jb] = 1.0f - t[jb] * t[jb];
    X3_FENCE();
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) d[4 * half + jb] = d[4 * half + jb] * t[jb];
    X3_FENCE();
  }
}

// 16 rows (row0 + r) of a staged tensor, row-major [rows][LDS2]; inside every 16-column block the four 8-byte pieces
// are XOR-swizzled by (row>>2)&3 (pv_sdec_fused_bf16.hip: fb_stage_store)
__device__ __forceinline__ void x3_stage_store(__bf16* __restrict__ sh, const bf16x4 (&h)[8], int row, int q) {
  row |= x3_opaque0();
  const int e = row * LDS2 + 4 * (q ^ ((row >> 2) & 3));
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) *reinterpret_cast<bf16x4*>(sh + e + 16 * jb) = h[jb];
}
// lane offset of the transposing read of staged rows R0 + 4q .. 4q+3 (R0 a multiple of 16), columns 16*blk ..
__device__ __forceinline__ int x3_stage_toff(int r, int q) { return (4 * q + (r >> 2)) * LDS2 + 4 * ((r & 3) ^ q); }

// wgrad over one staged exchange (64 rows = up to 2 k-steps).  Region: [dpre hi | dpre lo | h hi | h lo], 64 x LDS2 each.
// Wave (jp = wave >> 1, kh = wave & 1) owns the (16 SB) x 64 block dW[16 SB jp .. ][64kh .. +63] (SB = 16 / NW: 32 rows
// with 8 waves, 64 with 4) and the bias sums of its 16 NB rows 16 SB jp + 16 NB kh .. (NB = 8 / NW; an MFMA against ones):
//   dW[j][k] += sum_rows dpre[row][j] h[row][k]  (dh hh + dh hl + dl hh);   db[j] += sum_rows dpre[row][j]
// The wave holds its SB row blocks ROTATED by NB kh — accW[s] is row block (s + NB kh) mod SB — so that its bias blocks are
// always operands 0 .. NB-1: a wave-uniform choice between REGISTER operands (kh ? a[NB + t] : a[t]) made hipcc put the
// operand arrays in scratch memory and index them there (14 scratch transfers per k-step: 8.7 k cycles per exchange).
template <int NW>
__device__ __forceinline__ void x3_wgrad_consume(const __bf16* st, f32x4 (&accW)[16 / NW][4], f32x4 (&accB)[8 / NW], int wave,
                                                 int r, int q, int ksteps) {
  constexpr int SB = 16 / NW, NB = 8 / NW;
  const __bf16* sah = st;
  const __bf16* sal = st + X3_ARR;
  const __bf16* sbh = st + 2 * X3_ARR;
  const __bf16* sbl = st + 3 * X3_ARR;
  const int toff = x3_stage_toff(r | x3_opaque0(), q);
  const int jp = wave >> 1, kh = wave & 1;
  const short one = 0x3f80;                           // bf16 1.0
  const short8_ ones_s = {one, one, one, one, one, one, one, one};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
  for (int ks = 0; ks < ksteps; ++ks) {
    const int koff = toff + 32 * ks * LDS2;
    bf16x8 ah[SB], al[SB];
#pragma unroll
    for (int s_ = 0; s_ < SB; ++s_) {
      const int off = koff + 16 * SB * jp + 16 * ((s_ + NB * kh) & (SB - 1));
      ah[s_] = x3_cat(x3_tr(sah + off), x3_tr(sah + off + 16 * LDS2));
      al[s_] = x3_cat(x3_tr(sal + off), x3_tr(sal + off + 16 * LDS2));
    }
    bf16x8 bh[2], bl[2];
    auto load = [&](int o, bf16x8& h, bf16x8& l) {
      const int off = koff + 64 * kh + 16 * o;
      h = x3_cat(x3_tr(sbh + off), x3_tr(sbh + off + 16 * LDS2));
      l = x3_cat(x3_tr(sbl + off), x3_tr(sbl + off + 16 * LDS2));
    };
    load(0, bh[0], bl[0]);
    X3_FENCE();
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      accB[t] = MFMA32(ah[t], ones, accB[t]);
      accB[t] = MFMA32(al[t], ones, accB[t]);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o + 1 < 4) load(o + 1, bh[(o + 1) & 1], bl[(o + 1) & 1]);
      X3_FENCE();
#pragma unroll
      for (int s_ = 0; s_ < SB; ++s_) accW[s_][o] = MFMA32(ah[s_], bh[o & 1], accW[s_][o]);
#pragma unroll
      for (int s_ = 0; s_ < SB; ++s_) accW[s_][o] = MFMA32(ah[s_], bl[o & 1], accW[s_][o]);
#pragma unroll
      for (int s_ = 0; s_ < SB; ++s_) accW[s_][o] = MFMA32(al[s_], bh[o & 1], accW[s_][o]);
      X3_FENCE();
    }
  }
}

// wave-local column sums on the matrix cores: accS[jb][.] (D[j][n]) += sum over the unit's 16 rows of
//   t_hi[row][j] * b1[row][n] + t_lo[row][j] * b2[row][n].
// The wave stages its tile (hi at `sc`, lo 128 rows further) in its own 16 rows of a region nobody else touches at this
// point, reads it back transposed as the A operand and contracts against the B operands (lane (n, kq): B[4kq..4kq+3][n]).
__device__ __forceinline__ void x3_colsum_mfma(__bf16* __restrict__ sc, const bf16x4 (&th)[8], const bf16x4 (&tl)[8],
                                               const bf16x4& b1, const bf16x4& b2, f32x4 (&accS)[8], int wave, int r, int q) {
  __bf16* sl = sc + X3_ROWS * LDS2;
  x3_stage_store(sc, th, 16 * wave + r, q);
  x3_stage_store(sl, tl, 16 * wave + r, q);
  x3_wait_lgkm0();
  const int toff = (16 * wave) * LDS2 + x3_stage_toff(r | x3_opaque0(), q);
  bf16x4 a[8];
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) a[jb] = x3_tr(sc + toff + 16 * jb);
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) accS[jb] = x3_mfma16(a[jb], b1, accS[jb]);
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) a[jb] = x3_tr(sl + toff + 16 * jb);
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) accS[jb] = x3_mfma16(a[jb], b2, accS[jb]);
}

#ifdef X3_TRACE
__device__ long long x3_trace[512];
#define X3_STAMP(k)                                                                              \
  do {                                                                                           \
    if (g == 0 && lane == 0 && (wave == 0 || wave == NW - 1) && tile_no < 8)                          \
      x3_trace[(wave ? 256 : 0) + tile_no * 32 + (k)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
extern "C" int pv_debug_read_trace_w8x3(long long* out, int n) {
  if (n > 512) n = 512;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(x3_trace), n * sizeof(long long));
}
#else
#define X3_STAMP(k) do { } while (0)
#endif

// LIK: the likelihood is a compile-time choice
template <bool GRADS, int LIK, int NW>
__global__ __launch_bounds__(64 * NW) void pv_sdec_w8x3_kernel(PvFused f) {
  constexpr int SB = 16 / NW, NB = 8 / NW;           // 16-row blocks of dW / of the bias sums a wave owns
  constexpr int NH = NW / 4;                         // staged exchanges per layer and tile (64 rows each)
  extern __shared__ __attribute__((aligned(16))) char smb[];
  const int tid = threadIdx.x, lane0 = tid & 63, lane = lane0, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x, G = gridDim.x;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smb);
  const __bf16* W1h = reinterpret_cast<const __bf16*>(smb + XO_R1);
  const __bf16* W2h = reinterpret_cast<const __bf16*>(smb + XO_R2);
  __bf16* st2 = reinterpret_cast<__bf16*>(smb + XO_ST2);
  __bf16* st1 = reinterpret_cast<__bf16*>(smb + XO_ST1);
  float* vec = reinterpret_cast<float*>(smb + XO_VEC);
  float* info = reinterpret_cast<float*>(smb + XO_INFO);
  float* red = reinterpret_cast<float*>(smb + XO_RED);
  const char* gimg = reinterpret_cast<const char*>(f.wimg);

  // ---- prologue: weight images by LDS-DMA (prepared set: W1h W1l W2h W2l), vectors and tables ----
  x3_reload<NW>(gimg, lds0 + XO_R1, wave, lane);
  x3_reload<NW>(gimg + 2 * IMG_BYTES, lds0 + XO_R2, wave, lane);
  if (tid < FD_H) {
    vec[tid] = f.wo[tid];
    vec[FD_H + tid] = X3_C * f.b1[tid];
    vec[2 * FD_H + tid] = X3_C * f.b2[tid];
  }
  {
    // coordinate layer A operands (v_mfma_f32_16x16x16_bf16: lane (m, kq) holds A[m][4kq .. 4kq+3]), k slots:
    //   kq 0: [wh0 wh0 wl0 0] x [xh0 xl0 xh0 0]   kq 1: the same for coordinate 1   kq 2: [bch bcl 0 0] x [1 1 0 0]
    for (int jb = tid >> 6; jb < 8; jb += NW) {
      const int m = lane & 15, kq = lane >> 4, j = 16 * jb + m;
      float v = 0.0f;
      if (kq == 0) v = X3_C * f.Wc[j * f.cd];
      else if (kq == 1) v = f.cd == 2 ? X3_C * f.Wc[j * 2 + 1] : 0.0f;
      else if (kq == 2) v = X3_C * f.bc[j];
      __bf16 hi, lo;
      fb_split(v, hi, lo);
      bf16x4 a = x3_zero4();
      if (kq < 2) { a[0] = hi; a[1] = hi; a[2] = lo; }
      else if (kq == 2) { a[0] = hi; a[1] = lo; }
      reinterpret_cast<bf16x4*>(smb + XO_ATAB)[64 * jb + lane] = a;
    }
  }
  if (tid < 256) {
    // row-local dgrad A operands (16x16x32: lane (m, kq) holds A[m][k], k = the 8 logical columns a lane feeds as B:
    // 32mm + 4kq + e (e < 4), 32mm + 16 + 4kq + (e - 4)); rows m: 0 Wc0 hi, 1 Wc0 lo, 2 Wc1 hi, 3 Wc1 lo, others 0
    const int mm = tid >> 6, m = lane & 15, kq = lane >> 4;
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = 32 * mm + 4 * kq + (e < 4 ? e : 16 + e - 4);
      float w = 0.0f;
      if (m < 2) w = f.Wc[j * f.cd];
      else if (m < 4 && f.cd == 2) w = f.Wc[j * 2 + 1];
      __bf16 hi, lo;
      fb_split(w, hi, lo);
      a[e] = m >= 4 ? (__bf16)0.0f : ((m & 1) ? lo : hi);
    }
    reinterpret_cast<bf16x8*>(smb + XO_TTAB)[tid] = a;
  }
  x3_wait_vm0();
  __syncthreads();
  const float bo = f.bo[0];

  // persistent accumulators: the wave's 32 x 64 blocks of dW1 (x C) and dW2, the bias sums, and the wave-local column
  // sums D[j][n]: n = 0 dL/d(hz) | 1, 5 dWc0 (hi, lo) | 2, 6 dWc1 | 3, 4 d(wo)   (n 0,1,2,5,6 carry C^2)
  f32x4 accW1[SB][4], accW2[SB][4], accS[8], accB1[NB], accB2[NB];
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) accS[kb] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int s_ = 0; s_ < SB; ++s_)
#pragma unroll
    for (int o = 0; o < 4; ++o) { accW1[s_][o] = f32x4{0, 0, 0, 0}; accW2[s_][o] = f32x4{0, 0, 0, 0}; }
#pragma unroll
  for (int t = 0; t < NB; ++t) { accB1[t] = f32x4{0, 0, 0, 0}; accB2[t] = f32x4{0, 0, 0, 0}; }
  float dbo = 0.0f;
  int cur_b = -1;                                    // the sample whose dL/d(hz) this WAVE is accumulating
  const int upb = f.N / FD_UNIT;
  float* rec = f.part + (int64_t)g * FD_REC;
  // the wave's parking slot for h0 (hi/lo), rewritten every tile
  // (uniform base + 32-bit lane offset everywhere a lane touches global memory: one address register instead of a 64-bit
  //  pair per access point — per-lane pointers hoisted out of the tile loop were what the first build spilled)
  char* const park_base = reinterpret_cast<char*>(f.park) + ((int64_t)g * NW + wave) * X3_PARK_BYTES_PER_WAVE;   // (NW = 8 only)
  auto park_at = [&](int jb) -> uint4_* {
    return reinterpret_cast<uint4_*>(park_base + (unsigned)((lane0 | x3_opaque0()) * 16 + 1024 * jb));
  };

  auto flush_hz = [&](int b) {
    const int64_t ub = (int64_t)b * upb;
    const int gfirst = (int)(((ub + 1) * G + f.units - 1) / f.units) - 1;
    float* dst = f.part_hz + ((int64_t)b * f.kmax + (g - gfirst) * NW + wave) * FD_H + 4 * q;
    if (r == 0) {
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) *reinterpret_cast<f32x4*>(dst + 16 * jb) = accS[jb] * X3_RC2;
    }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
      for (int i = 0; i < 4; ++i) accS[jb][i] = r == 0 ? 0.0f : accS[jb][i];
  };

  const int u_lo = (int)((int64_t)g * f.units / G), u_hi = (int)((int64_t)(g + 1) * f.units / G);
  const int xun = (int)f.x_units;
  struct Pos { int unit, b, loc, xu; };                 // unit = b * upb + loc ; xu = unit mod x_units (x_units > 0)
  auto pos_of = [&](int unit_) {
    Pos p_;
    p_.unit = unit_; p_.b = unit_ / upb; p_.loc = unit_ - p_.b * upb; p_.xu = xun > 0 ? unit_ % xun : unit_;
    return p_;
  };
  auto advance = [&](Pos& p_, int by) {
    p_.unit += by; p_.loc += by; p_.xu += by;
    while (p_.loc >= upb) { p_.loc -= upb; ++p_.b; }
    if (xun > 0) { while (p_.xu >= xun) p_.xu -= xun; }
  };
  const Pos pos_lo = pos_of(u_lo);                      // what an out-of-range wave fetches instead (valid, unused)
  Pos pos_cur = pos_of(u_lo + wave < u_hi ? u_lo + wave : u_lo);
  Pos pos_nx = pos_cur;
  float sw_next = 1.0f;
  auto x_of = [&](const Pos& p_) -> float {
    if (f.sw) sw_next = f.sw[p_.b];
    return f.x[(int64_t)p_.xu * FD_UNIT + r];
  };
  float xv_next = x_of(pos_cur);
  float* chz = reinterpret_cast<float*>(smb + XO_CHZ) + wave * FD_H;
  float* ctp = reinterpret_cast<float*>(smb + XO_CTP) + wave * 64;
  float* cgr = reinterpret_cast<float*>(smb + XO_CGR) + wave * 64;
  auto fetch_unit_inputs = [&](const Pos& p_) {
    const int n0 = p_.loc * FD_UNIT;
    x3_glds4(f.hz + (int64_t)p_.b * FD_H + lane, lds0 + XO_CHZ + wave * (FD_H * 4));
    x3_glds4(f.hz + (int64_t)p_.b * FD_H + 64 + lane, lds0 + XO_CHZ + wave * (FD_H * 4) + 256);
    x3_glds4(f.tp + (int64_t)p_.b * 8 + (lane & 7), lds0 + XO_CTP + wave * 256);
    x3_glds4(f.grid + (int64_t)n0 * f.cd + (lane & (16 * f.cd - 1)), lds0 + XO_CGR + wave * 256);
  };
  x3_wait_vm0();                                        // (the observation load above: nothing compiler-visible in flight)
  fetch_unit_inputs(pos_cur);
  const X3Addr wad0 = x3_addr(r, q);
  (void)wad0;
  int tile_no = -1;
  for (int ut = u_lo; ut < u_hi; ut += NW) {
    ++tile_no;
    (void)tile_no;
    asm volatile("; X3_TILE_BEGIN");
    X3_STAMP(0);
    const int nact = (u_hi - ut) < NW ? (u_hi - ut) : NW;
    if (ut + NW + wave < u_hi) advance(pos_nx, NW);
    else pos_nx = pos_lo;
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const int lane = lane0 | opq, r = lane & 15, q = lane >> 4;
#if X3_WAD_LOCAL
#define wad x3_addr((lane0 | x3_opaque0()) & 15, (lane0 | x3_opaque0()) >> 4)
#else
    const X3Addr& wad = wad0;
#endif
    const float* wos = vec;
    const float* b1s = vec + FD_H;
    const float* b2s = vec + 2 * FD_H;
    const bool act = wave < nact;
    const int unit = act ? pos_cur.unit : ut;
    const int bu = pos_cur.b;
    const unsigned rowb = ((unsigned)unit * FD_UNIT + (unsigned)r) * 4u;      // byte offset of the lane's row (rows < 2^30)
    float x0, x1, u0c, u1c, sc;
    x3_wait_vm0();                        // this wave's LDS-DMA of the tile's inputs (issued a tile ago)
    {
      const float* t = ctp;
      const float* gr = cgr;
      if (f.cd == 2) {
        const float gx = gr[2 * r], gy = gr[2 * r + 1];
        u0c = gx * t[0] - gy * t[1];
        u1c = gx * t[1] + gy * t[0];
        sc = t[2];
        x0 = u0c * sc + t[3];
        x1 = u1c * sc + t[4];
      } else {
        u0c = gr[r]; u1c = 0.0f; sc = 1.0f;
        x0 = u0c + t[3]; x1 = 0.0f;
      }
    }
    const float xv = xv_next, swv = sw_next;
    float* inf_x0 = info + 16 * wave;
    float* inf_x1 = info + X3_ROWS + 16 * wave;
    float* inf_dl = info + 2 * X3_ROWS + 16 * wave;

    f32x4 tC[8], tD[8];                        // (tD: the 4-wave form keeps a layer's pre-activations while the next layer consumes them)
    bf16x4 h0h[8], h0l[8], h1h[8], h1l[8], pAh[8], pAl[8];
    float dlda = 0.0f;
    {
      // ---- coordinate layer on the matrix cores: C h0pre = (C Wc) x' + C bc + C hz[b] ----
      bf16x4 bx = x3_zero4();
      {
        const float v = q == 0 ? x0 : x1;
        __bf16 vh, vl;
        fb_split(v, vh, vl);
        const __bf16 one = (__bf16)1.0f;
        if (q < 2) { bx[0] = vh; bx[1] = vl; bx[2] = vh; }
        else if (q == 2) { bx[0] = one; bx[1] = one; }
      }
      const bf16x4* atab = reinterpret_cast<const bf16x4*>(smb + XO_ATAB) + lane;
      const float* hzb = chz;
      bf16x4 aop[8];
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        tC[jb] = *reinterpret_cast<const f32x4*>(hzb + 16 * jb + 4 * q);
        aop[jb] = atab[64 * jb];
      }
      if (GRADS && q == 0) { inf_x0[r] = x0; inf_x1[r] = x1; }
      X3_FENCE();
      if (f.hz_scale == 0.0f) {                  // hz arrives unscaled only when the generic encoder path produced it
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) tC[jb] = tC[jb] * X3_C;
      }
      X3_FENCE();
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) tC[jb] = x3_mfma16(aop[jb], bx, tC[jb]);
      X3_FENCE();
      if (NW == 4 && X3_FOLD) {
#pragma unroll
        for (int c = 0; c < 4; ++c) x3_tanh_split_chunk(tC, h0h, h0l, c);      // (k-block 0; the rest rides in layer 1's loop)
      } else {
        x3_tanh8(tC);
        x3_split8(tC, h0h, h0l);
      }
    }
    asm volatile("; X3_P1_coord_done");
    X3_STAMP(1);
    fetch_unit_inputs(pos_nx);                 // the slots were consumed by the coordinate layer above
    if (GRADS && tile_no > 0) x3_reload<NW>(gimg + 2 * IMG_BYTES, lds0 + XO_R2, wave, lane);   // W2 was the previous tile's staging area
    {
      if (NW == 4 && X3_FOLD) {
        x3_layer_fwd<X3_PF4>(W1h, b1s, h0h, h0l, tD, wad, q, [&](int g_) { if (g_ < 12) x3_tanh_split_chunk(tC, h0h, h0l, 4 + g_); });
#pragma unroll
        for (int c = 0; c < 4; ++c) x3_tanh_split_chunk(tD, h1h, h1l, c);
      } else {
        x3_layer_fwd<(NW == 4 ? X3_PF4 : 1)>(W1h, b1s, h0h, h0l, tC, wad, q);
      }
      if (GRADS && NW == 8) {
        // park h0 (needed again at the end of the tile only)
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
          typedef unsigned uint2_ __attribute__((ext_vector_type(2)));
          const uint2_ a = __builtin_bit_cast(uint2_, h0h[jb]), b = __builtin_bit_cast(uint2_, h0l[jb]);
          *park_at(jb) = uint4_{a[0], a[1], b[0], b[1]};
        }
      }
      if (!(NW == 4 && X3_FOLD)) {
        x3_tanh8(tC);
        x3_split8(tC, h1h, h1l);                                   // feeds layer 2 and its wgrad
      }
    }
    asm volatile("; X3_P2_l1_done");
    X3_STAMP(2);
    if (GRADS) {
      x3_wait_vm0();
      __syncthreads();      // barrier 0: W2 landed everywhere; every wave is past its reads of W1 (region ST2 is free)
    }
    X3_STAMP(3);
    {
      if (NW == 4 && X3_FOLD)
        x3_layer_fwd<X3_PF4>(W2h, b2s, h1h, h1l, tC, wad, q, [&](int g_) { if (g_ < 12) x3_tanh_split_chunk(tD, h1h, h1l, 4 + g_); });
      else
        x3_layer_fwd<(NW == 4 ? X3_PF4 : 1)>(W2h, b2s, h1h, h1l, tC, wad, q);
      // ---- h2, output layer + likelihood (fp32); tC <- g = wo (1 - h2^2), pA <- split(h2) ----
      x3_tanh8(tC);                                                // tC = h2
      if (GRADS) x3_split8(tC, pAh, pAl);
      f32x4 part4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wos + 16 * jb + 4 * q);
        part4 = part4 + tC[jb] * wv;
        if (GRADS) {
          const f32x4 t2 = tC[jb] * tC[jb];
          tC[jb] = wv - wv * t2;
        }
      }
      const float a = x3_sum_q((part4[0] + part4[1]) + (part4[2] + part4[3])) + bo;
      float ll, locv;
      if (LIK == PV_LIK_BERNOULLI) {
        const float pr = x3_rcp(1.0f + x3_exp(-a));
        const float pc = fminf(fmaxf(pr, BERN_EPS), 1.0f - BERN_EPS);
        // -BCEWithLogits(lg, x) with lg = logit(pc) (torch: probs_to_logits, then binary_cross_entropy_with_logits), written with
        // the identities 1 + exp(-|lg|) = 1 / max(pc, 1 - pc) and sigmoid(lg) = pc: the two logarithms lg is made of serve the
        // softplus term too, and the row's dependent chain is exp -> rcp -> 2 log instead of seven transcendentals (round 5)
        const float lpc = x3_log(pc), l1pc = x3_log(1.0f - pc);
        const float lg = lpc - l1pc;
        ll = -(fmaxf(lg, 0.0f) - lg * xv - fmaxf(lpc, l1pc));
        const float mask = (pr >= BERN_EPS && pr <= 1.0f - BERN_EPS) ? 1.0f : 0.0f;
        dlda = (pc - xv) * mask;
        locv = pr;
      } else if (LIK == PV_LIK_CBERNOULLI) {
        pv_cbern(a, xv, ll, dlda, locv);
      } else {
        const float pr = f.sigmoid_out ? x3_rcp(1.0f + x3_exp(-a)) : a;
        const float d = xv - pr;
        ll = -(d * d) / (2.0f * f.sig * f.sig) - x3_log(f.sig) - LOG_SQRT_2PI;
        dlda = -d / (f.sig * f.sig) * (f.sigmoid_out ? pr * (1.0f - pr) : 1.0f);
        locv = pr;
      }
      dlda *= act ? swv : 0.0f;
      if (q == 0) {
        if (act) {
          if (f.llrow) *reinterpret_cast<float*>(reinterpret_cast<char*>(f.llrow) + rowb) = ll;
          if (f.loc) *reinterpret_cast<float*>(reinterpret_cast<char*>(f.loc) + rowb) = locv;
        }
        if (GRADS) { dbo += dlda; inf_dl[r] = dlda; }
      }
      xv_next = x_of(pos_nx);
    }
    pos_cur = pos_nx;                          // (unit, bu, row of THIS tile were taken above)
    asm volatile("; X3_P3_fwd_done");
    X3_STAMP(4);
    if (!GRADS) continue;
    // k-steps (32 rows) of staged exchange h: units 4h .. 4h+3 of the tile
    auto ks_of = [&](int h) { const int n = nact - 4 * h; return n <= 0 ? 0 : (n >= 3 ? 2 : 1); };
    {
      // ---- d(wo) += sum_rows dlda h2 : wave-local MFMAs through the wave's own scratch rows of region ST2;
      // B = dlda of rows 4q..4q+3 in columns 3 (hi) and 4 (lo) for h2_hi, column 3 (hi) for h2_lo
      x3_wait_lgkm0();
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(inf_dl + 4 * q);
      bf16x4 bw1 = x3_zero4(), bw2 = x3_zero4();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __bf16 hi, lo;
        fb_split(d4[i], hi, lo);
        bw1[i] = r == 3 ? hi : (r == 4 ? lo : (__bf16)0.0f);
        bw2[i] = r == 3 ? hi : (__bf16)0.0f;
      }
      x3_colsum_mfma(st2, pAh, pAl, bw1, bw2, accS, wave, r, q);
      // dpre2 = dlda * wo (1 - h2^2)
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) tC[jb] = tC[jb] * dlda;
      x3_split8(tC, pAh, pAl);                                    // feeds the wgrad and the dgrad of layer 2
      x3_wait_lgkm0();                                            // (own scratch reads done)
    }
    asm volatile("; X3_P4_dwo_done");
    X3_STAMP(5);
    // ---- wgrad of layer 2: (dpre2, h1), 64 rows per staged exchange ----
    // (8 waves: waves 4-7's column-sum scratch rows are rows waves 0-3 stage into; 4 waves: a wave's scratch rows are its
    //  own staging rows in arrays 0 and 2, so its own lgkmcnt wait above is all the ordering it needs)
    if (NW == 8) __syncthreads();                                 // every wave's column sums are out of the region
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (ks_of(h) == 0) break;                                   // (workgroup-uniform)
      if (h > 0) __syncthreads();
      if ((wave >> 2) == h) {
        x3_stage_store(st2, pAh, 16 * (wave & 3) + r, q);
        x3_stage_store(st2 + X3_ARR, pAl, 16 * (wave & 3) + r, q);
        x3_stage_store(st2 + 2 * X3_ARR, h1h, 16 * (wave & 3) + r, q);
        x3_stage_store(st2 + 3 * X3_ARR, h1l, 16 * (wave & 3) + r, q);
      }
      __syncthreads();
      X3_STAMP(6 + h);
      x3_wgrad_consume<NW>(st2, accW2, accB2, wave, r, q, ks_of(h));
    }
    __syncthreads();                                              // region ST2 consumed everywhere
    asm volatile("; X3_P6_cons2");
    X3_STAMP(8);
    x3_wait_vm0();                                                // (nothing compiler-visible may be in flight)
    x3_reload<NW>(gimg, lds0 + XO_R1, wave, lane);                // W1 comes back under the dgrad of layer 2
    bf16x4 p0h[8], p0l[8], d1h[8], d1l[8];
    if (NW == 4 && X3_FOLD) {
      // 4 waves: dgrad of layer 2 in one piece; its epilogue C dpre1 = (C dL/dh1)(1 - h1^2) -> (hi, lo) rides in the k-loop
      // of the dgrad of layer 1, which consumes it block by block
      x3_layer_dgrad<X3_PF4>(W2h, pAh, pAl, tC, wad);              // tC = C dL/dh1
#pragma unroll
      for (int c = 0; c < 4; ++c) x3_dtanh_split_chunk(tC, h1h, h1l, d1h, d1l, c);
      asm volatile("; X3_P7_dgrad2");
      X3_STAMP(9);
      x3_wait_vm0();
      __syncthreads();      // W1 landed everywhere; every wave is past its reads of W2 (region ST1 is free)
      X3_STAMP(10);
      x3_layer_dgrad<X3_PF4>(W1h, d1h, d1l, tD, wad, [&](int g_) { if (g_ < 12) x3_dtanh_split_chunk(tC, h1h, h1l, d1h, d1l, 4 + g_); });
      f32x4 t4[4];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) t4[jb] = tD[4 * hf + jb];
        x3_dtanh_split4(t4, h0h + 4 * hf, h0l + 4 * hf, p0h + 4 * hf, p0l + 4 * hf);   // C^2 dpre0
      }
    } else {
    {
      // C dpre1 = (C dL/dh1) (1 - h1^2), split: feeds the dgrad and the wgrad of layer 1
      f32x4 t4[4];
      x3_layer_dgrad_half<(NW == 4 ? X3_PF4 : 0)>(W2h, pAh, pAl, t4, 0, wad);
      x3_dtanh_split4(t4, h1h, h1l, d1h, d1l);
      x3_layer_dgrad_half<(NW == 4 ? X3_PF4 : 0)>(W2h, pAh, pAl, t4, 1, wad);
      x3_dtanh_split4(t4, h1h + 4, h1l + 4, d1h + 4, d1l + 4);
    }
    asm volatile("; X3_P7_dgrad2");
    X3_STAMP(9);
    x3_wait_vm0();
    __syncthreads();        // W1 landed everywhere; every wave is past its reads of W2 (region ST1 is free)
    X3_STAMP(10);
    {
      // NW = 8: h0 comes back from its parking slot half by half, each under a half of the dgrad of layer 1
      typedef unsigned uint2_ __attribute__((ext_vector_type(2)));
      uint4_ pk[4];
      f32x4 t4[4];
      if (NW == 8) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) pk[jb] = *park_at(jb);
      }
      x3_layer_dgrad_half<(NW == 4 ? X3_PF4 : 0)>(W1h, d1h, d1l, t4, 0, wad);              // C^2 dL/dh0
      if (NW == 8) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          h0h[jb] = __builtin_bit_cast(bf16x4, uint2_{pk[jb][0], pk[jb][1]});
          h0l[jb] = __builtin_bit_cast(bf16x4, uint2_{pk[jb][2], pk[jb][3]});
        }
      }
      x3_dtanh_split4(t4, h0h, h0l, p0h, p0l);                     // C^2 dpre0
      if (NW == 8) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) pk[jb] = *park_at(4 + jb);
      }
      x3_layer_dgrad_half<(NW == 4 ? X3_PF4 : 0)>(W1h, d1h, d1l, t4, 1, wad);
      if (NW == 8) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          h0h[4 + jb] = __builtin_bit_cast(bf16x4, uint2_{pk[jb][0], pk[jb][1]});
          h0l[4 + jb] = __builtin_bit_cast(bf16x4, uint2_{pk[jb][2], pk[jb][3]});
        }
      }
      x3_dtanh_split4(t4, h0h + 4, h0l + 4, p0h + 4, p0l + 4);
    }
    }
    asm volatile("; X3_P8_dgrad1");
    X3_STAMP(11);
    {
      // ---- coordinate layer backward, row-local part on the matrix cores: D[m][row] = sum_j T[m][j] dpre0[row][j] ----
      f32x4 dd = {0.0f, 0.0f, 0.0f, 0.0f};
      const bf16x8* ttab = reinterpret_cast<const bf16x8*>(smb + XO_TTAB) + lane;
#pragma unroll
      for (int mm = 0; mm < 4; ++mm) {
        const bf16x8 tt = ttab[64 * mm];
        dd = MFMA32(tt, x3_cat(p0h[2 * mm], p0h[2 * mm + 1]), dd);
        dd = MFMA32(tt, x3_cat(p0l[2 * mm], p0l[2 * mm + 1]), dd);
      }
      if (q == 0 && act) {
        const float d0 = (dd[0] + dd[1]) * X3_RC2, d1 = (dd[2] + dd[3]) * X3_RC2;
        char* const tp0 = reinterpret_cast<char*>(f.rowtp);
        *reinterpret_cast<float*>(tp0 + rowb) = sc * (d1 * u0c - d0 * u1c);
        *reinterpret_cast<float*>(tp0 + f.M * 4 + rowb) = d0 * u0c + d1 * u1c;
        *reinterpret_cast<float*>(tp0 + f.M * 8 + rowb) = d0;
        *reinterpret_cast<float*>(tp0 + f.M * 12 + rowb) = d1;
      }
      if (act && bu != cur_b) {
        if (cur_b >= 0) flush_hz(cur_b);
        cur_b = bu;
      }
      // ---- dL/d(hz[b]) = sum_rows dpre0, dWc_k = sum_rows dpre0 x'_k : wave-local MFMAs, own scratch rows of region ST1;
      // B columns: 0 ones | 1, 5 x0 (hi, lo) | 2, 6 x1 (hi, lo) for dpre0_hi; 0 | 1 | 2 (hi) for dpre0_lo
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(inf_x0 + 4 * q);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(inf_x1 + 4 * q);
      bf16x4 bc1 = x3_zero4(), bc2 = x3_zero4();
      const bool use1 = r == 2 || r == 6, lo_col = r >= 5;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __bf16 hi, lo;
        fb_split(use1 ? a1[i] : a0[i], hi, lo);
        __bf16 v = lo_col ? lo : hi;
        if (r == 0) v = (__bf16)1.0f;
        if (r == 3 || r == 4 || r > 6) v = (__bf16)0.0f;
        bc1[i] = v;
        bc2[i] = r <= 2 ? v : (__bf16)0.0f;
      }
      x3_colsum_mfma(st1, p0h, p0l, bc1, bc2, accS, wave, r, q);
      x3_wait_lgkm0();                                              // (own reads done before the rows are re-staged)
    }
    asm volatile("; X3_P12_rowlocal");
    X3_STAMP(12);
    // ---- wgrad of layer 1: (C dpre1, h0), 64 rows per staged exchange ----
    if (NW == 8) __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (ks_of(h) == 0) break;
      if (h > 0) __syncthreads();
      if ((wave >> 2) == h) {
        x3_stage_store(st1, d1h, 16 * (wave & 3) + r, q);
        x3_stage_store(st1 + X3_ARR, d1l, 16 * (wave & 3) + r, q);
        x3_stage_store(st1 + 2 * X3_ARR, h0h, 16 * (wave & 3) + r, q);
        x3_stage_store(st1 + 3 * X3_ARR, h0l, 16 * (wave & 3) + r, q);
      }
      __syncthreads();
      X3_STAMP(13 + h);
      x3_wgrad_consume<NW>(st1, accW1, accB1, wave, r, q, ks_of(h));
    }
    __syncthreads();                      // region ST1 consumed everywhere (the next tile's W2 reload lands there)
    asm volatile("; X3_P13_cons1");
    X3_STAMP(15);
    asm volatile("; X3_TILE_END");
  }
  if (!GRADS) return;

  if (cur_b >= 0) flush_hz(cur_b);
  // ---- the workgroup's gradient record (pv_sdec_fused.h: FD_REC) ----
  {
    const int jp = wave >> 1, kh = wave & 1;
#pragma unroll
    for (int s_ = 0; s_ < SB; ++s_)
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // C/D layout: lane (col = r, q), reg i -> dW[16 SB jp + 16 blk + 4q + i][64kh + 16o + r], blk = (s + NB kh) mod SB
          const int e = (16 * SB * jp + 16 * ((s_ + NB * kh) & (SB - 1)) + 4 * q + i) * FD_H + 64 * kh + 16 * o + r;
          rec[e] = accW1[s_][o][i] * X3_RC;
          rec[FD_H * FD_H + e] = accW2[s_][o][i];
        }
    if (r == 0) {
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int j0 = 16 * SB * jp + 16 * (NB * kh + t);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rec[2 * FD_H * FD_H + j0 + 4 * q + i] = accB1[t][i] * X3_RC;
          rec[2 * FD_H * FD_H + FD_H + j0 + 4 * q + i] = accB2[t][i];
        }
      }
    }
  }
  // per-wave column sums -> LDS (W1's images are dead: every wave is past the last tile's barriers) -> summed over the
  // waves in ascending order
  __syncthreads();
  {
    float* scr = reinterpret_cast<float*>(smb + XO_R1);            // [wave][n][128] floats = 64 KB
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
      *reinterpret_cast<f32x4*>(scr + ((wave * 16 + r) * FD_H) + 16 * jb + 4 * q) = accS[jb];
  }
  const float tb = pv_wave_sum(dbo);
  if (lane == 0) red[wave] = tb;
  __syncthreads();
  if (tid < FD_H) {
    const float* scr = reinterpret_cast<const float*>(smb + XO_R1);
    float vo = 0.0f, v0 = 0.0f, v1 = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* s_ = scr + (w * 16) * FD_H + tid;
      v0 += s_[1 * FD_H] + s_[5 * FD_H];
      v1 += s_[2 * FD_H] + s_[6 * FD_H];
      vo += s_[3 * FD_H] + s_[4 * FD_H];
    }
    rec[2 * FD_H * FD_H + 2 * FD_H + tid] = v0 * X3_RC2;
    rec[2 * FD_H * FD_H + 3 * FD_H + tid] = v1 * X3_RC2;
    rec[2 * FD_H * FD_H + 4 * FD_H + tid] = vo;
  }
  if (tid == 0) {
    float v = 0.0f;
    for (int w = 0; w < NW; ++w) v += red[w];
    rec[2 * FD_H * FD_H + 5 * FD_H] = v;
  }
}

int64_t pv_sdec_fused_w8x3_park_bytes(int grid) { return (int64_t)grid * 8 * X3_PARK_BYTES_PER_WAVE; }

// waves: 8 or 4 (the template's NW)
int pv_sdec_fused_w8x3_launch(const PvFused& f_in, int grid, bool grads, hipStream_t s, int waves) {
  PvFused f = f_in;
  f.ablate = 0;
  if (waves != 4 && waves != 8) return PV_EINVAL;
  if (grads && waves == 8 && !f.park) return PV_EINVAL;
  const size_t lds = X3_LDS_BYTES;
  const void* fn = nullptr;
#define X3_PICK(G, L) fn = waves == 8 ? reinterpret_cast<const void*>(&pv_sdec_w8x3_kernel<G, L, 8>) \
                                      : reinterpret_cast<const void*>(&pv_sdec_w8x3_kernel<G, L, 4>)
  if (grads) {
    // the TRAINING forms of this source measured no faster than pv_sdec_fused_bf16.hip's: they exist in
    // the experiments build only; the shipped library instantiates the forward-only forms (decode, evaluate)
#ifdef PV_EXPERIMENTS
    if (f.lik == PV_LIK_BERNOULLI) X3_PICK(true, PV_LIK_BERNOULLI);
    else if (f.lik == PV_LIK_GAUSSIAN) X3_PICK(true, PV_LIK_GAUSSIAN);
    else X3_PICK(true, PV_LIK_CBERNOULLI);
#else
    return PV_EINVAL;
#endif
  } else {
    if (f.lik == PV_LIK_BERNOULLI) X3_PICK(false, PV_LIK_BERNOULLI);
    else if (f.lik == PV_LIK_GAUSSIAN) X3_PICK(false, PV_LIK_GAUSSIAN);
    else X3_PICK(false, PV_LIK_CBERNOULLI);
  }
#undef X3_PICK
  PV_TRY(pv_set_dynamic_lds(fn, (int)lds));          // (per device and kernel)
  void* args[] = {&f};
  hipError_t e2 = hipLaunchKernel(fn, dim3(grid), dim3(64 * waves), args, lds, s);
  if (e2 != hipSuccess) return (int)e2;
  PV_LAUNCH_CHECK();
  return 0;
}
