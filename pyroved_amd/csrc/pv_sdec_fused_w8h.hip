// pv_sdec_fused_w8h.hip — the fused persistent spatial-decoder forward+backward kernel of the FP32-CLASS path (plan.fused = 2)
// re-cut for TWO waves per SIMD (round 4).  Arithmetic of pv_sdec_fused_bf16.hip's fp16 builds (FB_P_H221 / H231: the WEIGHTS
// are two exact fp16 pieces, activations one piece; H231 splits dL/dpre for the two dgrads), geometry of pv_sdec_fused_w8.hip
// (512-thread workgroups, one per CU, a wave carries one 16-row unit of a 128-row tile and owns one 32 x 64 block of dW1 / dW2;
// every small contraction on the matrix cores; elementwise phases as stages; a branch-free tile body).
//
// Why this form.  One wave per SIMD issues a VALU instruction every ~6.5 cycles, two waves double the SIMD's rate, and the
// 4-wave kernel's tile is ~2.1 k VALU instructions per wave next to 420 matrix instructions (profiles/r04d_pmc_summary.txt):
// it is issue bound.  Rounds 2-3 could not re-cut the three-product bf16 kernel this way (pv_sdec_fused_w8x3.hip: 245 us
// against 188 — its activations are two pieces everywhere: registers); with ONE-piece activations the 8-wave form fits:
//   * saved activations are one fp16 piece (h0h, h1h: the next layer's operand and the staged wgrad operand) plus the
//     derivative 1 - h^2 as one fp16 piece (d0, d1), formed from the fp32 value before it is dropped: a derivative formed
//     from the ROUNDED activation is off by 2 h^2 / (1 - h^2) x 2^-12 near saturation, the rounded derivative by 2^-12;
//   * weight operands are read where they are used (no second register buffer): the partner wave covers the LDS latency;
//   * W1, W2 as hi and lo images are 128 KB of the CU's 160 KB: a 128-row tile's (dpre, h) staging (72 KB) lies OVER the
//     images that are dead at that moment — layer 2's over W1 (+ an 8 KB gap), layer 1's over (gap +) W2 — and the
//     overwritten images come back by LDS-DMA under the dgrad of layer 2 / the next tile's coordinate layer + forward of
//     layer 1, as in the 4-wave kernel: six workgroup barriers per 128 rows (the 4-wave kernel: six per 64).
// Scaling (exact powers of two, pv_sdec_fused_bf16.hip): the images hold C s W (C = 2 log2 e, so that tanh needs no multiply
// when s = 1: pv_fb_layout.h mode 2 keeps s = 1 while max |C W| lies in [2^-6, 2^10)); a row's dL/dlogit = m 2^e sends m down
// the dgrad chain and 2^(e + dl_exp) into the row's staged activation; per-row results get 2^e back in fp32.
// Layout, row -> lane mapping, image swizzles, staging swizzles and the per-workgroup gradient record are those of
// pv_sdec_fused_bf16.hip / pv_sdec_fused_w8.hip (pv_fb_layout.h), so the rest of the step is unchanged.
#include "pv_sdec_fused.h"
#include "pv_fb_layout.h"
#include <stdlib.h>

typedef short short4_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4_ lds_short4;
typedef _Float16 half4_ __attribute__((ext_vector_type(4)));
typedef _Float16 half8_ __attribute__((ext_vector_type(8)));

#define H8_WAVES 8
#define H8_ROWS (H8_WAVES * FD_UNIT)       // 128
#define H8_THREADS (64 * H8_WAVES)
#define LDS2 144                           // staging rows: 72 dwords -> conflict-free 4x16 transposing reads
#define H8_ARR (H8_ROWS * LDS2)            // elements of one staging array
#define H8_ARR_BYTES (2 * H8_ARR)          // 36,864
#define H8_GAP (2 * H8_ARR_BYTES - 2 * IMG_BYTES)      // 8,192
#define HO_W1H 0
#define HO_W1L IMG_BYTES
#define HO_W2H (2 * IMG_BYTES + H8_GAP)
#define HO_W2L (3 * IMG_BYTES + H8_GAP)
#define HO_ST2 0                           // layer 2's exchange (dpre2 | h1): over W1's images + the gap
#define HO_ST1 (2 * IMG_BYTES)             // layer 1's exchange (dpre1 | h0): over the gap + W2's images
#define HO_VEC (4 * IMG_BYTES + H8_GAP)    // fp32: wo[128], C s1 b1[128], C s2 b2[128]
#define HO_ATAB (HO_VEC + 3 * FD_H * 4)    // coordinate layer, A operands: 8 blocks x 64 lanes x 4 halves
#define HO_TTAB (HO_ATAB + 8 * 64 * 8)     // row-local dgrad, A operands: 4 k-blocks x 64 lanes x 8 halves
#define HO_INFO (HO_TTAB + 4 * 64 * 16)    // per row of the tile: ph x0[128], ph x1[128], dlda[128], ph[128]
#define HO_RED (HO_INFO + 4 * H8_ROWS * 4)
#define HO_CHZ (HO_RED + 256)              // next tile's per-unit inputs by LDS-DMA: hz[b] (128 floats) per wave
#define HO_CTP (HO_CHZ + H8_WAVES * FD_H * 4)
#define HO_CGR (HO_CTP + H8_WAVES * 256)
#define H8_LDS_BYTES (HO_CGR + H8_WAVES * 256)
static_assert(H8_GAP >= 0, "staging overlays");
static_assert(HO_ST2 + 2 * H8_ARR_BYTES <= HO_W2H && HO_ST1 + 2 * H8_ARR_BYTES <= HO_VEC, "staging overlays");
static_assert(H8_LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(2 * IMG_BYTES % (H8_WAVES * 1024) == 0, "image load: whole 1 KB LDS-DMA pieces per wave");

#define H8_C 2.8853900817779268f           // 2 log2(e): tanh(x) = 1 - 2 / (exp2(C x) + 1)
#define H8_KAPPA 16.0f                     // dL/dpre2 operands are kappa * s_o * mantissa(dL/dlogit) * wo * (1 - h2^2)
#define LOG_SQRT_2PI 0.91893853320467274178f
#define BERN_EPS 1.1920928955078125e-07f
#define H8_FENCE() __builtin_amdgcn_sched_barrier(0)

// 16-bit operands travel as bf16x4 / bf16x8 bit containers (pv_fb_layout.h's types); the instructions read them as fp16
__device__ __forceinline__ f32x4 h8_mma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_, a), __builtin_bit_cast(half8_, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 h8_mma16(const bf16x4& a, const bf16x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4_, a), __builtin_bit_cast(half4_, b), c, 0, 0, 0);
}
__device__ __forceinline__ float h8_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float h8_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float h8_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ bf16x8 h8_cat(const bf16x4& a, const bf16x4& b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ int h8_opaque0() { int z = 0; asm volatile("" : "+v"(z)); return z; }
__device__ __forceinline__ bf16x4 h8_tr(const __bf16* p) {
  const short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)p);
  return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ bf16x4 h8_tr_at(unsigned lds_byte_addr) {
  const short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(size_t)lds_byte_addr);
  return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ bf16x4 h8_zero4() { const short4_ z = {0, 0, 0, 0}; return __builtin_bit_cast(bf16x4, z); }
__device__ __forceinline__ unsigned short h8_bits(_Float16 v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ void h8_put(bf16x4& v, int i, _Float16 x) { v[i] = __builtin_bit_cast(__bf16, x); }
// x -> (hi, lo) fp16 with hi + lo = x to 2^-22 (lo subnormal below |x| ~ 0.125: 2^-25 absolute)
__device__ __forceinline__ void h8_split(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}
// LDS-DMA (see pv_sdec_fused_bf16.hip: not in hipcc's waitcnt bookkeeping; drain explicitly)
__device__ __forceinline__ void h8_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void h8_glds4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void h8_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void h8_wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }
__device__ __forceinline__ float h8_sum_q(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// one layer's hi + lo images (64 KB, adjacent in LDS and in the global copy): 8 one-KB pieces per wave
__device__ __forceinline__ void h8_reload(const char* __restrict__ gimg, unsigned lds_dst, int wave, int lane) {
  constexpr int PIECES = 2 * IMG_BYTES / (H8_WAVES * 1024);
#pragma unroll
  for (int c = 0; c < PIECES; ++c) {
    const int off = (wave * PIECES + c) * 1024;
    h8_glds16(gimg + off + lane * 16, lds_dst + off);
  }
}

// lane offsets (elements) of the weight reads (pv_sdec_fused_w8.hip: W8Addr)
struct H8Addr { int fb, fx[4], db, dx[4]; };
__device__ __forceinline__ H8Addr h8_addr(int r, int q) {
  H8Addr a;
  a.fb = r * LDB + 8 * (q ^ fb_sl(r >> 2));
  a.db = (4 * q + (r >> 2)) * LDB + 8 * ((r & 3) ^ fb_sl(q));
#pragma unroll
  for (int m = 0; m < 4; ++m) { a.fx[m] = 32 * (m ^ (r & 3)); a.dx[m] = 32 * (m ^ (r >> 2)); }
  return a;
}

#ifndef H8_PF
#define H8_PF 0                            // weight-operand prefetch distance of the layer loops, in groups (registers: 16 per step)
#endif
// forward layer of the wave's unit: out = bias + W in (times C s): two products per block, hi x in and lo x in
__device__ __forceinline__ void h8_layer_fwd(const __bf16* __restrict__ Wh, const float* __restrict__ bs, const bf16x4 (&ih)[8],
                                             f32x4 (&out)[8], const H8Addr& ad, int q) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) out[ob] = *reinterpret_cast<const f32x4*>(bs + 16 * ob + 4 * q);
  const __bf16* ah = Wh + ad.fb;
  const __bf16* al = ah + W_IMG;
  const int (&xm)[4] = ad.fx;
  bf16x8 wh[H8_PF + 1][2], wl[H8_PF + 1][2];
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
  for (int g = 0; g < H8_PF; ++g) load(g, wh[g], wl[g]);
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int m = g >> 2, op = (g & 3) * 2, cur = g % (H8_PF + 1);
    if (g + H8_PF < 16) load(g + H8_PF, wh[(g + H8_PF) % (H8_PF + 1)], wl[(g + H8_PF) % (H8_PF + 1)]);
    H8_FENCE();
    const bf16x8 bh = h8_cat(ih[2 * m], ih[2 * m + 1]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[op + o] = h8_mma(wh[cur][o], bh, out[op + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[op + o] = h8_mma(wl[cur][o], bh, out[op + o]);
    H8_FENCE();
  }
}

// dgrad of the wave's unit: out[k] = sum_j (C s W)[j][k] dp[j]; A = W^T via the transposing LDS read; DS: dp arrives split
template <bool DS>
__device__ __forceinline__ void h8_layer_dgrad(const __bf16* __restrict__ Wh, const bf16x4 (&ih)[8], const bf16x4 (&il)[8],
                                               f32x4 (&out)[8], const H8Addr& ad) {
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) out[kb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const __bf16* ah = Wh + ad.db;
  const __bf16* al = ah + W_IMG;
  const int (&xk)[4] = ad.dx;
  bf16x8 wh[H8_PF + 1][2], wl[H8_PF + 1][2];
  auto load = [&](int g, bf16x8 (&h)[2], bf16x8 (&l)[2]) {
    const int m = g >> 2, kp = (g & 3) * 2;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int off = 32 * m * LDB + xk[(kp + o) >> 1] + 4 * ((kp + o) & 1);
      h[o] = h8_cat(h8_tr(ah + off), h8_tr(ah + off + 16 * LDB));
      l[o] = h8_cat(h8_tr(al + off), h8_tr(al + off + 16 * LDB));
    }
  };
#pragma unroll
  for (int g = 0; g < H8_PF; ++g) load(g, wh[g], wl[g]);
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int m = g >> 2, kp = (g & 3) * 2, cur = g % (H8_PF + 1);
    if (g + H8_PF < 16) load(g + H8_PF, wh[(g + H8_PF) % (H8_PF + 1)], wl[(g + H8_PF) % (H8_PF + 1)]);
    H8_FENCE();
    const bf16x8 bh = h8_cat(ih[2 * m], ih[2 * m + 1]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[kp + o] = h8_mma(wh[cur][o], bh, out[kp + o]);
    if (DS) {
      const bf16x8 bl = h8_cat(il[2 * m], il[2 * m + 1]);
#pragma unroll
      for (int o = 0; o < 2; ++o) out[kp + o] = h8_mma(wh[cur][o], bl, out[kp + o]);
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) out[kp + o] = h8_mma(wl[cur][o], bh, out[kp + o]);
    H8_FENCE();
  }
}

// tanh of x given C s x (MUL: s != 1, rc = 1 / s), in place, written as stages (pv_sdec_fused_w8.hip)
template <bool MUL>
__device__ __forceinline__ void h8_tanh8(f32x4 (&v)[8], float rc) {
  if (MUL) {
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) v[jb] = v[jb] * rc;
    H8_FENCE();
  }
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[jb][i] = __builtin_amdgcn_exp2f(v[jb][i]);
  H8_FENCE();
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) v[jb] = v[jb] + 1.0f;
  H8_FENCE();
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[jb][i] = __builtin_amdgcn_rcpf(v[jb][i]);
  H8_FENCE();
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) v[jb] = 1.0f - 2.0f * v[jb];
  H8_FENCE();
}
// v -> one fp16 piece per C/D block
__device__ __forceinline__ void h8_cvt8(const f32x4 (&v)[8], bf16x4 (&h)[8]) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) h[jb] = __builtin_bit_cast(bf16x4, __builtin_convertvector(v[jb], half4_));
}
// v -> (hi, lo) fp16 per C/D block
__device__ __forceinline__ void h8_split8(const f32x4 (&v)[8], bf16x4 (&h)[8], bf16x4 (&l)[8]) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    const half4_ hh = __builtin_convertvector(v[jb], half4_);
    h[jb] = __builtin_bit_cast(bf16x4, hh);
    l[jb] = __builtin_bit_cast(bf16x4, __builtin_convertvector(v[jb] - __builtin_convertvector(hh, f32x4), half4_));
  }
}
// h (fp32) -> its fp16 piece and the fp16 piece of 1 - h^2
__device__ __forceinline__ void h8_save8(const f32x4 (&v)[8], bf16x4 (&h)[8], bf16x4 (&d)[8]) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    h[jb] = __builtin_bit_cast(bf16x4, __builtin_convertvector(v[jb], half4_));
    d[jb] = __builtin_bit_cast(bf16x4, __builtin_convertvector(1.0f - v[jb] * v[jb], half4_));
  }
}
// t *= d (the saved derivative piece)
__device__ __forceinline__ void h8_mul_d(f32x4 (&t)[8], const bf16x4 (&d)[8]) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    t[jb] = t[jb] * __builtin_convertvector(__builtin_bit_cast(half4_, d[jb]), f32x4);
  }
}
// the wave's 16 rows (row = 16 * wave + r) of a staged tensor, row-major [128][LDS2]; inside every 16-column block the
// four 8-byte pieces are XOR-swizzled by (row>>2)&3 (pv_sdec_fused_bf16.hip: fb_stage_store).  SCALED: every piece times the
// row's power of two ph (exact), and ph itself into column block 8 (the rows' padding): the bias gradient's operand
template <bool SCALED>
__device__ __forceinline__ void h8_stage_store(__bf16* __restrict__ sh, const bf16x4 (&h)[8], int row, int q, half4_ ph = half4_{}) {
  row |= h8_opaque0();
  const int e = row * LDS2 + 4 * (q ^ ((row >> 2) & 3));
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    if constexpr (SCALED) *reinterpret_cast<half4_*>(sh + e + 16 * jb) = __builtin_bit_cast(half4_, h[jb]) * ph;
    else *reinterpret_cast<bf16x4*>(sh + e + 16 * jb) = h[jb];
  }
  if constexpr (SCALED) *reinterpret_cast<half4_*>(sh + e + 16 * 8) = ph;
}
__device__ __forceinline__ int h8_stage_toff(int r, int q) { return (4 * q + (r >> 2)) * LDS2 + 4 * ((r & 3) ^ q); }

// wgrad over the staged tile (pv_sdec_fused_w8.hip: w8_wgrad_consume): wave (jp = wave >> 1, kh = wave & 1) owns the 32 x 64
// block dW[32jp .. +31][64kh .. +63] and the bias sums of rows 32jp + 16kh .. +15 — contracted against the rows' own
// factor (column block 8 of the staged activations) instead of ones
__device__ __forceinline__ void h8_wgrad_consume(const __bf16* sa, const __bf16* sb, f32x4 (&accW)[2][4], f32x4& accB,
                                                 int wave, int r, int q, int ksteps) {
  const int toff = h8_stage_toff(r | h8_opaque0(), q);
  const int jp = wave >> 1, kh = wave & 1;
  unsigned la0 = (unsigned)(size_t)sa + 2u * (unsigned)(toff + 32 * jp + 16 * kh);
  unsigned la1 = (unsigned)(size_t)sa + 2u * (unsigned)(toff + 32 * jp + 16 * (1 ^ kh));
  unsigned lb = (unsigned)(size_t)sb + 2u * (unsigned)(toff + 64 * kh);
  unsigned lp = (unsigned)(size_t)sb + 2u * (unsigned)(toff + 128);
  asm volatile("" : "+v"(la0), "+v"(la1), "+v"(lb), "+v"(lp));
  constexpr unsigned ROW16 = 2u * 16 * LDS2;         // bytes of 16 staged rows
  for (int ks = 0; ks < ksteps; ++ks) {
    bf16x8 a[2], b[4];
    a[0] = h8_cat(h8_tr_at(la0), h8_tr_at(la0 + ROW16));
    a[1] = h8_cat(h8_tr_at(la1), h8_tr_at(la1 + ROW16));
    const bf16x8 bp = h8_cat(h8_tr_at(lp), h8_tr_at(lp + ROW16));
#pragma unroll
    for (int o = 0; o < 4; ++o) b[o] = h8_cat(h8_tr_at(lb + 32u * o), h8_tr_at(lb + 32u * o + ROW16));
    H8_FENCE();
    accB = h8_mma(a[0], bp, accB);
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) accW[s_][o] = h8_mma(a[s_], b[o], accW[s_][o]);
    H8_FENCE();
    la0 += 2 * ROW16; la1 += 2 * ROW16; lb += 2 * ROW16; lp += 2 * ROW16;
  }
}

// wave-local column sums on the matrix cores: accS[jb][.] (D[j][n]) += sum over the unit's 16 rows of t[row][j] * Bn[row][n].
// The wave stages its fp16 tile `t` in its own rows of `sc` (nobody else reads them at this point of the tile), reads it
// back transposed as the A operand and contracts against `bop` (lane (n, kq): B[4kq..4kq+3][n]).
__device__ __forceinline__ void h8_colsum_mfma(__bf16* __restrict__ sc, const bf16x4 (&t)[8], const bf16x4& bop,
                                               f32x4 (&accS)[8], int wave, int r, int q) {
  h8_stage_store<false>(sc, t, 16 * wave + r, q);
  h8_wait_lgkm0();
  const __bf16* base = sc + (16 * wave) * LDS2 + h8_stage_toff(r | h8_opaque0(), q);
  bf16x4 a[8];
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) a[jb] = h8_tr(base + 16 * jb);
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) accS[jb] = h8_mma16(a[jb], bop, accS[jb]);
  h8_wait_lgkm0();                                   // (own reads done before the rows are written again)
}

// LIK: the likelihood is a compile-time choice; DS: dL/dpre split in both dgrads (H231) or one piece everywhere (H221)
template <int LIK, bool DS>
__global__ __launch_bounds__(H8_THREADS) void pv_sdec_w8h_kernel(PvFused f) {
  extern __shared__ __attribute__((aligned(16))) char smb[];
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x, G = gridDim.x;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smb);
  const __bf16* W1h = reinterpret_cast<const __bf16*>(smb + HO_W1H);
  const __bf16* W2h = reinterpret_cast<const __bf16*>(smb + HO_W2H);
  __bf16* sA2 = reinterpret_cast<__bf16*>(smb + HO_ST2);
  __bf16* sB2 = sA2 + H8_ARR;
  __bf16* sA1 = reinterpret_cast<__bf16*>(smb + HO_ST1);
  __bf16* sB1 = sA1 + H8_ARR;
  float* vec = reinterpret_cast<float*>(smb + HO_VEC);
  float* info = reinterpret_cast<float*>(smb + HO_INFO);
  float* red = reinterpret_cast<float*>(smb + HO_RED);
  const char* gimg = reinterpret_cast<const char*>(f.wimg);

  // ---- prologue: weight images by LDS-DMA, scales, vectors and tables ----
  h8_reload(gimg, lds0 + HO_W1H, wave, lane0);
  h8_reload(gimg + 2 * IMG_BYTES, lds0 + HO_W2H, wave, lane0);
  const float* scg = reinterpret_cast<const float*>(gimg + FB_SCALE_OFF);
  const float s1 = __builtin_amdgcn_readfirstlane(scg[0]), s2 = __builtin_amdgcn_readfirstlane(scg[1]);
  const float kso = H8_KAPPA * __builtin_amdgcn_readfirstlane(scg[2]);
  const float rc1 = 1.0f / s1, rc2 = 1.0f / s2;
  const bool mul1 = s1 != 1.0f, mul2 = s2 != 1.0f;
  // what the backward's carried scales amount to where a gradient leaves the kernel (all powers of two but C):
  //   dW2, db2: kappa s_o 2^dl_exp ; dW1, db1: that times C s2 ; dpre0 (per row, with 2^e): kappa s_o C^2 s1 s2
  const float uw2 = __builtin_amdgcn_ldexpf(1.0f / kso, -f.dl_exp);
  const float uw1 = uw2 / (H8_C * s2);
  const float u0 = 1.0f / (kso * s1 * s2 * H8_C * H8_C);          // times 2^e per row
  const float u0p = __builtin_amdgcn_ldexpf(u0, -f.dl_exp);       // ... for sums weighted by the row's staged factor 2^(e + dl_exp)
  if (tid < FD_H) {
    vec[tid] = f.wo[tid];
    vec[FD_H + tid] = H8_C * s1 * f.b1[tid];
    vec[2 * FD_H + tid] = H8_C * s2 * f.b2[tid];
  }
  {
    // coordinate layer A operands (v_mfma_f32_16x16x16_f16: lane (m, kq) holds A[m][4kq .. 4kq+3]), k slots:
    //   kq 0: [wh0 wh0 wl0 0] x [xh0 xl0 xh0 0]   kq 1: the same for coordinate 1   kq 2: [bch bcl 0 0] x [1 1 0 0]
    const int jb = tid >> 6, m = lane0 & 15, kq = lane0 >> 4, j = 16 * jb + m;
    float v = 0.0f;
    if (kq == 0) v = H8_C * f.Wc[j * f.cd];
    else if (kq == 1) v = f.cd == 2 ? H8_C * f.Wc[j * 2 + 1] : 0.0f;
    else if (kq == 2) v = H8_C * f.bc[j];
    _Float16 hi, lo;
    h8_split(v, hi, lo);
    bf16x4 a = h8_zero4();
    if (kq < 2) { h8_put(a, 0, hi); h8_put(a, 1, hi); h8_put(a, 2, lo); }
    else if (kq == 2) { h8_put(a, 0, hi); h8_put(a, 1, lo); }
    reinterpret_cast<bf16x4*>(smb + HO_ATAB)[tid] = a;
  }
  if (tid < 256) {
    // row-local dgrad A operands (16x16x32: lane (m, kq) holds A[m][k], k = the 8 logical columns a lane feeds as B:
    // 32mm + 4kq + e (e < 4), 32mm + 16 + 4kq + (e - 4)); rows m: 0 Wc0 hi, 1 Wc0 lo, 2 Wc1 hi, 3 Wc1 lo, others 0
    const int mm = tid >> 6, m = lane0 & 15, kq = lane0 >> 4;
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = 32 * mm + 4 * kq + (e < 4 ? e : 16 + e - 4);
      float w = 0.0f;
      if (m < 2) w = f.Wc[j * f.cd];
      else if (m < 4 && f.cd == 2) w = f.Wc[j * 2 + 1];
      _Float16 hi, lo;
      h8_split(w, hi, lo);
      a[e] = __builtin_bit_cast(__bf16, m >= 4 ? (_Float16)0.0f : ((m & 1) ? lo : hi));
    }
    reinterpret_cast<bf16x8*>(smb + HO_TTAB)[tid] = a;
  }
  h8_wait_vm0();
  __syncthreads();
  const float bo = f.bo[0];
  typedef __attribute__((address_space(1))) float gfloat;         // (explicitly global: an opaque pointer would be stored through flat_*)
  unsigned long long u_llrow, u_loc, u_rowtp, u_part_hz;
  int64_t a_M;
  asm volatile("s_mov_b64 %0, %5\n\ts_mov_b64 %1, %6\n\ts_mov_b64 %2, %7\n\ts_mov_b64 %3, %8\n\ts_mov_b64 %4, %9"
               : "=&s"(u_llrow), "=&s"(u_loc), "=&s"(u_rowtp), "=&s"(u_part_hz), "=&s"(a_M)
               : "s"((unsigned long long)f.llrow), "s"((unsigned long long)f.loc), "s"((unsigned long long)f.rowtp),
                 "s"((unsigned long long)f.part_hz), "s"(f.M));
  gfloat* a_llrow = (gfloat*)u_llrow; gfloat* a_loc = (gfloat*)u_loc; gfloat* a_rowtp = (gfloat*)u_rowtp;
  gfloat* a_part_hz = (gfloat*)u_part_hz;

  // persistent accumulators: the wave's block of dW1 / dW2 (carried scales: see uw1 / uw2), the bias sums, and the wave-local
  // column sums D[j][n]: n = 0 dL/d(hz) | 1, 5 dWc0 (hi, lo) | 2, 6 dWc1 | 3, 4 d(wo)
  f32x4 accW1[2][4], accW2[2][4], accS[8], accB1 = {0, 0, 0, 0}, accB2 = {0, 0, 0, 0};
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    accW1[kb >> 2][kb & 3] = f32x4{0, 0, 0, 0}; accW2[kb >> 2][kb & 3] = f32x4{0, 0, 0, 0}; accS[kb] = f32x4{0, 0, 0, 0};
  }
  float dbo = 0.0f;
  int cur_b = -1;                                    // the sample whose dL/d(hz) this WAVE is accumulating
  const int upb = f.N / FD_UNIT;
  float* rec = f.part + (int64_t)g * FD_REC;

  auto flush_hz = [&](int b, int r_, int q_) {
    // the wave's rows of sample b end: publish its partial dL/d(hz[b]) (column 0 of accS: lanes r == 0) in its own slot
    const int64_t ub = (int64_t)b * upb;
    const int gfirst = (int)(((ub + 1) * G + f.units - 1) / f.units) - 1;
    gfloat* dst = a_part_hz + ((int64_t)b * f.kmax + (g - gfirst) * H8_WAVES + wave) * FD_H + 4 * q_;
    if (r_ == 0) {
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) *(__attribute__((address_space(1))) f32x4*)(dst + 16 * jb) = accS[jb] * u0p;
    }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
      for (int i = 0; i < 4; ++i) accS[jb][i] = r_ == 0 ? 0.0f : accS[jb][i];
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
    return f.x[(int64_t)p_.xu * FD_UNIT + (lane0 & 15)];
  };
  float xv_next = x_of(pos_cur);
  float* chz = reinterpret_cast<float*>(smb + HO_CHZ) + wave * FD_H;
  float* ctp = reinterpret_cast<float*>(smb + HO_CTP) + wave * 64;
  float* cgr = reinterpret_cast<float*>(smb + HO_CGR) + wave * 64;
  auto fetch_unit_inputs = [&](const Pos& p_) {
    const int n0 = p_.loc * FD_UNIT;
    h8_glds4(f.hz + (int64_t)p_.b * FD_H + lane0, lds0 + HO_CHZ + wave * (FD_H * 4));
    h8_glds4(f.hz + (int64_t)p_.b * FD_H + 64 + lane0, lds0 + HO_CHZ + wave * (FD_H * 4) + 256);
    h8_glds4(f.tp + (int64_t)p_.b * 8 + (lane0 & 7), lds0 + HO_CTP + wave * 256);
    h8_glds4(f.grid + (int64_t)n0 * f.cd + (lane0 & (16 * f.cd - 1)), lds0 + HO_CGR + wave * 256);
  };
  fetch_unit_inputs(pos_cur);
  const H8Addr wad = h8_addr(lane0 & 15, lane0 >> 4);
  const float hzs = f.hz_scale == 0.0f ? H8_C : 1.0f;             // (hz arrives as C hz when the compact encoder produced it)
  int tile_no = -1;
  for (int ut = u_lo; ut < u_hi; ut += H8_WAVES) {
    ++tile_no;
    const int nact = (u_hi - ut) < H8_WAVES ? (u_hi - ut) : H8_WAVES;
    if (ut + H8_WAVES + wave < u_hi) advance(pos_nx, H8_WAVES);
    else pos_nx = pos_lo;
    int opq = 0;
    asm volatile("" : "+v"(opq));       // a zero the compiler cannot see through: lane-dependent addresses are recomputed per tile
    const int lane = lane0 | opq, r = lane & 15, q = lane >> 4;
    const float* wos = vec;
    const float* b1s = vec + FD_H;
    const float* b2s = vec + 2 * FD_H;
    const bool act = wave < nact;
    const int unit = act ? pos_cur.unit : ut;
    const int bu = pos_cur.b;
    const int64_t row = (int64_t)unit * FD_UNIT + r;
    float x0, x1, u0c, u1c, sc;
    h8_wait_vm0();                        // this wave's LDS-DMA of the tile's inputs (issued a tile ago)
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
    float* inf_x0 = info + 16 * wave;                  // (ph x0, ph x1, dlda, ph of the wave's rows)
    float* inf_x1 = info + H8_ROWS + 16 * wave;
    float* inf_dl = info + 2 * H8_ROWS + 16 * wave;
    float* inf_ph = info + 3 * H8_ROWS + 16 * wave;

    f32x4 tC[8];
    bf16x4 pA[8], pAl[8], h0h[8], d0[8], h1h[8], d1[8];       // (pAl: DS only)
    float dlda = 0.0f, frow = 0.0f;
    half4_ ph4 = half4_{};
    {
      // ---- coordinate layer on the matrix cores: C h0pre = (C Wc) x' + C bc + C hz[b] ----
      bf16x4 bx = h8_zero4();
      {
        const float v = q == 0 ? x0 : x1;
        _Float16 vh, vl;
        h8_split(v, vh, vl);
        if (q < 2) { h8_put(bx, 0, vh); h8_put(bx, 1, vl); h8_put(bx, 2, vh); }
        else if (q == 2) { h8_put(bx, 0, (_Float16)1.0f); h8_put(bx, 1, (_Float16)1.0f); }
      }
      const bf16x4* atab = reinterpret_cast<const bf16x4*>(smb + HO_ATAB) + lane;
      bf16x4 aop[8];
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        tC[jb] = *reinterpret_cast<const f32x4*>(chz + 16 * jb + 4 * q);
        aop[jb] = atab[64 * jb];
      }
      H8_FENCE();
      if (hzs != 1.0f) {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) tC[jb] = tC[jb] * hzs;
      }
      H8_FENCE();
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) tC[jb] = h8_mma16(aop[jb], bx, tC[jb]);
      H8_FENCE();
      h8_tanh8<false>(tC, 1.0f);
      h8_save8(tC, h0h, d0);
    }
    asm volatile("; H8_P1_coord");
    fetch_unit_inputs(pos_nx);                 // the slots were consumed by the coordinate layer above
    if (tile_no > 0) {
      // W2's images were the previous tile's staging area: bring them back under the forward of layer 1
      // (every compiler-visible load above has been consumed; none is issued before the barrier below)
      h8_wait_vm0();
      h8_reload(gimg + 2 * IMG_BYTES, lds0 + HO_W2H, wave, lane);
    }
    {
      h8_layer_fwd(W1h, b1s, h0h, tC, wad, q);
      if (mul1) h8_tanh8<true>(tC, rc1); else h8_tanh8<false>(tC, 1.0f);
      h8_save8(tC, h1h, d1);
    }
    asm volatile("; H8_P2_l1");
    h8_wait_vm0();
    __syncthreads();      // barrier (a): W2 landed everywhere; every wave is past its reads of W1 (staging may overwrite it)
    {
      h8_layer_fwd(W2h, b2s, h1h, tC, wad, q);
      if (mul2) h8_tanh8<true>(tC, rc2); else h8_tanh8<false>(tC, 1.0f);        // tC = h2
      // ---- output layer + likelihood (fp32); tC <- g = wo (1 - h2^2), pA <- fp16(h2) ----
      f32x4 part4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wos + 16 * jb + 4 * q);
        part4 = part4 + tC[jb] * wv;
        pA[jb] = __builtin_bit_cast(bf16x4, __builtin_convertvector(tC[jb], half4_));
        const f32x4 t2 = tC[jb] * tC[jb];
        tC[jb] = wv - wv * t2;
      }
      const float a = h8_sum_q((part4[0] + part4[1]) + (part4[2] + part4[3])) + bo;
      float ll, locv;
      if (LIK == PV_LIK_BERNOULLI) {
        const float pr = h8_rcp(1.0f + h8_exp(-a));
        const float pc = fminf(fmaxf(pr, BERN_EPS), 1.0f - BERN_EPS);
        // -BCEWithLogits(lg, x) with lg = logit(pc) (torch: probs_to_logits, then binary_cross_entropy_with_logits), written with
        // the identities 1 + exp(-|lg|) = 1 / max(pc, 1 - pc) and sigmoid(lg) = pc: the two logarithms lg is made of serve the
        // softplus term too, and the row's dependent chain is exp -> rcp -> 2 log instead of seven transcendentals (round 5)
        const float lpc = h8_log(pc), l1pc = h8_log(1.0f - pc);
        const float lg = lpc - l1pc;
        ll = -(fmaxf(lg, 0.0f) - lg * xv - fmaxf(lpc, l1pc));
        const float mask = (pr >= BERN_EPS && pr <= 1.0f - BERN_EPS) ? 1.0f : 0.0f;
        dlda = (pc - xv) * mask;
        locv = pr;
      } else if (LIK == PV_LIK_CBERNOULLI) {
        pv_cbern(a, xv, ll, dlda, locv);
      } else {
        const float pr = f.sigmoid_out ? h8_rcp(1.0f + h8_exp(-a)) : a;
        const float d = xv - pr;
        ll = -(d * d) / (2.0f * f.sig * f.sig) - h8_log(f.sig) - LOG_SQRT_2PI;
        dlda = -d / (f.sig * f.sig) * (f.sigmoid_out ? pr * (1.0f - pr) : 1.0f);
        locv = pr;
      }
      dlda *= act ? swv : 0.0f;
      // dL/dlogit = m 2^e: the mantissa (times kappa s_o) goes down the dgrad chain, the exponent into the staged rows
      const int e = __builtin_amdgcn_frexp_expf(dlda);
      const float dn = __builtin_amdgcn_frexp_mantf(dlda) * kso;
      const float phf = __builtin_amdgcn_ldexpf(1.0f, e + f.dl_exp);
      const _Float16 ph = (_Float16)phf;
      ph4 = half4_{ph, ph, ph, ph};
      frow = __builtin_amdgcn_ldexpf(u0, e);
      if (q == 0) {
        if (act) {
          if (a_llrow) a_llrow[row] = ll;
          if (a_loc) a_loc[row] = locv;
        }
        dbo += dlda;
        inf_dl[r] = dlda; inf_ph[r] = (float)ph; inf_x0[r] = (float)ph * x0; inf_x1[r] = (float)ph * x1;
      }
      xv_next = x_of(pos_nx);
      {
        // ---- d(wo) += sum_rows dlda h2 : wave-local MFMA through the wave's own rows of the layer-2 staging area (W1's
        // images are dead since barrier (a)); B = dlda of rows 4q..4q+3 in columns 3 (hi) and 4 (lo)
        h8_wait_lgkm0();
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(inf_dl + 4 * q);
        bf16x4 bw = h8_zero4();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          _Float16 hi, lo;
          h8_split(d4[i], hi, lo);
          h8_put(bw, i, r == 3 ? hi : (r == 4 ? lo : (_Float16)0.0f));
        }
        h8_colsum_mfma(sA2, pA, bw, accS, wave, r, q);
      }
      // dpre2 (normalised) = dn * wo (1 - h2^2)
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) tC[jb] = tC[jb] * dn;
      if (DS) h8_split8(tC, pA, pAl); else h8_cvt8(tC, pA);      // feeds the wgrad and the dgrad of layer 2
    }
    asm volatile("; H8_P3_fwd");
    pos_cur = pos_nx;                          // (unit, bu, row of THIS tile were taken above)
    const int ksteps = (nact + 1) >> 1;
    // ---- wgrad of layer 2: stage (dpre2, h1 2^(e + dl_exp)) of all 128 rows over W1's images, one pass ----
    h8_stage_store<false>(sA2, pA, 16 * wave + r, q);
    h8_stage_store<true>(sB2, h1h, 16 * wave + r, q, ph4);
    asm volatile("; H8_P4_stage2");
    __syncthreads();                                                // barrier (b)
    h8_wgrad_consume(sA2, sB2, accW2, accB2, wave, r, q, ksteps);
    asm volatile("; H8_P5_cons2");
    __syncthreads();                                                // barrier (c): consumed everywhere
    h8_wait_vm0();                                                  // (stores only: nothing the compiler still waits for)
    h8_reload(gimg, lds0 + HO_W1H, wave, lane);                     // W1 comes back under the dgrad of layer 2
    {
      h8_layer_dgrad<DS>(W2h, pA, pAl, tC, wad);                    // tC = (carried scales) dL/dh1
      h8_mul_d(tC, d1);
      if (DS) h8_split8(tC, pA, pAl); else h8_cvt8(tC, pA);        // dpre1: feeds the dgrad and the wgrad of layer 1
    }
    asm volatile("; H8_P6_dgrad2");
    h8_wait_vm0();
    __syncthreads();      // barrier (d): W1 landed everywhere; every wave is past its reads of W2
    bf16x4 p0h[8], p0l[8];
    {
      h8_layer_dgrad<DS>(W1h, pA, pAl, tC, wad);
      h8_mul_d(tC, d0);                                             // dpre0 (normalised; carried scales: u0 2^e)
      h8_split8(tC, p0h, p0l);
    }
    {
      // ---- coordinate layer backward, row-local part on the matrix cores: D[m][row] = sum_j T[m][j] dpre0[row][j] ----
      f32x4 dd = {0.0f, 0.0f, 0.0f, 0.0f};
      const bf16x8* ttab = reinterpret_cast<const bf16x8*>(smb + HO_TTAB) + lane;
#pragma unroll
      for (int mm = 0; mm < 4; ++mm) {
        const bf16x8 ta = ttab[64 * mm];
        dd = h8_mma(ta, h8_cat(p0h[2 * mm], p0h[2 * mm + 1]), dd);
        dd = h8_mma(ta, h8_cat(p0l[2 * mm], p0l[2 * mm + 1]), dd);
      }
      if (q == 0 && act) {
        const float d0_ = (dd[0] + dd[1]) * frow, d1_ = (dd[2] + dd[3]) * frow;
        a_rowtp[row] = sc * (d1_ * u0c - d0_ * u1c);
        a_rowtp[a_M + row] = d0_ * u0c + d1_ * u1c;
        a_rowtp[2 * a_M + row] = d0_;
        a_rowtp[3 * a_M + row] = d1_;
      }
      if (act && bu != cur_b) {
        if (cur_b >= 0) flush_hz(cur_b, r, q);
        cur_b = bu;
      }
      // ---- dL/d(hz[b]) = sum_rows dpre0, dWc_k = sum_rows dpre0 x'_k : wave-local MFMAs through the wave's own rows of the
      // layer-1 staging area (W2's images are dead since barrier (d)); every B column carries the row's 2^(e + dl_exp):
      // 0 ph | 1, 5 ph x0 (hi, lo) | 2, 6 ph x1 (hi, lo); both pieces of dpre0
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(inf_x0 + 4 * q);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(inf_x1 + 4 * q);
      const f32x4 ap = *reinterpret_cast<const f32x4*>(inf_ph + 4 * q);
      bf16x4 bc_ = h8_zero4();
      const bool use1 = r == 2 || r == 6, lo_col = r >= 5;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        _Float16 hi, lo;
        h8_split(use1 ? a1[i] : a0[i], hi, lo);
        _Float16 v = lo_col ? lo : hi;
        if (r == 0) v = (_Float16)ap[i];
        if (r == 3 || r == 4 || r > 6) v = (_Float16)0.0f;
        h8_put(bc_, i, v);
      }
      h8_colsum_mfma(sA1, p0h, bc_, accS, wave, r, q);
      h8_colsum_mfma(sA1, p0l, bc_, accS, wave, r, q);
    }
    asm volatile("; H8_P7_dgrad1_colsum");
    // ---- wgrad of layer 1: stage (dpre1, h0 2^(e + dl_exp)) over W2's images ----
    h8_stage_store<false>(sA1, pA, 16 * wave + r, q);
    h8_stage_store<true>(sB1, h0h, 16 * wave + r, q, ph4);
    asm volatile("; H8_P8_stage1");
    __syncthreads();                                                // barrier (e)
    h8_wgrad_consume(sA1, sB1, accW1, accB1, wave, r, q, ksteps);
    asm volatile("; H8_P9_cons1");
    __syncthreads();                                                // barrier (f): the staging area is free (next tile's W2 reload)
  }

  {
    const int r = lane0 & 15, q = lane0 >> 4;
    if (cur_b >= 0) flush_hz(cur_b, r, q);
    // ---- the workgroup's gradient record (pv_sdec_fused.h: FD_REC) ----
    const int jp = wave >> 1, kh = wave & 1;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // C/D layout: lane (col = r, q), reg i -> dW[32jp + 16 (s ^ kh) + 4q + i][64kh + 16o + r]  (the consume's rotation)
          const int e = (32 * jp + 16 * (s_ ^ kh) + 4 * q + i) * FD_H + 64 * kh + 16 * o + r;
          rec[e] = accW1[s_][o][i] * uw1;
          rec[FD_H * FD_H + e] = accW2[s_][o][i] * uw2;
        }
    if (r == 0) {
      const int j0 = 32 * jp + 16 * kh;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rec[2 * FD_H * FD_H + j0 + 4 * q + i] = accB1[i] * uw1;
        rec[2 * FD_H * FD_H + FD_H + j0 + 4 * q + i] = accB2[i] * uw2;
      }
    }
    // per-wave column sums -> LDS (W1's images are dead: every wave is past the last tile's barriers) -> summed over the
    // waves in ascending order
    __syncthreads();
    {
      float* scr = reinterpret_cast<float*>(smb + HO_ST2);           // [wave][n][128] floats = 64 KB
#pragma unroll
      for (int jb = 0; jb < 8; ++jb)
        *reinterpret_cast<f32x4*>(scr + ((wave * 16 + r) * FD_H) + 16 * jb + 4 * q) = accS[jb];
    }
    const float tb = pv_wave_sum(dbo);
    if (lane0 == 0) red[wave] = tb;
    __syncthreads();
    if (tid < FD_H) {
      const float* scr = reinterpret_cast<const float*>(smb + HO_ST2);
      float vo = 0.0f, v0 = 0.0f, v1 = 0.0f;
#pragma unroll
      for (int w = 0; w < H8_WAVES; ++w) {
        const float* s_ = scr + (w * 16) * FD_H + tid;
        v0 += s_[1 * FD_H] + s_[5 * FD_H];
        v1 += s_[2 * FD_H] + s_[6 * FD_H];
        vo += s_[3 * FD_H] + s_[4 * FD_H];
      }
      rec[2 * FD_H * FD_H + 2 * FD_H + tid] = v0 * u0p;
      rec[2 * FD_H * FD_H + 3 * FD_H + tid] = v1 * u0p;
      rec[2 * FD_H * FD_H + 4 * FD_H + tid] = vo;
    }
    if (tid == 0) {
      float v = 0.0f;
      for (int w = 0; w < H8_WAVES; ++w) v += red[w];
      rec[2 * FD_H * FD_H + 5 * FD_H] = v;
    }
  }
}

// ds: dL/dpre split in both dgrads (H231) or not (H221); training launches only
int pv_sdec_fused_w8h_launch(const PvFused& f_in, int grid, bool ds, hipStream_t s) {
  PvFused f = f_in;
  f.ablate = 0;
  const size_t lds = H8_LDS_BYTES;
  const void* fn = nullptr;
#define H8_PICK(L) fn = ds ? reinterpret_cast<const void*>(&pv_sdec_w8h_kernel<L, true>) \
                           : reinterpret_cast<const void*>(&pv_sdec_w8h_kernel<L, false>)
  if (f.lik == PV_LIK_BERNOULLI) H8_PICK(PV_LIK_BERNOULLI);
  else if (f.lik == PV_LIK_GAUSSIAN) H8_PICK(PV_LIK_GAUSSIAN);
  else H8_PICK(PV_LIK_CBERNOULLI);
#undef H8_PICK
  PV_TRY(pv_set_dynamic_lds(fn, (int)lds));          // (per device and kernel)
  void* args[] = {&f};
  hipError_t e2 = hipLaunchKernel(fn, dim3(grid), dim3(H8_THREADS), args, lds, s);
  if (e2 != hipSuccess) return (int)e2;
  PV_LAUNCH_CHECK();
  return 0;
}
