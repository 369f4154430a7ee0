// pv_sdec_fused_bf16.hip — the fused persistent spatial-decoder forward+backward kernel with the 128x128
// contractions on the bf16 matrix cores in SPLIT precision ("bf16x3"):
//      x = hi + lo  (hi = bf16(x), lo = bf16(x - hi));   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi   (fp32 accumulate)
// Each product carries ~2^-16 relative error (the dropped lo*lo term and the residual of the split), i.e.
// fp32-class results (the parity tests hold it to the same 1e-4 bar as the f32-MFMA kernel) at 3/16 of the
// f32-input MFMA's matrix time: v_mfma_f32_16x16x32_bf16 does 8x the k of v_mfma_f32_16x16x4_f32 in about half
// the cycles.  Everything outside the two hidden layers' GEMMs (coordinate layer, tanh, logit, likelihood,
// reductions) stays fp32.
//
// Mapping.  One persistent 256-thread workgroup per CU = 4 waves, ONE per SIMD, each with the full 512-register
// budget (the 8-wave / 256-register form of this kernel spilled its persistent accumulators to scratch every
// tile).  A tile is 64 rows: each wave carries one 16-row unit.  As in pv_sdec_fused.hip:
//   * layers are computed transposed, D[j][r] = sum_k W[j][k] h[r][k], so the 16x16 C/D layout of one layer is
//     the B-operand layout of the next: activations stay in registers through forward and dgrad;
//   * a wave owns two 16x128 slices of dW1 and dW2 in accumulators for the whole kernel and the workgroup
//     exchanges (dpre, h) through LDS for the wgrad;
//   * per-workgroup partial gradients are reduced in a fixed order afterwards (no float atomics).
// What the bf16 split changes:
//   * weights live in LDS as bf16 hi and lo images, row-major [128][128] with permuted columns and XOR-swizzled
//     16-byte chunks (pv_fb_layout.h); the
//     forward reads a lane's A operand with one ds_read_b128, the dgrad reads the TRANSPOSED operand from the same
//     image with ds_read_b64_tr_b16 (hardware 4x16 transpose), so one image serves both orientations;
//   * activations are split to (hi, lo) once per tensor (3 VALU per element) and the split feeds both the next
//     contraction and the wgrad exchange;
//   * LDS OVERLAY.  The two layers' weight images fill 128 of the 160 KB, which leaves no room to stage a whole
//     tile's (dpre, h) for the wgrad.  So the staging area of layer 2's wgrad lies OVER W1's images (idle between
//     the forward of layer 1 and the dgrad of layer 1) and that of layer 1's wgrad and of the coordinate layer's
//     reductions OVER W2's images (idle from the dgrad of layer 2 to the next tile's forward of layer 2); the
//     overwritten images come back by LDS-DMA (global_load_lds_dwordx4, no registers, asynchronous) from a
//     pre-split global copy made once per step (pv_fb_prep_kernel), under the dgrad of layer 2 resp. the next
//     tile's coordinate layer + forward of layer 1.  One staging pass per layer then covers all 64 rows: the
//     wgrad contracts 32 rows per v_mfma_f32_16x16x32_bf16 (operands by ds_read_b64_tr_b16; bias gradients ride
//     along as an MFMA against ones) and a tile needs 6 workgroup barriers instead of 24;
//   * column sums that need no other wave's rows (d(wo), and dhz / dWc of the coordinate layer) are wave-local
//     transposes through the wave's own rows of whichever staging area is dead at that moment (fb_colsum).
#include "pv_sdec_fused.h"
#include "pv_fb_layout.h"
#include "pv_kernels.h"        // PvHeadBwd, pv_head_dz / pv_head_bwd_math: the own-sample epilogue's latent backward
#include <stdlib.h>
#include <stdio.h>

typedef short short4_ __attribute__((ext_vector_type(4)));
typedef short short8_ __attribute__((ext_vector_type(8)));
typedef int int4_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4_ lds_short4;

#define FB_WAVES 4               // waves per workgroup (one per SIMD), one 16-row unit each
#define TILE_UNITS FB_WAVES      // units per workgroup tile
#define TILE_ROWS (TILE_UNITS * FD_UNIT)
#define LDS2 144                 // staging arrays: 72-dword rows -> the 4x16 transposing reads are conflict-free
#define ST_ARR (TILE_ROWS * LDS2)          // elements of one staging array (64 rows)
#define ST_BYTES (2 * ST_ARR)              // 18,432
// byte offsets in dynamic LDS:  [ weight images and staging areas (FbLds<PREC>) | vectors ... ]
#define FB_TOP_BYTES (4 * IMG_BYTES + 3 * ST_BYTES - IMG_BYTES)   // 153,600: the largest map (three staging arrays over a lo image)
#define BO_VEC FB_TOP_BYTES                // fp32 vectors: Wc0, Wc1, bc, wo, b1, b2 (128 each)
#define BO_INFO (BO_VEC + 6 * FD_H * 4)    // per row of the tile (each wave its own 16): x0[64], x1[64], (fp16 modes) 2^e[64]
#define BO_RED (BO_INFO + 768)
#define BO_CHZ (BO_RED + 256)              // next tile's per-unit inputs, fetched by LDS-DMA a tile ahead:
#define BO_CTP (BO_CHZ + FB_WAVES * FD_H * 4)   //   hz[b] (128 floats), tp[b] (8 of 64 floats), grid rows (16*cd of 64)
#define BO_CGR (BO_CTP + FB_WAVES * 256)
#define BO_TAILV (BO_CGR + FB_WAVES * 256)  // column-parallel tail (H231 build): column-sum vectors [3][128] + per-wave row partials [2][4][16]
#define FB_LDS_BYTES (BO_TAILV + 2048)
#define FB_THREADS (64 * FB_WAVES)
#ifndef FB_GB
#define FB_GB 2                  // output blocks per operand group of the layer loops (8 / FB_GB groups per k-block)
#endif
#define FB_GPM (8 / FB_GB)
#ifndef FB_DGD
#define FB_DGD 1                 // operand prefetch distance (groups) of the dgrad loops
#endif
// ---- precision modes (template parameter PREC) -------------------------------------------------------------------------------
//   FB_P_BF16  plain bf16 operands, one product per contraction (plan.fused = 3 on small problems)
//   FB_P_X3    bf16 hi + lo everywhere, three products (rounds 1-3's fp32-class kernel)
//   FB_P_H221  (round 4) fp16: the WEIGHTS are two exact pieces, activations and dL/dpre ONE piece: forward 2 products, dgrad 2,
//              wgrad 1.  A weight's rounding error is systematic (the same perturbation in every one of the 2e5 rows), an
//              activation's is independent from row to row and averages out of every sum the step forms — so only the weights
//              keep their second piece.  5 product passes instead of 9, no 3-instruction split of every activation, half the
//              staging traffic.
//   FB_P_H223 / H321 / H333   the same with the wgrad (both operands split), the forward (split activations) or everything at
//              three products: the other corners of the error table (scripts/fb_prec_table.py, profiles/r04_fb_prec_table.txt)
// fp16's narrow exponent is dealt with by exact power-of-two scaling, never by a rounding:
//   * weight images hold s W with max |s W| in [1, 2) (pv_fb_layout.h); the forward un-scales inside tanh's own multiply, the
//     backward CARRIES the scales (dgrad outputs are s x larger) and removes them where a gradient leaves the kernel;
//   * a row's dL/dlogit = m 2^e is split: the mantissa m goes down the dgrad chain (every dL/dpre operand is then O(1..2^12)
//     whatever the data's scale or the row's weight), 2^(e + dl_exp) is folded into the row's STAGED activation (h 2^(e+dl_exp),
//     exact) so the wgrad product is unchanged; the bias gradient contracts dL/dpre against that factor (a 9th column block of
//     the staged rows) instead of against ones; per-row results (dL/dpre0) get 2^e back in fp32.
#define FB_P_BF16 0
#define FB_P_X3 1
#define FB_P_H221 2
#define FB_P_H223 3
#define FB_P_H321 4
#define FB_P_H333 5
#define FB_P_H2A1 6              // dgrad of layer 2 with dL/dpre2 split (3 products), the rest as H221
#define FB_P_H2B1 7              // dgrad of layer 1 with dL/dpre1 split
#define FB_P_H231 8              // both dgrads with split dL/dpre
#define FB_P_H131 9              // H231 whose FORWARD contracts the weights' hi piece only (experiment: is the forward's second product needed?)
template <int P> struct FbP {
  static constexpr bool F16 = P >= 2;                          // fp16 pieces, normalised weights, per-row exponent
  static constexpr int WP = P == FB_P_BF16 ? 1 : 2;            // weight pieces (LDS images per layer)
  static constexpr bool FWD_LO = P == FB_P_X3 || P == FB_P_H321 || P == FB_P_H333;   // forward also contracts the activation's lo piece
  static constexpr bool DGR2_LO = P == FB_P_X3 || P == FB_P_H333 || P == FB_P_H2A1 || P == FB_P_H231 || P == FB_P_H131;   // dgrad of layer 2 also contracts dL/dpre2's lo piece
  static constexpr bool DGR1_LO = P == FB_P_X3 || P == FB_P_H333 || P == FB_P_H2B1 || P == FB_P_H231 || P == FB_P_H131;   // dgrad of layer 1: dL/dpre1's
  static constexpr bool FWD_WLO = WP == 2 && P != FB_P_H131;   // the forward contracts the weights' lo piece
  static constexpr bool WG3 = P == FB_P_X3 || P == FB_P_H223 || P == FB_P_H333;      // wgrad: both operands split, three products
  static constexpr bool HL = FWD_LO || WG3;                    // activations carry a lo piece
  static constexpr bool DL2 = DGR2_LO || WG3;                  // dL/dpre2 / dL/dpre1 carry a lo piece
  static constexpr bool DL1 = DGR1_LO || WG3;
  static constexpr bool DL = DL2 || DL1;
  static constexpr bool KEEP_WO = !HL && !DL;                  // registers to keep d(wo) per lane for the whole kernel
  // fp16 modes whose dL/dpre is split anyway stage the lo pieces too, for the BIAS gradients only: db = sum_rows dpre is a
  // column sum whose terms cancel (to 1e-3 of their size on jiVAE's db2) and one 16-bit piece left it at 5.9e-4
  static constexpr bool BIAS_LO = F16 && DL2 && DL1 && !WG3;
};
#define FB_KAPPA 16.0f           // fp16 modes: dL/dpre2 operands are kappa * s_o * mantissa(dL/dlogit) * wo * (1 - h2^2): |.| <= 32

// LDS map of the weight images and the two staging areas (NST arrays each: NAD_D of dL/dpre pieces, then NAD_H of activation pieces).
//   one piece:                  [ W1h | W2h | st2 | st1 ]                      (no overlay, no reloads)
//   two pieces, NST = 4:        [ W1h W1l | gap 8K | W2h W2l ]   st2 over W1h W1l + gap, st1 over gap + W2h W2l
//   two pieces, NST = 2 or 3:   [ W1h | W1l | gap | W2l | W2h ]   st2 over W1l + gap, st1 over gap + W2l: only the lo images
//                               are overwritten and come back by LDS-DMA (32 KB per layer and tile instead of 64)
// Everything from BO_VEC on sits at the same offsets in all modes.
template <int P> struct FbLds {
  static constexpr bool OV = FbP<P>::WP == 2;
  static constexpr int NAD_D = (FbP<P>::WG3 || FbP<P>::BIAS_LO) ? 2 : 1;      // staging arrays of dL/dpre, of the activations
  static constexpr int NAD_H = FbP<P>::WG3 ? 2 : 1;
  static constexpr int NST = NAD_D + NAD_H;
  static constexpr int NOV = NST == 4 ? 2 : 1;                 // images under a staging area
  static constexpr int GAP = OV ? NST * ST_BYTES - NOV * IMG_BYTES : 0;
  static constexpr int W1H = 0;
  static constexpr int W1L = IMG_BYTES;
  static constexpr int W2H = !OV ? IMG_BYTES : (NST == 4 ? 2 * IMG_BYTES + GAP : 3 * IMG_BYTES + GAP);
  static constexpr int W2L = !OV ? 0 : (NST == 4 ? 3 * IMG_BYTES + GAP : 2 * IMG_BYTES + GAP);
  static constexpr int ST2 = !OV ? 2 * IMG_BYTES : (NST == 4 ? 0 : IMG_BYTES);
  static constexpr int ST1 = !OV ? 2 * IMG_BYTES + 2 * ST_BYTES : 2 * IMG_BYTES;
  // what a staging area overwrites: LDS offset, offset in the global image copy (W1h W1l W2h W2l), bytes
  static constexpr int RL1_LDS = NST == 4 ? 0 : IMG_BYTES, RL1_SRC = RL1_LDS, RL_BYTES = NOV * IMG_BYTES;
  static constexpr int RL2_LDS = NST == 4 ? W2H : W2L, RL2_SRC = NST == 4 ? 2 * IMG_BYTES : 3 * IMG_BYTES;
  static_assert(GAP >= 0, "staging overlays");
  static_assert(!OV || (ST2 + NST * ST_BYTES <= 2 * IMG_BYTES + GAP && ST1 + NST * ST_BYTES <= 4 * IMG_BYTES + GAP), "overlay");
  static_assert((OV ? 4 * IMG_BYTES + GAP : 2 * IMG_BYTES + 4 * ST_BYTES) <= BO_VEC, "images and staging end before the vectors");
};
static_assert(IMG_BYTES % (FB_WAVES * 1024) == 0, "image reload: whole 1 KB LDS-DMA pieces per wave");
static_assert(FB_LDS_BYTES <= 160 * 1024, "LDS budget");

#define LOG_SQRT_2PI 0.91893853320467274178f
#define BERN_EPS 1.1920928955078125e-07f
// The stage fences of the elementwise phases and layer loops: __builtin_amdgcn_sched_barrier(mask), mask 0 = nothing crosses.
// NOT to be relaxed under the iterative-ILP strategy this file is compiled with (Makefile): masks 0x2 / 0x6 (VALU, SALU may cross)
// measure another 0.9-1.4 % faster there and produce WRONG gradients in the H231 builds (35 parity tests fail; the same masks
// with the default scheduler, and mask 0 with iterative-ILP, pass everything) — profiles/r06l_sched_strategy.txt.
#ifndef FB_FENCE_MASK
#define FB_FENCE_MASK 0
#endif
#ifdef FB_NO_FENCE
#define FB_FENCE() do { } while (0)
#else
#define FB_FENCE() __builtin_amdgcn_sched_barrier(FB_FENCE_MASK)
#endif
// (per group of sites, for the experiment of profiles/r06l_sched_strategy.txt: A forward layer loop, B dgrad layer loop, C tanh stages,
//  D weight-gradient consume loop)
#ifndef FB_FENCE_MASK_A
#define FB_FENCE_MASK_A FB_FENCE_MASK
#endif
#ifndef FB_FENCE_MASK_B
#define FB_FENCE_MASK_B FB_FENCE_MASK
#endif
#ifndef FB_FENCE_MASK_C
#define FB_FENCE_MASK_C FB_FENCE_MASK
#endif
#ifndef FB_FENCE_MASK_D
#define FB_FENCE_MASK_D FB_FENCE_MASK
#endif
#define FB_FENCE_A() __builtin_amdgcn_sched_barrier(FB_FENCE_MASK_A)
#ifndef FB_FENCE_MASK_B1
#define FB_FENCE_MASK_B1 FB_FENCE_MASK_B
#endif
#ifndef FB_FENCE_MASK_B2
#define FB_FENCE_MASK_B2 FB_FENCE_MASK_B
#endif
#define FB_FENCE_B1() __builtin_amdgcn_sched_barrier(FB_FENCE_MASK_B1)      // dgrad loop: between the operand requests and the MFMA group
#define FB_FENCE_B2() __builtin_amdgcn_sched_barrier(FB_FENCE_MASK_B2)      // dgrad loop: behind the MFMA group
#define FB_FENCE_C() __builtin_amdgcn_sched_barrier(FB_FENCE_MASK_C)
#define FB_FENCE_D() __builtin_amdgcn_sched_barrier(FB_FENCE_MASK_D)
typedef _Float16 half4_ __attribute__((ext_vector_type(4)));
typedef _Float16 half8_ __attribute__((ext_vector_type(8)));
// 16-bit operands travel as bf16x4 / bf16x8 bit containers in every mode; F16 picks the instruction that reads them
template <bool F16> __device__ __forceinline__ f32x4 fb_mma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_, a), __builtin_bit_cast(half8_, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float fb_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float fb_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fb_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fb_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ bf16x8 fb_cat(const bf16x4& a, const bf16x4& b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
// the forward's activation operand under the q-swapped image layout (pv_fb_layout.h): lanes of groups q >= 2 hold their weight
// chunk's halves in the other order (sw: per lane, loop-invariant)
__device__ __forceinline__ bf16x8 fb_catq(bf16x4 a, bf16x4 b, bool sw) {
  typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
  const u32x2_ ua = __builtin_bit_cast(u32x2_, a), ub = __builtin_bit_cast(u32x2_, b);
  const u32x2_ lo = {sw ? ub[0] : ua[0], sw ? ub[1] : ua[1]}, hi = {sw ? ua[0] : ub[0], sw ? ua[1] : ub[1]};
  return fb_cat(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi));
}

// a zero the compiler cannot see through: lane-address arithmetic that depends on it is redone where it is used
// instead of being hoisted out of the tile loop and held in (or spilled from) registers for the whole kernel
__device__ __forceinline__ int fb_opaque0() { int z = 0; asm volatile("" : "+v"(z)); return z; }

__device__ __forceinline__ bf16x4 fb_tr(const __bf16* p) {
  const short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)p);
  return __builtin_bit_cast(bf16x4, v);
}

// ---- LDS-DMA: 16 B per lane from global straight into LDS at (wave-uniform byte address) + 16 * lane.  hipcc
// does not count these in its s_waitcnt bookkeeping: fb_wait_vm0() before the landed data is read, and no
// compiler-visible global load may be pending when one is issued (the callers drain first).
__device__ __forceinline__ void fb_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void fb_glds4(const void* gsrc, unsigned lds_dst) {     // 4 B per lane, 256 B per wave
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void fb_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// BYTES of weight images (whole 32 KB images) from their global copy: BYTES / 4 KB one-KB pieces per wave
template <int BYTES>
__device__ __forceinline__ void fb_reload(const char* __restrict__ gimg, unsigned lds_dst, int wave, int lane) {
  constexpr int PIECES = BYTES / (FB_WAVES * 1024);
#pragma unroll
  for (int c = 0; c < PIECES; ++c) {
    const int off = (wave * PIECES + c) * 1024;
    fb_glds16(gimg + off + lane * 16, lds_dst + off);
  }
}

// forward layer of the wave's unit: out = bias + W in (pre-activation on return; times the images' scale in the fp16
// modes, whose bias vector is stored scaled); `in` arrives as 16-bit pieces (hi, and lo where the mode has one) per C/D
// block.  Stream: per k-block m (32 k's) and group of FB_GB output blocks read the weight operands (hi / lo, one group
// ahead) and issue FB_GB MFMAs per product: hi x hi, [hi x lo], [lo x hi] — independent accumulator chains.
template <int P>
__device__ __forceinline__ void fb_layer_fwd(const __bf16* __restrict__ Wh, const __bf16* __restrict__ Wl,
                                             const float* __restrict__ bs, const bf16x4 (&ih)[8],
                                             const bf16x4 (&il)[8], f32x4 (&out)[8], int r, int q, int qs) {
  constexpr bool F16 = FbP<P>::F16, WL = FbP<P>::FWD_WLO, AL = FbP<P>::FWD_LO;
  const bool sw = qs && q >= 2;                          // q-swapped images: this lane's chunk holds [h = 1 | h = 0]
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) out[ob] = *reinterpret_cast<const f32x4*>(bs + 16 * ob + 4 * q);
  // row 16*ob + r, logical chunk 4m + q  ->  physical chunk 4*(m ^ (r&3)) + (q ^ SL[r>>2])
  r |= fb_opaque0();
  const int lbase = r * LDB + 8 * (q ^ fb_sl(r >> 2));
  const __bf16* ah = Wh + lbase;
  const __bf16* al = Wl + lbase;
  int xm[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) xm[m] = 32 * (m ^ (r & 3));
  bf16x8 wh[2][FB_GB], wl[2][FB_GB];
  auto load = [&](int g, bf16x8 (&h)[FB_GB], bf16x8 (&l)[FB_GB]) {
    const int m = g / FB_GPM, op = (g % FB_GPM) * FB_GB;
#pragma unroll
    for (int o = 0; o < FB_GB; ++o) {
      const int off = 16 * (op + o) * LDB + xm[m];
      h[o] = *reinterpret_cast<const bf16x8*>(ah + off);
      if (WL) l[o] = *reinterpret_cast<const bf16x8*>(al + off);
    }
  };
  load(0, wh[0], wl[0]);
#pragma unroll
  for (int g = 0; g < 4 * FB_GPM; ++g) {
    const int m = g / FB_GPM, op = (g % FB_GPM) * FB_GB;
    if (g + 1 < 4 * FB_GPM) load(g + 1, wh[(g + 1) & 1], wl[(g + 1) & 1]);
    FB_FENCE_A();
    const bf16x8 bh = fb_catq(ih[2 * m], ih[2 * m + 1], sw);
    const bf16x8(&h)[FB_GB] = wh[g & 1];
    const bf16x8(&l)[FB_GB] = wl[g & 1];
#pragma unroll
    for (int o = 0; o < FB_GB; ++o) out[op + o] = fb_mma<F16>(h[o], bh, out[op + o]);
    if (AL) {
      const bf16x8 bl = fb_catq(il[2 * m], il[2 * m + 1], sw);
#pragma unroll
      for (int o = 0; o < FB_GB; ++o) out[op + o] = fb_mma<F16>(h[o], bl, out[op + o]);
    }
    if (WL) {
#pragma unroll
      for (int o = 0; o < FB_GB; ++o) out[op + o] = fb_mma<F16>(l[o], bh, out[op + o]);
    }
    FB_FENCE_A();
  }
}

// dgrad of the wave's unit: out[k] = sum_j W[j][k] dp[j]; A = W^T via the transposing LDS read; dp arrives as pieces
template <int P, bool AL>
__device__ __forceinline__ void fb_layer_dgrad(const __bf16* __restrict__ Wh, const __bf16* __restrict__ Wl,
                                               const bf16x4 (&ih)[8], const bf16x4 (&il)[8], f32x4 (&out)[8], int r,
                                               int q, int qs) {
  constexpr bool F16 = FbP<P>::F16, WL = FbP<P>::WP == 2;
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) out[kb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // lane i of 16-lane group q points at W[j0 + i/4][16*kb + 4*(i%4)], j0 = 32m + 4q (+16): after the transpose
  // lane k' holds W[j0 .. j0+3][16*kb + k']
  // (rows j0 + r/4 with j0 = 32m + 4q (+16): swizzle 4*(r>>2) + SL[q]; logical chunk 4*kk + (r&3), kk = kb/2)
  r |= fb_opaque0();
  // (q-swapped images: piece r&3 of column block 2 kk + h sits in half h ^ ((r&3) >> 1))
  const int toff = (4 * q + (r >> 2)) * LDB + 8 * ((r & 3) ^ fb_sl(q));
  const int hs = qs ? 4 * ((r >> 1) & 1) : 0;
  const __bf16* ahx[2] = {Wh + toff + hs, Wh + toff + 4 - hs};
  const __bf16* alx[2] = {Wl + toff + hs, Wl + toff + 4 - hs};
  int xk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) xk[kk] = 32 * (kk ^ (r >> 2));
  bf16x8 wh[FB_DGD + 1][FB_GB], wl[FB_DGD + 1][FB_GB];      // operands FB_DGD groups ahead (transposing reads are slow)
  auto load = [&](int g, bf16x8 (&h)[FB_GB], bf16x8 (&l)[FB_GB]) {
    const int m = g / FB_GPM, kp = (g % FB_GPM) * FB_GB;
#pragma unroll
    for (int o = 0; o < FB_GB; ++o) {
      const int off = 32 * m * LDB + xk[(kp + o) >> 1];
      const __bf16* ah = ahx[(kp + o) & 1];
      const __bf16* al = alx[(kp + o) & 1];
      h[o] = fb_cat(fb_tr(ah + off), fb_tr(ah + off + 16 * LDB));
      if (WL) l[o] = fb_cat(fb_tr(al + off), fb_tr(al + off + 16 * LDB));
    }
  };
#pragma unroll
  for (int g = 0; g < FB_DGD; ++g) load(g, wh[g], wl[g]);
#pragma unroll
  for (int g = 0; g < 4 * FB_GPM; ++g) {
    const int m = g / FB_GPM, kp = (g % FB_GPM) * FB_GB;
    if (g + FB_DGD < 4 * FB_GPM) load(g + FB_DGD, wh[(g + FB_DGD) % (FB_DGD + 1)], wl[(g + FB_DGD) % (FB_DGD + 1)]);
    FB_FENCE_B1();
    const bf16x8 bh = fb_cat(ih[2 * m], ih[2 * m + 1]);
    const bf16x8(&h)[FB_GB] = wh[g % (FB_DGD + 1)];
    const bf16x8(&l)[FB_GB] = wl[g % (FB_DGD + 1)];
#pragma unroll
    for (int o = 0; o < FB_GB; ++o) out[kp + o] = fb_mma<F16>(h[o], bh, out[kp + o]);
    if (AL) {
      const bf16x8 bl = fb_cat(il[2 * m], il[2 * m + 1]);
#pragma unroll
      for (int o = 0; o < FB_GB; ++o) out[kp + o] = fb_mma<F16>(h[o], bl, out[kp + o]);
    }
    if (WL) {
#pragma unroll
      for (int o = 0; o < FB_GB; ++o) out[kp + o] = fb_mma<F16>(l[o], bh, out[kp + o]);
    }
    FB_FENCE_B2();
  }
}

// Written as STAGES over all 32 values of a lane with scheduling fences in between (round 2): left alone, hipcc walks the
// values two at a time through the dependent chain mul -> exp -> add -> rcp -> fma, and the one wave of a SIMD then pays
// every instruction's latency (~10 cycles each instead of ~6.5; scripts/ubench/valu_rates.hip).
#define FB_C 2.8853900817779268f           // 2 log2(e): tanh(x) = 1 - 2 / (exp2(C x) + 1)
__device__ __forceinline__ void fb_tanh8(f32x4 (&v)[8], float c = FB_C) {       // c: C / (scale carried by v)
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) v[ob] = v[ob] * c;
  FB_FENCE_C();
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[ob][i] = __builtin_amdgcn_exp2f(v[ob][i]);
  FB_FENCE_C();
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) v[ob] = v[ob] + 1.0f;
  FB_FENCE_C();
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[ob][i] = __builtin_amdgcn_rcpf(v[ob][i]);
  FB_FENCE_C();
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) v[ob] = 1.0f - 2.0f * v[ob];
  FB_FENCE_C();
}

__device__ __forceinline__ void fb_mul_dtanh(f32x4 (&out)[8], const f32x4 (&h)[8]) {
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) out[kb] = out[kb] * (1.0f - h[kb] * h[kb]);
}

// a unit's rows -> 16-bit pieces per C/D block: hi, and lo = 16-bit(v - hi) where the mode wants it
template <bool F16, bool LO>
__device__ __forceinline__ void fb_presplit(const f32x4 (&v)[8], bf16x4 (&h)[8], bf16x4 (&l)[8]) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    if constexpr (F16) {
      const half4_ hh = __builtin_convertvector(v[jb], half4_);
      h[jb] = __builtin_bit_cast(bf16x4, hh);
      if (LO) {
        // lo = fp16(v - hi): the difference is exact, so one mixed-precision fma per value (f16 hi and f32 v in, one rounding into
        // its half of the pair) gives the same bits as cvt, sub, cvt — 4 instructions per four values instead of 10
        typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
        const u32x2_ hu = __builtin_bit_cast(u32x2_, hh);
        u32x2_ lu;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lu[0]) : "v"(hu[0]), "v"(v[jb][0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu[0]) : "v"(hu[0]), "v"(v[jb][1]));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lu[1]) : "v"(hu[1]), "v"(v[jb][2]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu[1]) : "v"(hu[1]), "v"(v[jb][3]));
        l[jb] = __builtin_bit_cast(bf16x4, lu);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (LO) { __bf16 a, b; fb_split(v[jb][i], a, b); h[jb][i] = a; l[jb][i] = b; }
        else h[jb][i] = (__bf16)v[jb][i];
      }
    }
  }
}
__device__ __forceinline__ void fb_zero8(bf16x4 (&h)[8], bf16x4 (&l)[8]) {
  const short4_ z = {0, 0, 0, 0};
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) { h[jb] = __builtin_bit_cast(bf16x4, z); l[jb] = __builtin_bit_cast(bf16x4, z); }
}
// the wave's 16 rows (row = 16 * wave + r) of one staged tensor: hi and lo arrays, row-major [64][LDS2].  Inside
// every 16-column block the four 8-byte pieces are XOR-swizzled by (row>>2)&3: ds_write_b64 is banked mod 32 and
// serviced 16 lanes (16 rows, one q) at a time, and 72-dword rows alone would put rows r and r+4 on the same banks
// (4-way); the transposing reads (fb_stage_toff) undo the swizzle and stay conflict-free.
// SCALED (fp16 modes, the activation operand): every piece times the row's power of two ph (exact), and the row's ph itself
// into column block 8 (the rows' padding) of the hi array: the bias gradient contracts against it (fb_wgrad_consume)
template <bool LO, bool SCALED>
__device__ __forceinline__ void fb_stage_store(__bf16* __restrict__ sh, __bf16* __restrict__ sl, const bf16x4 (&h)[8],
                                               const bf16x4 (&l)[8], int row, int q, half4_ ph = half4_{}) {
  row |= fb_opaque0();
  const int e = row * LDS2 + 4 * (q ^ ((row >> 2) & 3));
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    if constexpr (SCALED) {
      *reinterpret_cast<half4_*>(sh + e + 16 * jb) = __builtin_bit_cast(half4_, h[jb]) * ph;
      if (LO) *reinterpret_cast<half4_*>(sl + e + 16 * jb) = __builtin_bit_cast(half4_, l[jb]) * ph;
    } else {
      *reinterpret_cast<bf16x4*>(sh + e + 16 * jb) = h[jb];
      if (LO) *reinterpret_cast<bf16x4*>(sl + e + 16 * jb) = l[jb];
    }
  }
  if constexpr (SCALED) *reinterpret_cast<half4_*>(sh + e + 16 * 8) = ph;
}
// lane offset of the transposing read of staged rows R0 + 4q .. 4q+3 (R0 a multiple of 16), columns 16*blk ..
__device__ __forceinline__ int fb_stage_toff(int r, int q) { return (4 * q + (r >> 2)) * LDS2 + 4 * ((r & 3) ^ q); }

// wgrad of the wave's two 16-row slices (rows 16*(2*wave + s) ..) over the staged tile's 64 rows:
//   dW[j][k] += sum_rows dpre[row][j] h[row][k];   db[j] += sum_rows dpre[row][j]  (MFMA against ones)
// k-step ks contracts rows 32ks .. 32ks+31: lane group q feeds rows 32ks + {4q..4q+3, 16+4q..16+4q+3} of BOTH
// operands (two transposing reads each), which is all the contraction needs.
// (fp16 modes: the bias gradient's second operand is the rows' own factor — column block 8 of the staged activations — the
//  product then being what the weight gradient's is: dpre_n[row][j] * 2^(e_row + dl_exp); BIAS_LO: dL/dpre's lo pieces are
//  staged for this product alone)
template <int P>
__device__ __forceinline__ void fb_wgrad_consume(const __bf16* st, f32x4 (&accW)[2][8], f32x4 (&accB)[2], int wave,
                                                 int r, int q, int ksteps) {
  constexpr bool F16 = FbP<P>::F16, WG3 = FbP<P>::WG3, AL = WG3 || FbP<P>::BIAS_LO;
  const int toff = fb_stage_toff(r | fb_opaque0(), q);
  const short one = 0x3f80;                           // bf16 1.0
  const short8_ ones_s = {one, one, one, one, one, one, one, one};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
  // two LDS byte addresses (the dpre arrays' and the activation arrays') carry everything that is not a compile-time
  // constant; opaque, so that every read of a k-step is base + immediate — one staging area lies beyond the 16-bit offset
  // field of a ds_read and cost a v_add_u32 per read (pv_sdec_fused_w8.hip: w8_wgrad_consume)
  unsigned la = (unsigned)(size_t)st + 2u * (unsigned)(toff + 32 * wave);
  unsigned lb = (unsigned)(size_t)st + 2u * (unsigned)(FbLds<P>::NAD_D * ST_ARR + toff);
  asm volatile("" : "+v"(la), "+v"(lb));
  constexpr unsigned ROW16 = 2u * 16 * LDS2, ARR = 2u * ST_ARR;
  auto tr_at = [](unsigned addr) {
    const short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(size_t)addr);
    return __builtin_bit_cast(bf16x4, v);
  };
  for (int ks = 0; ks < ksteps; ++ks) {
    bf16x8 a_h[2], a_l[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      a_h[s] = fb_cat(tr_at(la + 32u * s), tr_at(la + 32u * s + ROW16));
      if (AL) a_l[s] = fb_cat(tr_at(la + ARR + 32u * s), tr_at(la + ARR + 32u * s + ROW16));
    }
    bf16x8 bh[2][2], bl[2][2];
    auto load = [&](int kp, bf16x8 (&h)[2], bf16x8 (&l)[2]) {
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const unsigned off = 32u * (unsigned)(kp + o);
        h[o] = fb_cat(tr_at(lb + off), tr_at(lb + off + ROW16));
        if (WG3) l[o] = fb_cat(tr_at(lb + ARR + off), tr_at(lb + ARR + off + ROW16));
      }
    };
    load(0, bh[0], bl[0]);
    const bf16x8 bias_b = F16 ? fb_cat(tr_at(lb + 32u * 8), tr_at(lb + 32u * 8 + ROW16)) : ones;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      accB[s] = fb_mma<F16>(a_h[s], bias_b, accB[s]);
      if (AL) accB[s] = fb_mma<F16>(a_l[s], bias_b, accB[s]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int kp = 2 * g;
      if (g + 1 < 4) load(kp + 2, bh[(g + 1) & 1], bl[(g + 1) & 1]);
      FB_FENCE_D();
      const bf16x8(&h)[2] = bh[g & 1];
      const bf16x8(&l)[2] = bl[g & 1];
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int s = 0; s < 2; ++s) accW[s][kp + o] = fb_mma<F16>(a_h[s], h[o], accW[s][kp + o]);
      if (WG3) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int s = 0; s < 2; ++s) accW[s][kp + o] = fb_mma<F16>(a_h[s], l[o], accW[s][kp + o]);
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int s = 0; s < 2; ++s) accW[s][kp + o] = fb_mma<F16>(a_l[s], h[o], accW[s][kp + o]);
      }
      FB_FENCE_D();
    }
    la += 2 * ROW16; lb += 2 * ROW16;
  }
}

__device__ __forceinline__ float fb_sum_q(float v) {
  return pv_sum_rows(v);                             // (pv_common.h: v_permlane16/32_swap, the bits of the two shfl_xor sums)
}
// Column sums over a unit's 16 rows of a C/D-layout tensor v (lane (r, q): row r, columns 16*jb + 4q + i), optionally
// also weighted by two per-row scalars: wave-local transpose through LDS.  The wave writes its 16 x 128 fp32 tile
// into `tmp` (its OWN 16 rows of two adjacent staging arrays: 72 + 72 floats per row, which it is about to overwrite
// with its staging anyway), then lane l sums columns 2l, 2l+1 over the rows and adds them into its accumulator slots
// (round 4: the accumulators are two registers each per lane — lane l owns columns 2l, 2l+1 — where they were LDS words updated by
// read-modify-write: 8 KB of LDS freed for the third staging array of the fp16 build).  No workgroup barrier, no cross-lane VALU.
// NWT: 0 plain column sums; 2 also the sums weighted by w1 and by w2; 3 the plain sum weighted by w0 as well (fp16 modes:
// w0 = the row's 2^e, w1 / w2 = 2^e x0 / 2^e x1 — the rows come normalised)
template <int NWT>
__device__ __forceinline__ void fb_colsum(const f32x4 (&v)[8], float* __restrict__ tmp /* &arrA[16*wave][0] */,
                                          float2& acc0, float2& acc1, float2& acc2,
                                          const float* __restrict__ w1, const float* __restrict__ w2, int lane, int r,
                                          int q, const float* __restrict__ w0 = nullptr) {
  // row r: columns 0..63 at tmp[r*72 ..], columns 64..127 at tmp[ST_ARR/2 .. ] (the next array's same row; float units)
  constexpr int HALF = ST_ARR / 2;                       // one bf16 staging array, in floats
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    float* p = tmp + (jb < 4 ? 0 : HALF) + r * (LDS2 / 2) + 16 * (jb & 3) + 4 * q;
    *reinterpret_cast<f32x4*>(p) = v[jb];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the wave's own stores have landed
  const float* c = tmp + (lane < 32 ? 0 : HALF) + 2 * (lane & 31);
  float s0 = 0.0f, s1 = 0.0f, a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
  f32x4 wv0[4], wv1[4], wv2[4];                          // the 16 rows' weights: broadcast ds_read_b128
  if (NWT >= 2) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      wv1[k] = *reinterpret_cast<const f32x4*>(w1 + 4 * k);
      wv2[k] = *reinterpret_cast<const f32x4*>(w2 + 4 * k);
      if (NWT == 3) wv0[k] = *reinterpret_cast<const f32x4*>(w0 + 4 * k);
    }
  }
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const float2 x = *reinterpret_cast<const float2*>(c + rr * (LDS2 / 2));
    if (NWT == 3) { const float w = wv0[rr >> 2][rr & 3]; s0 += x.x * w; s1 += x.y * w; }
    else { s0 += x.x; s1 += x.y; }
    if (NWT >= 2) {
      const float u = wv1[rr >> 2][rr & 3], t = wv2[rr >> 2][rr & 3];
      a0 += x.x * u; a1 += x.y * u;
      b0 += x.x * t; b1 += x.y * t;
    }
  }
  acc0.x += s0; acc0.y += s1;
  if (NWT >= 2) {
    acc1.x += a0; acc1.y += a1;
    acc2.x += b0; acc2.y += b1;
  }
}

// ---- column-parallel tail (round 6; the H231 build) --------------------------------------------------------------------------
// At batch 256 a workgroup owns 49 units: twelve full 4-wave tiles and ONE unit more, which row-parallel costs a whole tile's
// dependent chain (1/13 of the launch) for one wave's work.  Here the four waves SHARE that unit, as the 8-wave kernel's tail
// does (pv_sdec_fused_w8.hip): wave w computes the column blocks 2w, 2w + 1 of every layer (a quarter of the matrix and of the
// transcendental instructions) and the waves exchange their 16-bit pieces through LDS.  After an exchange every wave holds the
// unit's whole row set in exactly the registers the row-parallel form keeps, so the weight-gradient staging is the row-parallel
// code run by one wave (wave 0: the 16 rows; wave 1: 16 zero rows — a k-step contracts 32), with the same barriers, LDS overlay
// and image reloads as a full tile.  What differs from the row-parallel arithmetic: sums over a row's 128 columns (the logit,
// the row-local coordinate backward) are four per-wave partial sums added in wave order, and the wave-local column sums
// (d(wo), dL/d(hz), dWc) are 16-lane butterflies — other summation orders of the same fp32 terms.
#ifndef FB_TAIL
#define FB_TAIL 1                // 0: the odd last unit row-parallel, as in rounds 1-5 (A/B builds)
#endif
// LDS of the tail: everything lives in the GAP between the two layers' images (never an image: no reload inside the tail) — one
// 8 KB exchange buffer (halves alternate: h0 | h1, then dpre hi | lo), three COMPACT 16-row staging arrays (the tail's weight
// gradient contracts its 16 rows with v_mfma_f32_16x16x16_f16: no zero rows, no 64-row arrays) and the unit's d(wo) sums; the
// other column sums and the per-wave row partials sit behind the prefetch slots (BO_TAILV)
template <int P> struct FbTailLds {
  using LL = FbLds<P>;
  static constexpr int G0 = 2 * IMG_BYTES;                                       // the gap (W1h | W1l | gap | W2l | W2h)
  static constexpr int E = G0;                                                   // exchange buffer, 2 x 4 KB
  static constexpr int ST_ROWS16 = 16 * LDS2 * 2;                                // bytes of a compact staging array
  static constexpr int SD = E + 8192;                                            // dL/dpre hi, lo
  static constexpr int SA = SD + 2 * ST_ROWS16;                                  // activations (scaled rows + their factor column)
  static constexpr int WOV = SA + ST_ROWS16;                                     // d(wo) column sums [128]
  static constexpr int V = BO_TAILV;                                             // dL/d(hz), dWc0, dWc1 [3][128]
  static constexpr int RP = V + 3 * FD_H * 4;                                    // row partials [2][4][16] (the logit's first, then d0 | d1)
  static_assert(LL::NST == 3 && LL::OV && LL::W1L + IMG_BYTES == G0 && LL::W2L == G0 + LL::GAP, "the tail's LDS map is written for the three-array overlay (H231)");
  static_assert(WOV + FD_H * 4 <= G0 + LL::GAP, "exchange buffer + compact staging + d(wo) fit in the gap");
  static_assert(FB_WAVES * 16 * 36 * 4 <= 2 * ST_ROWS16 && FB_WAVES * 16 * 36 * 4 <= IMG_BYTES, "fb_tail_colsum's transpose buffers");
  static_assert(RP + 2 * 4 * 16 * 4 <= FB_LDS_BYTES && FB_LDS_BYTES <= 160 * 1024, "tail vectors");
};
__device__ __forceinline__ float fb_row16_sum(float v) {       // sum over the 16 lanes of a row (all lanes end with it)
#define FB_DPP_F(X, CTRL) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), (CTRL), 0xF, 0xF, false))
  v += FB_DPP_F(v, 0x128);                            // row_ror:8            lane ^ 8
  { const float t = FB_DPP_F(v, 0x141); v += FB_DPP_F(t, 0x1B); }   // lane ^ 4
  v += FB_DPP_F(v, 0x4E);                             // lane ^ 2
  v += FB_DPP_F(v, 0xB1);                             // lane ^ 1
#undef FB_DPP_F
  return v;
}
// Column sums over the unit's 16 rows of the wave's two blocks v (lane (r, q): row r, columns 32 wave + 16 o + 4q + i), weighted
// per row by up to three weight vectors (w[k][row]; null: ones): a wave-local transpose through `tmp` (16 rows x 36 floats of
// this wave's own) — lane l sums column 32 wave + (l & 31) over rows 8 (l >> 5) .. +7, the two halves meet by one half-wave swap.
// A 16-lane DPP butterfly per value (the first cut of this tail) cost 6 dependent VALU per value: 2.4 k cycles for the 24 sums
// of the coordinate layer against ~0.5 k this way.
template <int NW>
__device__ __forceinline__ void fb_tail_colsum(const f32x4 (&v)[2], float* __restrict__ tmp, const float* __restrict__ w0,
                                               const float* __restrict__ w1, const float* __restrict__ w2, int lane, int r, int q,
                                               float (&out)[3]) {
#pragma unroll
  for (int o = 0; o < 2; ++o) *reinterpret_cast<f32x4*>(tmp + r * 36 + 16 * o + 4 * q) = v[o];
  __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the wave's own stores have landed
  const int col = lane & 31, r0 = 8 * (lane >> 5);
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float x = tmp[(r0 + k) * 36 + col];
    s0 += w0 ? x * w0[r0 + k] : x;
    if (NW >= 2) s1 += x * w1[r0 + k];
    if (NW >= 3) s2 += x * w2[r0 + k];
  }
  auto meet = [](float a) {
    const unsigned b = __float_as_uint(a);
    auto sw = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  };
  out[0] = meet(s0);
  if (NW >= 2) out[1] = meet(s1);
  if (NW >= 3) out[2] = meet(s2);
}
__device__ __forceinline__ void fb_xchg_put(char* smb, int off, const bf16x4 (&mine)[2], int wave, int lane) {
#pragma unroll
  for (int o = 0; o < 2; ++o) reinterpret_cast<bf16x4*>(smb + off)[(2 * wave + o) * 64 + lane] = mine[o];
}
__device__ __forceinline__ void fb_xchg_get(const char* smb, int off, bf16x4 (&all)[8], int lane) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) all[jb] = reinterpret_cast<const bf16x4*>(smb + off)[jb * 64 + lane];
}
// blocks 2 wave, 2 wave + 1 of a forward layer (fb_layer_fwd's arithmetic for two of its eight output blocks)
template <int P>
__device__ __forceinline__ void fb_tail_fwd(const __bf16* __restrict__ Wh, const __bf16* __restrict__ Wl,
                                            const float* __restrict__ bs, const bf16x4 (&ih)[8], f32x4 (&out)[2], int wave,
                                            int r, int q, int qs) {
  constexpr bool F16 = FbP<P>::F16, WL = FbP<P>::FWD_WLO;
  const bool sw = qs && q >= 2;
#pragma unroll
  for (int o = 0; o < 2; ++o) out[o] = *reinterpret_cast<const f32x4*>(bs + 16 * (2 * wave + o) + 4 * q);
  const int lbase = r * LDB + 8 * (q ^ fb_sl(r >> 2)) + 32 * wave * LDB;
  bf16x8 wh[4][2], wl[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int off = lbase + 16 * o * LDB + 32 * (m ^ (r & 3));
      wh[m][o] = *reinterpret_cast<const bf16x8*>(Wh + off);
      if (WL) wl[m][o] = *reinterpret_cast<const bf16x8*>(Wl + off);
    }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const bf16x8 bh = fb_catq(ih[2 * m], ih[2 * m + 1], sw);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[o] = fb_mma<F16>(wh[m][o], bh, out[o]);
    if (WL) {
#pragma unroll
      for (int o = 0; o < 2; ++o) out[o] = fb_mma<F16>(wl[m][o], bh, out[o]);
    }
  }
}
// blocks 2 wave, 2 wave + 1 of a dgrad layer (fb_layer_dgrad's arithmetic: h x dp_hi, h x dp_lo, l x dp_hi)
template <int P, bool AL>
__device__ __forceinline__ void fb_tail_dgrad(const __bf16* __restrict__ Wh, const __bf16* __restrict__ Wl, const bf16x4 (&ih)[8],
                                              const bf16x4 (&il)[8], f32x4 (&out)[2], int wave, int r, int q, int qs) {
  constexpr bool F16 = FbP<P>::F16, WL = FbP<P>::WP == 2;
  const int hs = qs ? 4 * ((r >> 1) & 1) : 0;
#pragma unroll
  for (int o = 0; o < 2; ++o) out[o] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // (kb = 2 wave + o: kb >> 1 = wave, kb & 1 = o)
  const int toff = (4 * q + (r >> 2)) * LDB + 8 * ((r & 3) ^ fb_sl(q)) + 32 * (wave ^ (r >> 2));
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    bf16x8 h[2], l[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int off = toff + 32 * m * LDB + (o ? 4 - hs : hs);
      h[o] = fb_cat(fb_tr(Wh + off), fb_tr(Wh + off + 16 * LDB));
      if (WL) l[o] = fb_cat(fb_tr(Wl + off), fb_tr(Wl + off + 16 * LDB));
    }
    const bf16x8 bh = fb_cat(ih[2 * m], ih[2 * m + 1]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[o] = fb_mma<F16>(h[o], bh, out[o]);
    if (AL) {
      const bf16x8 bl = fb_cat(il[2 * m], il[2 * m + 1]);
#pragma unroll
      for (int o = 0; o < 2; ++o) out[o] = fb_mma<F16>(h[o], bl, out[o]);
    }
    if (WL) {
#pragma unroll
      for (int o = 0; o < 2; ++o) out[o] = fb_mma<F16>(l[o], bh, out[o]);
    }
  }
}
// the tail's weight gradient: dW[j][k] += sum over the unit's 16 rows of dpre[row][j] h[row][k] (+ the bias gradient against the
// rows' factor column) from COMPACT staging arrays (dpre hi | dpre lo at sd, activations at sa; fb_stage_store's row layout) with
// the 16-row matrix instruction — fb_wgrad_consume's arithmetic for one half k-step
template <int P>
__device__ __forceinline__ void fb_wgrad_consume16(const __bf16* sd, const __bf16* sa, f32x4 (&accW)[2][8], f32x4 (&accB)[2],
                                                   int wave, int r, int q) {
  static_assert(FbP<P>::F16 && FbP<P>::BIAS_LO && !FbP<P>::WG3, "H231");
  const int toff = fb_stage_toff(r, q);
  auto tr4 = [](const __bf16* p_) { return __builtin_bit_cast(half4_, fb_tr(p_)); };
  half4_ a_h[2], a_l[2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) {
    a_h[s_] = tr4(sd + toff + 32 * wave + 16 * s_);
    a_l[s_] = tr4(sd + 16 * LDS2 + toff + 32 * wave + 16 * s_);
  }
  const half4_ bias_b = tr4(sa + toff + 16 * 8);
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) {
    accB[s_] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_h[s_], bias_b, accB[s_], 0, 0, 0);
    accB[s_] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_l[s_], bias_b, accB[s_], 0, 0, 0);
  }
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    const half4_ b = tr4(sa + toff + 16 * kb);
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) accW[s_][kb] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_h[s_], b, accW[s_][kb], 0, 0, 0);
  }
}
__device__ __forceinline__ void fb_tanh2(f32x4 (&v)[2], float c) {
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[o][i] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v[o][i] * c) + 1.0f);
}
// two blocks -> fp16 pieces (fb_presplit's arithmetic)
template <bool LO>
__device__ __forceinline__ void fb_presplit2(const f32x4 (&v)[2], bf16x4 (&h)[2], bf16x4 (&l)[2]) {
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const half4_ hh = __builtin_convertvector(v[o], half4_);
    h[o] = __builtin_bit_cast(bf16x4, hh);
    if (LO) {
      typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
      const u32x2_ hu = __builtin_bit_cast(u32x2_, hh);
      u32x2_ lu;
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lu[0]) : "v"(hu[0]), "v"(v[o][0]));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu[0]) : "v"(hu[0]), "v"(v[o][1]));
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lu[1]) : "v"(hu[1]), "v"(v[o][2]));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu[1]) : "v"(hu[1]), "v"(v[o][3]));
      l[o] = __builtin_bit_cast(bf16x4, lu);
    }
  }
}

// phase-timing trace (profiling only, enabled by PV_FD_ABLATE bit 256): shader-clock stamps of workgroup 0 /
// wave 0 for its first tiles; read back with pv_debug_read_trace()
__device__ long long fb_trace[256];
#ifdef FB_TRACE
#define FB_STAMP(k)                                                                        \
  do {                                                                                     \
    if ((f.ablate & 256) && g == 0 && tid == 0 && tile_no < 4)                             \
      fb_trace[tile_no * 32 + (k)] = (long long)__builtin_readcyclecounter();              \
  } while (0)
#define FB_KSTAMP(k)                                                                       \
  do {                                                                                     \
    if ((f.ablate & 256) && blockIdx.x == 0 && threadIdx.x == 0)                           \
      fb_trace[200 + (k)] = (long long)__builtin_readcyclecounter();                       \
  } while (0)
#define FB_TSTAMP(k)                                                                       \
  do {                                                                                     \
    if ((f.ablate & 256) && blockIdx.x == 0 && threadIdx.x == 0)                           \
      fb_trace[160 + (k)] = (long long)__builtin_readcyclecounter();                       \
  } while (0)
#else
#define FB_STAMP(k) do { } while (0)     // (the stamps split basic blocks: compiled in only for scripts/gpu_trace.py)
#define FB_KSTAMP(k) do { } while (0)
#define FB_TSTAMP(k) do { } while (0)    // (the column-parallel tail's phases)
#endif

// stand-alone form of the per-step preparation (pv_fb_layout.h); the SVI step runs it inside the encoder's
// first-layer launch instead (pv_encoder.hip)
__global__ __launch_bounds__(256) void pv_fb_prep_kernel(PvFbPrep p) {
  pv_fb_prep(p, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256, 4, (int)(threadIdx.x >> 6));
}

// LIK: the likelihood is a compile-time choice (the rarely used ones must not cost the Bernoulli kernel registers)
// PREC: FB_P_* (above)
template <bool GRADS, int LIK, int PREC>
__global__ __launch_bounds__(FB_THREADS, 1) void pv_sdec_fused_bf16_kernel(PvFused f, PvEncFold e_arg) {
  (void)e_arg;                                   // (read in the epilogue only, through the kernarg pointer: see there)
  using PP = FbP<PREC>;
  using LL = FbLds<PREC>;
  constexpr bool F16 = PP::F16, OV = LL::OV;
  extern __shared__ __attribute__((aligned(16))) char smb[];
  FB_KSTAMP(0);                                                          // kernel entry
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x, G = gridDim.x;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smb);
  const __bf16* W1h = reinterpret_cast<const __bf16*>(smb + LL::W1H);
  const __bf16* W1l = reinterpret_cast<const __bf16*>(smb + LL::W1L);
  const __bf16* W2h = reinterpret_cast<const __bf16*>(smb + LL::W2H);
  const __bf16* W2l = reinterpret_cast<const __bf16*>(smb + LL::W2L);
  __bf16* st2 = reinterpret_cast<__bf16*>(smb + LL::ST2);          // wgrad-2 staging (two-piece weights: over W1's images)
  __bf16* st1 = reinterpret_cast<__bf16*>(smb + LL::ST1);          // wgrad-1 staging (two-piece weights: over W2's images)
  constexpr int SB = LL::NAD_D * ST_ARR;                           // the second operand's arrays follow the first's
  float* vec = reinterpret_cast<float*>(smb + BO_VEC);
  float* info = reinterpret_cast<float*>(smb + BO_INFO);
  float* red = reinterpret_cast<float*>(smb + BO_RED);
  const char* gimg = reinterpret_cast<const char*>(f.wimg);

  // ---- weight images by LDS-DMA, fp32 vectors by hand ----
  // (round 6) Training launches of the two-piece builds wait for W2's images in every tile anyway (the barrier in front of the
  // forward of layer 2: they are reloaded per tile), so the prologue only waits for W1's: W2's 64 KB are requested LAST — behind
  // the vectors, the first unit's observations and its prefetch slots — and land under the first tile's coordinate layer and
  // forward of layer 1.
  constexpr bool LATE_W2 = OV && GRADS;
  fb_reload<IMG_BYTES>(gimg, lds0 + LL::W1H, wave, lane);
  if (!LATE_W2) fb_reload<IMG_BYTES>(gimg + 2 * IMG_BYTES, lds0 + LL::W2H, wave, lane);
  if (PP::WP == 2) {
    fb_reload<IMG_BYTES>(gimg + IMG_BYTES, lds0 + LL::W1L, wave, lane);
    if (!LATE_W2) fb_reload<IMG_BYTES>(gimg + 3 * IMG_BYTES, lds0 + LL::W2L, wave, lane);
  }
  // fp16 modes: the images' power-of-two scales (written by the preparation, pv_fb_layout.h); everything derived from them
  // is wave-uniform.  The forward un-scales in tanh's multiply (c1, c2); the backward carries kso * m down the dgrad chain
  // and removes uw2 / uw1 / u0 where a gradient leaves the kernel.
  float s1 = 1.0f, s2 = 1.0f, kso = 1.0f, c1 = FB_C, c2 = FB_C, uw2 = 1.0f, uw1 = 1.0f, u0 = 1.0f;
  if (F16) {
    const float* sc = reinterpret_cast<const float*>(gimg + FB_SCALE_OFF);
    s1 = __builtin_amdgcn_readfirstlane(sc[0]); s2 = __builtin_amdgcn_readfirstlane(sc[1]);
    kso = FB_KAPPA * __builtin_amdgcn_readfirstlane(sc[2]);
    c1 = FB_C / s1; c2 = FB_C / s2;
    uw2 = __builtin_amdgcn_ldexpf(1.0f / kso, -f.dl_exp);
    uw1 = uw2 / s2;
    u0 = 1.0f / (kso * s1 * s2);
  }
  for (int j = tid; j < FD_H; j += FB_THREADS) {
    vec[j] = f.Wc[j * f.cd];
    vec[FD_H + j] = f.cd == 2 ? f.Wc[j * 2 + 1] : 0.0f;
    vec[2 * FD_H + j] = f.bc[j];
    vec[3 * FD_H + j] = f.wo[j];
    vec[4 * FD_H + j] = f.b1[j] * s1;
    vec[5 * FD_H + j] = f.b2[j] * s2;
  }
  if (!LATE_W2) {
    fb_wait_vm0();
    __syncthreads();
  }
  const float bo = f.bo[0];

  // persistent accumulators: dW1 / dW2 rows 16*(2*wave + s) .. +15, s = 0, 1 ; bias sums likewise
  f32x4 accW1[2][8], accW2[2][8], accB1[2], accB2[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    accB1[s] = f32x4{0, 0, 0, 0};
    accB2[s] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) { accW1[s][kb] = f32x4{0, 0, 0, 0}; accW2[s][kb] = f32x4{0, 0, 0, 0}; }
  }
  float dbo = 0.0f;
  // plain bf16 has the registers to keep d(wo) per lane (row r's share of columns 16*jb + 4q + i) for the whole kernel:
  // one cross-row reduction at the end instead of one per tile
  f32x4 accWo[PP::KEEP_WO ? 8 : 1];
#pragma unroll
  for (int jb = 0; jb < (PP::KEEP_WO ? 8 : 1); ++jb) accWo[jb] = f32x4{0, 0, 0, 0};
  int cur_b = -1;                                    // the sample whose dL/d(hz) this WAVE is accumulating
  const int upb = f.N / FD_UNIT;
  float* rec = f.part + (int64_t)g * FD_REC;
  // this wave's column-sum accumulators (lane l owns columns 2l, 2l+1 of each): d(wo), dL/d(hz[cur_b]) (flushed when the wave's
  // sample changes), dWc0, dWc1
  float2 cs_wo = float2{0.0f, 0.0f}, cs_hz = float2{0.0f, 0.0f}, cs_c0 = float2{0.0f, 0.0f}, cs_c1 = float2{0.0f, 0.0f};
  float2 cs_none = float2{0.0f, 0.0f};

  // (round 6) f.part_rs: the wave's running sums over its rows of sample cur_b of {ll, d(phi), d(scale), d(tx), d(ty)} — every lane
  // (r, q) carries row r's share (the four q copies are equal), published with dL/d(hz) instead of five stores per row and tile
  float rs0 = 0.0f, rs1 = 0.0f, rs2 = 0.0f, rs3 = 0.0f, rs4 = 0.0f;
  auto flush_hz = [&](int b) {
    // the wave's rows of sample b end (or the workgroup's do): publish its partial dL/d(hz[b]) in its own slot
    // (kmax counts FB_WAVES slots per workgroup that can touch a sample; unused slots stay zero)
    const int64_t ub = (int64_t)b * upb;
    const int gfirst = (int)(((ub + 1) * G + f.units - 1) / f.units) - 1;
    const int64_t slot = (int64_t)b * f.kmax + (g - gfirst) * FB_WAVES + wave;
    *reinterpret_cast<float2*>(f.part_hz + slot * FD_H + 2 * lane) = cs_hz;
    cs_hz = float2{0.0f, 0.0f};
    if (GRADS && f.part_rs) {
      const float v0 = fb_row16_sum(rs0), v1 = fb_row16_sum(rs1), v2 = fb_row16_sum(rs2), v3 = fb_row16_sum(rs3), v4 = fb_row16_sum(rs4);
      if (lane == 0) {
        float* d = f.part_rs + slot * PV_RS_W;
        *reinterpret_cast<f32x4*>(d) = f32x4{v0, v1, v2, v3};
        d[4] = v4;
      }
      rs0 = rs1 = rs2 = rs3 = rs4 = 0.0f;
    }
  };

  // A unit's sample b / offset inside the sample / observation unit are carried incrementally from tile to tile (round 2):
  // the int64 divisions this replaced expanded to ~900 scalar instructions per tile.  Units stay 64-bit here (this kernel
  // also serves problems beyond 2^31 rows).
  const int64_t u_lo = (int64_t)g * f.units / G, u_hi = (int64_t)(g + 1) * f.units / G;
  struct Pos { int64_t unit; int b, loc; int64_t xu; };   // unit = b * upb + loc ; xu = unit mod x_units (x_units > 0)
  auto pos_of = [&](int64_t unit_) {
    Pos p_;
    p_.unit = unit_; p_.b = (int)(unit_ / upb); p_.loc = (int)(unit_ - (int64_t)p_.b * upb);
    p_.xu = f.x_units > 0 ? unit_ % f.x_units : unit_;
    return p_;
  };
  auto advance = [&](Pos& p_, int by) {
    p_.unit += by; p_.loc += by; p_.xu += by;
    while (p_.loc >= upb) { p_.loc -= upb; ++p_.b; }
    if (f.x_units > 0) { while (p_.xu >= f.x_units) p_.xu -= f.x_units; }
  };
  // (H231 build) a range of 4 n + 1 units ends with a column-parallel tail (below): the row-parallel tiles cover [u_lo, u_end), and
  // what a wave without a further unit prefetches is the TAIL unit's inputs (every wave takes part in the tail)
  constexpr bool TAIL = FB_TAIL && GRADS && PREC == FB_P_H231;
#ifdef PV_EXPERIMENTS
  const int qsw = f.qswap;                  // the images' column order (pv_fb_layout.h): plain in the shipped build
#else
  constexpr int qsw = 0;
#endif
  const bool has_tail = TAIL && ((u_hi - u_lo) & (TILE_UNITS - 1)) == 1 && !(f.ablate & 512);   // (ablate: experiments build only)
  const int64_t u_end = has_tail ? u_hi - 1 : u_hi;
  const Pos pos_lo = pos_of(has_tail ? u_end : u_lo);   // what an out-of-range wave fetches instead (valid; unused without a tail)
  Pos pos_cur = pos_of(u_lo + wave < u_end ? u_lo + wave : (has_tail ? u_end : u_lo));
  Pos pos_nx = pos_cur;
  // the wave's observations are fetched one tile ahead (an HBM miss, and loads retire in order: fetched in the
  // tile itself it would hold up the coordinate layer's own small loads)
  float sw_next = 1.0f;                  // (jiVAE) weight of the unit's sample, fetched with its observations
  auto x_of = [&](const Pos& p_) -> float {
    if (f.sw) sw_next = f.sw[p_.b];
    return f.x[p_.xu * FD_UNIT + r];
  };
  float xv_next = x_of(pos_cur);
  // ... and so are its other per-unit inputs (hz[b], tp[b], the unit's grid rows), by LDS-DMA into the wave's own
  // slots: the coordinate layer then starts from LDS instead of waiting ~1.5k cycles on dependent global loads
  float* chz = reinterpret_cast<float*>(smb + BO_CHZ) + wave * FD_H;
  float* ctp = reinterpret_cast<float*>(smb + BO_CTP) + wave * 64;
  float* cgr = reinterpret_cast<float*>(smb + BO_CGR) + wave * 64;
  auto fetch_unit_inputs = [&](const Pos& p_) {
    const int b_ = p_.b;
    const int n0 = p_.loc * FD_UNIT;
    fb_glds4(f.hz + (int64_t)b_ * FD_H + lane, lds0 + BO_CHZ + wave * (FD_H * 4));
    fb_glds4(f.hz + (int64_t)b_ * FD_H + 64 + lane, lds0 + BO_CHZ + wave * (FD_H * 4) + 256);
    fb_glds4(f.tp + (int64_t)b_ * 8 + (lane & 7), lds0 + BO_CTP + wave * 256);
    fb_glds4(f.grid + (int64_t)n0 * f.cd + (lane & (16 * f.cd - 1)), lds0 + BO_CGR + wave * 256);
  };
  fetch_unit_inputs(pos_cur);
  if (LATE_W2) {
    // everything requested so far lands (W1's images, the first unit's observations and slots) — the observation registers are
    // touched here so that the compiler's own wait for them stands HERE and not at their first use inside the tile, where it
    // would drain W2's requests as well — then W2's images are requested and the tile loop starts without waiting for them
    fb_wait_vm0();
    asm volatile("" : "+v"(xv_next), "+v"(sw_next));
    fb_reload<IMG_BYTES>(gimg + 2 * IMG_BYTES, lds0 + LL::W2H, wave, lane);
    fb_reload<IMG_BYTES>(gimg + 3 * IMG_BYTES, lds0 + LL::W2L, wave, lane);
    __syncthreads();
  }
  int tile_no = -1;
  FB_KSTAMP(1);                                                          // prologue done
  for (int64_t ut = u_lo; ut < u_end; ut += TILE_UNITS) {
    ++tile_no;
    FB_STAMP(0);
    const int nact = (int)((u_end - ut) < TILE_UNITS ? (u_end - ut) : TILE_UNITS);
    if (ut + TILE_UNITS + wave < u_hi) advance(pos_nx, TILE_UNITS);     // the unit this wave fetches for the NEXT tile
    else pos_nx = pos_lo;
    int opq = 0;
    asm volatile("" : "+v"(opq));       // loop-variant zero: keeps LICM from hoisting the LDS-resident vectors
    const float* Wc0 = vec + opq;
    const float* Wc1 = vec + FD_H + opq;
    const float* bcs = vec + 2 * FD_H + opq;
    const float* wos = vec + 3 * FD_H + opq;
    const float* b1s = vec + 4 * FD_H + opq;
    const float* b2s = vec + 5 * FD_H + opq;

    // the wave's unit; an inactive wave (partial last tile) computes nothing and stages zeros
    const bool act = wave < nact;
    const int unit = (int)ut + (act ? wave : 0);
    const int bu = pos_cur.b;
    // the wave's sample changes with this tile: publish the old one's sums first (round 6: at the TOP of the tile — nothing between
    // here and the coordinate layer's column sums touches them — so that this tile's row sums can be added where they arise)
    if (GRADS && act && bu != cur_b) {
      if (cur_b >= 0) flush_hz(cur_b);
      cur_b = bu;
    }
    const int64_t row = (int64_t)unit * FD_UNIT + r;
    float x0, x1, u0c, u1c, sc;
    // this wave's LDS-DMA of the tile's inputs (issued a tile ago; first tile: in front of W2's images, which stay in flight)
    if (!(LATE_W2 && tile_no == 0)) fb_wait_vm0();
    {
      const float* t = ctp + opq;
      const float* gr = cgr + opq;
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
    const float* hzb = chz + opq;
    const float xv = xv_next, swv = sw_next;

    f32x4 h0[8], tA[8], tB[8], tC[8];
    bf16x4 pAh[8], pAl[8], pBh[8], pBl[8];
    // (round 2) A wave without a unit of its own (partial last tile) runs the same straight-line code on a valid unit of
    // the workgroup's range (what its prefetch slots hold) with its dL/dlogit forced to zero — every gradient it stages or
    // accumulates is then zero, its per-row outputs are not stored — instead of skipping the phases under wave-uniform
    // branches: the merges those branches create cost register copies in every tile (pv_sdec_fused_w8.hip).
    {
      // ---- coordinate layer (fp32): h0 = tanh(Wc x' + bc + hz[b]) ----
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        const int j = 16 * jb + 4 * q;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wc0 + j);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(Wc1 + j);
        const f32x4 bc = *reinterpret_cast<const f32x4*>(bcs + j);
        const f32x4 hz = *reinterpret_cast<const f32x4*>(hzb + j);
#pragma unroll
        for (int i = 0; i < 4; ++i) h0[jb][i] = w0[i] * x0 + w1[i] * x1 + bc[i] + hz[i];
      }
      fb_tanh8(h0);
      fb_presplit<F16, PP::FWD_LO>(h0, pBh, pBl);
    }
    FB_STAMP(1);
    fetch_unit_inputs(pos_nx);                   // the slots were consumed by the coordinate layer above
    if (OV && GRADS && tile_no > 0) {
      // W2's images were the previous tile's staging area: bring them back under the forward of layer 1.
      // (every compiler-visible load above has been consumed; none is issued before the barrier below)
      fb_wait_vm0();
      fb_reload<LL::RL_BYTES>(gimg + LL::RL2_SRC, lds0 + LL::RL2_LDS, wave, lane);
    }
    {
      fb_layer_fwd<PREC>(W1h, W1l, b1s, pBh, pBl, tB, r, q, qsw);
      fb_tanh8(tB, c1);                                          // tB = h1
      fb_presplit<F16, PP::HL>(tB, pBh, pBl);                    // feeds layer 2 and its wgrad
    }
    FB_STAMP(2);
    if (OV && GRADS) {
      fb_wait_vm0();
      __syncthreads();      // W2 landed everywhere; every wave is past its reads of W1 (staging may overwrite it)
    }
    FB_STAMP(3);
    float dlda = 0.0f;
    float frow = 0.0f;                                  // fp16 modes: what turns the row's normalised dL/dpre0 into the true one
    half4_ ph4 = half4_{};                              //             the row's 2^(e + dl_exp) as fp16 (staged activations)
    {
      fb_layer_fwd<PREC>(W2h, W2l, b2s, pBh, pBl, tC, r, q, qsw);
      FB_STAMP(16);
      fb_tanh8(tC, c2);                                          // tC = h2
      FB_STAMP(17);
      // ---- output layer + likelihood (fp32) ----
      // (four independent accumulation chains: one chain of 32 dependent multiply-adds is ~250 cycles of pure latency for the
      //  single wave of a SIMD)
      f32x4 part4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wos + 16 * jb + 4 * q);
        part4 = part4 + tC[jb] * wv;
      }
      const float a = fb_sum_q((part4[0] + part4[1]) + (part4[2] + part4[3])) + bo;
      float ll, locv;
      if (LIK == PV_LIK_BERNOULLI) {
        const float pr = fb_rcp(1.0f + fb_exp(-a));
        const float pc = fminf(fmaxf(pr, BERN_EPS), 1.0f - BERN_EPS);
        // -BCEWithLogits(lg, x) with lg = logit(pc) (torch: probs_to_logits, then binary_cross_entropy_with_logits), written with
        // the identities 1 + exp(-|lg|) = 1 / max(pc, 1 - pc) and sigmoid(lg) = pc: the two logarithms lg is made of serve the
        // softplus term too, and the row's dependent chain is exp -> rcp -> 2 log instead of seven transcendentals (round 5)
        const float lpc = fb_log(pc), l1pc = fb_log(1.0f - pc);
        const float lg = lpc - l1pc;
        ll = -(fmaxf(lg, 0.0f) - lg * xv - fmaxf(lpc, l1pc));
        const float mask = (pr >= BERN_EPS && pr <= 1.0f - BERN_EPS) ? 1.0f : 0.0f;
        dlda = (pc - xv) * mask;
        locv = pr;
      } else if (LIK == PV_LIK_CBERNOULLI) {
        pv_cbern(a, xv, ll, dlda, locv);
      } else {
        const float pr = f.sigmoid_out ? fb_rcp(1.0f + fb_exp(-a)) : a;
        const float d = xv - pr;
        ll = -(d * d) / (2.0f * f.sig * f.sig) - fb_log(f.sig) - LOG_SQRT_2PI;
        dlda = -d / (f.sig * f.sig) * (f.sigmoid_out ? pr * (1.0f - pr) : 1.0f);
        locv = pr;
      }
      FB_STAMP(18);
      dlda *= act ? swv : 0.0f;
      if (GRADS && f.part_rs) {
        rs0 += act ? ll : 0.0f;
        if (q == 0 && act && f.loc) f.loc[row] = locv;
      } else if (q == 0 && act) {
        if (f.llrow) f.llrow[row] = ll;
        if (f.loc) f.loc[row] = locv;
      }
      xv_next = x_of(pos_nx);                   // lands long before the next LDS-DMA issue point drains loads
      if (GRADS) {
        if (q == 0) dbo += dlda;
        if (PP::KEEP_WO) {
#pragma unroll
          for (int jb = 0; jb < 8; ++jb) accWo[jb] += dlda * tC[jb];
        } else {
          // d(wo)[j] += sum_rows dlda * h2[row][j]: W1's images are dead since the barrier above (the wgrad-2 staging
          // goes there next), so the wave's own staging rows serve as the transpose buffer
          f32x4 pv_[8];
#pragma unroll
          for (int jb = 0; jb < 8; ++jb) pv_[jb] = dlda * tC[jb];
          fb_colsum<0>(pv_, reinterpret_cast<float*>(st2) + (16 * wave) * (LDS2 / 2), cs_wo, cs_none, cs_none, nullptr,
                       nullptr, lane, r, q);
        }
        FB_STAMP(19);
        float dn = dlda;                       // the row factor of dpre2
        if (F16) {
          // dL/dlogit = m 2^e: the mantissa (times kappa s_o) goes down the dgrad chain, the exponent into the staged rows
          const int e = __builtin_amdgcn_frexp_expf(dlda);
          dn = __builtin_amdgcn_frexp_mantf(dlda) * kso;
          const _Float16 ph = (_Float16)__builtin_amdgcn_ldexpf(1.0f, e + f.dl_exp);
          ph4 = half4_{ph, ph, ph, ph};
          frow = __builtin_amdgcn_ldexpf(u0, e);
        }
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wos + 16 * jb + 4 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) tC[jb][i] = dn * wv[i] * (1.0f - tC[jb][i] * tC[jb][i]);  // dpre2
        }
        FB_STAMP(20);
        fb_presplit<F16, PP::DL2>(tC, pAh, pAl);                     // feeds the wgrad and the dgrad of layer 2
      }
    }
    FB_STAMP(4);
    pos_cur = pos_nx;                     // (unit, bu, row of THIS tile were taken above)
    if (!GRADS) continue;
    const int ksteps = nact > 2 ? 2 : 1;          // rows 32.. are only staged (as zeros or not) when a unit owns them

    // ---- wgrad of layer 2: stage (dpre2, h1) of all 64 rows over W1's images, one pass ----
    fb_stage_store<LL::NAD_D == 2, false>(st2, st2 + ST_ARR, pAh, pAl, 16 * wave + r, q);
    fb_stage_store<PP::WG3, F16>(st2 + SB, st2 + SB + ST_ARR, pBh, pBl, 16 * wave + r, q, ph4);
    __syncthreads();
    FB_STAMP(5);
    fb_wgrad_consume<PREC>(st2, accW2, accB2, wave, r, q, ksteps);
    if (OV) {
      __syncthreads();
      FB_STAMP(6);
      fb_wait_vm0();                                   // (stores only: nothing the compiler still waits for)
      fb_reload<LL::RL_BYTES>(gimg + LL::RL1_SRC, lds0 + LL::RL1_LDS, wave, lane);   // W1 comes back under the dgrad of layer 2
    }
    {
      fb_layer_dgrad<PREC, PP::DGR2_LO>(W2h, W2l, pAh, pAl, tA, r, q, qsw);
      fb_mul_dtanh(tA, tB);                                      // tA = dpre1 (fp16 modes: times s2 kso / m 2^e ... carried)
      fb_presplit<F16, PP::DL1>(tA, pAh, pAl);                   // feeds the dgrad and the wgrad of layer 1
    }
    FB_STAMP(7);
    if (OV) {
      fb_wait_vm0();
      __syncthreads();      // W1 landed everywhere; every wave is past its reads of W2
    }
    FB_STAMP(8);
    {
      fb_layer_dgrad<PREC, PP::DGR1_LO>(W1h, W1l, pAh, pAl, tC, r, q, qsw);
      fb_mul_dtanh(tC, h0);                                      // tC = dpre0 (fp16 modes: normalised; frow restores the row)
      // ---- coordinate layer, cross-row part: dhz[b] = sum_rows dpre0, dWc_k = sum_rows dpre0 * x'_k.  Wave-local
      // (the unit's rows all belong to this wave and to one sample): W2's images are dead since the barrier above, so
      // the wave's own rows of the wgrad-1 staging area serve as the transpose buffer.  No workgroup barrier.
      if (F16) {
        if (q == 0) {
          info[16 * wave + r] = frow * x0; info[TILE_ROWS + 16 * wave + r] = frow * x1; info[2 * TILE_ROWS + 16 * wave + r] = frow;
        }
        fb_colsum<3>(tC, reinterpret_cast<float*>(st1) + (16 * wave) * (LDS2 / 2), cs_hz, cs_c0, cs_c1,
                     info + 16 * wave, info + TILE_ROWS + 16 * wave, lane, r, q, info + 2 * TILE_ROWS + 16 * wave);
      } else {
        if (q == 0) { info[16 * wave + r] = x0; info[TILE_ROWS + 16 * wave + r] = x1; }
        fb_colsum<2>(tC, reinterpret_cast<float*>(st1) + (16 * wave) * (LDS2 / 2), cs_hz, cs_c0, cs_c1,
                     info + 16 * wave, info + TILE_ROWS + 16 * wave, lane, r, q);
      }
      fb_presplit<F16, PP::WG3>(h0, pBh, pBl);
    }
    FB_STAMP(9);
    // ---- wgrad of layer 1: stage (dpre1, h0) over W2's images ----
    fb_stage_store<LL::NAD_D == 2, false>(st1, st1 + ST_ARR, pAh, pAl, 16 * wave + r, q);
    fb_stage_store<PP::WG3, F16>(st1 + SB, st1 + SB + ST_ARR, pBh, pBl, 16 * wave + r, q, ph4);
    __syncthreads();
    FB_STAMP(10);
    fb_wgrad_consume<PREC>(st1, accW1, accB1, wave, r, q, ksteps);
    FB_STAMP(21);
    // ---- coordinate layer backward (fp32): row-local part ----
    {
      f32x4 d04 = {0.0f, 0.0f, 0.0f, 0.0f}, d14 = {0.0f, 0.0f, 0.0f, 0.0f};      // (independent chains, as the logit's)
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wc0 + 16 * jb + 4 * q);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(Wc1 + 16 * jb + 4 * q);
        d04 = d04 + tC[jb] * w0;
        d14 = d14 + tC[jb] * w1;
      }
      float d0 = fb_sum_q((d04[0] + d04[1]) + (d04[2] + d04[3]));
      float d1 = fb_sum_q((d14[0] + d14[1]) + (d14[2] + d14[3]));
      if (F16) { d0 *= frow; d1 *= frow; }
      if (f.part_rs) {
        if (act) {
          rs1 += sc * (d1 * u0c - d0 * u1c);
          rs2 += d0 * u0c + d1 * u1c;
          rs3 += d0;
          rs4 += d1;
        }
      } else if (q == 0 && act) {
        f.rowtp[row] = sc * (d1 * u0c - d0 * u1c);
        f.rowtp[f.M + row] = d0 * u0c + d1 * u1c;
        f.rowtp[2 * f.M + row] = d0;
        f.rowtp[3 * f.M + row] = d1;
      }
    }
    FB_STAMP(22);
    if (OV) __syncthreads();   // the staging area is free again (the next tile's W2 reload lands here)
    FB_STAMP(13);
  }
  if constexpr (TAIL) {
    if (has_tail) {
      // ================= column-parallel tail: ONE unit, four waves (helpers above the kernel) =================
      using TL = FbTailLds<PREC>;
      ++tile_no;
      const float* Wc0 = vec;
      const float* Wc1 = vec + FD_H;
      const float* bcs = vec + 2 * FD_H;
      const float* wos = vec + 3 * FD_H;
      const float* b1s = vec + 4 * FD_H;
      const float* b2s = vec + 5 * FD_H;
      const int bu = pos_cur.b;                          // (every wave's prefetch slots hold the tail unit's inputs)
      const int64_t row = (int64_t)u_end * FD_UNIT + r;
      float x0, x1, u0c, u1c, sc;
      fb_wait_vm0();
      if (f.cd == 2) {
        const float gx = cgr[2 * r], gy = cgr[2 * r + 1];
        u0c = gx * ctp[0] - gy * ctp[1];
        u1c = gx * ctp[1] + gy * ctp[0];
        sc = ctp[2];
        x0 = u0c * sc + ctp[3];
        x1 = u1c * sc + ctp[4];
      } else {
        u0c = cgr[r]; u1c = 0.0f; sc = 1.0f;
        x0 = u0c + ctp[3]; x1 = 0.0f;
      }
      FB_TSTAMP(0);
      const float xv = xv_next, swv = sw_next;
      f32x4 h0o[2], h1o[2], h2o[2], t2[2];
      bf16x4 oh[2], ol[2], pH0[8], pH1[8], pAh[8], pAl[8];
      // ---- coordinate layer, own blocks ----
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const int j = 16 * (2 * wave + o) + 4 * q;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wc0 + j);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(Wc1 + j);
        const f32x4 bc = *reinterpret_cast<const f32x4*>(bcs + j);
        const f32x4 hz = *reinterpret_cast<const f32x4*>(chz + j);
#pragma unroll
        for (int i = 0; i < 4; ++i) h0o[o][i] = w0[i] * x0 + w1[i] * x1 + bc[i] + hz[i];
      }
      fb_tanh2(h0o, FB_C);
      fb_presplit2<false>(h0o, oh, ol);
      fb_xchg_put(smb, TL::E, oh, wave, lane);
      if (tile_no > 0) {                                  // W2's lo image was the previous tile's staging area
        fb_wait_vm0();
        fb_reload<LL::RL_BYTES>(gimg + LL::RL2_SRC, lds0 + LL::RL2_LDS, wave, lane);
      }
      FB_TSTAMP(1);
      __syncthreads();                                                   // exchange 0: h0
      FB_TSTAMP(2);
      fb_xchg_get(smb, TL::E, pH0, lane);
      fb_tail_fwd<PREC>(W1h, W1l, b1s, pH0, h1o, wave, r, q, qsw);
      fb_tanh2(h1o, c1);
      fb_presplit2<false>(h1o, oh, ol);
      fb_xchg_put(smb, TL::E + 4096, oh, wave, lane);
      FB_TSTAMP(3);
      fb_wait_vm0();
      FB_TSTAMP(4);
      __syncthreads();                                                   // exchange 1: h1; W2 landed
      FB_TSTAMP(5);
      fb_xchg_get(smb, TL::E + 4096, pH1, lane);
      fb_tail_fwd<PREC>(W2h, W2l, b2s, pH1, h2o, wave, r, q, qsw);
      fb_tanh2(h2o, c2);
      float* rp = reinterpret_cast<float*>(smb + TL::RP);
      {
        f32x4 p4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int o = 0; o < 2; ++o) p4 = p4 + h2o[o] * *reinterpret_cast<const f32x4*>(wos + 16 * (2 * wave + o) + 4 * q);
        const float part = fb_sum_q((p4[0] + p4[1]) + (p4[2] + p4[3]));
        if (q == 0) rp[16 * wave + r] = part;
      }
      FB_TSTAMP(6);
      __syncthreads();                                                   // exchange 2: the logit's per-wave partial sums
      FB_TSTAMP(7);
      const float a = ((rp[r] + rp[16 + r]) + (rp[32 + r] + rp[48 + r])) + bo;
      float ll, locv, dlda;
      if (LIK == PV_LIK_BERNOULLI) {
        const float pr = fb_rcp(1.0f + fb_exp(-a));
        const float pc = fminf(fmaxf(pr, BERN_EPS), 1.0f - BERN_EPS);
        const float lpc = fb_log(pc), l1pc = fb_log(1.0f - pc);
        const float lg = lpc - l1pc;
        ll = -(fmaxf(lg, 0.0f) - lg * xv - fmaxf(lpc, l1pc));
        const float mask = (pr >= BERN_EPS && pr <= 1.0f - BERN_EPS) ? 1.0f : 0.0f;
        dlda = (pc - xv) * mask;
        locv = pr;
      } else if (LIK == PV_LIK_CBERNOULLI) {
        pv_cbern(a, xv, ll, dlda, locv);
      } else {
        const float pr = f.sigmoid_out ? fb_rcp(1.0f + fb_exp(-a)) : a;
        const float d = xv - pr;
        ll = -(d * d) / (2.0f * f.sig * f.sig) - fb_log(f.sig) - LOG_SQRT_2PI;
        dlda = -d / (f.sig * f.sig) * (f.sigmoid_out ? pr * (1.0f - pr) : 1.0f);
        locv = pr;
      }
      dlda *= swv;
      if (wave == 0) {
        // (the tail's row sums ride in wave 0's slot: its sample check first — the tile loop's, at the top of the tile)
        if (bu != cur_b) {
          if (cur_b >= 0) flush_hz(cur_b);
          cur_b = bu;
        }
        if (f.part_rs) rs0 += ll;
        if (q == 0) {
          if (f.llrow && !f.part_rs) f.llrow[row] = ll;
          if (f.loc) f.loc[row] = locv;
          dbo += dlda;
        }
      }
      // d(wo)[j] += sum_rows dlda h2[row][j], own blocks (transpose buffer: this wave's quarter of the not yet staged dpre arrays)
      {
        float* wov = reinterpret_cast<float*>(smb + TL::WOV);
        f32x4 pv_[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) pv_[o] = dlda * h2o[o];
        float cs[3];
        fb_tail_colsum<1>(pv_, reinterpret_cast<float*>(smb + TL::SD) + wave * (16 * 36), nullptr, nullptr, nullptr, lane, r, q, cs);
        if (lane < 32) wov[32 * wave + lane] = cs[0];
      }
      // dL/dlogit = m 2^e: the mantissa (times kappa s_o) goes down the dgrad chain, the exponent into the staged rows
      const int ex = __builtin_amdgcn_frexp_expf(dlda);
      const float dn = __builtin_amdgcn_frexp_mantf(dlda) * kso;
      const _Float16 ph = (_Float16)__builtin_amdgcn_ldexpf(1.0f, ex + f.dl_exp);
      const half4_ ph4 = half4_{ph, ph, ph, ph};
      const float frow = __builtin_amdgcn_ldexpf(u0, ex);
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wos + 16 * (2 * wave + o) + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) t2[o][i] = dn * wv[i] * (1.0f - h2o[o][i] * h2o[o][i]);     // dpre2
      }
      fb_presplit2<true>(t2, oh, ol);
      fb_xchg_put(smb, TL::E, oh, wave, lane);
      fb_xchg_put(smb, TL::E + 4096, ol, wave, lane);
      FB_TSTAMP(8);
      __syncthreads();                                                   // exchange 3: dpre2 (hi, lo)
      FB_TSTAMP(9);
      fb_xchg_get(smb, TL::E, pAh, lane);
      fb_xchg_get(smb, TL::E + 4096, pAl, lane);
      // ---- weight gradients: the unit's 16 rows, staged by wave 0 in compact arrays, contracted with the 16-row instruction ----
      __bf16* tsd = reinterpret_cast<__bf16*>(smb + TL::SD);
      __bf16* tsa = reinterpret_cast<__bf16*>(smb + TL::SA);
      auto stage = [&](const bf16x4 (&dh)[8], const bf16x4 (&dl)[8], const bf16x4 (&hh)[8]) {
        if (wave == 0) {
          fb_stage_store<true, false>(tsd, tsd + 16 * LDS2, dh, dl, r, q);
          fb_stage_store<false, true>(tsa, tsa, hh, dl, r, q, ph4);
        }
      };
      stage(pAh, pAl, pH1);
      FB_TSTAMP(10);
      __syncthreads();
      FB_TSTAMP(11);
      fb_wgrad_consume16<PREC>(tsd, tsa, accW2, accB2, wave, r, q);
      FB_TSTAMP(12);
      // ---- dgrad of layer 2, own blocks ----
      fb_tail_dgrad<PREC, PP::DGR2_LO>(W2h, W2l, pAh, pAl, t2, wave, r, q, qsw);
#pragma unroll
      for (int o = 0; o < 2; ++o) t2[o] = t2[o] * (1.0f - h1o[o] * h1o[o]);          // dpre1 (carried scales as in the tile loop)
      fb_presplit2<true>(t2, oh, ol);
      fb_xchg_put(smb, TL::E, oh, wave, lane);                          // (every wave read exchange 3 before the staging barrier)
      fb_xchg_put(smb, TL::E + 4096, ol, wave, lane);
      FB_TSTAMP(13);
      __syncthreads();                                                   // exchange 4: dpre1; the staged rows are consumed
      FB_TSTAMP(14);
      fb_xchg_get(smb, TL::E, pAh, lane);
      fb_xchg_get(smb, TL::E + 4096, pAl, lane);
      // ---- dgrad of layer 1, own blocks ----
      fb_tail_dgrad<PREC, PP::DGR1_LO>(W1h, W1l, pAh, pAl, t2, wave, r, q, qsw);
#pragma unroll
      for (int o = 0; o < 2; ++o) t2[o] = t2[o] * (1.0f - h0o[o] * h0o[o]);          // dpre0, normalised (frow restores the row)
      FB_TSTAMP(15);
      stage(pAh, pAl, pH0);
      {
        // coordinate layer: the column sums dL/d(hz), dWc0, dWc1 of the own blocks and the row-local partial sums
        // (transpose buffer: W2's lo image — the tail is the last tile and every wave is past the dgrad of layer 2; the rows'
        //  weights 2^e, 2^e x0, 2^e x1 in the wave's own slots of the tile loop's `info` rows)
        float* vv = reinterpret_cast<float*>(smb + TL::V);
        if (q == 0) {
          info[16 * wave + r] = frow * x0; info[TILE_ROWS + 16 * wave + r] = frow * x1; info[2 * TILE_ROWS + 16 * wave + r] = frow;
        }
        float cs[3];
        fb_tail_colsum<3>(t2, reinterpret_cast<float*>(smb + LL::W2L) + wave * (16 * 36), info + 2 * TILE_ROWS + 16 * wave,
                          info + 16 * wave, info + TILE_ROWS + 16 * wave, lane, r, q, cs);
        if (lane < 32) {
          vv[32 * wave + lane] = cs[0];
          vv[FD_H + 32 * wave + lane] = cs[1];
          vv[2 * FD_H + 32 * wave + lane] = cs[2];
        }
        f32x4 d04 = {0.0f, 0.0f, 0.0f, 0.0f}, d14 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int j = 16 * (2 * wave + o) + 4 * q;
          d04 = d04 + t2[o] * *reinterpret_cast<const f32x4*>(Wc0 + j);
          d14 = d14 + t2[o] * *reinterpret_cast<const f32x4*>(Wc1 + j);
        }
        const float d0 = fb_sum_q((d04[0] + d04[1]) + (d04[2] + d04[3]));
        const float d1 = fb_sum_q((d14[0] + d14[1]) + (d14[2] + d14[3]));
        if (q == 0) { rp[16 * wave + r] = d0; rp[64 + 16 * wave + r] = d1; }      // (the logit's partials were read before barrier 3)
      }
      FB_TSTAMP(16);
      __syncthreads();
      FB_TSTAMP(17);
      fb_wgrad_consume16<PREC>(tsd, tsa, accW1, accB1, wave, r, q);
      FB_TSTAMP(18);
      if (wave == 0) {
        const float* vv = reinterpret_cast<const float*>(smb + TL::V);
        const float* wov = reinterpret_cast<const float*>(smb + TL::WOV);
        const float2 a0 = *reinterpret_cast<const float2*>(vv + 2 * lane);
        const float2 a1 = *reinterpret_cast<const float2*>(vv + FD_H + 2 * lane);
        const float2 a2 = *reinterpret_cast<const float2*>(vv + 2 * FD_H + 2 * lane);
        const float2 aw = *reinterpret_cast<const float2*>(wov + 2 * lane);
        cs_hz.x += a0.x; cs_hz.y += a0.y;
        cs_c0.x += a1.x; cs_c0.y += a1.y;
        cs_c1.x += a2.x; cs_c1.y += a2.y;
        cs_wo.x += aw.x; cs_wo.y += aw.y;
        {
          const float* dp = rp;
          const float d0 = ((dp[r] + dp[16 + r]) + (dp[32 + r] + dp[48 + r])) * frow;
          const float d1 = ((dp[64 + r] + dp[80 + r]) + (dp[96 + r] + dp[112 + r])) * frow;
          if (f.part_rs) {
            rs1 += sc * (d1 * u0c - d0 * u1c);
            rs2 += d0 * u0c + d1 * u1c;
            rs3 += d0;
            rs4 += d1;
          } else if (q == 0) {
            f.rowtp[row] = sc * (d1 * u0c - d0 * u1c);
            f.rowtp[f.M + row] = d0 * u0c + d1 * u1c;
            f.rowtp[2 * f.M + row] = d0;
            f.rowtp[3 * f.M + row] = d1;
          }
        }
      }
    }
  }
  FB_KSTAMP(2);                                                          // last tile done
  if (!GRADS) return;

  // (round 6) this workgroup's units are exactly ONE sample (batch == grid): it hands the latent backward the sample's dL/d(hz),
  // dL/dz and row sums itself — its four waves' partials summed next to the other column sums — instead of four slots each
  const bool own = f.dhz_out != nullptr && u_lo == (int64_t)g * upb && u_hi - u_lo == upb;
  if (cur_b >= 0 && !own) flush_hz(cur_b);
  if (PP::KEEP_WO) {
    // (every wave is past the last tile's barriers: the wgrad-2 staging rows are free)
    f32x4 t[8];
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) t[jb] = accWo[PP::KEEP_WO ? jb : 0];
    fb_colsum<0>(t, reinterpret_cast<float*>(st2) + (16 * wave) * (LDS2 / 2), cs_wo, cs_none, cs_none, nullptr, nullptr,
                 lane, r, q);
  }
  // (round 6, third cut: PvEncFold::chain) a workgroup that owns its sample also runs the sample's latent backward and encoder
  // chain below (pv_sdec_fused_w8.hip has the story); the operands are requested here, ahead of the 138 KB of record stores
  // (`e` is read through an opaque pointer into the kernarg segment HERE: named directly, the compiler fetches its fields at kernel
  //  entry and carries them — spilled, 51 -> 92 SGPRs — through the tile loop: +3 us on the launch even with the chain switched off)
  const PvEncFoldArg ep_ = pv_kernarg_fold();
  const bool own_chain = own && ep_->chain && f.part_rs && f.dzc_out;
  float ch_whd[16], ch_a1 = 0.0f, ch_a0 = 0.0f, ch_z = 0.0f, ch_sig = 1.0f, ch_ep = 0.0f, ch_sp = 0.0f;
  f32x4 ch_w1[16];
  if (own_chain) {
    const int ho = ep_->head.out_dim;
    if (tid < FD_H) {
      const float* Wh = ep_->params + ep_->head.w_off + tid;
#pragma unroll
      for (int o = 0; o < 16; ++o) ch_whd[o] = Wh[(o < ho ? o : ho - 1) * FD_H];          // (clamped: multiplied by a zero below)
      ch_a1 = ep_->eact1[(int64_t)g * FD_H + tid];
      ch_a0 = ep_->eact0[(int64_t)g * FD_H + tid];
    }
    // (second hidden layer: thread (c = tid & 31, jg = tid >> 5) holds W1[16 jg .. 16 jg + 15][4c .. 4c + 3])
    const float* W1c = ep_->params + ep_->enc1.w_off + 4 * (tid & 31) + (16 * (tid >> 5)) * FD_H;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) ch_w1[jj] = *reinterpret_cast<const f32x4*>(W1c + jj * FD_H);
    if (lane < ep_->z_dim && wave < 2) {
      ch_z = ep_->z[(int64_t)g * ep_->z_dim + lane];
      ch_sig = ep_->z_scale[(int64_t)g * ep_->z_dim + lane];
      ch_ep = ep_->eps[(int64_t)g * ep_->z_dim + lane];
      ch_sp = ep_->head_out[(int64_t)g * ep_->ldh + ep_->z_dim + lane];
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j0 = 16 * (2 * wave + s);
    if (f.ablate & 1024) {                   // (experiments build: the row-major form of rounds 1-5, 128 scattered 4-byte stores per wave)
#pragma unroll
      for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // C/D layout: lane (col k' = r, q), reg i -> dW[j0 + 4*q + i][16*kb + r]
          rec[(j0 + 4 * q + i) * FD_H + 16 * kb + r] = accW1[s][kb][i] * uw1;
          rec[FD_H * FD_H + (j0 + 4 * q + i) * FD_H + 16 * kb + r] = accW2[s][kb][i] * uw2;
        }
    } else {
      // LANE-NATIVE (pv_sdec_fused.h PV_REC_LANE_F32): every accumulator block as it sits in registers, one 1 KB-contiguous
      // 16-byte-per-lane store each — 32 store instructions per wave instead of 128
      f32x4* rec4 = reinterpret_cast<f32x4*>(rec);
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        rec4[(((0 * FB_WAVES + wave) * 2 + s) * 8 + kb) * 64 + lane] = accW1[s][kb] * uw1;
        rec4[(((1 * FB_WAVES + wave) * 2 + s) * 8 + kb) * 64 + lane] = accW2[s][kb] * uw2;
      }
    }
    // bias gradients: every column of accB holds the same sums; lane (col 0, q), reg i -> row j0 + 4q + i
    if (r == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rec[2 * FD_H * FD_H + j0 + 4 * q + i] = accB1[s][i] * uw1;
        rec[2 * FD_H * FD_H + FD_H + j0 + 4 * q + i] = accB2[s][i] * uw2;
      }
    }
  }
  const float tb = pv_wave_sum(dbo);
  if (lane == 0) red[wave] = tb;
  __syncthreads();          // every wave is done with every staging area
  {
    // per-wave column sums -> the record (waves in ascending order): dWc0 | dWc1 | dwo, through the (free) first staging area
    float* d = reinterpret_cast<float*>(st2);
    *reinterpret_cast<float2*>(d + (3 * wave + 0) * FD_H + 2 * lane) = cs_wo;
    *reinterpret_cast<float2*>(d + (3 * wave + 1) * FD_H + 2 * lane) = cs_c0;
    *reinterpret_cast<float2*>(d + (3 * wave + 2) * FD_H + 2 * lane) = cs_c1;
    float* dh = d + 3 * FB_WAVES * FD_H;                  // own: [wave][128] dL/d(hz) partials, then [wave][8] row sums, then [2][16] dz partials
    float wzv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (own) {
      *reinterpret_cast<float2*>(dh + wave * FD_H + 2 * lane) = cs_hz;
      if (f.part_rs) {
        const float v0 = fb_row16_sum(rs0), v1 = fb_row16_sum(rs1), v2 = fb_row16_sum(rs2), v3 = fb_row16_sum(rs3), v4 = fb_row16_sum(rs4);
        if (lane == 0) {
          float* rr = dh + FB_WAVES * FD_H + 8 * wave;
          rr[0] = v0; rr[1] = v1; rr[2] = v2; rr[3] = v3; rr[4] = v4;
        }
      }
      if (f.dzc_out && tid < FD_H) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wzv[i] = i < f.lat_in ? f.Wz[(int64_t)tid * f.lat_in + i] : 0.0f;
      }
    }
    __syncthreads();
    float dhz_j = 0.0f;
    if (tid < FD_H) {
      float vo = 0.0f, v0 = 0.0f, v1 = 0.0f;
#pragma unroll
      for (int w = 0; w < FB_WAVES; ++w) {
        vo += d[(3 * w + 0) * FD_H + tid];
        v0 += d[(3 * w + 1) * FD_H + tid];
        v1 += d[(3 * w + 2) * FD_H + tid];
        if (own) dhz_j += dh[w * FD_H + tid];
      }
      rec[2 * FD_H * FD_H + 2 * FD_H + tid] = v0;
      rec[2 * FD_H * FD_H + 3 * FD_H + tid] = v1;
      rec[2 * FD_H * FD_H + 4 * FD_H + tid] = vo;
      if (own) f.dhz_out[(int64_t)g * FD_H + tid] = dhz_j;
    }
    if (own) {
      if (f.part_rs && tid < 5) {
        const float* rr = dh + FB_WAVES * FD_H + tid;
        f.part_rs[((int64_t)g * f.kmax) * PV_RS_W + tid] = (rr[0] + rr[8]) + (rr[16] + rr[24]);      // (slot 0 of the sample; the others stay zero)
      }
      if (f.dzc_out) {
        // dL/dz[i] = sum_j dL/d(hz[j]) Wz[j][i]: threads 0 .. 127 hold dL/d(hz[j]) (waves 0, 1)
        float* dzp = dh + FB_WAVES * FD_H + 8 * FB_WAVES;
        if (wave < 2) {
          for (int i = 0; i < f.lat_in; ++i) {
            const float pz = pv_wave_sum(dhz_j * (i < 4 ? wzv[i] : f.Wz[(int64_t)tid * f.lat_in + i]));
            if (lane == 0) dzp[16 * wave + i] = pz;
          }
        }
        __syncthreads();
        if (tid < f.lat_in) f.dzc_out[(int64_t)g * f.lat_in + tid] = dzp[tid] + dzp[16 + tid];
      }
      if (own_chain) {
        // ---- the sample's latent backward + encoder chain (pv_sdec_fused_w8.hip's epilogue, for this kernel's four waves):
        //   dhead from the row sums and dL/dz;  edp1 = (dhead Whead) * act'(eact1);  edp0 = (edp1 W1) * act'(eact0)
        // waves 0 and 1 run the head backward each for itself (wave-private LDS words, no barrier) and take 64 entries of edp1;
        // every thread contracts its 16 x 4 block of W1; threads 0 .. 127 add the 8 partial sums of their column.
        const float* rr = dh + FB_WAVES * FD_H;
        const float* dzp = dh + FB_WAVES * FD_H + 8 * FB_WAVES;
        float* cs = dh + FB_WAVES * FD_H + 8 * FB_WAVES + 64;        // [0..127] two waves' words, [128..255] edp1, [256 + 128 jg ..] partials
        if (wave < 2) {
          float* csw = cs + 64 * wave;
          if (lane < 5) {
            const float v = (rr[lane] + rr[8 + lane]) + (rr[16 + lane] + rr[24 + lane]);
            csw[lane] = v;
            if (tid == 0) ep_->llb[g] = v;
          }
          if (lane >= 8 && lane < 8 + f.lat_in) csw[lane] = dzp[lane - 8] + dzp[16 + lane - 8];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          PvHeadBwd hb{};
          hb.coord_dim = ep_->coord_dim; hb.has_r = ep_->has_r; hb.has_t = ep_->has_t; hb.has_s = ep_->has_s;
          hb.tp0 = ep_->tp0; hb.tp1 = ep_->tp1; hb.sc_prior = ep_->sc_prior;
          if (lane < 16) {
            if (lane < ep_->z_dim) {
              float g_, ds_;
              const float dz = pv_head_dz(hb, lane, [&](int c) { return csw[1 + c]; }, [&](int k) { return csw[8 + k]; });
              pv_head_bwd_math(dz, ch_z, ch_sig, ch_ep, ch_sp, ep_->beta, 0, g_, ds_);
              if (wave == 0) {
                ep_->dhead[(int64_t)g * ep_->ldh + lane] = g_;
                ep_->dhead[(int64_t)g * ep_->ldh + ep_->z_dim + lane] = ds_;
              }
              csw[16 + lane] = g_;
              csw[16 + ep_->z_dim + lane] = ds_;
            } else if (lane + ep_->z_dim < 16) {
              csw[16 + ep_->z_dim + lane] = 0.0f;                      // (entries 2 z_dim .. 15)
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          float v = 0.0f;
#pragma unroll
          for (int o = 0; o < 16; ++o) v += csw[16 + o] * ch_whd[o];
          v *= pv_act_grad2(ch_a1, 0.0f, ep_->enc1.act);
          ep_->edp1[(int64_t)g * FD_H + tid] = v;
          cs[128 + tid] = v;
        }
        __syncthreads();
        {
          const int c = tid & 31, jg = tid >> 5;
          f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const f32x4 ev = *reinterpret_cast<const f32x4*>(cs + 128 + 16 * jg + 4 * u);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += ch_w1[4 * u + i] * ev[i];
          }
          *reinterpret_cast<f32x4*>(cs + 256 + 128 * jg + 4 * c) = acc;
        }
        __syncthreads();
        if (tid < FD_H) {
          float y = 0.0f;
#pragma unroll
          for (int jg = 0; jg < 8; ++jg) y += cs[256 + 128 * jg + tid];
          y *= pv_act_grad2(ch_a0, 0.0f, ep_->enc0.act);
          ep_->edp0[(int64_t)g * FD_H + tid] = y;
        }
      }
    }
  }
  if (tid == 0) {
    float v = 0.0f;
    for (int w = 0; w < FB_WAVES; ++w) v += red[w];
    rec[2 * FD_H * FD_H + 5 * FD_H] = v;
  }
#ifdef FB_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  FB_KSTAMP(3);                                                          // record written
}

extern "C" int pv_debug_read_trace(long long* out, int n) {
  if (n > 256) n = 256;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fb_trace), n * sizeof(long long));
}

// Kernel choice.  Plain bf16 (fused = 3): the 8-wave kernel (pv_sdec_fused_w8.hip) when a workgroup gets at least ~6 of the
// 8 units a tile takes (batch >= ~32 at 28x28; below that most of a 128-row tile would idle and the 4-wave / 64-row kernel
// here is faster: 26 vs 30 us at batch 16).  Split precision (fused = 2): training launches run this file's 4-wave kernel;
// forward-only launches of enough units (decode, evaluate) the 8-wave build of pv_sdec_fused_w8x3.hip (round 3: 6.3 vs 5.0 M
// decoded images/s).  That source's training builds — 8 waves: 245 us, 4 waves: 188 us at batch 256, against 190 us here —
// stay selectable for A/B runs and are parity-tested (tests/test_gpu_parity.py: force_w8x3).
// Round 4: training launches of the fp32-class path run this file's fp16 builds — weights as two exact scaled pieces,
// activations one piece — once the rows a gradient sums are enough for the one-piece operands' independent rounding errors to
// average out (error table: profiles/r04a_fb_prec_table.txt, bars: every gradient tensor 1e-4 vs the oracle):
//   >= FB_H231_MIN_UNITS units (16 384 rows):   H231 (kind 28: dL/dpre split in both dgrads; worst tensor 1.6e-5 at C2)
//   >= FB_H221_MIN_UNITS units (524 288 rows):  H221 (kind 21: one piece everywhere; 6e-5 at C2's 2e5 rows, ~1 / sqrt(rows):
//                                               round 5 moved the gate from 2 M rows to where the model predicts <= 4e-5 and three
//                                               draws of the 64x64 config measure 4.3e-5 ... 5.1e-5 on the worst tensor —
//                                               profiles/r05u_grad_margin_h221_*.txt — C4's decoder launch 357 -> 327 us)
// smaller problems keep the bf16 three-product kernel (kind 0).
// Which build runs is a function of (fused mode, units, launch kind) and of pv_ivae_plan.dec_kernel (`sel`, 0 = by size: ABI v15,
// a PLAN field — no process-wide state).  sel, plain (fused 3): 1 the 4-wave kernel, 2 the 8-wave kernel.  sel, split precision
// (fused 2): 1 the bf16 three-product kernel (kind 0), 21 / 28 the fp16 builds; the experiments build (-DPV_EXPERIMENTS) also
// knows 4 / 8 (pv_sdec_fused_w8x3.hip's training forms), 23 / 31 / 33 / 26 / 27 / 29 (the error table's other corners), 41 / 48
// (pv_sdec_fused_w8h.hip) and, with sel == 0, the environment: PV_W8=0 / 1, PV_X3_KERNEL=old | 4 | 8 | h221 | h223 | ... | w221 | w231.
#ifndef FB_H231_MIN_UNITS
#define FB_H231_MIN_UNITS 1024         // 16 384 rows
#endif
#ifndef FB_H221_MIN_UNITS
#define FB_H221_MIN_UNITS 32768        // 524 288 rows (round 5; rounds 4: 131072 = 2 097 152 rows)
#endif
#ifndef FB_EXPERIMENTS
#ifdef PV_EXPERIMENTS
#define FB_EXPERIMENTS 1               // the error table's other corners (H223 / H321 / H333; Bernoulli training launches only)
#else
#define FB_EXPERIMENTS 0
#endif
#endif
static const int64_t fb_row_cap = (int64_t)1 << 30;          // (the pv_sdec_fused_w8*.hip kernels address rows by 32-bit BYTE offsets)
#ifdef PV_EXPERIMENTS
static int fb_env_plain() {
  static const int v = [] { const char* e = pv_exp_str("PV_W8"); return e ? (atoi(e) == 0 ? 0 : 1) : 2; }();
  return v;
}
static int fb_env_x3() {
  static const int v = [] {
    const char* e = pv_exp_str("PV_X3_KERNEL");
    return !e ? 2 : (e[0] == 'w' ? (atoi(e + 1) == 231 ? 48 : 41) : e[0] == 'o' ? 0 : (e[0] == 'h' ? (atoi(e + 1) == 221 ? 21 : atoi(e + 1) == 223 ? 23 : atoi(e + 1) == 321 ? 31 : atoi(e + 1) == 333 ? 33 : atoi(e + 1) == 231 ? 28 : 2)
                                                    : (atoi(e) == 8 ? 8 : (atoi(e) == 4 ? 4 : 2))));
  }();
  return v;
}
#else
static constexpr int fb_env_plain() { return 2; }
static constexpr int fb_env_x3() { return 2; }
#endif
bool pv_sdec_fused_sel_valid(int fused, int sel) {
  if (sel == 0) return true;
  if (fused == 3) return sel == 1 || sel == 2;
  if (fused != 2) return false;
  if (sel == 1 || sel == 21 || sel == 28) return true;
#ifdef PV_EXPERIMENTS
  return sel == 4 || sel == 8 || sel == 23 || sel == 31 || sel == 33 || sel == 26 || sel == 27 || sel == 29 || sel == 41 || sel == 48;
#else
  return false;
#endif
}
static bool fb_use_w8(int64_t units, int sel) {
  const int v = sel == 1 ? 0 : (sel == 2 ? 1 : fb_env_plain());          // 0 / 1 forced, 2 by problem size
  if (v != 2) return v != 0 && units * FD_UNIT < fb_row_cap;
  return units >= 6 * (int64_t)pv_sdec_fused_grid(units) && units * FD_UNIT < fb_row_cap;
}
// split precision: 0 = this file's 4-wave bf16 kernel, 21 (23 / 31 / 33) = its fp16 builds, 4 / 8 = pv_sdec_fused_w8x3.hip with
// that many waves
static int fb_x3_kind(int64_t units, bool grads, int sel) {
  const int v = sel == 0 ? fb_env_x3() : (sel == 1 ? 0 : sel);
  if (v > 8) return grads ? v : (units * FD_UNIT < fb_row_cap && units >= 6 * (int64_t)pv_sdec_fused_grid(units) ? 8 : 0);
  if (units * FD_UNIT >= fb_row_cap) return 0;
  if (v != 2) return v;
  if (grads && units >= FB_H221_MIN_UNITS) return 21;
  if (grads && units >= FB_H231_MIN_UNITS) return 28;
  // training: this file's kernel.  The new source's 4-wave build measures the same (187.9 vs 189.7 us at batch 256 with every
  // offload and the epilogues folded into the consuming k-loops: profiles/r03_decoder_schedule_experiments.txt) — not worth a
  // switch of the two-rounds-tested default; its 8-wave build is slower (245 us).  Forward-only: the 8-wave build by size.
  if (grads) return 0;
  return units >= 6 * (int64_t)pv_sdec_fused_grid(units) ? 8 : 0;
}
static bool fb_kind_here(int kind) { return kind == 0 || (kind > 8 && kind < 40); }   // kinds this file's 4-wave kernel serves
static bool fb_kind_w8h(int kind) { return kind == 41 || kind == 48; }   // pv_sdec_fused_w8h.hip: H221 / H231 arithmetic, 8 waves
static int fb_kind_prec(int kind) {
  return kind == 21 ? FB_P_H221 : kind == 23 ? FB_P_H223 : kind == 31 ? FB_P_H321 : kind == 33 ? FB_P_H333
       : kind == 26 ? FB_P_H2A1 : kind == 27 ? FB_P_H2B1 : kind == 28 ? FB_P_H231 : kind == 29 ? FB_P_H131 : FB_P_X3;
}
int pv_sdec_fused_bf16_waves(bool x3, int64_t units, int sel) {
  const int kind = x3 ? fb_x3_kind(units, true, sel) : 0;
  return (x3 ? (kind == 8 || fb_kind_w8h(kind)) : fb_use_w8(units, sel)) ? 8 : FB_WAVES;
}
int pv_sdec_fused_bf16_record_fmt(bool x3, int64_t units, int sel) {
  static const int ablate = pv_exp_int("PV_FD_ABLATE", 0);          // (experiments build, bit 1024: the row-major fp32 form from both kernels)
  if (ablate & 1024) return PV_REC_ROWMAJOR;
  if (!x3) return fb_use_w8(units, sel) ? PV_REC_LANE_BF16 : PV_REC_LANE_F32;
  return fb_kind_here(fb_x3_kind(units, true, sel)) ? PV_REC_LANE_F32 : PV_REC_ROWMAJOR;   // (pv_sdec_fused_w8x3 / w8h.hip: row-major)
}
int64_t pv_sdec_fused_bf16_park_bytes(bool x3, int64_t units, int grid, int sel) {
#ifdef PV_EXPERIMENTS
  return (x3 && fb_x3_kind(units, true, sel) == 8) ? pv_sdec_fused_w8x3_park_bytes(grid) : 0;
#else
  (void)x3; (void)units; (void)grid; (void)sel;
  return 0;                                             // (parking slots: the 8-wave split-precision TRAINING form only)
#endif
}

// measurement hook (not in include/): the kernel a decoder launch of (fused mode, units, grads, likelihood) dispatches, spelled
// the way rocprofv3 prints it — bench.py puts it in `roofline.kernel` so that the line can be matched against
// profiles/*_kernel_stats.csv by name
extern "C" const char* pv_debug_decoder_kernel_name_fold(int grads, int lik) {      // (the launch that hosts the guide: pv_ivae_guide_folds)
  static thread_local char buf[128];
  // (training launches of the plain iVAE step also host the latent backward: build 2, pv_plan.hip's own_chain)
  snprintf(buf, sizeof buf, "void pv_sdec_w8_kernel<%s, %d, %d>(PvFused, PvEncFold)", grads ? "true" : "false", lik, grads ? 2 : 1);
  return buf;
}
extern "C" const char* pv_debug_decoder_kernel_name(int fused, int64_t units, int grads, int lik) {
  static thread_local char buf[128];
  const char* g = grads ? "true" : "false";
  if (fused == 1) snprintf(buf, sizeof buf, "void pv_sdec_fused_kernel<%s>(PvFused)", g);
  else if (fused == 2 && fb_kind_w8h(fb_x3_kind(units, grads != 0, 0))) snprintf(buf, sizeof buf, "void pv_sdec_w8h_kernel<%d, %s>(PvFused)", lik, fb_x3_kind(units, grads != 0, 0) == 48 ? "true" : "false");
  else if (fused == 2 && !fb_kind_here(fb_x3_kind(units, grads != 0, 0))) snprintf(buf, sizeof buf, "void pv_sdec_w8x3_kernel<%s, %d, %d>(PvFused)", g, lik, fb_x3_kind(units, grads != 0, 0));
  else if (fused == 3 && fb_use_w8(units, 0)) snprintf(buf, sizeof buf, "void pv_sdec_w8_kernel<%s, %d, 0>(PvFused, PvEncFold)", g, lik);
  else if (fused >= 2) snprintf(buf, sizeof buf, "void pv_sdec_fused_bf16_kernel<%s, %d, %d>(PvFused, PvEncFold)", g, lik, fused == 2 ? fb_kind_prec(fb_x3_kind(units, grads != 0, 0)) : FB_P_BF16);
  else snprintf(buf, sizeof buf, "pv_gemm_kernel (layer-by-layer path)");
  return buf;
}

// which column order a launch's weight images are in (pv_fb_layout.h): the 8-wave plain-bf16 kernel reads the q-swapped one
// (-0.6 % of its step); this file's 4-wave kernel the plain one — its transposing reads lose 80 % of their bank conflicts too
// (SQ_LDS_BANK_CONFLICT 8.0 M -> 1.6 M per launch, LDS-array cycles -19 %) and the launch does not move (144.3 vs 144.8 us:
// profiles/r06i_qswap_ab.txt), so it keeps the form without the operand selects.  PV_QSWAP=1 (experiments build) turns it on.
static int fb_qswap_on() {
  static const int v = pv_exp_int("PV_QSWAP", 0);
  return v != 0;
}
PvFbPrep pv_sdec_fused_bf16_prep_args(const PvFused& f, bool grads, bool x3) {
  static_assert(FB_WIMG_BYTES >= FB_SCALE_OFF + 16, "pv_sdec_fused.h and the LDS image layout disagree");
  PvFbPrep p{};
  p.W1 = f.W1; p.W2 = f.W2; p.img = f.wimg; p.zero = f.part_hz; p.wo = f.wo;
  p.nzero4 = grads ? (int64_t)f.B * f.kmax * (FD_H + (f.part_rs ? PV_RS_W : 0)) / 4 : 0;      // (dL/d(hz) slots + the row-sum slots behind them)
  const int kind = x3 ? fb_x3_kind(f.units, grads, f.sel) : -1;
  p.mode = fb_kind_w8h(kind) ? 2 : (kind > 8 ? 1 : 0);
  p.scale = (x3 ? !fb_kind_here(kind) : fb_use_w8(f.units, f.sel)) ? 2.8853900817779268f : 0.0f;     // (also what hz arrives multiplied by)
  p.qswap = p.scale == 0.0f ? fb_qswap_on() : (!x3 && pv_sdec_fused_w8_qswap());
  return p;
}

int pv_sdec_fused_bf16_prep(const PvFused& f, bool grads, bool x3, hipStream_t s) {
  if (!f.wimg) return PV_EINVAL;
  const PvFbPrep p = pv_sdec_fused_bf16_prep_args(f, grads, x3);
  const int64_t work = p.nzero4 > FD_H * (FD_H / 4) ? p.nzero4 : FD_H * (FD_H / 4);
  int blocks = (int)((work + 255) / 256);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(pv_fb_prep_kernel, dim3(blocks), dim3(256), 0, s, p);
  PV_LAUNCH_CHECK();
  return 0;
}

bool pv_sdec_fused_fold_ok(const PvFused& f, int grid, bool x3) {
  return !x3 && fb_use_w8(f.units, f.sel) && pv_sdec_fused_w8_fold_ok(f, grid);
}

int pv_sdec_fused_bf16_launch(const PvFused& f_in, int grid, bool grads, bool x3, hipStream_t s, const PvEncFold* fold,
                              const PvEncFold* chain) {
  if (fold && (x3 || !fb_use_w8(f_in.units, f_in.sel))) return PV_EINVAL;      // (the folded guide lives in the 8-wave plain-bf16 kernel)
  int prec = FB_P_BF16;
  if (x3) {
    const int kind = fb_x3_kind(f_in.units, grads, f_in.sel);
#ifdef PV_EXPERIMENTS
    if (fb_kind_w8h(kind)) return grads ? pv_sdec_fused_w8h_launch(f_in, grid, kind == 48, s) : PV_EINVAL;
#else
    if (fb_kind_w8h(kind) || (grads && !fb_kind_here(kind))) return PV_EINVAL;   // (dropped variants: the experiments build)
#endif
    if (!fb_kind_here(kind)) return pv_sdec_fused_w8x3_launch(f_in, grid, grads, s, kind);
    prec = fb_kind_prec(kind);
  } else if (fb_use_w8(f_in.units, f_in.sel)) {
    return pv_sdec_fused_w8_launch(f_in, grid, grads, s, fold);
  }
  PvFused f = f_in;
  static const int ablate = pv_exp_int("PV_FD_ABLATE", 0);
  f.ablate = ablate;
  f.qswap = fb_qswap_on();
  const size_t lds = FB_LDS_BYTES;
  const void* fn = nullptr;
#define FB_PICK_P(G, L, P) fn = reinterpret_cast<const void*>(&pv_sdec_fused_bf16_kernel<G, L, P>)
#define FB_PICK(G, L)                                                                          \
  do {                                                                                         \
    if (prec == FB_P_BF16) FB_PICK_P(G, L, FB_P_BF16);                                         \
    else FB_PICK_P(G, L, FB_P_X3);                                                             \
  } while (0)
#define FB_PICK_G(L)                                                                           \
  do {                                                                                         \
    if (prec == FB_P_H231) FB_PICK_P(true, L, FB_P_H231);                                      \
    else if (prec == FB_P_H221) FB_PICK_P(true, L, FB_P_H221);                                 \
    else FB_PICK(true, L);                                                                     \
  } while (0)
  if (prec > FB_P_H221 && prec != FB_P_H231 && !(FB_EXPERIMENTS && grads && f.lik == PV_LIK_BERNOULLI)) return PV_EINVAL;
  if (prec >= FB_P_H221 && !grads) return PV_EINVAL;          // (forward-only launches never select the fp16 builds)
  if (grads) {
    if (f.lik == PV_LIK_BERNOULLI) {
      FB_PICK_G(PV_LIK_BERNOULLI);
#if FB_EXPERIMENTS
      if (prec == FB_P_H223) FB_PICK_P(true, PV_LIK_BERNOULLI, FB_P_H223);
      else if (prec == FB_P_H321) FB_PICK_P(true, PV_LIK_BERNOULLI, FB_P_H321);
      else if (prec == FB_P_H333) FB_PICK_P(true, PV_LIK_BERNOULLI, FB_P_H333);
      else if (prec == FB_P_H2A1) FB_PICK_P(true, PV_LIK_BERNOULLI, FB_P_H2A1);
      else if (prec == FB_P_H2B1) FB_PICK_P(true, PV_LIK_BERNOULLI, FB_P_H2B1);
      else if (prec == FB_P_H131) FB_PICK_P(true, PV_LIK_BERNOULLI, FB_P_H131);
#endif
    }
    else if (f.lik == PV_LIK_GAUSSIAN) FB_PICK_G(PV_LIK_GAUSSIAN);
    else FB_PICK_G(PV_LIK_CBERNOULLI);
  } else {
    if (f.lik == PV_LIK_BERNOULLI) FB_PICK(false, PV_LIK_BERNOULLI);
    else if (f.lik == PV_LIK_GAUSSIAN) FB_PICK(false, PV_LIK_GAUSSIAN);
    else FB_PICK(false, PV_LIK_CBERNOULLI);
  }
#undef FB_PICK_G
#undef FB_PICK
#undef FB_PICK_P
  // (per device: a process may drive several; idempotent: a race between host threads only repeats the call)
  PV_TRY(pv_set_dynamic_lds(fn, (int)lds));          // (per device and kernel)
  PvEncFold ec{};                                   // (chain: the own-sample epilogue's latent backward, PvEncFold::chain)
  if (chain && grads) ec = *chain;
  void* args[] = {&f, &ec};
  hipError_t e2 = hipLaunchKernel(fn, dim3(grid), dim3(FB_THREADS), args, lds, s);
  if (e2 != hipSuccess) return (int)e2;
  PV_LAUNCH_CHECK();
  return 0;
}
