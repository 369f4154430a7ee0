// pv_sdec_fused_w8.hip — the fused persistent spatial-decoder forward+backward kernel, plain-bf16 operands
// (SVItrainer(precision="bf16"), plan.fused = 3), re-cut for TWO waves per SIMD.
//
// Why.  The 4-wave form of this kernel (pv_sdec_fused_bf16.hip, X3 = false) measured 14 % of the bf16 MFMA peak with the
// matrix pipe busy 16 % of the time: one wave per SIMD issues about one instruction every 4-5 cycles, and a 64-row tile
// is ~3.2 k instructions per wave (1.7 k of them VALU) — the tile was ISSUE bound, with nothing to hide fill / drain
// latencies behind.  This form
//   * runs 8 waves per workgroup (512 threads, one workgroup per CU, two waves per SIMD, <= 256 registers each), a tile
//     is 128 rows, each wave carries one 16-row unit and owns ONE 16-row slice of dW1 / dW2 in accumulators;
//   * moves every small contraction that sat on the VALU onto the matrix cores, which have slack:
//       - the coordinate layer (K = 2 plus bias) is one v_mfma_f32_16x16x16_bf16 per output block with hi/lo split
//         operands (w x ~= wh xh + wh xl + wl xh), the per-sample fc_latent term enters as the accumulator's initial value;
//       - its row-local input gradient (d0, d1 = dpre0 . Wc[:, 0 / 1]) is a 4-MFMA "dgrad" against a 16-row table
//         [Wc0 hi; Wc0 lo; Wc1 hi; Wc1 lo; 0 ...];
//       - the column sums over a unit's rows — d(wo) = sum dlda h2, dL/d(hz) = sum dpre0, dWc_k = sum dpre0 x'_k — are
//         wave-local MFMAs: the wave writes its 16 x 128 bf16 tile into its OWN rows of the staging area, reads it back
//         transposed (ds_read_b64_tr_b16) as the A operand and contracts against B = [1 | x0 | x1 | dlda (hi / lo columns)];
//         one 32-register accumulator holds all four sums for the whole kernel (dL/d(hz) is flushed per sample);
//   * drops the multiply of every tanh: the weight images, the biases and the coordinate layer's operands are stored
//     pre-scaled by c = 2 log2(e), so the MFMAs deliver c * pre-activation and tanh = 1 - 2 rcp(exp2(.) + 1) is four
//     instructions; the backward pass then carries c * dpre1 and c^2 * dpre0 and un-scales where a gradient leaves the
//     kernel (dW1, db1, dWc, dL/d(hz), the per-row transform gradients).
// Layout, row -> lane mapping, weight images, staging swizzles and the per-workgroup gradient record are those of
// pv_sdec_fused_bf16.hip (pv_fb_layout.h), so the rest of the step is unchanged.
//
// LDS (158,976 B): W1 | W2 images (bf16, 32 KB each) | staging A | staging B (128 rows x 144 bf16 each, shared by the
// two wgrad rounds of a tile) | wo, c b1, c b2 | coordinate-layer A table | row-local dgrad A table | per-row scalars
// | per-wave prefetch slots.  Four workgroup barriers per 128-row tile.
#include "pv_sdec_fused.h"
#include "pv_fb_layout.h"
#include "pv_kernels.h"        // PvHeadBwd, pv_head_dz / pv_head_bwd_math: the image's latent backward in the hosting launch's epilogue
#include <stdlib.h>

typedef short short4_ __attribute__((ext_vector_type(4)));
typedef short short8_ __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) short4_ lds_short4;

#define W8_WAVES 8
#define W8_ROWS (W8_WAVES * FD_UNIT)       // 128
#define W8_THREADS (64 * W8_WAVES)
#define LDS2 144                           // staging rows: 72 dwords -> conflict-free 4x16 transposing reads
#define W8_ARR (W8_ROWS * LDS2)            // elements of one staging array
#define W8_ARR_BYTES (2 * W8_ARR)          // 36,864
#define WO_W1 0
#define WO_W2 IMG_BYTES
#define WO_SA (2 * IMG_BYTES)
#define WO_SB (WO_SA + W8_ARR_BYTES)
#define WO_VEC (WO_SB + W8_ARR_BYTES)      // fp32: wo[128], c*b1[128], c*b2[128]
#define WO_ATAB (WO_VEC + 3 * FD_H * 4)    // coordinate layer, A operands: 8 blocks x 64 lanes x bf16x4
#define WO_TTAB (WO_ATAB + 8 * 64 * 8)     // row-local dgrad, A operands: 4 k-blocks x 64 lanes x bf16x8
#define WO_INFO (WO_TTAB + 4 * 64 * 16)    // per row of the tile: x0[128], x1[128], dlda[128]
#define WO_RED (WO_INFO + 3 * W8_ROWS * 4)
#define WO_CHZ (WO_RED + 256)              // next tile's per-unit inputs by LDS-DMA: hz[b] (128 floats) per wave
#define WO_CTP (WO_CHZ + W8_WAVES * FD_H * 4)
#define WO_CGR (WO_CTP + W8_WAVES * 256)
#define W8_LDS_BYTES (WO_CGR + W8_WAVES * 256)
// column-parallel tail (a workgroup's last tile when it holds ONE unit): the exchange buffers live in the rows of staging B
// that a one-k-step consume never reads (rows 32 ..): six 4 KB piece buffers [wave][lane] of bf16x4 + the partial logits
#define WO_TX (WO_SB + 2 * 32 * LDS2)
#define W8_TX_BYTES (W8_WAVES * 64 * 8)
#define WO_TLOG (WO_TX + 6 * W8_TX_BYTES)
static_assert(WO_TLOG + W8_WAVES * 16 * 4 <= WO_SB + W8_ARR_BYTES, "tail exchange buffers fit behind the staged rows");
static_assert(W8_LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(2 * IMG_BYTES % (W8_WAVES * 1024) == 0, "image load: whole 1 KB LDS-DMA pieces per wave");

#define W8_C 2.8853900817779268f           // 2 log2(e): tanh(x) = 1 - 2 / (exp2(C x) + 1)
#define W8_RC (1.0f / W8_C)
#define W8_RC2 (W8_RC * W8_RC)
#define LOG_SQRT_2PI 0.91893853320467274178f
#define BERN_EPS 1.1920928955078125e-07f
#ifndef W8_FENCE_MASK
#define W8_FENCE_MASK 0          // __builtin_amdgcn_sched_barrier's mask: 0 = nothing crosses a stage fence
#endif
#define W8_FENCE() __builtin_amdgcn_sched_barrier(W8_FENCE_MASK)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 w8_mfma16(const bf16x4& a, const bf16x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4_, a), __builtin_bit_cast(short4_, b), c, 0, 0, 0);
}
// tanh of x given C*x
__device__ __forceinline__ float w8_tanhc(float cx) {
  const float e = __builtin_amdgcn_exp2f(cx);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float w8_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float w8_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float w8_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#ifndef W8_HSWAP
#define W8_HSWAP 0            // (A/B build only)
#endif
#ifndef W8_QSWAP
#define W8_QSWAP 1            // the weight images' column order (below, at w8_addr)
#endif
bool pv_sdec_fused_w8_qswap() { return W8_QSWAP != 0; }
__device__ __forceinline__ bf16x8 w8_cat(const bf16x4& a, const bf16x4& b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 w8_catq(const bf16x4& a, const bf16x4& b, int q) {
#if W8_QSWAP
  typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
  const bool sw = q >= 2;
  const u32x2_ ua = __builtin_bit_cast(u32x2_, a), ub = __builtin_bit_cast(u32x2_, b);
  const u32x2_ lo = {sw ? ub[0] : ua[0], sw ? ub[1] : ua[1]}, hi = {sw ? ua[0] : ub[0], sw ? ua[1] : ub[1]};
  return w8_cat(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi));
#else
  (void)q;
  return w8_cat(a, b);
#endif
}
__device__ __forceinline__ int w8_opaque0() { int z = 0; asm volatile("" : "+v"(z)); return z; }
__device__ __forceinline__ bf16x4 w8_tr(const __bf16* p) {
  const short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)p);
  return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ bf16x4 w8_zero4() { const short4_ z = {0, 0, 0, 0}; return __builtin_bit_cast(bf16x4, z); }
// LDS-DMA (see pv_sdec_fused_bf16.hip: not in hipcc's waitcnt bookkeeping; drain explicitly)
__device__ __forceinline__ void w8_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void w8_glds4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void w8_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void w8_wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }
__device__ __forceinline__ float w8_sum_q(float v) {
  return pv_sum_rows(v);                             // (pv_common.h: v_permlane16/32_swap, the bits of the two shfl_xor sums)
}

// timing ablations (profiling builds only, results are WRONG; profiles/r03e_w8_ablations.txt): -DW8_ABL=1 the forward loops
// read every second weight group from LDS and use it twice, 2 the same in the dgrad loops, 4 the wgrad consume reads every
// second k-step, 8 no dW1 / dW2 record stores
#ifndef W8_ABL
#define W8_ABL 0
#endif
#ifndef W8_TAIL
#define W8_TAIL 1                    // 0: the seventh tile row-parallel as in rounds 2-4 (A/B builds)
#endif
#define W8_ABL_LOADS(bit, g) (!(W8_ABL & (bit)) || (((g) & 1) == 0))
#define W8_ABL_BUF(bit, g) ((W8_ABL & (bit)) ? (((g) >> 1) & 1) : ((g) & 1))

// forward layer of the wave's unit: out = bias + W in, both pre-scaled by C (C * pre-activation on return)
// lane offsets (elements) of the weight reads, computed once per kernel and kept in registers (10 of them):
//   forward: row 16*ob + r, logical chunk 4m + q -> physical chunk 4*(m ^ (r&3)) + (q ^ SL[r>>2])   (pv_fb_layout.h)
//   dgrad  : lane i of 16-lane group q points at W[32m + 4q (+16) + r/4][...], swizzle 4*(r>>2) + SL[q]
// W8_HSWAP (A/B build only; profiles/r06*_lds_swizzle_ab.txt): alternate 4-row groups of an image keep the two
// 8-byte halves of every 16-byte chunk SWAPPED.  The dgrad's transposing 8-byte reads — 32 lanes = 8 rows x 4 chunks, all of
// them wanting the same half: 2-way on every read in the shipped layout — then cover both halves of 16 chunks: conflict-free.
// The price is the forward's operand read: its 8 k-values are no longer one ds_read_b128 but two ds_read_b64 at lane-dependent
// halves (fs / hs below: which half comes first for this lane's rows).  Valid only where the kernel builds its own images (FOLD).
// W8_QSWAP (the shipped form since round 6's second cut; profiles/r06i_qswap_ab.txt): the q-swapped column order of
// pv_fb_layout.h — the half a column block sits in is h ^ (q >> 1).  The transposing reads are conflict-free as under W8_HSWAP, and
// the forward keeps its ONE ds_read_b128: lanes of groups q >= 2 feed the activation pieces in the swapped order instead (w8_catq:
// 4 v_cndmask per k-block).  Images written by the kernel itself (FOLD) and by pv_fb_prep (PvFbPrep::qswap) alike.

struct W8Addr { int fb, fx[4], db, dx[4], fs, hs; };
__device__ __forceinline__ W8Addr w8_addr(int r, int q) {
  W8Addr a;
  a.fs = W8_HSWAP ? ((r >> 2) & 1) : 0;               // forward: rows 16 ob + r
  a.hs = W8_HSWAP ? (q & 1) : (W8_QSWAP ? ((r >> 1) & 1) : 0);   // dgrad: rows 32 m + 4 q (+ 16) + r / 4, piece r & 3
  a.fb = r * LDB + 8 * (q ^ fb_sl(r >> 2));
  a.db = (4 * q + (r >> 2)) * LDB + 8 * ((r & 3) ^ fb_sl(q));
#pragma unroll
  for (int m = 0; m < 4; ++m) { a.fx[m] = 32 * (m ^ (r & 3)); a.dx[m] = 32 * (m ^ (r >> 2)); }
  return a;
}

__device__ __forceinline__ void w8_layer_fwd(const __bf16* __restrict__ Wh, const float* __restrict__ bs,
                                             const bf16x4 (&ih)[8], f32x4 (&out)[8], const W8Addr& ad, int q) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) out[ob] = *reinterpret_cast<const f32x4*>(bs + 16 * ob + 4 * q);
  const __bf16* ah = Wh + ad.fb;
#if W8_HSWAP
  const __bf16* ah0 = ah + 4 * ad.fs;
  const __bf16* ah1 = ah + 4 - 4 * ad.fs;
  asm volatile("" : "+v"(ah0), "+v"(ah1));
#endif
  const int (&xm)[4] = ad.fx;
  bf16x8 wh[2][2];
  auto load = [&](int g, bf16x8 (&h)[2]) {
    const int m = g >> 2, op = (g & 3) * 2;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
#if W8_HSWAP
      // (two base registers: from ONE base the compiler merges the pair into a ds_read2_b64, whose banking — mod 32 over 16-lane
      //  groups — conflicts where two ds_read_b64 do not: first cut of this experiment, conflict ratio 0.216 -> 0.283)
      // (volatile: the compiler also pairs the reads of the two output blocks into ds_read2st64_b64 — the same half-rate form)
      typedef const volatile bf16x4* vp4;
      const bf16x4 lo4 = *reinterpret_cast<vp4>(ah0 + 16 * (op + o) * LDB + xm[m]);
      const bf16x4 hi4 = *reinterpret_cast<vp4>(ah1 + 16 * (op + o) * LDB + xm[m]);
      h[o] = w8_cat(lo4, hi4);
#else
      h[o] = *reinterpret_cast<const bf16x8*>(ah + 16 * (op + o) * LDB + xm[m]);
#endif
    }
  };
  load(0, wh[0]);
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int m = g >> 2, op = (g & 3) * 2;
    if (g + 1 < 16 && W8_ABL_LOADS(1, g + 1)) load(g + 1, wh[W8_ABL_BUF(1, g + 1)]);
    W8_FENCE();
    const bf16x8 bh = w8_catq(ih[2 * m], ih[2 * m + 1], q);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[op + o] = MFMA32(wh[W8_ABL_BUF(1, g)][o], bh, out[op + o]);
    W8_FENCE();
  }
}

// dgrad of the wave's unit: out[k] = sum_j (C W)[j][k] dp[j]; A = W^T via the transposing LDS read
__device__ __forceinline__ void w8_layer_dgrad(const __bf16* __restrict__ Wh, const bf16x4 (&ih)[8], f32x4 (&out)[8],
                                               const W8Addr& ad) {
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) out[kb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const __bf16* ah = Wh + ad.db;
  const int (&xk)[4] = ad.dx;
  bf16x8 wh[2][2];
  auto load = [&](int g, bf16x8 (&h)[2]) {
    const int m = g >> 2, kp = (g & 3) * 2;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int off = 32 * m * LDB + xk[(kp + o) >> 1] + (((kp + o) & 1) ? 4 - 4 * ad.hs : 4 * ad.hs);
      h[o] = w8_cat(w8_tr(ah + off), w8_tr(ah + off + 16 * LDB));
    }
  };
  load(0, wh[0]);
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int m = g >> 2, kp = (g & 3) * 2;
    if (g + 1 < 16 && W8_ABL_LOADS(2, g + 1)) load(g + 1, wh[W8_ABL_BUF(2, g + 1)]);
    W8_FENCE();
    const bf16x8 bh = w8_cat(ih[2 * m], ih[2 * m + 1]);
#pragma unroll
    for (int o = 0; o < 2; ++o) out[kp + o] = MFMA32(wh[W8_ABL_BUF(2, g)][o], bh, out[kp + o]);
    W8_FENCE();
  }
}

// Elementwise phases are written as STAGES over all 32 values of a lane with scheduling fences in between: left alone,
// the compiler (at the 256-register limit) walks the values two at a time through the whole dependent chain
// (exp -> add -> rcp -> fma -> cvt), and an in-order wave then pays every instruction's latency (~10 cycles each).
// tanh of x given C*x, in place: 1 - 2 rcp(exp2(.) + 1)
__device__ __forceinline__ void w8_tanh8(f32x4 (&v)[8]) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[jb][i] = __builtin_amdgcn_exp2f(v[jb][i]);
  W8_FENCE();
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) v[jb] = v[jb] + 1.0f;
  W8_FENCE();
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[jb][i] = __builtin_amdgcn_rcpf(v[jb][i]);
  W8_FENCE();
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) v[jb] = 1.0f - 2.0f * v[jb];
  W8_FENCE();
}
__device__ __forceinline__ f32x4 w8_f32_of(const bf16x4& h) {
  typedef unsigned uint2_ __attribute__((ext_vector_type(2)));
  const uint2_ u = __builtin_bit_cast(uint2_, h);
  f32x4 f;
  f[0] = __builtin_bit_cast(float, u[0] << 16);
  f[1] = __builtin_bit_cast(float, u[0] & 0xffff0000u);
  f[2] = __builtin_bit_cast(float, u[1] << 16);
  f[3] = __builtin_bit_cast(float, u[1] & 0xffff0000u);
  return f;
}
// d *= h^2 - 1 = -(1 - h^2) with h saved as bf16, four blocks at a time.  The SIGN is deliberate: `1 - t t` costs a v_xor per value
// in front of the packed fma (its operand negation is not used by the compiler: 64 instructions per tile), `t t - 1` is the
// packed fma alone.  Everything downstream is linear, so after layer 2's call the kernel carries -C dpre1 (the wgrad of layer 1
// accumulates -dW1 / -db1: undone with the un-scaling where the record is written), and layer 1's call flips the sign back
// (C^2 dpre0 as before).  Negation commutes with every rounding on the way: bit-identical results.
__device__ __forceinline__ void w8_mul_dtanh(f32x4 (&d)[8], const bf16x4 (&hb)[8]) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 t[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) t[jb] = w8_f32_of(hb[4 * half + jb]);
    W8_FENCE();
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) t[jb] = t[jb] * t[jb] - 1.0f;
    W8_FENCE();
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) d[4 * half + jb] = d[4 * half + jb] * t[jb];
    W8_FENCE();
  }
}

__device__ __forceinline__ void w8_cvt8(const f32x4 (&v)[8], bf16x4 (&h)[8]) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) h[jb][i] = (__bf16)v[jb][i];
}
// the wave's 16 rows (row = 16 * wave + r) of a staged tensor, row-major [128][LDS2]; inside every 16-column block the
// four 8-byte pieces are XOR-swizzled by (row>>2)&3 (pv_sdec_fused_bf16.hip: fb_stage_store)
__device__ __forceinline__ void w8_stage_store(__bf16* __restrict__ sh, const bf16x4 (&h)[8], int row, int q) {
  row |= w8_opaque0();
  const int e = row * LDS2 + 4 * (q ^ ((row >> 2) & 3));
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) *reinterpret_cast<bf16x4*>(sh + e + 16 * jb) = h[jb];
}
// lane offset of the transposing read of staged rows R0 + 4q .. 4q+3 (R0 a multiple of 16), columns 16*blk ..
__device__ __forceinline__ int w8_stage_toff(int r, int q) { return (4 * q + (r >> 2)) * LDS2 + 4 * ((r & 3) ^ q); }

// wgrad over the staged tile.  Wave (jp = wave >> 1, kh = wave & 1) owns the 32 x 64 block dW[32jp .. +31][64kh .. +63]
// (2 A operands x 4 B operands per 32 staged rows: 12 transposing reads per 8 MFMAs; a 16 x 128 slice per wave needs
// 18) and the bias sums of rows 32jp + 16kh .. +15 (an MFMA against ones):
//   dW[j][k] += sum_rows dpre[row][j] h[row][k];   db[j] += sum_rows dpre[row][j]
__device__ __forceinline__ bf16x4 w8_tr_at(unsigned lds_byte_addr) {
  const short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(size_t)lds_byte_addr);
  return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ void w8_wgrad_consume(const __bf16* sa, const __bf16* sb, f32x4 (&accW)[2][4], f32x4& accB,
                                                 int wave, int r, int q, int ksteps) {
  const int toff = w8_stage_toff(r | w8_opaque0(), q);
  const int jp = wave >> 1, kh = wave & 1;
  const short one = 0x3f80;                           // bf16 1.0
  const short8_ ones_s = {one, one, one, one, one, one, one, one};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
  // Three LDS byte addresses carry everything that is not a compile-time constant (the arrays' bases — beyond the 16-bit
  // offset field of a ds_read — the lane's transposing-read offset, the wave's block); made opaque so that the twelve reads
  // of a k-step are base + immediate instead of a v_add_u32 each (21 address instructions per k-step before).
  // (the wave holds its two row blocks ROTATED by kh — a[s] / accW[s] are row block s ^ kh — so that its bias block is always
  //  operand 0: no wave-uniform select between register operands, 8 v_cndmask per k-step)
  unsigned la0 = (unsigned)(size_t)sa + 2u * (unsigned)(toff + 32 * jp + 16 * kh);
  unsigned la1 = (unsigned)(size_t)sa + 2u * (unsigned)(toff + 32 * jp + 16 * (1 ^ kh));
  unsigned lb = (unsigned)(size_t)sb + 2u * (unsigned)(toff + 64 * kh);
  asm volatile("" : "+v"(la0), "+v"(la1), "+v"(lb));
  constexpr unsigned ROW16 = 2u * 16 * LDS2;         // bytes of 16 staged rows
#if W8_ABL & 4
  bf16x8 a[2], b[4];
#endif
  for (int ks = 0; ks < ksteps; ++ks) {
#if W8_ABL & 4
    if ((ks & 1) == 0) {
#else
    bf16x8 a[2], b[4];
    {
#endif
    a[0] = w8_cat(w8_tr_at(la0), w8_tr_at(la0 + ROW16));
    a[1] = w8_cat(w8_tr_at(la1), w8_tr_at(la1 + ROW16));
#pragma unroll
    for (int o = 0; o < 4; ++o) b[o] = w8_cat(w8_tr_at(lb + 32u * o), w8_tr_at(lb + 32u * o + ROW16));
    }
    W8_FENCE();
    accB = MFMA32(a[0], ones, accB);
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) accW[s_][o] = MFMA32(a[s_], b[o], accW[s_][o]);
    W8_FENCE();
    la0 += 2 * ROW16; la1 += 2 * ROW16; lb += 2 * ROW16;
  }
}

// wave-local column sums on the matrix cores: accS[jb][.] (D[j][n]) += sum over the unit's 16 rows of t[row][j] * Bn[row][n].
// The wave stages its bf16 tile `t` in its own rows of `sc` (nobody else reads them at this point of the tile), reads it
// back transposed as the A operand and contracts against `bop` (lane (n, kq): B[4kq..4kq+3][n]).
__device__ __forceinline__ void w8_colsum_mfma(__bf16* __restrict__ sc, const bf16x4 (&t)[8], const bf16x4& bop,
                                               f32x4 (&accS)[8], int wave, int r, int q) {
  w8_stage_store(sc, t, 16 * wave + r, q);
  w8_wait_lgkm0();
  const __bf16* base = sc + (16 * wave) * LDS2 + w8_stage_toff(r | w8_opaque0(), q);
  bf16x4 a[8];
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) a[jb] = w8_tr(base + 16 * jb);
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) accS[jb] = w8_mfma16(a[jb], bop, accS[jb]);
}

// phase-timing trace (profiling builds only: -DW8_TRACE): shader-clock stamps of workgroup 0, waves 0 and 7, first tiles
#ifdef W8_TRACE
__device__ long long w8_trace[512];
#define W8_STAMP(k)                                                                              \
  do {                                                                                           \
    if (g == 0 && lane == 0 && (wave == 0 || wave == 7) && tile_no < 8)                          \
      w8_trace[(wave ? 256 : 0) + tile_no * 16 + (k)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
extern "C" int pv_debug_read_trace_w8(long long* out, int n) {
  if (n > 512) n = 512;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(w8_trace), n * sizeof(long long));
}
// ... and of the launch around the tile loop: 0 kernel entry, 1 prologue done, 2 tile loop done, 3 record written (shader
// cycles at [128 + k], the constant 100 MHz counter at [136 + k]: their ratio is the clock the launch ran at)
#define W8_STAMP_K(k)                                                                             \
  do {                                                                                           \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 0) {                 \
      w8_trace[128 + (k)] = (long long)__builtin_readcyclecounter();                             \
      w8_trace[136 + (k)] = (long long)__builtin_amdgcn_s_memrealtime();                         \
    }                                                                                            \
  } while (0)
// ... and of the epilogue's phases ([144 + k])
#define W8_STAMP_E(k)                                                                             \
  do {                                                                                           \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 0)                   \
      w8_trace[144 + (k)] = (long long)__builtin_readcyclecounter();                             \
  } while (0)
#else
#define W8_STAMP(k) do { } while (0)
#define W8_STAMP_K(k) do { } while (0)
#define W8_STAMP_E(k) do { } while (0)
#endif

// ---- column-parallel tail helpers --------------------------------------------------------------------------------------
// At batch 256 a workgroup owns 49 units: six full 8-wave tiles and ONE unit more.  Row-parallel, that seventh tile costs a
// whole tile's dependent chain (20.9 k of the launch's 178 k cycles: profiles/r03e_w8_ablations.txt) for one wave's work.
// Here the eight waves share the unit instead: wave w computes the 16-column block w of every layer (4 MFMAs where the
// row-parallel wave issues 32, 4 tanh per lane instead of 32) and the waves exchange their bf16x4 pieces through LDS — after
// an exchange every wave holds the unit's whole activation row set in exactly the registers (ih[8]) the row-parallel form
// keeps, so weight-gradient staging, the wave-local column sums and the row-local coordinate backward are the row-parallel
// code run by ONE wave each.
__device__ __forceinline__ void w8_xchg_put(char* smb, int k, const bf16x4& mine, int wave, int lane) {
  reinterpret_cast<bf16x4*>(smb + WO_TX + k * W8_TX_BYTES)[wave * 64 + lane] = mine;
}
__device__ __forceinline__ void w8_xchg_get(const char* smb, int k, bf16x4 (&all)[8], int lane) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) all[jb] = reinterpret_cast<const bf16x4*>(smb + WO_TX + k * W8_TX_BYTES)[jb * 64 + lane];
}
// block `wave` of a forward layer: C * pre-activation of outputs 16 wave + 4q .. +3 for row r
__device__ __forceinline__ f32x4 w8_tail_fwd(const __bf16* __restrict__ Wh, const float* __restrict__ bs, const bf16x4 (&ih)[8],
                                             const W8Addr& ad, int wave, int q) {
  f32x4 out = *reinterpret_cast<const f32x4*>(bs + 16 * wave + 4 * q);
  const __bf16* ah = Wh + ad.fb + 16 * wave * LDB;
  bf16x8 wh[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#if W8_HSWAP
    wh[m] = w8_cat(*reinterpret_cast<const bf16x4*>(ah + ad.fx[m] + 4 * ad.fs), *reinterpret_cast<const bf16x4*>(ah + ad.fx[m] + 4 - 4 * ad.fs));
#else
    wh[m] = *reinterpret_cast<const bf16x8*>(ah + ad.fx[m]);
#endif
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) out = MFMA32(wh[m], w8_catq(ih[2 * m], ih[2 * m + 1], q), out);
  return out;
}
// block `wave` of a dgrad layer: out[k = 16 wave + 4q .. +3] = sum_j (C W)[j][k] dp[j]
__device__ __forceinline__ f32x4 w8_tail_dgrad(const __bf16* __restrict__ Wh, const bf16x4 (&ih)[8], const W8Addr& ad, int wave,
                                               int r) {
  f32x4 out = {0.0f, 0.0f, 0.0f, 0.0f};
  const __bf16* ah = Wh + ad.db + 32 * ((wave >> 1) ^ (r >> 2)) + ((wave & 1) ? 4 - 4 * ad.hs : 4 * ad.hs);
  bf16x8 wh[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) wh[m] = w8_cat(w8_tr(ah + 32 * m * LDB), w8_tr(ah + 32 * m * LDB + 16 * LDB));
#pragma unroll
  for (int m = 0; m < 4; ++m) out = MFMA32(wh[m], w8_cat(ih[2 * m], ih[2 * m + 1]), out);
  return out;
}
__device__ __forceinline__ f32x4 w8_tanh4(f32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = w8_tanhc(v[i]);
  return v;
}
__device__ __forceinline__ bf16x4 w8_cvt4(const f32x4& v) {
  bf16x4 h;
#pragma unroll
  for (int i = 0; i < 4; ++i) h[i] = (__bf16)v[i];
  return h;
}
// d * (h^2 - 1) for one block (w8_mul_dtanh's sign convention)
__device__ __forceinline__ f32x4 w8_mul_dtanh4(const f32x4& d, const bf16x4& hb) {
  const f32x4 t = w8_f32_of(hb);
  return d * (t * t - 1.0f);
}

// ---- the guide in the prologue (PvEncFold, pv_sdec_fused.h) ------------------------------------------------------------------
// LDS scratch of the folded guide: the staging arrays, which the tile loop has not touched yet
#define WF_H1 WO_SA                                 // hidden activations, 128 floats each
#define WF_H2 (WF_H1 + 512)
#define WF_HD (WF_H2 + 512)                         // head pre-activations (<= 64), then z (<= 32) at +256
#define WF_P (WF_HD + 512)                          // per wave: 64 lanes x 17 partial sums
#define WF_P_WAVE (64 * 17 * 4)
static_assert(WF_P + W8_WAVES * WF_P_WAVE <= WO_VEC, "guide scratch fits in the staging arrays");
#include "pv_gemv16.h"      // w8_gemv16, w8_gemv16_k128(_load): one wave's 16 rows of a matrix-vector product from L2-resident weights

// LIK: the likelihood is a compile-time choice.  FOLD: the workgroup runs its images' guide itself (PvEncFold e; else unused)
// the shared first layer's bounded wait (FOLDK == 3): polls of ~0.5 us before a consumer computes its layer alone, and how often that
// happened in this process (observability, tests: pv_debug_coop_late_count)
#define W8_COOP_SPINS 4096
__device__ unsigned w8_coop_late_total;
extern "C" int pv_debug_coop_late_count(unsigned* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(w8_coop_late_total), sizeof(unsigned));
}
// FOLDK: 0 plain; 1 = FOLD; 2 = FOLD + the image's latent backward and encoder chain in the epilogue (PvEncFold::chain) — a build
// of its own: compiled into the FOLD build, the chain's registers cost the launch 2 us even when it is switched off; 3 = 2 + the
// guide's first layer shared among the 32 workgroups of a group (PvEncFold::coop, below; experiments build: measured SLOWER —
// 86.7 vs 85.0 us, the hand-off's 7 k cycles cost what the smaller L2 stream saves: profiles/r06k_coop_first_layer.txt)
template <bool GRADS, int LIK, int FOLDK>
__global__ __launch_bounds__(W8_THREADS) void pv_sdec_w8_kernel(PvFused f, PvEncFold e) {
  constexpr bool FOLD = FOLDK != 0, CHAIN = FOLDK >= 2, COOP = FOLDK == 3;
  extern __shared__ __attribute__((aligned(16))) char smb[];
  const int tid = threadIdx.x, lane0 = tid & 63, lane = lane0, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x, G = gridDim.x;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smb);
  const __bf16* W1h = reinterpret_cast<const __bf16*>(smb + WO_W1);
  const __bf16* W2h = reinterpret_cast<const __bf16*>(smb + WO_W2);
  __bf16* sA = reinterpret_cast<__bf16*>(smb + WO_SA);
  __bf16* sB = reinterpret_cast<__bf16*>(smb + WO_SB);
  float* vec = reinterpret_cast<float*>(smb + WO_VEC);
  float* info = reinterpret_cast<float*>(smb + WO_INFO);
  float* red = reinterpret_cast<float*>(smb + WO_RED);
  const char* gimg = reinterpret_cast<const char*>(f.wimg);
  W8_STAMP_K(0);
  f32x4 xr[4], xpk;                                   // FOLD: the workgroup's image, float4 columns lane + 64 c (c < 4); w8_gemv16's packed group
  // COOP: the FOUR images this wave serves in the shared first layer (group k = g & 7: images 8 mm + k, mm = 4 wave + i), float4
  // columns lane + 64 c; and the tag this launch's hand-offs carry (a word the step's closing launch increments: no value repeats)
  f32x4 xc[4][4];
  unsigned coop_tag = 0;
  if (COOP) {
    coop_tag = e.coop_flags[G] + 1u;
    const int K4 = (int)(e.ldx >> 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* xg = e.x + (int64_t)(8 * (4 * wave + i) + (g & 7)) * e.ldx;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k4 = lane + 64 * c;
        xc[i][c] = *reinterpret_cast<const f32x4*>(xg + 4 * (k4 < K4 ? k4 : 0));
      }
    }
  }
  if (FOLD && !COOP) {
    const int K4 = (int)(e.ldx >> 2);
    const float* xg = e.x + (int64_t)g * e.ldx;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k4 = lane + 64 * c;
      xr[c] = *reinterpret_cast<const f32x4*>(xg + 4 * (k4 < K4 ? k4 : 0));
    }
    const int kp = 64 * (((K4 + 63) >> 6) - 1) + (lane & 3);
    xpk = *reinterpret_cast<const f32x4*>(xg + 4 * (kp < K4 ? kp : 0));
  }
  // FOLD: the operands of the vectors and tables below, requested here (they are written after the guide: their round trips
  // would otherwise follow it)
  float pt_v[3] = {0.0f, 0.0f, 0.0f}, pt_a = 0.0f, pt_t[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (FOLD) {
    if (tid < FD_H) { pt_v[0] = f.wo[tid]; pt_v[1] = f.b1[tid]; pt_v[2] = f.b2[tid]; }
    {
      const int jb = tid >> 6, m = lane & 15, kq = lane >> 4, j = 16 * jb + m;
      if (kq == 0) pt_a = f.Wc[j * f.cd];
      else if (kq == 1) pt_a = f.cd == 2 ? f.Wc[j * 2 + 1] : 0.0f;
      else if (kq == 2) pt_a = f.bc[j];
    }
    if (tid < 256) {
      const int mm = tid >> 6, m = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int e_ = 0; e_ < 8; ++e_) {
        const int j = 32 * mm + 4 * kq + (e_ < 4 ? e_ : 16 + e_ - 4);
        if (m < 2) pt_t[e_] = f.Wc[j * f.cd];
        else if (m < 4 && f.cd == 2) pt_t[e_] = f.Wc[j * 2 + 1];
      }
    }
  }

  // ---- prologue: weight images by LDS-DMA (W1 at image 0, W2 at image 2 of the prepared set), vectors and tables ----
  f32x4 w1v[8], w2v[8];                               // (FOLD: the fp32 weights in flight while the guide runs)
  if (!FOLD) {
    constexpr int PIECES = IMG_BYTES / (W8_WAVES * 1024);
#pragma unroll
    for (int c = 0; c < PIECES; ++c) {
      const int off = (wave * PIECES + c) * 1024;
      w8_glds16(gimg + off + lane * 16, lds0 + WO_W1 + off);
      w8_glds16(gimg + 2 * IMG_BYTES + off + lane * 16, lds0 + WO_W2 + off);
    }
  } else {
    // FOLD: the images straight from the fp32 weights (pv_fb_layout.h pv_fb_prep, mode 0: bf16(C w), permuted columns,
    // swizzled chunks) — 8 float4 of each matrix per thread; and this workgroup's dL/d(hz) slots cleared
    if (GRADS) {
      const int64_t b0 = (int64_t)g * e.img_per_wg;
      f32x4* zp = reinterpret_cast<f32x4*>(f.part_hz + b0 * f.kmax * FD_H);
      const int n4 = e.img_per_wg * f.kmax * (FD_H / 4);
      for (int i = tid; i < n4; i += W8_THREADS) zp[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (f.part_rs) {                                                // ... and its row-sum slots (filled in the epilogue below)
        f32x4* zr = reinterpret_cast<f32x4*>(f.part_rs + b0 * f.kmax * PV_RS_W);
        for (int i = tid; i < e.img_per_wg * f.kmax * (PV_RS_W / 4); i += W8_THREADS) zr[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
  }
  // (FOLD: the guide first — its operand requests head the memory queue; the vectors and tables below need nothing from it)
  if (FOLD) {
    // ---- the guide of this workgroup's images (nets/fc.py:51-61, models/ivae.py:204-221, models/base.py:97-119) ----
    // (barriers here are LDS-only: the global stores of this phase are read back after the closing __syncthreads alone)
    float* h1s = reinterpret_cast<float*>(smb + WF_H1);
    float* h2s = reinterpret_cast<float*>(smb + WF_H2);
    float* hds = reinterpret_cast<float*>(smb + WF_HD);
    float* zs = hds + 64;
    float* P = reinterpret_cast<float*>(smb + WF_P + wave * WF_P_WAVE);
    const int N = (int)e.ldx, zd = e.z_dim;
    const int r_ = lane & 15, q_ = lane >> 4;
    {
      const int64_t b = g;                                         // one image per workgroup (pv_sdec_fused_w8_fold_ok)
      // small operands of the later phases, requested up front (each would otherwise head its phase with an L2 / HBM round trip)
      const int jw = 16 * wave + r_;
      const float pb1 = (e.enc1.b_off >= 0 && jw < e.enc1.out_dim) ? e.params[e.enc1.b_off + jw] : 0.0f;
      const float pbh = (e.head.b_off >= 0 && jw < e.head.out_dim) ? e.params[e.head.b_off + jw] : 0.0f;
      const float pep = (wave == 0 && lane < zd) ? e.eps[b * zd + lane] : 0.0f;
      f32x4 wl1[8], wlh[8];
      float pwz[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (tid < FD_H) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pwz[i] = i < e.lat_in ? e.Wz[(int64_t)tid * e.lat_in + i] : 0.0f;
      }
      // the first layer for THIS image by this workgroup alone: every workgroup streams the whole 400 KB matrix from L2 (12.8 MB per
      // XCD through an L2 that delivers ~1-2 KB per clock: the layer is bound by that, ~19 k cycles)
      auto layer0_alone = [&](const f32x4 (&xr_)[4], const f32x4& xpk_, bool load_next) {
        // (every workgroup reads the same 400 KB matrix at the same time: which wave takes which 16 rows rotates with the
        //  workgroup index, so that the chip's requests spread over the L2 channels instead of marching through them in step)
        const int jr = 16 * ((wave + g) & (W8_WAVES - 1));
        const float v = w8_gemv16(e.params + e.enc0.w_off, N, e.enc0.out_dim, jr, xr_, xpk_, P, lane,
                                  [&]() {     // (issued behind the first pass's loads)
                                    if (load_next) w8_gemv16_k128_load(e.params + e.enc1.w_off, e.enc1.in_dim, e.enc1.out_dim, 16 * wave, lane, wl1);
                                  });
        if (load_next && 16 * wave < e.head.out_dim)     // (the head's, once the first layer's operand registers are free: under layer 1)
          w8_gemv16_k128_load(e.params + e.head.w_off, e.head.in_dim, e.head.out_dim, 16 * wave, lane, wlh);
        const int j = jr + r_;
        if (q_ == 0 && j < e.enc0.out_dim) {
          const float y = pv_act_fwd2(v + (e.enc0.b_off >= 0 ? e.params[e.enc0.b_off + j] : 0.0f), e.enc0.act);
          h1s[j] = y;
          e.eact0[b * e.enc0.out_dim + j] = y;
        }
      };
      if constexpr (!COOP) {
        layer0_alone(xr, xpk, true);
      } else {
        // ---- the first layer SHARED (round 6, fourth cut): the 32 workgroups g = 8 mm + k of a group (one XCD's, where workgroups are
        // dealt round-robin — placement only matters for speed) each take FOUR rows of the matrix (12.5 KB, staged in LDS) for all
        // 32 images of the group (100 KB of observations, read by the whole group from the same L2): 3.6 MB per XCD instead of 12.8.
        // Member mm writes act(W0[4mm .. 4mm+3] x_b + b0) for the group's images with agent-scope stores and publishes the launch's
        // tag; every member then waits for its 32 producers' tags (bounded: a consumer whose producers do not show up — a GPU
        // shared with another process, CUs masked off — computes its image's layer ALONE, as the other build does) and reads its
        // image's 128 activations past its L1.  Same fp32 multiply-adds in another order than the alone form: equal to rounding.
        const int kg = g & 7, mm = g >> 3, K4 = N >> 2;
        float* w0s = reinterpret_cast<float*>(smb + WO_SB);            // [4][N]: the wgrad-1 staging area is free until the tile loop
        {
          const float* wsrc = e.params + e.enc0.w_off + (int64_t)(4 * mm) * N;
          for (int idx = tid; idx < K4 * 4; idx += W8_THREADS) reinterpret_cast<f32x4*>(w0s)[idx] = reinterpret_cast<const f32x4*>(wsrc)[idx];
        }
        w8_gemv16_k128_load(e.params + e.enc1.w_off, e.enc1.in_dim, e.enc1.out_dim, 16 * wave, lane, wl1);
        if (16 * wave < e.head.out_dim)
          w8_gemv16_k128_load(e.params + e.head.w_off, e.head.in_dim, e.head.out_dim, 16 * wave, lane, wlh);
        const float pb0 = (lane < 16 && e.enc0.b_off >= 0) ? e.params[e.enc0.b_off + 4 * mm + (lane & 3)] : 0.0f;
        W8_STAMP_E(20);
        pv_lds_barrier();                                              // the four rows are staged
        W8_STAMP_E(21);
        float acc[16];                                                 // [image i][row]: 4 i + row
#pragma unroll
        for (int a = 0; a < 16; ++a) acc[a] = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k4 = lane + 64 * c;
          const bool okc = k4 < K4;
          if (64 * c >= K4) break;                                     // (wave-uniform)
#pragma unroll
          for (int row = 0; row < 4; ++row) {
            f32x4 wv = *reinterpret_cast<const f32x4*>(w0s + row * N + 4 * (okc ? k4 : 0));
            if (!okc) wv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};             // (a column that does not exist contributes 0 * x)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[4 * i + row] = w8_dot4(wv, xc[i][c], acc[4 * i + row]);
          }
        }
        // 64 x 16 partial sums -> this wave's transpose buffer -> lane (r, q) sums column r over its 16 lanes, then over q
        W8_STAMP_E(22);
#pragma unroll
        for (int a = 0; a < 16; ++a) P[lane * 17 + a] = acc[a];
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += P[(16 * q_ + i) * 17 + r_];
        v = pv_sum_rows(v);
        if (q_ == 0) {                                                 // lane r: image i = r >> 2 of this wave, row r & 3
          const int bb = 8 * (4 * wave + (r_ >> 2)) + kg, j = 4 * mm + (r_ & 3);
          const float y = pv_act_fwd2(v + pb0, e.enc0.act);
          __hip_atomic_store(e.eact0 + (int64_t)bb * e.enc0.out_dim + j, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        W8_STAMP_E(23);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // every thread's write-through stores acknowledged
        W8_STAMP_E(24);
        pv_lds_barrier();
        if (tid == 0) __hip_atomic_store(e.coop_flags + g, coop_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        W8_STAMP_E(25);
        int late = 0;
        if (tid < 32) {
          const unsigned* fl = e.coop_flags + 8 * tid + kg;
          late = 1;
          for (int spin = 0; spin < W8_COOP_SPINS; ++spin) {
            if (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == coop_tag) { late = 0; break; }
            __builtin_amdgcn_s_sleep(2);
          }
        }
        W8_STAMP_E(26);
        if (__syncthreads_or(late) != 0) {                            // (workgroup-uniform) the bounded wait ran out: alone
          if (tid == 0) atomicAdd(&w8_coop_late_total, 1u);
          f32x4 xa[4], xpa;
          const float* xg = e.x + (int64_t)g * e.ldx;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int k4 = lane + 64 * c;
            xa[c] = *reinterpret_cast<const f32x4*>(xg + 4 * (k4 < K4 ? k4 : 0));
          }
          const int kp = 64 * (((K4 + 63) >> 6) - 1) + (lane & 3);
          xpa = *reinterpret_cast<const f32x4*>(xg + 4 * (kp < K4 ? kp : 0));
          layer0_alone(xa, xpa, false);
        } else if (tid < FD_H) {
          h1s[tid] = __hip_atomic_load(e.eact0 + b * e.enc0.out_dim + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      W8_STAMP_K(4);
      {
        // the decoder's fp32 hidden weights, requested now (the first layer's 34 operand registers per lane are free again)
        // and converted into the LDS images after the guide: 8 float4 of each matrix per thread
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = tid + W8_THREADS * u;
          w1v[u] = reinterpret_cast<const f32x4*>(f.W1)[idx];
          w2v[u] = reinterpret_cast<const f32x4*>(f.W2)[idx];
        }
      }
      pv_lds_barrier();
      {
        const float v = w8_gemv16_k128(wl1, e.enc1.in_dim, h1s, P, lane);
        const int j = 16 * wave + r_;
        if (q_ == 0 && j < e.enc1.out_dim) {
          const float y = pv_act_fwd2(v + pb1, e.enc1.act);
          h2s[j] = y;
          e.eact1[b * e.enc1.out_dim + j] = y;
        }
      }
      W8_STAMP_K(5);
      pv_lds_barrier();
      if (16 * wave < e.head.out_dim) {                            // [mu | softplus input]: 16 rows per wave
        const float v = w8_gemv16_k128(wlh, e.head.in_dim, h2s, P, lane);
        const int j = 16 * wave + r_;
        if (q_ == 0 && j < e.head.out_dim) {
          const float y = v + pbh;
          hds[j] = y;
          e.head_out[b * e.head.out_dim + j] = y;
        }
      }
      W8_STAMP_K(6);
      pv_lds_barrier();
      if (wave == 0) {
        // z = mu + softplus(s) eps and the sampled-KL terms (torch Normal.log_prob), one lane per latent coordinate
        float lp = 0.0f, lq = 0.0f;
        float rc_ = 1.0f, rs_ = 0.0f;              // (round 6) cos / sin of the rotation: by the lane that holds phi (coordinate 0), next to
        if (lane < zd) {                            // the other lanes' log-density terms instead of after the wave sums; one range reduction
          const float mu = hds[lane], sig = pv_softplus(hds[zd + lane]);
          const float ep = pep;
          const float z = mu + sig * ep;
          if (lane == 0 && e.coord_dim == 2 && e.has_r) sincosf(z, &rs_, &rc_);
          e.z[b * zd + lane] = z;
          e.z_scale[b * zd + lane] = sig;
          if (e.z_loc_out) e.z_loc_out[b * zd + lane] = mu;
          if (e.z_scale_out) e.z_scale_out[b * zd + lane] = sig;
          const float d = z - mu;
          lq = -(d * d) / (2.0f * (sig * sig)) - logf(sig) - LOG_SQRT_2PI;
          lp = -(z * z) / 2.0f - LOG_SQRT_2PI;
          zs[lane] = z;
        }
        lp = pv_wave_sum(lp);
        lq = pv_wave_sum(lq);
        if (lane == 0) {
          e.kl_part[2 * b] = e.beta * lp;
          e.kl_part[2 * b + 1] = e.beta * lq;
          // _split_latent -> the transform parameters (models/base.py:97-119; the t / s priors of models/ivae.py:187-191)
          const float* zb = zs;
          int idx = 0;
          float c = 1.0f, sn = 0.0f, sc = 1.0f, tx = 0.0f, ty = 0.0f;
          if (e.coord_dim == 1) {
            if (e.has_t) { tx = zb[0] * e.tp0; idx = 1; }
          } else if (e.coord_dim == 2) {
            if (e.has_r) { ++idx; c = rc_; sn = rs_; }
            if (e.has_t) { tx = zb[idx] * e.tp0; ty = zb[idx + 1] * e.tp1; idx += 2; }
            if (e.has_s) { sc = 1.0f + e.sc_prior * zb[idx++]; }
          }
          float* t = e.tp + b * 8;
          t[0] = c; t[1] = sn; t[2] = sc; t[3] = tx; t[4] = ty;
        }
      }
      pv_lds_barrier();
      if (tid < FD_H) {
        // hz = C fc_latent(z content) (nets/fc.py:217,230: no bias)
        int coord = 0;
        if (e.coord_dim == 1) coord = e.has_t ? 1 : 0;
        else if (e.coord_dim == 2) coord = e.has_r + 2 * e.has_t + e.has_s;
        const float* wz = e.Wz + (int64_t)tid * e.lat_in;
        float v = 0.0f;
        for (int i = 0; i < e.lat_in; ++i) v += zs[coord + i] * (i < 4 ? pwz[i] : wz[i]);
        e.hz[b * FD_H + tid] = v * W8_C;
      }
      W8_STAMP_K(7);
    }
  }
  if (tid < FD_H) {
    vec[tid] = FOLD ? pt_v[0] : f.wo[tid];
    vec[FD_H + tid] = W8_C * (FOLD ? pt_v[1] : f.b1[tid]);
    vec[2 * FD_H + tid] = W8_C * (FOLD ? pt_v[2] : f.b2[tid]);
  }
  {
    // coordinate layer A operands (v_mfma_f32_16x16x16_bf16: lane (m, kq) holds A[m][4kq .. 4kq+3]), k slots:
    //   kq 0: [wh0 wh0 wl0 0] x [xh0 xl0 xh0 0]   kq 1: the same for coordinate 1   kq 2: [bch bcl 0 0] x [1 1 0 0]
    const int jb = tid >> 6, m = lane & 15, kq = lane >> 4, j = 16 * jb + m;
    float v = 0.0f;
    if (FOLD) v = W8_C * pt_a;
    else if (kq == 0) v = W8_C * f.Wc[j * f.cd];
    else if (kq == 1) v = f.cd == 2 ? W8_C * f.Wc[j * 2 + 1] : 0.0f;
    else if (kq == 2) v = W8_C * f.bc[j];
    __bf16 hi, lo;
    fb_split(v, hi, lo);
    bf16x4 a = w8_zero4();
    if (kq < 2) { a[0] = hi; a[1] = hi; a[2] = lo; }
    else if (kq == 2) { a[0] = hi; a[1] = lo; }
    reinterpret_cast<bf16x4*>(smb + WO_ATAB)[tid] = a;
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
      if (FOLD) w = pt_t[e];
      else if (m < 2) w = f.Wc[j * f.cd];
      else if (m < 4 && f.cd == 2) w = f.Wc[j * 2 + 1];
      __bf16 hi, lo;
      fb_split(w, hi, lo);
      a[e] = m >= 4 ? (__bf16)0.0f : ((m & 1) ? lo : hi);
    }
    reinterpret_cast<bf16x8*>(smb + WO_TTAB)[tid] = a;
  }
  if (FOLD) {
    // ... and the decoder's weight images from the fp32 values requested before the guide
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    __bf16* i1 = reinterpret_cast<__bf16*>(smb + WO_W1);
    __bf16* i2 = reinterpret_cast<__bf16*>(smb + WO_W2);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + W8_THREADS * u, row = idx >> 5, c4 = idx & 31;
      us4 h1, h2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        h1[i] = __builtin_bit_cast(unsigned short, (__bf16)(w1v[u][i] * W8_C));
        h2[i] = __builtin_bit_cast(unsigned short, (__bf16)(w2v[u][i] * W8_C));
      }
      const int el = fb_wel(row, fb_pcol(4 * c4)) ^ (W8_HSWAP ? 4 * ((row >> 2) & 1) : 0) ^ (W8_QSWAP ? 4 * ((c4 >> 1) & 1) : 0);
      *reinterpret_cast<us4*>(i1 + el) = h1;
      *reinterpret_cast<us4*>(i2 + el) = h2;
    }
  }
  w8_wait_vm0();
  __syncthreads();
  W8_STAMP_K(1);
  const float bo = f.bo[0];
  // The kernel arguments arrive as 16-dword scalar tuples; what the tile loop needs of them again lives in scalar values of
  // its own (made opaque here), so that a register-pressure spill saves and restores two lanes, not the argument tuple's
  // sixteen — 36 + 21 v_readlane per tile came from fetching llrow / loc / rowtp back that way.
  typedef __attribute__((address_space(1))) float gfloat;         // (explicitly global: an opaque pointer would be stored through flat_*)
  unsigned long long u_llrow, u_loc, u_rowtp, u_part_hz;
  int64_t a_M;
  // (real copies: an in-place "+s" keeps the value inside the tuple's registers, and the tuple is still spilled as a whole)
  asm volatile("s_mov_b64 %0, %5\n\ts_mov_b64 %1, %6\n\ts_mov_b64 %2, %7\n\ts_mov_b64 %3, %8\n\ts_mov_b64 %4, %9"
               : "=&s"(u_llrow), "=&s"(u_loc), "=&s"(u_rowtp), "=&s"(u_part_hz), "=&s"(a_M)
               : "s"((unsigned long long)f.llrow), "s"((unsigned long long)f.loc), "s"((unsigned long long)f.rowtp),
                 "s"((unsigned long long)f.part_hz), "s"(f.M));
  gfloat* a_llrow = (gfloat*)u_llrow; gfloat* a_loc = (gfloat*)u_loc; gfloat* a_rowtp = (gfloat*)u_rowtp;
  gfloat* a_part_hz = (gfloat*)u_part_hz;

  // persistent accumulators: the wave's slice (rows 16*wave .. +15) of dW1 (x C) and dW2, the bias sums, and the
  // wave-local column sums D[j][n]: n = 0 dL/d(hz) | 1, 5 dWc0 (hi, lo) | 2, 6 dWc1 | 3, 4 d(wo)   (n 0,1,2,5,6 carry C^2)
  f32x4 accW1[2][4], accW2[2][4], accS[8], accB1 = {0, 0, 0, 0}, accB2 = {0, 0, 0, 0};
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    accW1[kb >> 2][kb & 3] = f32x4{0, 0, 0, 0}; accW2[kb >> 2][kb & 3] = f32x4{0, 0, 0, 0}; accS[kb] = f32x4{0, 0, 0, 0};
  }
  float dbo = 0.0f;
  int cur_b = -1;                                    // the sample whose dL/d(hz) this WAVE is accumulating
  const int upb = f.N / FD_UNIT;
  float* rec = f.part + (int64_t)g * FD_REC;

  auto flush_hz = [&](int b) {
    // the wave's rows of sample b end: publish its partial dL/d(hz[b]) (column 0 of accS: lanes r == 0) in its own slot
    const int64_t ub = (int64_t)b * upb;
    const int gfirst = (int)(((ub + 1) * G + f.units - 1) / f.units) - 1;
    gfloat* dst = a_part_hz + ((int64_t)b * f.kmax + (g - gfirst) * W8_WAVES + wave) * FD_H + 4 * q;
    if (r == 0) {
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) *(__attribute__((address_space(1))) f32x4*)(dst + 16 * jb) = accS[jb] * W8_RC2;
    }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
      for (int i = 0; i < 4; ++i) accS[jb][i] = r == 0 ? 0.0f : accS[jb][i];
  };

  // units are 32-bit here (the launcher falls back to the 4-wave kernel beyond 2^31 rows), and a unit's sample b / offset
  // inside the sample / observation unit are carried incrementally from tile to tile: no integer division in the loop
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
  // a range of 8 n + 1 units ends with a column-parallel tail (below): the row-parallel tiles cover [u_lo, u_end), and what a
  // wave without a further unit prefetches is the tail unit's inputs (every wave takes part in the tail)
  const bool has_tail = W8_TAIL && ((u_hi - u_lo) & (W8_WAVES - 1)) == 1;
  const int u_end = has_tail ? u_hi - 1 : u_hi;
  const Pos pos_lo = pos_of(has_tail ? u_end : u_lo);   // what an out-of-range wave fetches instead (valid; unused without a tail)
  Pos pos_cur = pos_of(u_lo + wave < u_end ? u_lo + wave : (has_tail ? u_end : u_lo));
  Pos pos_nx = pos_cur;
  float sw_next = 1.0f;
  auto x_of = [&](const Pos& p_) -> float {
    if (f.sw) sw_next = f.sw[p_.b];
    return f.x[(int64_t)p_.xu * FD_UNIT + r];
  };
  float xv_next = x_of(pos_cur);
  float* chz = reinterpret_cast<float*>(smb + WO_CHZ) + wave * FD_H;
  float* ctp = reinterpret_cast<float*>(smb + WO_CTP) + wave * 64;
  float* cgr = reinterpret_cast<float*>(smb + WO_CGR) + wave * 64;
  auto fetch_unit_inputs = [&](const Pos& p_) {
    const int n0 = p_.loc * FD_UNIT;
    w8_glds4(f.hz + (int64_t)p_.b * FD_H + lane, lds0 + WO_CHZ + wave * (FD_H * 4));
    w8_glds4(f.hz + (int64_t)p_.b * FD_H + 64 + lane, lds0 + WO_CHZ + wave * (FD_H * 4) + 256);
    w8_glds4(f.tp + (int64_t)p_.b * 8 + (lane & 7), lds0 + WO_CTP + wave * 256);
    w8_glds4(f.grid + (int64_t)n0 * f.cd + (lane & (16 * f.cd - 1)), lds0 + WO_CGR + wave * 256);
  };
  fetch_unit_inputs(pos_cur);
  const W8Addr wad = w8_addr(r, q);
  int tile_no = -1;
  for (int ut = u_lo; ut < u_end; ut += W8_WAVES) {
    ++tile_no;
    (void)tile_no;                      // (used by the -DW8_TRACE stamps only)
    asm volatile("; W8_TILE_BEGIN");
    W8_STAMP(0);
    const int nact = (u_end - ut) < W8_WAVES ? (u_end - ut) : W8_WAVES;
    // the unit this wave fetches for the NEXT tile
    if (ut + W8_WAVES + wave < u_end) advance(pos_nx, W8_WAVES);
    else pos_nx = pos_lo;
    int opq = 0;
    asm volatile("" : "+v"(opq));       // a zero the compiler cannot see through, OR-ed into the lane id: every lane-dependent
    const int lane = lane0 | opq, r = lane & 15, q = lane >> 4;   // address below is then recomputed per tile (one or two
    // instructions each) instead of being hoisted out of the loop, kept in registers and — at the 256-register limit —
    // spilled to scratch and reloaded (the first build of this kernel moved 25 MB of scratch per launch that way)
    const float* wos = vec;
    const float* b1s = vec + FD_H;
    const float* b2s = vec + 2 * FD_H;
    const bool act = wave < nact;
    const int unit = act ? pos_cur.unit : ut;
    const int bu = pos_cur.b;
    const int64_t row = (int64_t)unit * FD_UNIT + r;
    float x0, x1, u0c, u1c, sc;
    w8_wait_vm0();                        // this wave's LDS-DMA of the tile's inputs (issued a tile ago)
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
    float* inf_x1 = info + W8_ROWS + 16 * wave;
    float* inf_dl = info + 2 * W8_ROWS + 16 * wave;

    // saved activations live as bf16 only (h0b, h1b): fp32 copies next to the 104 accumulator registers do not fit
    // two waves per SIMD; the backward pass forms 1 - h^2 from them (bf16-relative precision, like every MFMA operand here)
    f32x4 tC[8];
    bf16x4 pA[8], h0b[8], h1b[8];
    float dlda = 0.0f;
    // A wave without a unit of its own (partial last tile) runs the same straight-line code on a valid unit of the
    // workgroup's range (what its prefetch slots hold) with its dL/dlogit forced to zero: every gradient it stages or
    // accumulates is then zero and its per-row outputs are not stored.  No wave-uniform branches around the phases —
    // the merges they create cost register copies of the 104 accumulators' neighbours in every tile.
    {
      // ---- coordinate layer on the matrix cores: C h0pre = (C Wc) x' + C bc + C hz[b] ----
      bf16x4 bx = w8_zero4();
      {
        const float v = q == 0 ? x0 : x1;
        __bf16 vh, vl;
        fb_split(v, vh, vl);
        const __bf16 one = (__bf16)1.0f;
        if (q < 2) { bx[0] = vh; bx[1] = vl; bx[2] = vh; }
        else if (q == 2) { bx[0] = one; bx[1] = one; }
      }
      const bf16x4* atab = reinterpret_cast<const bf16x4*>(smb + WO_ATAB) + lane;
      const float* hzb = chz;
      bf16x4 aop[8];
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        tC[jb] = *reinterpret_cast<const f32x4*>(hzb + 16 * jb + 4 * q);
        aop[jb] = atab[64 * jb];
      }
      if (GRADS && q == 0) { inf_x0[r] = x0; inf_x1[r] = x1; }
      W8_FENCE();
      if (f.hz_scale == 0.0f) {                  // hz arrives unscaled only when the generic encoder path produced it
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) tC[jb] = tC[jb] * W8_C;
      }
      W8_FENCE();
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) tC[jb] = w8_mfma16(aop[jb], bx, tC[jb]);
      W8_FENCE();
      w8_tanh8(tC);
      w8_cvt8(tC, h0b);
    }
    asm volatile("; W8_P1_coord_done");
    W8_STAMP(1);
    fetch_unit_inputs(pos_nx);                 // the slots were consumed by the coordinate layer above
    {
      w8_layer_fwd(W1h, b1s, h0b, tC, wad, q);
      w8_tanh8(tC);
      w8_cvt8(tC, h1b);                                             // feeds layer 2 and its wgrad
      asm volatile("; W8_P2_l1_done");
    W8_STAMP(2);
      w8_layer_fwd(W2h, b2s, h1b, tC, wad, q);
      // ---- h2, output layer + likelihood (fp32); tC <- g = wo (1 - h2^2), pA <- bf16(h2) ----
      w8_tanh8(tC);                                                // tC = h2
      f32x4 part4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wos + 16 * jb + 4 * q);
        part4 = part4 + tC[jb] * wv;
        if (GRADS) {
#pragma unroll
          for (int i = 0; i < 4; ++i) pA[jb][i] = (__bf16)tC[jb][i];
          const f32x4 t2 = tC[jb] * tC[jb];
          tC[jb] = wv - wv * t2;
        }
      }
      const float a = w8_sum_q((part4[0] + part4[1]) + (part4[2] + part4[3])) + bo;
      float ll, locv;
      if (LIK == PV_LIK_BERNOULLI) {
        const float pr = w8_rcp(1.0f + w8_exp(-a));
        const float pc = fminf(fmaxf(pr, BERN_EPS), 1.0f - BERN_EPS);
        // -BCEWithLogits(lg, x) with lg = logit(pc) (torch: probs_to_logits, then binary_cross_entropy_with_logits), written with
        // the identities 1 + exp(-|lg|) = 1 / max(pc, 1 - pc) and sigmoid(lg) = pc: the two logarithms lg is made of serve the
        // softplus term too, and the row's dependent chain is exp -> rcp -> 2 log instead of seven transcendentals (round 5)
        const float lpc = w8_log(pc), l1pc = w8_log(1.0f - pc);
        const float lg = lpc - l1pc;
        ll = -(fmaxf(lg, 0.0f) - lg * xv - fmaxf(lpc, l1pc));
        const float mask = (pr >= BERN_EPS && pr <= 1.0f - BERN_EPS) ? 1.0f : 0.0f;
        dlda = (pc - xv) * mask;
        locv = pr;
      } else if (LIK == PV_LIK_CBERNOULLI) {
        pv_cbern(a, xv, ll, dlda, locv);
      } else {
        const float pr = f.sigmoid_out ? w8_rcp(1.0f + w8_exp(-a)) : a;
        const float d = xv - pr;
        ll = -(d * d) / (2.0f * f.sig * f.sig) - w8_log(f.sig) - LOG_SQRT_2PI;
        dlda = -d / (f.sig * f.sig) * (f.sigmoid_out ? pr * (1.0f - pr) : 1.0f);
        locv = pr;
      }
      dlda *= act ? swv : 0.0f;
      if (q == 0) {
        if (act) {
          if (a_llrow) a_llrow[row] = ll;
          if (a_loc) a_loc[row] = locv;
        }
        if (GRADS) { dbo += dlda; inf_dl[r] = dlda; }
      }
      xv_next = x_of(pos_nx);
    }
    pos_cur = pos_nx;                          // (unit, bu, row of THIS tile were taken above)
    asm volatile("; W8_P3_fwd_done");
    W8_STAMP(3);
    if (!GRADS) continue;
    const int ksteps = (nact + 1) >> 1;
    {
      // ---- d(wo) += sum_rows dlda h2 : wave-local MFMA through the wave's own rows of staging A (free: every wave
      // passed barrier 4 of the previous tile); B = dlda of rows 4q..4q+3 in columns 3 (hi) and 4 (lo)
      w8_wait_lgkm0();
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(inf_dl + 4 * q);
      bf16x4 bw = w8_zero4();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __bf16 hi, lo;
        fb_split(d4[i], hi, lo);
        bw[i] = r == 3 ? hi : (r == 4 ? lo : (__bf16)0.0f);
      }
      w8_colsum_mfma(sA, pA, bw, accS, wave, r, q);
      // dpre2 = dlda * wo (1 - h2^2)
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) tC[jb] = tC[jb] * dlda;
      w8_cvt8(tC, pA);                                            // feeds the wgrad and the dgrad of layer 2
    }
    asm volatile("; W8_P4_dwo_done");
    W8_STAMP(4);
    // ---- wgrad of layer 2: stage (dpre2, h1) of all 128 rows, one pass ----
    w8_stage_store(sA, pA, 16 * wave + r, q);
    w8_stage_store(sB, h1b, 16 * wave + r, q);
    __syncthreads();                                                // barrier 1
    asm volatile("; W8_P5_bar1");
    W8_STAMP(5);
    w8_wgrad_consume(sA, sB, accW2, accB2, wave, r, q, ksteps);
    asm volatile("; W8_P6_cons2");
    W8_STAMP(6);
    bf16x4 p0[8];
    {
      w8_layer_dgrad(W2h, pA, tC, wad);                            // tC = C dL/dh1
      w8_mul_dtanh(tC, h1b);                                      // -C dpre1 (see w8_mul_dtanh)
      w8_cvt8(tC, pA);                                            // feeds the dgrad and the wgrad of layer 1
      asm volatile("; W8_P7_dgrad2");
    W8_STAMP(7);
      w8_layer_dgrad(W1h, pA, tC, wad);                            // tC = -C^2 dL/dh0
      w8_mul_dtanh(tC, h0b);                                      // C^2 dpre0 (the sign is back)
      w8_cvt8(tC, p0);
    }
    asm volatile("; W8_P8_dgrad1");
    W8_STAMP(8);
    __syncthreads();                                                // barrier 2: round 1 consumed everywhere
    W8_STAMP(9);
    {
      // (the staging area is free between barrier 2 and the round-2 stores: the wave's own rows serve the dpre0 sums now,
      //  which ends p0's life before the round-2 consume)
      // ---- coordinate layer backward, row-local part on the matrix cores: D[m][row] = sum_j T[m][j] dpre0[row][j] ----
      f32x4 dd = {0.0f, 0.0f, 0.0f, 0.0f};
      const bf16x8* ttab = reinterpret_cast<const bf16x8*>(smb + WO_TTAB) + lane;
#pragma unroll
      for (int mm = 0; mm < 4; ++mm) dd = MFMA32(ttab[64 * mm], w8_cat(p0[2 * mm], p0[2 * mm + 1]), dd);
      if (q == 0 && act) {
        const float d0 = (dd[0] + dd[1]) * W8_RC2, d1 = (dd[2] + dd[3]) * W8_RC2;
        a_rowtp[row] = sc * (d1 * u0c - d0 * u1c);
        a_rowtp[a_M + row] = d0 * u0c + d1 * u1c;
        a_rowtp[2 * a_M + row] = d0;
        a_rowtp[3 * a_M + row] = d1;
      }
      if (act && bu != cur_b) {
        if (cur_b >= 0) flush_hz(cur_b);
        cur_b = bu;
      }
      // ---- dL/d(hz[b]) = sum_rows dpre0, dWc_k = sum_rows dpre0 x'_k : wave-local MFMA, own rows of staging A;
      // B columns: 0 ones | 1, 5 x0 (hi, lo) | 2, 6 x1 (hi, lo)
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(inf_x0 + 4 * q);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(inf_x1 + 4 * q);
      bf16x4 bc_ = w8_zero4();
      const bool use1 = r == 2 || r == 6, lo_col = r >= 5;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __bf16 hi, lo;
        fb_split(use1 ? a1[i] : a0[i], hi, lo);
        __bf16 v = lo_col ? lo : hi;
        if (r == 0) v = (__bf16)1.0f;
        if (r == 3 || r == 4 || r > 6) v = (__bf16)0.0f;
        bc_[i] = v;
      }
      w8_colsum_mfma(sA, p0, bc_, accS, wave, r, q);
      w8_wait_lgkm0();                                              // (own reads done before the rows are re-staged)
    }
    asm volatile("; W8_P12_rowlocal");
    W8_STAMP(12);
    // ---- wgrad of layer 1: stage (C dpre1, h0) ----
    w8_stage_store(sA, pA, 16 * wave + r, q);
    w8_stage_store(sB, h0b, 16 * wave + r, q);
    __syncthreads();                                                // barrier 3
    asm volatile("; W8_P10_bar3");
    W8_STAMP(10);
    w8_wgrad_consume(sA, sB, accW1, accB1, wave, r, q, ksteps);
    asm volatile("; W8_P11_cons1");
    W8_STAMP(11);
    __syncthreads();                                                // barrier 4: round 2 consumed everywhere
    asm volatile("; W8_P13_bar4");
    W8_STAMP(13);
    W8_STAMP(14);
    asm volatile("; W8_TILE_END");
  }

  if (has_tail) {
    // ================= column-parallel tail: ONE unit, eight waves (helpers above the kernel) =================
    asm volatile("; W8_TAIL_BEGIN");
    const int lane = lane0, r = lane & 15, q = lane >> 4;
    const float* wos = vec;
    const float* b1s = vec + FD_H;
    const float* b2s = vec + 2 * FD_H;
    const int unit = u_end, bu = pos_cur.b;            // (every wave's prefetch slots hold the tail unit's inputs: pos_lo)
    const int64_t row = (int64_t)unit * FD_UNIT + r;
    float x0, x1, u0c, u1c, sc;
    w8_wait_vm0();
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
    float* inf_x1 = info + W8_ROWS + 16 * wave;
    float* inf_dl = info + 2 * W8_ROWS + 16 * wave;
    bf16x4 h0b[8], h1b[8], own0, own1;
    {
      // ---- coordinate layer, block `wave` ----
      bf16x4 bx = w8_zero4();
      {
        const float v = q == 0 ? x0 : x1;
        __bf16 vh, vl;
        fb_split(v, vh, vl);
        const __bf16 one = (__bf16)1.0f;
        if (q < 2) { bx[0] = vh; bx[1] = vl; bx[2] = vh; }
        else if (q == 2) { bx[0] = one; bx[1] = one; }
      }
      f32x4 t0 = *reinterpret_cast<const f32x4*>(chz + 16 * wave + 4 * q);
      if (f.hz_scale == 0.0f) t0 = t0 * W8_C;
      const bf16x4 aop = (reinterpret_cast<const bf16x4*>(smb + WO_ATAB) + lane)[64 * wave];
      if (GRADS && q == 0) { inf_x0[r] = x0; inf_x1[r] = x1; }
      t0 = w8_mfma16(aop, bx, t0);
      own0 = w8_cvt4(w8_tanh4(t0));
      w8_xchg_put(smb, 0, own0, wave, lane);
    }
    __syncthreads();                                               // exchange 0: h0
    w8_xchg_get(smb, 0, h0b, lane);
    own1 = w8_cvt4(w8_tanh4(w8_tail_fwd(W1h, b1s, h0b, wad, wave, q)));
    w8_xchg_put(smb, 1, own1, wave, lane);
    __syncthreads();                                               // exchange 1: h1
    w8_xchg_get(smb, 1, h1b, lane);
    f32x4 gq;                                                      // wo (1 - h2^2), block `wave`
    float dlda = 0.0f;
    {
      const f32x4 t2 = w8_tanh4(w8_tail_fwd(W2h, b2s, h1b, wad, wave, q));
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wos + 16 * wave + 4 * q);
      const f32x4 pw = t2 * wv;
      const float part = w8_sum_q((pw[0] + pw[1]) + (pw[2] + pw[3]));
      if (GRADS) {
        w8_xchg_put(smb, 2, w8_cvt4(t2), wave, lane);               // (h2: read back by the wave that sums d(wo))
        gq = wv - wv * (t2 * t2);
      }
      float* tlog = reinterpret_cast<float*>(smb + WO_TLOG);
      if (q == 0) tlog[16 * wave + r] = part;
      __syncthreads();                                             // exchange 2: h2 pieces + partial logits
      float a = bo;
      {
        float sacc = 0.0f;
#pragma unroll
        for (int w = 0; w < W8_WAVES; ++w) sacc += tlog[16 * w + r];
        a += sacc;
      }
      float ll, locv;
      if (LIK == PV_LIK_BERNOULLI) {
        const float pr = w8_rcp(1.0f + w8_exp(-a));
        const float pc = fminf(fmaxf(pr, BERN_EPS), 1.0f - BERN_EPS);
        // -BCEWithLogits(lg, x) with lg = logit(pc) (torch: probs_to_logits, then binary_cross_entropy_with_logits), written with
        // the identities 1 + exp(-|lg|) = 1 / max(pc, 1 - pc) and sigmoid(lg) = pc: the two logarithms lg is made of serve the
        // softplus term too, and the row's dependent chain is exp -> rcp -> 2 log instead of seven transcendentals (round 5)
        const float lpc = w8_log(pc), l1pc = w8_log(1.0f - pc);
        const float lg = lpc - l1pc;
        ll = -(fmaxf(lg, 0.0f) - lg * xv - fmaxf(lpc, l1pc));
        const float mask = (pr >= BERN_EPS && pr <= 1.0f - BERN_EPS) ? 1.0f : 0.0f;
        dlda = (pc - xv) * mask;
        locv = pr;
      } else if (LIK == PV_LIK_CBERNOULLI) {
        pv_cbern(a, xv, ll, dlda, locv);
      } else {
        const float pr = f.sigmoid_out ? w8_rcp(1.0f + w8_exp(-a)) : a;
        const float d = xv - pr;
        ll = -(d * d) / (2.0f * f.sig * f.sig) - w8_log(f.sig) - LOG_SQRT_2PI;
        dlda = -d / (f.sig * f.sig) * (f.sigmoid_out ? pr * (1.0f - pr) : 1.0f);
        locv = pr;
      }
      dlda *= swv;
      if (q == 0) {
        if (wave == 0) {
          if (a_llrow) a_llrow[row] = ll;
          if (a_loc) a_loc[row] = locv;
          if (GRADS) dbo += dlda;
        }
        if (GRADS) inf_dl[r] = dlda;
      }
    }
    if (GRADS) {
      bf16x4 pA[8];
      // ---- dpre2 = dlda wo (1 - h2^2), block `wave` -> every wave ----
      w8_xchg_put(smb, 3, w8_cvt4(gq * dlda), wave, lane);
      __syncthreads();                                             // exchange 3: dpre2
      w8_xchg_get(smb, 3, pA, lane);
      if (wave == 0) {                                             // rows 0 .. 15 of the weight-gradient staging
        w8_stage_store(sA, pA, r, q);
        w8_stage_store(sB, h1b, r, q);
      } else if (wave == 1) {                                      // rows 16 .. 31: zero gradient rows (one k-step is 32 rows)
        bf16x4 z8[8];
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) z8[jb] = w8_zero4();
        w8_stage_store(sA, z8, 16 + r, q);
        w8_stage_store(sB, h1b, 16 + r, q);
      } else if (wave == 3) {
        // d(wo) += sum_rows dlda h2 in this wave's own rows of staging A (rows 48 .. 63: unused by the tail's staging)
        bf16x4 h2b[8];
        w8_xchg_get(smb, 2, h2b, lane);
        w8_wait_lgkm0();
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(inf_dl + 4 * q);
        bf16x4 bw = w8_zero4();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __bf16 hi, lo;
          fb_split(d4[i], hi, lo);
          bw[i] = r == 3 ? hi : (r == 4 ? lo : (__bf16)0.0f);
        }
        w8_colsum_mfma(sA, h2b, bw, accS, wave, r, q);
      }
      // ---- dgrad of layer 2, block `wave` ----
      const f32x4 d1 = w8_mul_dtanh4(w8_tail_dgrad(W2h, pA, wad, wave, r), own1);      // -C dpre1
      w8_xchg_put(smb, 4, w8_cvt4(d1), wave, lane);
      __syncthreads();                                             // exchange 4: dpre1 (+ the staged rows of round 1)
      w8_xchg_get(smb, 4, pA, lane);
      w8_wgrad_consume(sA, sB, accW2, accB2, wave, r, q, 1);
      // ---- dgrad of layer 1, block `wave` ----
      const f32x4 d0 = w8_mul_dtanh4(w8_tail_dgrad(W1h, pA, wad, wave, r), own0);      // C^2 dpre0
      w8_xchg_put(smb, 5, w8_cvt4(d0), wave, lane);
      __syncthreads();                                             // exchange 5: dpre0; round 1 consumed everywhere
      if (wave == 0) {
        w8_stage_store(sA, pA, r, q);
        w8_stage_store(sB, h0b, r, q);
      } else if (wave == 1) {
        bf16x4 z8[8];
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) z8[jb] = w8_zero4();
        w8_stage_store(sA, z8, 16 + r, q);
        w8_stage_store(sB, h0b, 16 + r, q);
      } else if (wave == 2) {
        // coordinate layer backward, row-local part
        bf16x4 p0[8];
        w8_xchg_get(smb, 5, p0, lane);
        f32x4 dd = {0.0f, 0.0f, 0.0f, 0.0f};
        const bf16x8* ttab = reinterpret_cast<const bf16x8*>(smb + WO_TTAB) + lane;
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) dd = MFMA32(ttab[64 * mm], w8_cat(p0[2 * mm], p0[2 * mm + 1]), dd);
        if (q == 0) {
          const float d0_ = (dd[0] + dd[1]) * W8_RC2, d1_ = (dd[2] + dd[3]) * W8_RC2;
          a_rowtp[row] = sc * (d1_ * u0c - d0_ * u1c);
          a_rowtp[a_M + row] = d0_ * u0c + d1_ * u1c;
          a_rowtp[2 * a_M + row] = d0_;
          a_rowtp[3 * a_M + row] = d1_;
        }
      } else if (wave == 3) {
        // dL/d(hz[b]) = sum_rows dpre0, dWc_k = sum_rows dpre0 x'_k in this wave's own rows of staging A
        bf16x4 p0[8];
        w8_xchg_get(smb, 5, p0, lane);
        if (bu != cur_b) {
          if (cur_b >= 0) flush_hz(cur_b);
          cur_b = bu;
        }
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(inf_x0 + 4 * q);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(inf_x1 + 4 * q);
        bf16x4 bc_ = w8_zero4();
        const bool use1 = r == 2 || r == 6, lo_col = r >= 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __bf16 hi, lo;
          fb_split(use1 ? a1[i] : a0[i], hi, lo);
          __bf16 v = lo_col ? lo : hi;
          if (r == 0) v = (__bf16)1.0f;
          if (r == 3 || r == 4 || r > 6) v = (__bf16)0.0f;
          bc_[i] = v;
        }
        w8_colsum_mfma(sA, p0, bc_, accS, wave, r, q);
      }
      __syncthreads();                                             // round 2 staged
      w8_wgrad_consume(sA, sB, accW1, accB1, wave, r, q, 1);
    }
    asm volatile("; W8_TAIL_END");
  }
  if (!GRADS) return;
  W8_STAMP_K(2);

  // (FOLD with dhz_out: one image per workgroup — no slot was ever published; the waves' partials are summed with the column sums below)
  // (from here on `e` is read through an opaque pointer into the kernarg segment: named directly, the fields the epilogue needs are
  //  fetched at kernel entry and carried — spilled — through the tile loop)
  const PvEncFoldArg ep_ = pv_kernarg_fold();
  const bool own_dhz = FOLD && f.dhz_out != nullptr;
  if (cur_b >= 0 && !own_dhz) flush_hz(cur_b);
  // Order of the epilogue (round 6, third cut): barrier -> every LOAD the rest of it needs -> the record stores -> LDS work.  The
  // records are 71 KB per workgroup, 18 MB chip-wide within a microsecond: the memory pipeline takes ~3.5 us to drain them, and a
  // load issued BEHIND them waits for that (scripts/gpu_trace_w8.py: 7.3 k cycles on ten row-sum loads); requested ahead of them,
  // the loads' data arrives while the stores drain under the column sums and the latent backward below.
  W8_STAMP_E(0);
  __syncthreads();                                     // (every wave is past the last tile: its rows' outputs are in L2, the staging area is free)
  W8_STAMP_E(1);
  // (round 6, FOLD: the image's per-row outputs are complete in L2 — the barrier above drains every wave's stores — and are requested
  //  HERE, so that their round trip runs under the column-sum phase below; they are added up behind its barrier)
  // (thread n takes rows 2n, 2n + 1 — one 8-byte load per quantity: a load instruction costs this CU ~10 cycles of issue whatever its
  //  width, and the epilogue is made of little else)
  float rsv[2][5] = {{0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};
  const bool fold_rs = FOLD && f.part_rs && f.N <= 2 * W8_THREADS && f.N % 2 == 0;
  if (fold_rs) {
    const int64_t r0 = (int64_t)g * f.N;
    if (2 * tid < f.N) {
      auto ld2 = [](const float* p_, float& a, float& b) {
        const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (past this CU's L1)
        a = __uint_as_float((unsigned)v); b = __uint_as_float((unsigned)(v >> 32));
      };
      ld2(f.llrow + r0 + 2 * tid, rsv[0][0], rsv[1][0]);
#pragma unroll
      for (int c = 0; c < 4; ++c) ld2(f.rowtp + (int64_t)c * f.M + r0 + 2 * tid, rsv[0][1 + c], rsv[1][1 + c]);
    }
  }
  W8_STAMP_E(10);
  float wzv[4] = {0.0f, 0.0f, 0.0f, 0.0f};                           // (FOLD: fc_latent's row of this thread's hidden unit, for dL/dz below)
  if (FOLD && f.dhz_out && f.dzc_out && tid < FD_H) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wzv[i] = i < ep_->lat_in ? ep_->Wz[(int64_t)tid * ep_->lat_in + i] : 0.0f;
  }
  // (third cut, PvEncFold::chain) the image's latent backward and encoder chain follow below: their operands — the head's and the
  // second hidden layer's weights (L2), the image's own activations, sample and noise — are requested HERE, three barriers and the
  // column sums ahead of their first use.  (What this workgroup itself wrote in its prologue is read past the CU's L1.)
  const bool own_chain = CHAIN && ep_->chain && f.part_rs && f.dhz_out && f.dzc_out;
  float ch_whd[16], ch_a1 = 0.0f, ch_a0 = 0.0f, ch_z = 0.0f, ch_sig = 1.0f, ch_ep = 0.0f, ch_sp = 0.0f;
  f32x4 ch_w1[8];
  if (own_chain) {
    auto ldc = [](const float* p_) { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const int ho = ep_->head.out_dim;
    if (tid < FD_H) {
      const float* Wh = ep_->params + ep_->head.w_off + tid;
#pragma unroll
      for (int o = 0; o < 16; ++o) ch_whd[o] = Wh[(o < ho ? o : ho - 1) * FD_H];          // (clamped: used below only for o < out_dim)
      ch_a1 = ldc(ep_->eact1 + (int64_t)g * FD_H + tid);
      ch_a0 = ldc(ep_->eact0 + (int64_t)g * FD_H + tid);
    }
    // (second hidden layer: thread (c = tid & 31, jg = tid >> 5) holds W1[8 jg .. 8 jg + 7][4c .. 4c + 3] — eight 16-byte loads)
    const float* W1c = ep_->params + ep_->enc1.w_off + 4 * (tid & 31) + (8 * (tid >> 5)) * FD_H;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) ch_w1[jj] = *reinterpret_cast<const f32x4*>(W1c + jj * FD_H);
    if (lane < ep_->z_dim && wave < 2) {                  // (waves 0 and 1 each run the head backward for themselves below)
      ch_z = ldc(ep_->z + (int64_t)g * ep_->z_dim + lane);
      ch_sig = ldc(ep_->z_scale + (int64_t)g * ep_->z_dim + lane);
      ch_ep = ep_->eps[(int64_t)g * ep_->z_dim + lane];
      ch_sp = ldc(ep_->head_out + (int64_t)g * ep_->ldh + ep_->z_dim + lane);
    }
  }
  // ---- the workgroup's gradient record (pv_sdec_fused.h: FD_REC) ----
  W8_STAMP_E(11);
  {
    const int jp = wave >> 1, kh = wave & 1;
    // LANE-NATIVE PACKED (pv_sdec_fused.h PV_REC_LANE_BF16): per accumulator block ONE 16-byte store per lane, 1 KB contiguous
    // per instruction — {W1 rows (i0, i1), W1 (i2, i3), W2 (i0, i1), W2 (i2, i3)} as bf16 pairs (round to nearest even): 8 store
    // instructions per wave where the row-major fp32 form issued 128 four-byte ones
    auto pack2 = [](float a, float b) {
      typedef __bf16 bf2_ __attribute__((ext_vector_type(2)));
      bf2_ h; h[0] = (__bf16)a; h[1] = (__bf16)b;
      return __builtin_bit_cast(unsigned, h);
    };
    if (f.ablate & 1024) {                       // (experiments build: the row-major fp32 form, for the A/B of profiles/r06*_records_ab.txt)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // C/D layout: lane (col = r, q), reg i -> dW[32jp + 16 (s ^ kh) + 4q + i][64kh + 16o + r]  (w8_wgrad_consume's rotation)
            const int e = (32 * jp + 16 * (s_ ^ kh) + 4 * q + i) * FD_H + 64 * kh + 16 * o + r;
            rec[e] = accW1[s_][o][i] * -W8_RC;                        // (accW1 / accB1 hold -C dW1 / -C db1)
            rec[FD_H * FD_H + e] = accW2[s_][o][i];
          }
    } else {
      typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
      u32x4_* rec4 = reinterpret_cast<u32x4_*>(rec);
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          u32x4_ v;
          v[0] = pack2(accW1[s_][o][0] * -W8_RC, accW1[s_][o][1] * -W8_RC);   // (accW1 / accB1 hold -C dW1 / -C db1)
          v[1] = pack2(accW1[s_][o][2] * -W8_RC, accW1[s_][o][3] * -W8_RC);
          v[2] = pack2(accW2[s_][o][0], accW2[s_][o][1]);
          v[3] = pack2(accW2[s_][o][2], accW2[s_][o][3]);
          rec4[((wave * 2 + s_) * 4 + o) * 64 + lane] = v;
        }
    }
    if (r == 0) {
      const int j0 = 32 * jp + 16 * kh;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rec[2 * FD_H * FD_H + j0 + 4 * q + i] = accB1[i] * -W8_RC;
        rec[2 * FD_H * FD_H + FD_H + j0 + 4 * q + i] = accB2[i];
      }
    }
  }
  W8_STAMP_E(12);
  // per-wave column sums -> LDS -> summed over the waves in ascending order.  (The barriers from here on wait for LDS traffic only
  // — pv_lds_barrier — not for the record stores in flight: everything the waves hand each other below goes through LDS.)
  {
    float* scr = reinterpret_cast<float*>(smb + WO_SA);            // [wave][n][128] floats = 64 KB
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
      *reinterpret_cast<f32x4*>(scr + ((wave * 16 + r) * FD_H) + 16 * jb + 4 * q) = accS[jb];
  }
  const float tb = pv_wave_sum(dbo);
  if (lane == 0) red[wave] = tb;
  W8_STAMP_E(2);
  pv_lds_barrier();
  W8_STAMP_E(3);
  // ---- the column sums over the waves (threads 0 .. 127), then — where the workgroup owns a whole image (FOLD) — the image's row
  // sums, dL/d(hz) and dL/dz for the latent backward (round 6: PvFused::part_rs / dhz_out / dzc_out), one more barrier for all three ----
  float dhz_j = 0.0f;                                                // this thread's dL/d(hz[g][tid]) (tid < 128)
  if (tid < FD_H) {
    const float* scr = reinterpret_cast<const float*>(smb + WO_SA);
    float vo = 0.0f, v0 = 0.0f, v1 = 0.0f, vh = 0.0f;
#pragma unroll
    for (int w = 0; w < W8_WAVES; ++w) {
      const float* s_ = scr + (w * 16) * FD_H + tid;
      vh += s_[0];                                                   // (column 0: the wave's dL/d(hz) partial, times C^2)
      v0 += s_[1 * FD_H] + s_[5 * FD_H];
      v1 += s_[2 * FD_H] + s_[6 * FD_H];
      vo += s_[3 * FD_H] + s_[4 * FD_H];
    }
    dhz_j = vh * W8_RC2;
    if (own_dhz) f.dhz_out[(int64_t)g * FD_H + tid] = dhz_j;
    rec[2 * FD_H * FD_H + 2 * FD_H + tid] = v0 * W8_RC2;
    rec[2 * FD_H * FD_H + 3 * FD_H + tid] = v1 * W8_RC2;
    rec[2 * FD_H * FD_H + 4 * FD_H + tid] = vo;
  }
  const bool own_dzc = own_dhz && f.dzc_out != nullptr;
  if (FOLD && (f.part_rs || own_dzc)) {
    // ONE image per workgroup.  Row sums: its rows' five per-row outputs {ll, d(phi), d(scale), d(tx), d(ty)} were all written by
    // this workgroup (requested above, behind the barrier that drains every wave's stores) — summed into the image's first row-sum
    // slot, so that the latent backward adds slots instead of loading and block-reducing 5 x N rows (the running sums the 4-wave
    // kernels keep in registers do not fit this kernel: profiles/r06h_row_sums_ab.txt).  Fixed order: thread n takes rows n,
    // n + 512; wave sums; waves 0..7.  dL/dz[i] = sum_j dL/d(hz[j]) Wz[j][i] (coord_latent.fc_latent, nets/fc.py:217,230): threads
    // 0 .. 127 hold dL/d(hz[j]); wave sums of waves 0, 1.
    if (f.part_rs) {
      const int64_t r0 = (int64_t)g * f.N;
      float a5[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (fold_rs) {
#pragma unroll
        for (int c = 0; c < 5; ++c) a5[c] = rsv[0][c] + rsv[1][c];       // (rows tid, tid + 512: requested above)
      } else {
        for (int n = tid; n < f.N; n += W8_THREADS) {
          a5[0] += __hip_atomic_load(f.llrow + r0 + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (past this CU's L1)
#pragma unroll
          for (int c = 0; c < 4; ++c) a5[1 + c] += __hip_atomic_load(f.rowtp + (int64_t)c * f.M + r0 + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int c = 0; c < 5; ++c) a5[c] = pv_wave_sum(a5[c]);
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 5; ++c) red[8 + 8 * c + wave] = a5[c];
      }
    }
    if (own_dzc && wave < 2) {
      for (int i = 0; i < ep_->lat_in; ++i) {
        const float pz = pv_wave_sum(dhz_j * (i < 4 ? wzv[i] : ep_->Wz[(int64_t)tid * ep_->lat_in + i]));
        if (lane == 0) info[16 * wave + i] = pz;
      }
    }
    W8_STAMP_E(4);
    pv_lds_barrier();
    W8_STAMP_E(5);
    float* cs = reinterpret_cast<float*>(smb + WO_SA);             // (the column-sum scratch is free behind the barrier above)
    if (f.part_rs && tid < 5) {
      float v = 0.0f;
      for (int w = 0; w < W8_WAVES; ++w) v += red[8 + 8 * tid + w];
      f.part_rs[((int64_t)g * f.kmax) * PV_RS_W + tid] = v;
    }
    if (own_dzc && tid < ep_->lat_in) f.dzc_out[(int64_t)g * ep_->lat_in + tid] = info[tid] + info[16 + tid];
    if (own_chain) {
      // ---- the image's latent backward (pv_elementwise.hip: pv_latent_bwd_block, for the one sample this workgroup owns): head
      // backward from the row sums {ll, d(phi), d(scale), d(tx), d(ty)} and dL/dz, then the encoder's input-gradient chain
      //   edp1 = (dhead Whead) * act'(eact1),  edp0 = (edp1 W1) * act'(eact0)     (nn.Linear backward)
      // Three phases, two barriers.  1: waves 0 and 1 — each runs the head backward for itself (lane i: coordinate i; the sums it
      // needs straight from the waves' partials in LDS), hands dhead round with v_readlane, and takes 64 of edp1's 128 entries.
      // 2: every thread contracts its 8 x 4 block of W1 with edp1.  3: threads 0 .. 127 add the 16 partial sums of their column.
      // cs: [128 ..] edp1, [256 + 128 jg ..] the partial sums of row group jg
      if (wave < 2) {
        // (wave-private LDS words csw[0 .. 63]: a wave's LDS operations complete in order, so lanes of ONE wave hand values to each
        //  other through them with a wait, no barrier.  [0..4] the five row sums, [8..] dL/dz content, [16..31] dhead, zero past its end)
        float* csw = cs + 64 * wave;
        if (lane < 5) {
          float v = 0.0f;
#pragma unroll
          for (int w = 0; w < W8_WAVES; ++w) v += red[8 + 8 * lane + w];
          csw[lane] = v;
          if (tid == 0) ep_->llb[g] = v;
        }
        if (lane >= 8 && lane < 8 + ep_->lat_in) csw[lane] = info[lane - 8] + info[16 + lane - 8];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PvHeadBwd hb{};
        hb.coord_dim = ep_->coord_dim; hb.has_r = ep_->has_r; hb.has_t = ep_->has_t; hb.has_s = ep_->has_s;
        hb.tp0 = ep_->tp0; hb.tp1 = ep_->tp1; hb.sc_prior = ep_->sc_prior;
        if (lane < 16) {
          float g_ = 0.0f, ds_ = 0.0f;
          if (lane < ep_->z_dim) {
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
        for (int o = 0; o < 16; ++o) v += csw[16 + o] * ch_whd[o];     // (head rows past out_dim: a clamped weight times the zero above)
        v *= pv_act_grad2(ch_a1, 0.0f, ep_->enc1.act);
        ep_->edp1[(int64_t)g * FD_H + tid] = v;
        cs[128 + tid] = v;
      }
      pv_lds_barrier();
      W8_STAMP_E(6);
      {
        const int c = tid & 31, jg = tid >> 5;
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(cs + 128 + 8 * jg), e1 = *reinterpret_cast<const f32x4*>(cs + 132 + 8 * jg);
        f32x4 acc = ch_w1[0] * e0[0];
        acc += ch_w1[1] * e0[1]; acc += ch_w1[2] * e0[2]; acc += ch_w1[3] * e0[3];
        acc += ch_w1[4] * e1[0]; acc += ch_w1[5] * e1[1]; acc += ch_w1[6] * e1[2]; acc += ch_w1[7] * e1[3];
        *reinterpret_cast<f32x4*>(cs + 256 + 128 * jg + 4 * c) = acc;
      }
      pv_lds_barrier();
      W8_STAMP_E(7);
      if (tid < FD_H) {
        float y = 0.0f;
#pragma unroll
        for (int jg = 0; jg < 16; ++jg) y += cs[256 + 128 * jg + tid];
        y *= pv_act_grad2(ch_a0, 0.0f, ep_->enc0.act);
        ep_->edp0[(int64_t)g * FD_H + tid] = y;
      }
    }
  }
  if (tid == 0) {
    float v = 0.0f;
    for (int w = 0; w < W8_WAVES; ++w) v += red[w];
    rec[2 * FD_H * FD_H + 5 * FD_H] = v;
  }
  W8_STAMP_K(3);
}

// the guide can ride in this launch when every workgroup's unit range is exactly one image (batch == grid: BASELINE's batch
// 256 on 256 CUs; with more images per workgroup their guides would run one after the other in front of the tile loop)
bool pv_sdec_fused_w8_fold_ok(const PvFused& f, int grid) {
  if (f.N % FD_UNIT != 0 || f.N > 1024 || f.N % 4 != 0 || grid < 1 || f.B % grid != 0 || f.x_units != 0) return false;
  const int ipw = f.B / grid;
  return ipw == 1 && f.units == (int64_t)f.B * (f.N / FD_UNIT);      // ONE image per workgroup: batch == grid
}

int pv_sdec_fused_w8_launch(const PvFused& f_in, int grid, bool grads, hipStream_t s, const PvEncFold* fold) {
  PvFused f = f_in;
  static const int ablate = pv_exp_int("PV_FD_ABLATE", 0) & 1024;    // (experiments build: fp32 records, see the record stores)
  f.ablate = ablate;
  const size_t lds = W8_LDS_BYTES;
  const void* fn = nullptr;
  PvEncFold e{};
  if (fold) {
    if (!pv_sdec_fused_w8_fold_ok(f, grid)) return PV_EINVAL;
    e = *fold;
    e.img_per_wg = f.B / grid;
    // (the shared first layer is written for 8 groups of 32 workgroups, four of the layer's 128 rows each)
    if (e.coop && !(grid == 256 && e.enc0.out_dim == FD_H && e.coop_flags && e.chain)) e.coop = 0;
#ifndef PV_EXPERIMENTS
    e.coop = 0;
#endif
  }
#ifdef PV_EXPERIMENTS                  // (build 3 — the shared first layer, measured slower: profiles/r06k_coop_first_layer.txt — exists in the experiments library only)
#define W8_COOP_BUILD 3
#else
#define W8_COOP_BUILD 2
#endif
#define W8_PICK(G, L) fn = !fold ? reinterpret_cast<const void*>(&pv_sdec_w8_kernel<G, L, 0>)                     \
                               : (G && e.chain && e.coop) ? reinterpret_cast<const void*>(&pv_sdec_w8_kernel<G, L, G ? W8_COOP_BUILD : 1>) \
                               : (G && e.chain) ? reinterpret_cast<const void*>(&pv_sdec_w8_kernel<G, L, G ? 2 : 1>) \
                                                : reinterpret_cast<const void*>(&pv_sdec_w8_kernel<G, L, 1>)
  if (grads) {
    if (f.lik == PV_LIK_BERNOULLI) W8_PICK(true, PV_LIK_BERNOULLI);
    else if (f.lik == PV_LIK_GAUSSIAN) W8_PICK(true, PV_LIK_GAUSSIAN);
    else W8_PICK(true, PV_LIK_CBERNOULLI);
  } else {
    if (f.lik == PV_LIK_BERNOULLI) W8_PICK(false, PV_LIK_BERNOULLI);
    else if (f.lik == PV_LIK_GAUSSIAN) W8_PICK(false, PV_LIK_GAUSSIAN);
    else W8_PICK(false, PV_LIK_CBERNOULLI);
  }
#undef W8_PICK
  PV_TRY(pv_set_dynamic_lds(fn, (int)lds));          // (per device and kernel)
  void* args[] = {&f, &e};
  hipError_t e2 = hipLaunchKernel(fn, dim3(grid), dim3(W8_THREADS), args, lds, s);
  if (e2 != hipSuccess) return (int)e2;
  PV_LAUNCH_CHECK();
  return 0;
}
