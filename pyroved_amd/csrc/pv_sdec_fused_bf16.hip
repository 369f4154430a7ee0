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
// tile: 243 MB of scratch writes per launch and 66 % of wave time in s_waitcnt).  A wave carries TWO 16-row
// units at a time: every weight operand read from LDS feeds both units' MFMAs (half the LDS traffic) and the
// two units' accumulators give the in-order wave independent MFMA chains.  As in pv_sdec_fused.hip:
//   * layers are computed transposed, D[j][r] = sum_k W[j][k] h[r][k], so the 16x16 C/D layout of one layer is
//     the B-operand layout of the next: activations stay in registers through forward and dgrad;
//   * a wave owns two 16x128 slices of dW1 and dW2 in accumulators for the whole kernel and the workgroup
//     exchanges (dpre, h) through LDS, one unit at a time, for the wgrad;
//   * per-workgroup partial gradients are reduced in a fixed order afterwards (no float atomics).
// What the bf16 split changes:
//   * weights live in LDS as bf16 hi and lo images, row-major [128][136]; the forward reads a lane's A operand
//     with two ds_read_b64 (k = 32m+4q.. and 32m+16+4q..), the dgrad reads the TRANSPOSED operand from the same
//     image with ds_read_b64_tr_b16 (hardware 4x16 transpose), so one image serves both orientations;
//   * activations are split to (hi, lo) on the fly (3 VALU per element);
//   * the wgrad stages bf16 hi/lo rows and uses v_mfma_f32_16x16x16_bf16 (k = the unit's 16 rows), operands by
//     ds_read_b64_tr_b16; bias gradients ride along as an MFMA against a column of ones.
#include "pv_sdec_fused.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4_ lds_short4;

#define FB_WAVES 4               // waves per workgroup (one per SIMD)
#ifndef UPW
#define UPW 1                    // units (16 rows) a wave carries at a time
#endif
#define TILE_UNITS (FB_WAVES * UPW)   // units per workgroup tile
#define LDB 136                  // bf16 elements per LDS row of the weight / staging images (272 B)
#define W_IMG (FD_H * LDB)       // elements of one weight image
#define LDS2 144                 // staging images: 72-dword rows -> the 4x16 transposing reads are conflict-free
#define ST_IMG (FD_UNIT * LDS2)  // elements of one staging image
// byte offsets in dynamic LDS
#define BO_W1H 0
#define BO_W1L (BO_W1H + 2 * W_IMG)
#define BO_W2H (BO_W1L + 2 * W_IMG)
#define BO_W2L (BO_W2H + 2 * W_IMG)
#define BO_SAH (BO_W2L + 2 * W_IMG)       // staged dpre: hi, lo
#define BO_SAL (BO_SAH + 2 * ST_IMG)
#define BO_SBH (BO_SAL + 2 * ST_IMG)      // staged h: hi, lo
#define BO_SBL (BO_SBH + 2 * ST_IMG)
#define BO_VEC (BO_SBL + 2 * ST_IMG)      // fp32 vectors: Wc0, Wc1, bc, wo, b1, b2 (128 each)
#define BO_INFO (BO_VEC + 6 * FD_H * 4)
#define BO_RED (BO_INFO + 256)
#define BO_DWO (BO_RED + 256)              // per-wave d(wo) partial sums: FB_WAVES x 128 floats
#define FB_LDS_BYTES (BO_DWO + FB_WAVES * FD_H * 4)
#define FB_THREADS (64 * FB_WAVES)

#define LOG_SQRT_2PI 0.91893853320467274178f
#define BERN_EPS 1.1920928955078125e-07f
#define FB_FENCE() __builtin_amdgcn_sched_barrier(0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float fb_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float fb_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fb_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fb_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// x -> (hi, lo) bf16 with hi + lo = x to ~2^-17 relative
__device__ __forceinline__ void fb_split(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

// two D-layout blocks (4 + 4 consecutive k of this lane) -> the lane's 8-element hi / lo operands
__device__ __forceinline__ void fb_split8(const f32x4& u, const f32x4& v, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    __bf16 h, l;
    fb_split(u[i], h, l); hi[i] = h; lo[i] = l;
    fb_split(v[i], h, l); hi[4 + i] = h; lo[4 + i] = l;
  }
}

__device__ __forceinline__ bf16x8 fb_cat(const bf16x4& a, const bf16x4& b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// LDS weight images are stored with their columns permuted inside every block of 32: logical column
// k = 32m + 16h + 4q + i (h in {0,1}, q in 0..3, i in 0..3) sits at physical column 32m + 8q + 4h + i, so that the
// 8 k's a lane feeds to one v_mfma_f32_16x16x32_bf16 ({32m+4q+i} and {32m+16+4q+i}: the C/D layout of the
// producing layer) are CONTIGUOUS: the forward A operand is one ds_read_b128 (a pair of ds_read_b64 gets fused
// into ds_read2_b64, whose 32-bank addressing runs at a quarter of the rate on this row stride).  Groups of 4
// consecutive logical columns stay contiguous, which is all the transposing dgrad read needs.
__device__ __forceinline__ int fb_pcol(int k) {
  return (k & ~31) | (((k >> 2) & 3) << 3) | (((k >> 4) & 1) << 2) | (k & 3);
}

__device__ __forceinline__ bf16x4 fb_tr(const __bf16* p) {
  const short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)p);
  return __builtin_bit_cast(bf16x4, v);
}

// forward layer for the wave's two units: out[u] = bias + W in[u]  (pre-activation on return).
// Stream: per k-block m (32 k's) split both units' inputs; per pair of output blocks read 4 weight operands
// (hi/lo x 2 blocks, double-buffered one group ahead) and issue 12 MFMAs (3 split terms x 2 blocks x 2 units).
__device__ __forceinline__ void fb_layer_fwd(const __bf16* __restrict__ Wh, const __bf16* __restrict__ Wl,
                                             const float* __restrict__ bs, const f32x4 (&in)[UPW][8],
                                             f32x4 (&out)[UPW][8], int r, int q) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bs + 16 * ob + 4 * q);
#pragma unroll
    for (int u = 0; u < UPW; ++u) out[u][ob] = bias;
  }
  const __bf16* ah = Wh + r * LDB + 8 * q;
  const __bf16* al = Wl + r * LDB + 8 * q;
  // operands are fetched TWO groups ahead (three rotating register sets): a group is only 6 MFMAs (~100 cycles),
  // shorter than the LDS latency, and this wave is alone on its SIMD
  bf16x8 wh[3][2], wl[3][2];
  auto load = [&](int g, bf16x8 (&h)[2], bf16x8 (&l)[2]) {
    const int m = g >> 2, op = (g & 3) * 2;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int off = 16 * (op + o) * LDB + 32 * m;
      h[o] = *reinterpret_cast<const bf16x8*>(ah + off);
      l[o] = *reinterpret_cast<const bf16x8*>(al + off);
    }
  };
  load(0, wh[0], wl[0]);
  load(1, wh[1], wl[1]);
  bf16x8 bh[UPW], bl[UPW];
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int m = g >> 2, op = (g & 3) * 2;
    if (g + 2 < 16) load(g + 2, wh[(g + 2) % 3], wl[(g + 2) % 3]);
    FB_FENCE();
    if ((g & 3) == 0) {
#pragma unroll
      for (int u = 0; u < UPW; ++u) fb_split8(in[u][2 * m], in[u][2 * m + 1], bh[u], bl[u]);
    }
    const bf16x8(&h)[2] = wh[g % 3];
    const bf16x8(&l)[2] = wl[g % 3];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int u = 0; u < UPW; ++u) out[u][op + o] = MFMA32(h[o], bh[u], out[u][op + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int u = 0; u < UPW; ++u) out[u][op + o] = MFMA32(h[o], bl[u], out[u][op + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int u = 0; u < UPW; ++u) out[u][op + o] = MFMA32(l[o], bh[u], out[u][op + o]);
    FB_FENCE();
  }
}

// dgrad for the wave's two units: out[u][k] = sum_j W[j][k] dp[u][j]; A = W^T via the transposing LDS read
__device__ __forceinline__ void fb_layer_dgrad(const __bf16* __restrict__ Wh, const __bf16* __restrict__ Wl,
                                               const f32x4 (&dp)[UPW][8], f32x4 (&out)[UPW][8], int r, int q) {
#pragma unroll
  for (int kb = 0; kb < 8; ++kb)
#pragma unroll
    for (int u = 0; u < UPW; ++u) out[u][kb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // lane i of 16-lane group q points at W[j0 + i/4][16*kb + 4*(i%4)], j0 = 32m + 4q (+16): after the transpose
  // lane k' holds W[j0 .. j0+3][16*kb + k']
  const int toff = (4 * q + (r >> 2)) * LDB + 8 * (r & 3);
  const __bf16* ah = Wh + toff;
  const __bf16* al = Wl + toff;
  bf16x8 wh[3][2], wl[3][2];
  auto load = [&](int g, bf16x8 (&h)[2], bf16x8 (&l)[2]) {
    const int m = g >> 2, kp = (g & 3) * 2;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int off = 32 * m * LDB + 32 * ((kp + o) >> 1) + 4 * ((kp + o) & 1);
      h[o] = fb_cat(fb_tr(ah + off), fb_tr(ah + off + 16 * LDB));
      l[o] = fb_cat(fb_tr(al + off), fb_tr(al + off + 16 * LDB));
    }
  };
  load(0, wh[0], wl[0]);
  load(1, wh[1], wl[1]);
  bf16x8 bh[UPW], bl[UPW];
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int m = g >> 2, kp = (g & 3) * 2;
    if (g + 2 < 16) load(g + 2, wh[(g + 2) % 3], wl[(g + 2) % 3]);
    FB_FENCE();
    if ((g & 3) == 0) {
#pragma unroll
      for (int u = 0; u < UPW; ++u) fb_split8(dp[u][2 * m], dp[u][2 * m + 1], bh[u], bl[u]);
    }
    const bf16x8(&h)[2] = wh[g % 3];
    const bf16x8(&l)[2] = wl[g % 3];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int u = 0; u < UPW; ++u) out[u][kp + o] = MFMA32(h[o], bh[u], out[u][kp + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int u = 0; u < UPW; ++u) out[u][kp + o] = MFMA32(h[o], bl[u], out[u][kp + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int u = 0; u < UPW; ++u) out[u][kp + o] = MFMA32(l[o], bh[u], out[u][kp + o]);
    FB_FENCE();
  }
}

__device__ __forceinline__ void fb_tanh8(f32x4 (&v)[8]) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[ob][i] = fb_tanh(v[ob][i]);
}

__device__ __forceinline__ void fb_mul_dtanh(f32x4 (&out)[8], const f32x4 (&h)[8]) {
#pragma unroll
  for (int kb = 0; kb < 8; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[kb][i] *= 1.0f - h[kb][i] * h[kb][i];
}

// rows of one unit -> (hi, lo) bf16 in registers, done by every wave BEFORE the exchange loop so that the
// owner's turn inside the loop is only 16 ds_write_b64 (the other waves wait at the barrier meanwhile)
__device__ __forceinline__ void fb_presplit(const f32x4 (&v)[8], bf16x4 (&h)[8], bf16x4 (&l)[8]) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) { __bf16 a, b; fb_split(v[jb][i], a, b); h[jb][i] = a; l[jb][i] = b; }
}
__device__ __forceinline__ void fb_stage_store(__bf16* __restrict__ sh, __bf16* __restrict__ sl, const bf16x4 (&h)[8],
                                               const bf16x4 (&l)[8], int r, int q) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    *reinterpret_cast<bf16x4*>(sh + r * LDS2 + 16 * jb + 4 * q) = h[jb];
    *reinterpret_cast<bf16x4*>(sl + r * LDS2 + 16 * jb + 4 * q) = l[jb];
  }
}

// wgrad for the wave's two 16-row slices (rows 16*(2*wave + s) ..) over the staged unit's 16 rows:
//   dW[j][k] += sum_rows dpre[row][j] h[row][k];   db[j] += sum_rows dpre[row][j]  (MFMA against ones)
__device__ __forceinline__ void fb_wgrad_consume(const __bf16* sah, const __bf16* sal, const __bf16* sbh,
                                                 const __bf16* sbl, f32x4 (&accW)[2][8], f32x4 (&accB)[2], int wave,
                                                 int r, int q) {
  const int toff = (4 * q + (r >> 2)) * LDS2 + 4 * (r & 3);
  short4_ a_h[2], a_l[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    a_h[s] = __builtin_bit_cast(short4_, fb_tr(sah + toff + 16 * (2 * wave + s)));
    a_l[s] = __builtin_bit_cast(short4_, fb_tr(sal + toff + 16 * (2 * wave + s)));
  }
  const short one = 0x3f80;                           // bf16 1.0
  const short4_ ones = {one, one, one, one};
  short4_ bh[2][2], bl[2][2];
  auto load = [&](int kp, short4_ (&h)[2], short4_ (&l)[2]) {
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      h[o] = __builtin_bit_cast(short4_, fb_tr(sbh + toff + 16 * (kp + o)));
      l[o] = __builtin_bit_cast(short4_, fb_tr(sbl + toff + 16 * (kp + o)));
    }
  };
  load(0, bh[0], bl[0]);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    accB[s] = MFMA16(a_h[s], ones, accB[s]);
    accB[s] = MFMA16(a_l[s], ones, accB[s]);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int kp = 2 * g;
    if (g + 1 < 4) load(kp + 2, bh[(g + 1) & 1], bl[(g + 1) & 1]);
    FB_FENCE();
    const short4_(&h)[2] = bh[g & 1];
    const short4_(&l)[2] = bl[g & 1];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int s = 0; s < 2; ++s) accW[s][kp + o] = MFMA16(a_h[s], h[o], accW[s][kp + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int s = 0; s < 2; ++s) accW[s][kp + o] = MFMA16(a_h[s], l[o], accW[s][kp + o]);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int s = 0; s < 2; ++s) accW[s][kp + o] = MFMA16(a_l[s], h[o], accW[s][kp + o]);
    FB_FENCE();
  }
}

__device__ __forceinline__ float fb_sum_q(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// sum over the 16 lanes of a DPP row (= one q group), result in every lane: quad swaps, then the two mirrors
template <int CTRL>
__device__ __forceinline__ float fb_dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float fb_sum_r(float v) {
  v += fb_dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
  v += fb_dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
  v += fb_dpp_mov<0x141>(v);     // row_half_mirror
  v += fb_dpp_mov<0x140>(v);     // row_mirror
  return v;
}

// phase-timing trace (profiling only, enabled by PV_FD_ABLATE bit 256): shader-clock stamps of workgroup 0 /
// wave 0 for its first tiles; read back with pv_debug_read_trace()
__device__ long long fb_trace[256];
#define FB_STAMP(k)                                                                        \
  do {                                                                                     \
    if ((f.ablate & 256) && g == 0 && tid == 0 && tile_no < 4)                             \
      fb_trace[tile_no * 16 + (k)] = (long long)__builtin_readcyclecounter();              \
  } while (0)

template <bool GRADS>
__global__ __launch_bounds__(FB_THREADS, 1) void pv_sdec_fused_bf16_kernel(PvFused f) {
  extern __shared__ __attribute__((aligned(16))) char smb[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x, G = gridDim.x;
  __bf16* W1h = reinterpret_cast<__bf16*>(smb + BO_W1H);
  __bf16* W1l = reinterpret_cast<__bf16*>(smb + BO_W1L);
  __bf16* W2h = reinterpret_cast<__bf16*>(smb + BO_W2H);
  __bf16* W2l = reinterpret_cast<__bf16*>(smb + BO_W2L);
  __bf16* sAh = reinterpret_cast<__bf16*>(smb + BO_SAH);
  __bf16* sAl = reinterpret_cast<__bf16*>(smb + BO_SAL);
  __bf16* sBh = reinterpret_cast<__bf16*>(smb + BO_SBH);
  __bf16* sBl = reinterpret_cast<__bf16*>(smb + BO_SBL);
  float* vec = reinterpret_cast<float*>(smb + BO_VEC);
  float* info = reinterpret_cast<float*>(smb + BO_INFO);
  float* red = reinterpret_cast<float*>(smb + BO_RED);

  // ---- split the weights into bf16 hi / lo LDS images (once per kernel) ----
  for (int idx = tid; idx < FD_H * (FD_H / 4); idx += FB_THREADS) {
    const int row = idx >> 5, c4 = idx & 31;
    const f32x4 w1 = reinterpret_cast<const f32x4*>(f.W1)[idx];
    const f32x4 w2 = reinterpret_cast<const f32x4*>(f.W2)[idx];
    bf16x4 h1, l1, h2, l2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __bf16 a, b;
      fb_split(w1[i], a, b); h1[i] = a; l1[i] = b;
      fb_split(w2[i], a, b); h2[i] = a; l2[i] = b;
    }
    const int pc = fb_pcol(4 * c4);
    *reinterpret_cast<bf16x4*>(W1h + row * LDB + pc) = h1;
    *reinterpret_cast<bf16x4*>(W1l + row * LDB + pc) = l1;
    *reinterpret_cast<bf16x4*>(W2h + row * LDB + pc) = h2;
    *reinterpret_cast<bf16x4*>(W2l + row * LDB + pc) = l2;
  }
  for (int j = tid; j < FD_H; j += FB_THREADS) {
    vec[j] = f.Wc[j * f.cd];
    vec[FD_H + j] = f.cd == 2 ? f.Wc[j * 2 + 1] : 0.0f;
    vec[2 * FD_H + j] = f.bc[j];
    vec[3 * FD_H + j] = f.wo[j];
    vec[4 * FD_H + j] = f.b1[j];
    vec[5 * FD_H + j] = f.b2[j];
  }
  __syncthreads();
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
  float aWc0[2] = {0.0f, 0.0f}, aWc1[2] = {0.0f, 0.0f}, ahz[2] = {0.0f, 0.0f}, dbo = 0.0f;
  int cur_b = -1;
  const int upb = f.N / FD_UNIT;
  float* rec = f.part + (int64_t)g * FD_REC;
  // this wave's private d(wo) slots in the record's tail (slot `wave` for its first unit, `4 + wave` for its
  // second): written and re-read only by lanes (r == 0, q) — one thread per address, so plain same-thread
  // ordering suffices
  float* dwo_g = reinterpret_cast<float*>(smb + BO_DWO) + wave * FD_H;     // this wave's LDS slot
  if (r == 0) {
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) *reinterpret_cast<f32x4*>(dwo_g + 16 * jb + 4 * q) = f32x4{0, 0, 0, 0};
  }

  auto flush_hz = [&](int b) {
    // sample b's rows end (or the workgroup's do): publish this workgroup's partial dL/d(hz[b])
    const int64_t ub = (int64_t)b * upb;
    const int gfirst = (int)(((ub + 1) * G + f.units - 1) / f.units) - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float t = fb_sum_q(ahz[s]);
      if (q == 0) f.part_hz[((int64_t)b * f.kmax + (g - gfirst)) * FD_H + 16 * (2 * wave + s) + r] = t;
      ahz[s] = 0.0f;
    }
  };

  const int64_t u_lo = (int64_t)g * f.units / G, u_hi = (int64_t)(g + 1) * f.units / G;
  int tile_no = -1;
  for (int64_t ut = u_lo; ut < u_hi; ut += TILE_UNITS) {
    ++tile_no;
    FB_STAMP(0);
    const int nact = (int)((u_hi - ut) < TILE_UNITS ? (u_hi - ut) : TILE_UNITS);
    int opq = 0;
    asm volatile("" : "+v"(opq));       // loop-variant zero: keeps LICM from hoisting the LDS-resident vectors
    const float* Wc0 = vec + opq;
    const float* Wc1 = vec + FD_H + opq;
    const float* bcs = vec + 2 * FD_H + opq;
    const float* wos = vec + 3 * FD_H + opq;
    const float* b1s = vec + 4 * FD_H + opq;
    const float* b2s = vec + 5 * FD_H + opq;

    // per-unit row bookkeeping; an inactive unit (partial last tile) re-reads the tile's first unit and has its
    // dL/dlogit forced to zero, so everything it would contribute vanishes
    bool act[UPW];
    int bu[UPW];
    int64_t row[UPW];
    float x0[UPW], x1[UPW], u0c[UPW], u1c[UPW], sc[UPW];
    const float* hzb[UPW];
    f32x4 tA[UPW][8], tB[UPW][8], tC[UPW][8];
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
      act[u] = UPW * wave + u < nact;
      const int unit = (int)ut + (act[u] ? UPW * wave + u : 0);
      bu[u] = unit / upb;
      const int n = (unit - bu[u] * upb) * FD_UNIT + r;
      row[u] = (int64_t)unit * FD_UNIT + r;
      const float* t = f.tp + (int64_t)bu[u] * 8;
      if (f.cd == 2) {
        const float gx = f.grid[2 * n], gy = f.grid[2 * n + 1];
        u0c[u] = gx * t[0] - gy * t[1];
        u1c[u] = gx * t[1] + gy * t[0];
        sc[u] = t[2];
        x0[u] = u0c[u] * sc[u] + t[3];
        x1[u] = u1c[u] * sc[u] + t[4];
      } else {
        u0c[u] = f.grid[n]; u1c[u] = 0.0f; sc[u] = 1.0f;
        x0[u] = u0c[u] + t[3]; x1[u] = 0.0f;
      }
      hzb[u] = f.hz + (int64_t)bu[u] * FD_H;
    }
    auto coord_layer = [&](f32x4 (&h0)[8], int u) {
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        const int j = 16 * jb + 4 * q;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wc0 + j);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(Wc1 + j);
        const f32x4 bc = *reinterpret_cast<const f32x4*>(bcs + j);
        const f32x4 hz = *reinterpret_cast<const f32x4*>(hzb[u] + j);
#pragma unroll
        for (int i = 0; i < 4; ++i) h0[jb][i] = fb_tanh(w0[i] * x0[u] + w1[i] * x1[u] + bc[i] + hz[i]);
      }
    };

    const bool wave_active = UPW * wave < nact;
    if (wave_active) {
#pragma unroll
      for (int u = 0; u < UPW; ++u) coord_layer(tA[u], u);          // tA = h0
      FB_STAMP(1);
      fb_layer_fwd(W1h, W1l, b1s, tA, tB, r, q);
      FB_STAMP(2);
#pragma unroll
      for (int u = 0; u < UPW; ++u) fb_tanh8(tB[u]);                // tB = h1
      fb_layer_fwd(W2h, W2l, b2s, tB, tC, r, q);
      FB_STAMP(3);
#pragma unroll
      for (int u = 0; u < UPW; ++u) fb_tanh8(tC[u]);                // tC = h2
      // ---- output layer + likelihood (fp32) ----
#pragma unroll
      for (int u = 0; u < UPW; ++u) {
        float part = 0.0f;
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wos + 16 * jb + 4 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) part += tC[u][jb][i] * wv[i];
        }
        const float a = fb_sum_q(part) + bo;
        const float xv = f.x[row[u]];
        float ll, dlda, locv;
        if (f.lik == PV_LIK_BERNOULLI) {
          const float pr = fb_rcp(1.0f + fb_exp(-a));
          const float pc = fminf(fmaxf(pr, BERN_EPS), 1.0f - BERN_EPS);
          const float lg = fb_log(pc) - fb_log(1.0f - pc);
          ll = -(fmaxf(lg, 0.0f) - lg * xv + fb_log(1.0f + fb_exp(-fabsf(lg))));
          const float mask = (pr >= BERN_EPS && pr <= 1.0f - BERN_EPS) ? 1.0f : 0.0f;
          dlda = (fb_rcp(1.0f + fb_exp(-lg)) - xv) * mask;
          locv = pr;
        } else {
          const float pr = f.sigmoid_out ? fb_rcp(1.0f + fb_exp(-a)) : a;
          const float d = xv - pr;
          ll = -(d * d) / (2.0f * f.sig * f.sig) - fb_log(f.sig) - LOG_SQRT_2PI;
          dlda = -d / (f.sig * f.sig) * (f.sigmoid_out ? pr * (1.0f - pr) : 1.0f);
          locv = pr;
        }
        if (!act[u]) dlda = 0.0f;
        if (q == 0 && act[u]) {
          f.llrow[row[u]] = ll;
          if (f.loc) f.loc[row[u]] = locv;
        }
        if (GRADS) {
          if (q == 0) dbo += dlda;
          float* dwo = dwo_g;
#pragma unroll
          for (int jb = 0; jb < 8; ++jb) {
            f32x4 tv;
#pragma unroll
            for (int i = 0; i < 4; ++i) tv[i] = fb_sum_r(dlda * tC[u][jb][i]);
            if (r == 0) {
              f32x4* p = reinterpret_cast<f32x4*>(dwo + 16 * jb + 4 * q);
              *p = *p + tv;
            }
          }
#pragma unroll
          for (int jb = 0; jb < 8; ++jb) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wos + 16 * jb + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) tC[u][jb][i] = dlda * wv[i] * (1.0f - tC[u][jb][i] * tC[u][jb][i]);  // dpre2
          }
        }
      }
    }
    FB_STAMP(4);
    if (!GRADS) continue;

    // ---- wgrad of layer 2: exchange (dpre2 = tC, h1 = tB) one unit at a time ----
    bf16x4 pAh[UPW][8], pAl[UPW][8], pBh[UPW][8], pBl[UPW][8];
    if (wave_active) {
#pragma unroll
      for (int u = 0; u < UPW; ++u) { fb_presplit(tC[u], pAh[u], pAl[u]); fb_presplit(tB[u], pBh[u], pBl[u]); }
    }
    FB_STAMP(5);
    for (int c = 0; c < nact; ++c) {
#pragma unroll
      for (int u = 0; u < UPW; ++u)
        if (c == UPW * wave + u) {
          fb_stage_store(sAh, sAl, pAh[u], pAl[u], r, q);
          fb_stage_store(sBh, sBl, pBh[u], pBl[u], r, q);
        }
      if ((f.ablate & 512) && g == 0 && tid == 0 && tile_no == 1) fb_trace[128 + 4 * c] = (long long)__builtin_readcyclecounter();
      __syncthreads();
      if ((f.ablate & 512) && g == 0 && tid == 0 && tile_no == 1) fb_trace[129 + 4 * c] = (long long)__builtin_readcyclecounter();
      fb_wgrad_consume(sAh, sAl, sBh, sBl, accW2, accB2, wave, r, q);
      if ((f.ablate & 512) && g == 0 && tid == 0 && tile_no == 1) fb_trace[130 + 4 * c] = (long long)__builtin_readcyclecounter();
      __syncthreads();
      if ((f.ablate & 512) && g == 0 && tid == 0 && tile_no == 1) fb_trace[131 + 4 * c] = (long long)__builtin_readcyclecounter();
    }
    FB_STAMP(6);
    if (wave_active) {
      fb_layer_dgrad(W2h, W2l, tC, tA, r, q);
      FB_STAMP(7);
#pragma unroll
      for (int u = 0; u < UPW; ++u) {
        fb_mul_dtanh(tA[u], tB[u]);                        // tA = dpre1
        coord_layer(tB[u], u);                             // tB = h0 (recomputed)
      }
      fb_layer_dgrad(W1h, W1l, tA, tC, r, q);
#pragma unroll
      for (int u = 0; u < UPW; ++u) fb_mul_dtanh(tC[u], tB[u]);   // tC = dpre0
    }
    FB_STAMP(8);
    // ---- wgrad of layer 1: exchange (dpre1 = tA, h0 = tB) ----
    if (wave_active) {
#pragma unroll
      for (int u = 0; u < UPW; ++u) { fb_presplit(tA[u], pAh[u], pAl[u]); fb_presplit(tB[u], pBh[u], pBl[u]); }
    }
    for (int c = 0; c < nact; ++c) {
#pragma unroll
      for (int u = 0; u < UPW; ++u)
        if (c == UPW * wave + u) {
          fb_stage_store(sAh, sAl, pAh[u], pAl[u], r, q);
          fb_stage_store(sBh, sBl, pBh[u], pBl[u], r, q);
        }
      __syncthreads();
      fb_wgrad_consume(sAh, sAl, sBh, sBl, accW1, accB1, wave, r, q);
      __syncthreads();
    }
    FB_STAMP(10);
    // ---- coordinate layer backward (fp32): row-local part ----
    if (wave_active) {
#pragma unroll
      for (int u = 0; u < UPW; ++u) {
        float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wc0 + 16 * jb + 4 * q);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(Wc1 + 16 * jb + 4 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) { d0 += tC[u][jb][i] * w0[i]; d1 += tC[u][jb][i] * w1[i]; }
        }
        d0 = fb_sum_q(d0);
        d1 = fb_sum_q(d1);
        if (q == 0 && act[u]) {
          f.rowtp[row[u]] = sc[u] * (d1 * u0c[u] - d0 * u1c[u]);
          f.rowtp[f.M + row[u]] = d0 * u0c[u] + d1 * u1c[u];
          f.rowtp[2 * f.M + row[u]] = d0;
          f.rowtp[3 * f.M + row[u]] = d1;
        }
      }
    }
    // ---- cross-row part: dWc, dbc / dhz from dpre0 staged in fp32 over the (now idle) staging images ----
    FB_STAMP(11);
    float* st32 = reinterpret_cast<float*>(smb + BO_SAH);      // 16 rows x 136 floats = 8704 B <= 4 x 4352 B
    for (int c = 0; c < nact; ++c) {
#pragma unroll
      for (int u = 0; u < UPW; ++u)
        if (c == UPW * wave + u) {
#pragma unroll
          for (int jb = 0; jb < 8; ++jb) *reinterpret_cast<f32x4*>(st32 + r * LDB + 16 * jb + 4 * q) = tC[u][jb];
          if (q == 0) { info[r] = x0[u]; info[16 + r] = x1[u]; }
          if (lane == 0) reinterpret_cast<int*>(info)[32] = bu[u];
        }
      __syncthreads();
      const int bcur = reinterpret_cast<const int*>(info)[32];
      if (bcur != cur_b) {
        if (cur_b >= 0) flush_hz(cur_b);
        cur_b = bcur;
      }
#pragma unroll
      for (int sgm = 0; sgm < 4; ++sgm) {
        const float xa = info[4 * sgm + q], xb = info[16 + 4 * sgm + q];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const float v = st32[(4 * sgm + q) * LDB + 16 * (2 * wave + s) + r];
          ahz[s] += v;
          aWc0[s] += v * xa;
          aWc1[s] += v * xb;
        }
      }
      __syncthreads();
    }
    FB_STAMP(12);
  }
  if (!GRADS) return;

  if (cur_b >= 0) flush_hz(cur_b);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j0 = 16 * (2 * wave + s);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // C/D layout: lane (col k' = r, q), reg i -> dW[j0 + 4*q + i][16*kb + r]
        rec[(j0 + 4 * q + i) * FD_H + 16 * kb + r] = accW1[s][kb][i];
        rec[FD_H * FD_H + (j0 + 4 * q + i) * FD_H + 16 * kb + r] = accW2[s][kb][i];
      }
    // bias gradients: every column of accB holds the same sums; lane (col 0, q), reg i -> row j0 + 4q + i
    if (r == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rec[2 * FD_H * FD_H + j0 + 4 * q + i] = accB1[s][i];
        rec[2 * FD_H * FD_H + FD_H + j0 + 4 * q + i] = accB2[s][i];
      }
    }
    const float tc0 = fb_sum_q(aWc0[s]), tc1 = fb_sum_q(aWc1[s]);
    if (q == 0) {
      rec[2 * FD_H * FD_H + 2 * FD_H + j0 + r] = tc0;
      rec[2 * FD_H * FD_H + 3 * FD_H + j0 + r] = tc1;
    }
  }
  const float tb = pv_wave_sum(dbo);
  if (lane == 0) red[wave] = tb;
  __syncthreads();
  if (tid < FD_H) {
    const float* d = reinterpret_cast<const float*>(smb + BO_DWO);
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < FB_WAVES; ++w) v += d[w * FD_H + tid];
    rec[2 * FD_H * FD_H + 4 * FD_H + tid] = v;
  }
  if (tid == 0) {
    float v = 0.0f;
    for (int w = 0; w < FB_WAVES; ++w) v += red[w];
    rec[2 * FD_H * FD_H + 5 * FD_H] = v;
  }
}

extern "C" int pv_debug_read_trace(long long* out, int n) {
  if (n > 256) n = 256;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fb_trace), n * sizeof(long long));
}

int pv_sdec_fused_bf16_launch(const PvFused& f_in, int grid, bool grads, hipStream_t s) {
  PvFused f = f_in;
  static int ablate = -1;
  if (ablate < 0) { const char* e = getenv("PV_FD_ABLATE"); ablate = e ? atoi(e) : 0; }
  f.ablate = ablate;
  const size_t lds = FB_LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&pv_sdec_fused_bf16_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&pv_sdec_fused_bf16_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e1 != hipSuccess) return (int)e1;
    if (e2 != hipSuccess) return (int)e2;
    attr_set = true;
  }
  if (grads)
    hipLaunchKernelGGL(pv_sdec_fused_bf16_kernel<true>, dim3(grid), dim3(FB_THREADS), lds, s, f);
  else
    hipLaunchKernelGGL(pv_sdec_fused_bf16_kernel<false>, dim3(grid), dim3(FB_THREADS), lds, s, f);
  PV_LAUNCH_CHECK();
  return 0;
}
