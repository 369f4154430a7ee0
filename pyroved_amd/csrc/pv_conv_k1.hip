// pv_conv_k1.hip — kernel-1 convolutions (nets/conv.py: UpsampleBlock's conv, the decoder's output layer) over
// channels-last maps: a Linear over the pixels, out[p][n] = act(sum_k in[p][k] w[n][k] + b[n]), with a SHORT contraction
// (32 ... 128 channels) and a few thousand pixels.  The LDS-tiled GEMM (pv_gemm.hip) runs such a problem as ~100
// workgroups that each walk K in 32-wide stages behind barriers: one global-memory round trip per stage, 8-22 us per
// launch for 10-70 MFLOP.  Here every operand of a wave's tile is requested at once and goes from L2 straight into
// the registers of v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate): one memory latency per launch, no LDS, no barrier.
//   forward / input gradient: a wave owns 16 pixels x 32 outputs; the weights are the MFMA's A operand (m = output), the
//     pixels its B operand (n = pixel), so a lane ends up with four CONSECUTIVE outputs of one pixel (16-byte stores).
//     The input gradient is the same kernel on the transposed weight (strided loads; the matrix is L2-resident) with the
//     producing layer's activation derivative in the epilogue.
//   weight gradient: dW[n][k] = sum_p g[p][n] in[p][k]: one 16 x 16 output tile and one chunk of the pixels per workgroup,
//     its four waves take 64-pixel register batches, partial tiles meet in LDS and go to the step's deferred-reduction
//     table (pv_conv.h: PvFinishList) — no launch of its own for the split-order sum.
#include "pv_common.h"
#include "pv_side.h"
#include "pv_conv.h"
#include <stdlib.h>

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define K1_KC 128                    // contraction chunk held in registers

struct K1Fwd {
  const float* in; const float* w; const float* bias; float* out; const float* eg_y;
  int64_t rows;
  int K, N, ld_in, ld_out, w_ns, w_ks, act, eg_act, ngroups, w_bytes;
  int simple; float slope, gslope;   // piecewise-linear activations (none / relu / lrelu, both ways): v > 0 ? v : slope v, one
                                     // branch-free path instead of a per-value switch over every activation's code
  int up_out, up_in;      // 1-D nearest 2x upsample fused: every output row stored twice (rows 2p, 2p + 1) / the input row is the sum of
                          // rows 2p and 2p + 1 (the upsample's backward)
};

// any activation / derivative: ONE out-of-line copy of the switch (inlined per value it was 6.5 k of a kernel's 6.8 k instructions)
__device__ __noinline__ f32x4 k1_act_generic(f32x4 v, const float* eg_y, int act, int eg_act) {
  for (int i = 0; i < 4; ++i) {
    v[i] = pv_act_fwd(v[i], act);
    if (eg_y) v[i] *= pv_act_grad(eg_y[i], 0.0f, eg_act);
  }
  return v;
}

__device__ __noinline__ float k1_act_one(float t, const float* eg_y, int act, int eg_act) {
  t = pv_act_fwd(t, act);
  if (eg_y) t *= pv_act_grad(eg_y[0], 0.0f, eg_act);
  return t;
}

// The vector form: K % 16 == 0, N % 4 == 0, 16-byte aligned pixel rows.  NJ 16-wide contraction groups per register chunk
// (K % (16 NJ) == 0), NB 16-output blocks per wave; every load and MFMA is unconditional (outputs past N read a clamped
// weight row and are not stored).  WV: the weight's contraction index has unit stride (forward form, 16-byte loads);
// otherwise (input-gradient form, the transposed matrix) dword buffer loads: uniform base + 32-bit lane offset.
template <bool WV, int NJ, int NB>
__global__ __launch_bounds__(256) void pv_k1_fwd_kernel(K1Fwd a) {
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  const int64_t rb = unit / a.ngroups;
  const int ng = (int)(unit - rb * a.ngroups);
  if (rb * 16 >= a.rows) return;                                   // (no barrier in this kernel)
  const int64_t row = rb * 16 + r;
  const bool rok = row < a.rows;
  const float* ip = a.in + (rok ? row : 0) * (a.up_in ? 2 : 1) * a.ld_in + 4 * q;
  const int n0 = 16 * NB * ng;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const float* wp[NB];
  int wo[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n = n0 + 16 * b + r, nc = n < a.N ? n : 0;
    wp[b] = a.w + (int64_t)nc * a.w_ns + 4 * q;
    wo[b] = 4 * (nc * a.w_ns + 4 * q * a.w_ks);
  }
  f32x4 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int kc = 0; kc < a.K; kc += 16 * NJ) {
    f32x4 x[NJ], w[NB][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) x[j] = *reinterpret_cast<const f32x4*>(ip + kc + 16 * j);
    if (a.up_in) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) x[j] = x[j] + *reinterpret_cast<const f32x4*>(ip + a.ld_in + kc + 16 * j);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (WV) {
          w[b][j] = *reinterpret_cast<const f32x4*>(wp[b] + kc + 16 * j);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            w[b][j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, wo[b], 4 * (kc + 16 * j + i) * a.w_ks, 0));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = MFMA4(w[b][j][i], x[j][i], acc[b]);
      }
    }
  }
  // C layout: lane (pixel r, q), register i <-> output n0 + 16 b + 4 q + i
  if (!rok) return;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n = n0 + 16 * b + 4 * q;
    if (n >= a.N) continue;
    f32x4 v = acc[b];
    if (a.bias) v = v + *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.simple) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.0f ? v[i] : v[i] * a.slope;
      if (a.eg_y) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(a.eg_y + row * a.ld_out + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= y[i] > 0.0f ? 1.0f : a.gslope;
      }
    } else {
      v = k1_act_generic(v, a.eg_y ? a.eg_y + row * a.ld_out + n : nullptr, a.act, a.eg_act);
    }
    if (a.up_out) {
      *reinterpret_cast<f32x4*>(a.out + 2 * row * a.ld_out + n) = v;
      *reinterpret_cast<f32x4*>(a.out + (2 * row + 1) * a.ld_out + n) = v;
    } else {
      *reinterpret_cast<f32x4*>(a.out + row * a.ld_out + n) = v;
    }
  }
}

// any K, N, alignment: scalar loads at clamped addresses, 16 pixels x 16 outputs per wave
__global__ __launch_bounds__(256) void pv_k1_gen_kernel(K1Fwd a) {
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  const int64_t rb = unit / a.ngroups;
  const int ng = (int)(unit - rb * a.ngroups);
  if (rb * 16 >= a.rows) return;
  const int64_t row = rb * 16 + r;
  const bool rok = row < a.rows;
  const float* ip = a.in + (rok ? row : 0) * (a.up_in ? 2 : 1) * a.ld_in;
  const int n = 16 * ng + r;
  const float* wp = a.w + (int64_t)(n < a.N ? n : 0) * a.w_ns;
  f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int k0 = 0; k0 < a.K; k0 += 16) {
    float x[4], w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + 4 * q + i, kk = k < a.K ? k : a.K - 1;
      const float xv = ip[kk] + (a.up_in ? ip[a.ld_in + kk] : 0.0f), wv = wp[(int64_t)kk * a.w_ks];
      x[i] = k < a.K ? xv : 0.0f;
      w[i] = k < a.K ? wv : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = MFMA4(w[i], x[i], acc);
  }
  if (!rok) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int no = 16 * ng + 4 * q + i;
    if (no >= a.N) break;
    float t = acc[i] + (a.bias ? a.bias[no] : 0.0f);
    if (a.simple) {
      t = t > 0.0f ? t : t * a.slope;
      if (a.eg_y) t *= a.eg_y[row * a.ld_out + no] > 0.0f ? 1.0f : a.gslope;
    } else {
      t = k1_act_one(t, a.eg_y ? a.eg_y + row * a.ld_out + no : nullptr, a.act, a.eg_act);
    }
    if (a.up_out) { a.out[2 * row * a.ld_out + no] = t; a.out[(2 * row + 1) * a.ld_out + no] = t; }
    else a.out[row * a.ld_out + no] = t;
  }
}

static inline bool k1_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <bool WV, int NJ>
static void k1_launch_nb(const K1Fwd& a, int nb, unsigned grid, hipStream_t s) {
  // (PV_LAUNCH_FORK: an input-gradient launch carries the fork event when the recorded weight gradients are about to be
  //  flushed onto the side stream — pv_side.h)
  if (nb == 4) PV_LAUNCH_FORK((pv_k1_fwd_kernel<WV, NJ, 4>), dim3(grid), dim3(256), 0, s, a);
  else if (nb == 2) PV_LAUNCH_FORK((pv_k1_fwd_kernel<WV, NJ, 2>), dim3(grid), dim3(256), 0, s, a);
  else PV_LAUNCH_FORK((pv_k1_fwd_kernel<WV, NJ, 1>), dim3(grid), dim3(256), 0, s, a);
}
template <bool WV>
static void k1_launch_nj(const K1Fwd& a, int nb, unsigned grid, hipStream_t s) {
  if (a.K % 128 == 0) k1_launch_nb<WV, 8>(a, nb, grid, s);
  else if (a.K % 64 == 0) k1_launch_nb<WV, 4>(a, nb, grid, s);
  else if (a.K % 32 == 0) k1_launch_nb<WV, 2>(a, nb, grid, s);
  else k1_launch_nb<WV, 1>(a, nb, grid, s);
}

static int k1_launch(K1Fwd a, int64_t w_elems, hipStream_t s) {
  if (a.rows <= 0 || a.N <= 0) return 0;
  if (a.K < 1) return PV_EINVAL;
  const bool xv = a.K % 16 == 0 && a.N % 4 == 0 && a.ld_in % 4 == 0 && a.ld_out % 4 == 0 && k1_al16(a.in) && k1_al16(a.out) &&
                  (!a.bias || k1_al16(a.bias)) && (!a.eg_y || k1_al16(a.eg_y)) && w_elems < (1 << 28);
  const bool wv = a.w_ks == 1 && a.w_ns % 4 == 0 && k1_al16(a.w);
  a.w_bytes = (int)(w_elems * 4);
  auto lin = [](int act) { return act == PV_ACT_NONE || act == PV_ACT_RELU || act == PV_ACT_LRELU; };
  auto slope = [](int act) { return act == PV_ACT_NONE ? 1.0f : act == PV_ACT_RELU ? 0.0f : 0.01f; };
  const int eg = a.eg_y ? a.eg_act : PV_ACT_NONE;
  a.simple = lin(a.act) && lin(eg);
  a.slope = slope(a.act); a.gslope = slope(eg);
  if (!xv) {
    a.ngroups = (a.N + 15) / 16;
    const int64_t units = ((a.rows + 15) / 16) * a.ngroups;
    PV_LAUNCH_FORK(pv_k1_gen_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, a);
  } else {
    static const int nb_env = pv_exp_int("PV_K1_NB", 0);   // PV_K1_NB=1|2|4: 16-output blocks per wave (experiments build)
    int nb = a.N > 16 ? 2 : 1;                         // (measured on VED C5: 2 blocks per wave -9 us per step against 4 — more waves)
    if (nb_env == 1 || nb_env == 2 || nb_env == 4) nb = (a.N > 16 * (nb_env / 2)) ? nb_env : nb;
    a.ngroups = (a.N + 16 * nb - 1) / (16 * nb);
    const int64_t units = ((a.rows + 15) / 16) * a.ngroups;
    const unsigned grid = (unsigned)((units + 3) / 4);
    if (wv) k1_launch_nj<true>(a, nb, grid, s);
    else k1_launch_nj<false>(a, nb, grid, s);
  }
  PV_LAUNCH_CHECK();
  return 0;
}

// out (rows, Co) = act(in (rows, Ci) w (Co, Ci)^T + bias); up: out has 2 rows rows, row p written to 2p and 2p + 1
int pv_k1_fwd(const float* in, int64_t rows, int Ci, const float* w, const float* bias, float* out, int Co, int act,
              hipStream_t s, int up) {
  K1Fwd a{};
  a.in = in; a.w = w; a.bias = bias; a.out = out; a.rows = rows; a.K = Ci; a.N = Co; a.ld_in = Ci; a.ld_out = Co;
  a.w_ns = Ci; a.w_ks = 1; a.act = act; a.up_out = up;
  return k1_launch(a, (int64_t)Co * Ci, s);
}

// gin (rows, Ci) = (g (rows, Co) w (Co, Ci)) * act'(eg_y), eg_y shaped like gin (null: no factor); up: g has 2 rows rows and
// g[p] stands for g[2p] + g[2p + 1]
int pv_k1_dgrad(const float* g, int64_t rows, int Co, const float* w, float* gin, int Ci, const float* eg_y, int eg_act,
                hipStream_t s, int up) {
  K1Fwd a{};
  a.in = g; a.w = w; a.out = gin; a.rows = rows; a.K = Co; a.N = Ci; a.ld_in = Co; a.ld_out = Ci;
  a.w_ns = 1; a.w_ks = Ci; a.act = PV_ACT_NONE; a.up_in = up;
  if (eg_y && eg_act != PV_ACT_NONE) { a.eg_y = eg_y; a.eg_act = eg_act; }
  return k1_launch(a, (int64_t)Co * Ci, s);
}

// ---- weight gradient ------------------------------------------------------------------------------------------------
typedef PvK1Wg K1Wg;                 // (pv_conv.h: the batch of deferred problems holds them)

#define K1_WB 64                     // pixels per register batch of a wave

__device__ __forceinline__ void k1_wgrad_body(const K1Wg& a, int blk, float (*part)[16][17], float (*rpart)[16]) {
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = a.mtiles * a.ntiles;
  const int sp = blk / (tiles * a.taps), tt = blk - sp * tiles * a.taps;
  const int tap = tt / tiles, t = tt - tap * tiles;
  const int mb = t / a.ntiles, nb = t - mb * a.ntiles;
  // A lane (m = output channel 16 mb + r, pixel slot q), B lane (n = input channel 16 nb + r, pixel slot q)
  const int m = 16 * mb + r, n = 16 * nb + r;
  const bool mok = m < a.Co, nok = n < a.Ci;
  const float* gp = a.g + (mok ? m : 0);
  const float* xp = a.in + (nok ? n : 0);
  const int64_t p0 = (int64_t)sp * a.chunk, p1 = p0 + a.chunk < a.rows ? p0 + a.chunk : a.rows;
  const int sh = a.taps == 3 ? tap - 1 : 0;                  // input position = output position + sh
  const bool pow2 = (a.L & (a.L - 1)) == 0;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float rs = 0.0f;
  for (int64_t b0 = p0 + (int64_t)K1_WB * wave; b0 < p1; b0 += 4 * K1_WB) {
    float gv[K1_WB / 4], xv[K1_WB / 4];
#pragma unroll
    for (int s = 0; s < K1_WB / 4; ++s) {
      const int64_t p = b0 + 4 * s + q;
      const int64_t pc = p < p1 ? p : p1 - 1;
      bool xok = p < p1 && nok;
      int64_t px = pc;
      if (sh != 0) {
        const int l = pow2 ? (int)(pc & (a.L - 1)) : (int)(pc % a.L);
        xok = xok && l + sh >= 0 && l + sh < a.L;
        px = pc + sh < 0 ? 0 : pc + sh >= a.rows ? a.rows - 1 : pc + sh;
      }
      const float x = a.up ? gp[2 * pc * a.Co] + gp[(2 * pc + 1) * a.Co] : gp[pc * a.Co], y = xp[px * a.Ci];
      gv[s] = (p < p1 && mok) ? x : 0.0f;
      xv[s] = xok ? y : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < K1_WB / 4; ++s) {
      acc[s & 3] = MFMA4(gv[s], xv[s], acc[s & 3]);
      rs += gv[s];
    }
  }
  const f32x4 c = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  // C layout: lane (n = r, q), register i <-> m = 4 q + i
#pragma unroll
  for (int i = 0; i < 4; ++i) part[wave][4 * q + i][r] = c[i];
  rs = pv_sum_rows(rs);
  if (q == 0) rpart[wave][r] = rs;
  __syncthreads();
  const int mm = tid >> 4, nn = tid & 15, mo = 16 * mb + mm, no = 16 * nb + nn;
  if (mo < a.Co && no < a.Ci)
    a.part[((int64_t)sp * a.Co * a.Ci + (int64_t)mo * a.Ci + no) * a.taps + tap] =
        (part[0][mm][nn] + part[1][mm][nn]) + (part[2][mm][nn] + part[3][mm][nn]);
  if (a.part_b && nb == 0 && tap == 0 && tid < 16 && 16 * mb + tid < a.Co)
    a.part_b[(int64_t)sp * a.Co + 16 * mb + tid] = (rpart[0][tid] + rpart[1][tid]) + (rpart[2][tid] + rpart[3][tid]);
}

__global__ __launch_bounds__(256) void pv_k1_wgrad_kernel(K1Wg a) {
  __shared__ float part[4][16][17];
  __shared__ float rpart[4][16];
  k1_wgrad_body(a, blockIdx.x, part, rpart);
}

// The batched launch is a THROUGHPUT kernel (~20 k workgroups, load-bound at one operand load per MFMA), where the lean
// tile's virtue — many short workgroups — buys nothing: 2 x 2 output blocks per workgroup with all TAPS of a Conv1d in it
// (g loaded once, the input at the three shifted positions) is a third of the loads per MFMA.  (As a launch of its own this
// form measured 20 us per step SLOWER: a quarter of the workgroups.)  Same partial layout as the lean form.
template <int TAPS>
__device__ __forceinline__ void k1_wgrad_fat_body(const K1Wg& a, int blk, float* lds) {
  constexpr int WB = TAPS == 3 ? 32 : 64;            // pixels per register batch of a wave
  float (*part)[TAPS * 32][33] = reinterpret_cast<float (*)[TAPS * 32][33]>(lds);
  float (*rpart)[32] = reinterpret_cast<float (*)[32]>(lds + 4 * TAPS * 32 * 33);
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = a.mtiles * a.ntiles;
  const int sp = blk / tiles, t = blk - sp * tiles;
  const int mb = t / a.ntiles, nb = t - mb * a.ntiles;
  bool mok[2], nok[2];
  const float* gp[2];
  const float* xp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = 16 * (2 * mb + i) + r; mok[i] = m < a.Co; gp[i] = a.g + (mok[i] ? m : 0);
    const int n = 16 * (2 * nb + i) + r; nok[i] = n < a.Ci; xp[i] = a.in + (nok[i] ? n : 0);
  }
  const int64_t p0 = (int64_t)sp * a.chunk, p1 = p0 + a.chunk < a.rows ? p0 + a.chunk : a.rows;
  const bool pow2 = (a.L & (a.L - 1)) == 0;
  f32x4 acc[TAPS][2][2];
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[tp][i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float rs[2] = {0.0f, 0.0f};
  // Operands by buffer loads: a 32-bit lane offset per operand and batch plus the (uniform) step within the batch —
  // no 64-bit address arithmetic and no selects per load (the lean form spends ~100 VALU instructions per 12 MFMAs on
  // them).  Rows past the tensor read 0 (buffer bounds; the chunk is a whole number of workgroup batches, so a batch never
  // straddles the next chunk), channels past Co / Ci read a valid channel whose outputs are never stored.
  const int gm = a.up ? 2 : 1;
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, (int)(a.rows * gm * a.Co * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)(a.rows * a.Ci * 4), 0x00020000);
  int mc[2], nc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { mc[i] = mok[i] ? 16 * (2 * mb + i) + r : 0; nc[i] = nok[i] ? 16 * (2 * nb + i) + r : 0; }
  for (int64_t b0 = p0 + (int64_t)WB * wave; b0 < p1; b0 += 4 * WB) {
    float gv[2][WB / 4], xv[TAPS][2][WB / 4];
    const int pb = (int)b0 + q;                        // this lane's first pixel of the batch
    int go[2], xo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { go[i] = 4 * (pb * gm * a.Co + mc[i]); xo[i] = 4 * (pb * a.Ci + nc[i]); }
    const int l0 = pow2 ? (pb & (a.L - 1)) : pb % a.L;
#pragma unroll
    for (int s = 0; s < WB / 4; ++s) {
      const int so_g = 16 * s * gm * a.Co, so_x = 16 * s * a.Ci;      // (uniform: 4 pixels x 4 bytes per step)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // (the step goes into the LANE offset: the scalar offset is not part of the buffer's range check, and a lane offset
        //  that is negative before the step is added would read 0 for a pixel that exists)
        float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, go[i] + so_g, 0, 0));
        if (a.up) x += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, go[i] + so_g + 4 * a.Co, 0, 0));
        gv[i][s] = x;
      }
      if (TAPS == 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) xv[0][j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xo[j] + so_x, 0, 0));
      } else {
        int l = l0 + 4 * s;
        l = pow2 ? (l & (a.L - 1)) : l % a.L;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {          // input position = output position + tp - 1, inside the sample
          const int sh = tp - 1;
          const bool ok = l + sh >= 0 && l + sh < a.L;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xo[j] + so_x + 4 * sh * a.Ci, 0, 0));
            xv[tp][j][s] = ok ? y : 0.0f;
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < WB / 4; ++s) {
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[tp][i][j] = MFMA4(gv[i][s], xv[tp][j][s], acc[tp][i][j]);
      rs[0] += gv[0][s]; rs[1] += gv[1][s];
    }
  }
  // C layout: lane (n = r, q), register e <-> m = 4 q + e
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) part[wave][(tp * 2 + i) * 16 + 4 * q + e][16 * j + r] = acc[tp][i][j][e];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float v = rs[i];
    v = pv_sum_rows(v);
    if (q == 0) rpart[wave][16 * i + r] = v;
  }
  __syncthreads();
  for (int e = tid; e < TAPS * 32 * 32; e += 256) {
    const int nn = e & 31, row = e >> 5, mm = row & 31, tp = row >> 5;
    const int mo = 32 * mb + mm, no = 32 * nb + nn;
    if (mo < a.Co && no < a.Ci)
      a.part[((int64_t)sp * a.Co * a.Ci + (int64_t)mo * a.Ci + no) * TAPS + tp] =
          (part[0][row][nn] + part[1][row][nn]) + (part[2][row][nn] + part[3][row][nn]);
  }
  if (a.part_b && nb == 0 && tid < 32 && 32 * mb + tid < a.Co)
    a.part_b[(int64_t)sp * a.Co + 32 * mb + tid] = (rpart[0][tid] + rpart[1][tid]) + (rpart[2][tid] + rpart[3][tid]);
}

// The WIDE form (round 5).  The fat form feeds every MFMA with two 4-byte loads per lane: its ~2 M load instructions per launch
// (each a 256-byte request) are what the launch takes its 100+ us for, and what it takes from the kernels it runs next to (VED at
// batch 256 without this launch: 0.772 -> 0.728 ms).  An fp32 MFMA wants ONE value per lane and operand, but nothing says WHICH
// 16 channels an MFMA's rows are: a lane loads MJ (NJ) consecutive channels of its pixel in one 8- / 16-byte load and component
// j feeds MFMA j, whose row r is channel MJ r + j — a (16 MJ) x (16 NJ) tile per wave and tap from one g load and TAPS input
// loads per four pixels: 4 load instructions per 24 MFMAs (three taps, 64 x 32) where the fat form issued 16.  Same split / partial
// layout, the four waves' tiles summed through LDS tap by tap.
template <int N> struct K1Vec;
template <> struct K1Vec<2> { typedef float T __attribute__((ext_vector_type(2))); typedef unsigned U __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ T ld(__amdgpu_buffer_rsrc_t rs, int off) { return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0)); } };
template <> struct K1Vec<4> { typedef float T __attribute__((ext_vector_type(4))); typedef unsigned U __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ T ld(__amdgpu_buffer_rsrc_t rs, int off) { return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0)); } };
template <int TAPS, int MJ, int NJ> struct K1Wide {
  static constexpr int WB = TAPS == 3 ? (NJ == 4 ? 16 : 32) : 64;    // pixels per register batch of a wave
  static constexpr int MT = 16 * MJ, NT = 16 * NJ;
};
template <int TAPS, int MJ, int NJ>
__device__ __forceinline__ void k1_wgrad_wide_body(const K1Wg& a, int blk, float* lds) {
  typedef K1Wide<TAPS, MJ, NJ> W;
  constexpr int WB = W::WB, MT = W::MT, NT = W::NT;
  typedef typename K1Vec<MJ>::T vm;
  typedef typename K1Vec<NJ>::T vn;
  float (*part)[MT][NT + 1] = reinterpret_cast<float (*)[MT][NT + 1]>(lds);
  float (*rpart)[MT] = reinterpret_cast<float (*)[MT]>(lds + 4 * MT * (NT + 1));
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = a.mtiles * a.ntiles;
  const int sp = blk / tiles, t = blk - sp * tiles;
  const int mb = t / a.ntiles, nb = t - mb * a.ntiles;
  const int m0 = MT * mb, n0 = NT * nb;               // (Co / Ci are whole tiles: k1_wgrad_launch picks the form)
  const int64_t p0 = (int64_t)sp * a.chunk, p1 = p0 + a.chunk < a.rows ? p0 + a.chunk : a.rows;
  const bool pow2 = (a.L & (a.L - 1)) == 0;
  f32x4 acc[TAPS][MJ][NJ];
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int i = 0; i < NJ; ++i) acc[tp][j][i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  vm rs = {};
  const int gm = a.up ? 2 : 1;
  // (rows past the tensor read 0 through the buffers' bounds, as in the fat form)
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, (int)(a.rows * gm * a.Co * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)(a.rows * a.Ci * 4), 0x00020000);
  for (int64_t b0 = p0 + (int64_t)WB * wave; b0 < p1; b0 += 4 * WB) {
    vm gv[WB / 4];
    vn xv[TAPS][WB / 4];
    const int pb = (int)b0 + q;                        // this lane's first pixel of the batch
    const int go = 4 * (pb * gm * a.Co + m0 + MJ * r), xo = 4 * (pb * a.Ci + n0 + NJ * r);
    const int l0 = pow2 ? (pb & (a.L - 1)) : pb % a.L;
#pragma unroll
    for (int s = 0; s < WB / 4; ++s) {
      const int so_g = 16 * s * gm * a.Co, so_x = 16 * s * a.Ci;      // (uniform: 4 pixels x 4 bytes per step)
      vm x = K1Vec<MJ>::ld(grs, go + so_g);
      if (a.up) x += K1Vec<MJ>::ld(grs, go + so_g + 4 * a.Co);
      gv[s] = x;
      if (TAPS == 1) {
        xv[0][s] = K1Vec<NJ>::ld(xrs, xo + so_x);
      } else {
        int l = l0 + 4 * s;
        l = pow2 ? (l & (a.L - 1)) : l % a.L;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {            // input position = output position + tp - 1, inside the sample
          const int sh = tp - 1;
          const bool ok = l + sh >= 0 && l + sh < a.L;
          const vn y = K1Vec<NJ>::ld(xrs, xo + so_x + 4 * sh * a.Ci);
#pragma unroll
          for (int i = 0; i < NJ; ++i) xv[tp][s][i] = ok ? y[i] : 0.0f;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < WB / 4; ++s) {
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
          for (int i = 0; i < NJ; ++i) acc[tp][j][i] = MFMA4(gv[s][j], xv[tp][s][i], acc[tp][j][i]);
      rs += gv[s];
    }
  }
  // C layout: lane (column r, q), register e <-> row 4 q + e; row rr of MFMA j is channel MJ rr + j, column r of MFMA i channel NJ r + i
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp) {
    if (tp) __syncthreads();                           // (the previous tap's sums are read)
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int i = 0; i < NJ; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) part[wave][MJ * (4 * q + e) + j][NJ * r + i] = acc[tp][j][i][e];
    if (tp == 0) {
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const float v = pv_sum_rows(rs[j]);
        if (q == 0) rpart[wave][MJ * r + j] = v;
      }
    }
    __syncthreads();
    for (int e = tid; e < MT * NT; e += 256) {
      const int nn = e % NT, mm = e / NT;
      a.part[((int64_t)sp * a.Co * a.Ci + (int64_t)(m0 + mm) * a.Ci + n0 + nn) * TAPS + tp] =
          (part[0][mm][nn] + part[1][mm][nn]) + (part[2][mm][nn] + part[3][mm][nn]);
    }
  }
  if (a.part_b && nb == 0 && tid < MT)
    a.part_b[(int64_t)sp * a.Co + m0 + tid] = (rpart[0][tid] + rpart[1][tid]) + (rpart[2][tid] + rpart[3][tid]);
}

// every recorded weight gradient of a stack's backward in ONE launch (PvK1Batch): workgroup b serves problem k with
// blk0[k] <= b < blk0[k + 1].  The problems are independent (each reads its own layer's g and input, which the stack keeps
// alive until the flush), so a dozen 8-23 us launches — each with its own cold start — become one that fills the GPU.
struct K1Tab { K1Wg e[PV_K1_BATCH]; int blk0[PV_K1_BATCH + 1]; int n; };
__global__ __launch_bounds__(256) void pv_k1_wgrad_table_kernel(K1Tab t) {
  __shared__ float lds[4 * 3 * 32 * 33 + 4 * 32];     // the fat three-tap form's partial tiles (50.7 KB); the others use a part of it
  int k = 0;
  while (k + 1 < t.n && (int)blockIdx.x >= t.blk0[k + 1]) ++k;
  const int blk = (int)blockIdx.x - t.blk0[k];
  if (t.e[k].fat == 2) {
    const int key = t.e[k].taps * 100 + t.e[k].mj * 10 + t.e[k].nj;
    if (key == 342) k1_wgrad_wide_body<3, 4, 2>(t.e[k], blk, lds);
    else if (key == 324) k1_wgrad_wide_body<3, 2, 4>(t.e[k], blk, lds);
    else if (key == 322) k1_wgrad_wide_body<3, 2, 2>(t.e[k], blk, lds);
    else if (key == 142) k1_wgrad_wide_body<1, 4, 2>(t.e[k], blk, lds);
    else if (key == 124) k1_wgrad_wide_body<1, 2, 4>(t.e[k], blk, lds);
    else k1_wgrad_wide_body<1, 2, 2>(t.e[k], blk, lds);
  } else if (t.e[k].fat) {
    if (t.e[k].taps == 3) k1_wgrad_fat_body<3>(t.e[k], blk, lds);
    else k1_wgrad_fat_body<1>(t.e[k], blk, lds);
  } else {
    k1_wgrad_body(t.e[k], blk, reinterpret_cast<float (*)[16][17]>(lds), reinterpret_cast<float (*)[16]>(lds + 4 * 16 * 17));
  }
}

static int k1_wg_splits(int64_t rows, int Ci, int Co, int taps) {
  const int64_t tiles = (int64_t)((Co + 15) / 16) * ((Ci + 15) / 16) * taps;
  int64_t ns = (rows + 4 * K1_WB - 1) / (4 * K1_WB);          // one register batch per wave
  static const int cap_env = pv_exp_int("PV_K1_WCAP", 0);      // PV_K1_WCAP=n: workgroup target of the weight gradient (experiments build)
  const int64_t target = cap_env > 0 ? cap_env : (taps == 3 ? 4096 : 1024);    // (measured on VED C5: 4096 for the three-tap form -7 us)
  const int64_t cap = (target + tiles - 1) / tiles;            // ~4 workgroups per CU in all
  if (ns > cap) ns = cap;
  if (ns < 1) ns = 1;
  return (int)ns;
}

static int64_t k1_wg_ws(int64_t rows, int Ci, int Co, int taps) {
  return (int64_t)k1_wg_splits(rows, Ci, Co, taps) * ((int64_t)Co * Ci * taps + Co) * (int64_t)sizeof(float);
}
int64_t pv_k1_wgrad_ws(int64_t rows, int Ci, int Co) { return k1_wg_ws(rows, Ci, Co, 1); }
int64_t pv_conv3_1d_wgrad_lean_ws(int B, int L, int Ci, int Co) { return k1_wg_ws((int64_t)B * L, Ci, Co, 3); }

static int k1_wgrad_launch(const float* g, const float* in, int64_t rows, int L, int taps, int Ci, int Co, float* dw, float* db,
                           void* ws, int64_t ws_bytes, hipStream_t s, PvFinishList* defer, int up) {
  if (rows < 1 || Ci < 1 || Co < 1 || L < 1) return PV_EINVAL;
  const int64_t need = k1_wg_ws(rows, Ci, Co, taps);
  const bool deferred = pv_wgrad_ws(defer, need, ws, ws_bytes);
  if (!ws || ws_bytes < need) return PV_EWS;
  K1Wg a{};
  a.g = g; a.in = in; a.rows = rows; a.Ci = Ci; a.Co = Co; a.up = up; a.taps = taps; a.L = L;
  a.mtiles = (Co + 15) / 16; a.ntiles = (Ci + 15) / 16;
  a.nsplit = k1_wg_splits(rows, Ci, Co, taps);
  a.chunk = (rows + a.nsplit - 1) / a.nsplit;
  a.chunk = (a.chunk + 3) / 4 * 4;
  a.nsplit = (int)((rows + a.chunk - 1) / a.chunk);
  a.part = (float*)ws;
  a.part_b = db ? a.part + (int64_t)a.nsplit * Co * Ci * taps : nullptr;
  a.nblk = a.mtiles * a.ntiles * taps * a.nsplit;
  if (deferred && defer->k1b && defer->k1b->n < PV_K1_BATCH) {          // recorded: launched by pv_k1_wgrad_flush
    static const int fat_env = pv_exp_int("PV_K1_NOFAT", 0) ? 0 : 1;
    if (fat_env && Ci > 16 && Co > 16 && rows * 2 * (int64_t)(Co > Ci ? Co : Ci) * 4 < (int64_t)1 << 31) {                                // the throughput form (never more splits: ws is sized for the lean one)
      const int wb = taps == 3 ? 32 : 64;
      static const int fatb = pv_exp_int("PV_K1_FATB", 2);   // register batches per wave (experiments)
      const int nbw = fatb >= 1 && fatb <= 16 ? fatb : 2;
      int64_t ns = (rows + 4 * nbw * wb - 1) / (4 * nbw * wb);          // two register batches per wave
      if (ns > a.nsplit) ns = a.nsplit;
      if (ns < 1) ns = 1;
      a.fat = 1;
      a.mtiles = (Co + 31) / 32; a.ntiles = (Ci + 31) / 32;
      a.chunk = ((rows + ns - 1) / ns + 4 * wb - 1) / (4 * wb) * (4 * wb);   // whole workgroup batches (the body relies on it)
      a.nsplit = (int)((rows + a.chunk - 1) / a.chunk);
      a.part_b = db ? a.part + (int64_t)a.nsplit * Co * Ci * taps : nullptr;
      a.nblk = a.mtiles * a.ntiles * a.nsplit;
      // the wide form where the channel counts are whole tiles of it (PV_K1_NOWIDE=1 in the experiments build: the fat form)
      static const int wide_env = pv_exp_int("PV_K1_NOWIDE", 0) ? 0 : 1;
      int mj = 0, nj = 0;
      if (Co % 64 == 0 && Ci % 32 == 0) { mj = 4; nj = 2; }
      else if (Co % 32 == 0 && Ci % 64 == 0) { mj = 2; nj = 4; }
      else if (Co % 32 == 0 && Ci % 32 == 0) { mj = 2; nj = 2; }
      if (wide_env && mj && (taps == 1 || taps == 3)) {
        const int wbw = taps == 3 ? (nj == 4 ? 16 : 32) : 64;          // (K1Wide::WB)
        static const int wideb = pv_exp_int("PV_K1_WIDEB", 1);          // register batches per wave
        const int nbq = wideb >= 1 && wideb <= 16 ? wideb : 1;
        int64_t nsw = (rows + 4 * nbq * wbw - 1) / (4 * nbq * wbw);
        if (nsw > k1_wg_splits(rows, Ci, Co, taps)) nsw = k1_wg_splits(rows, Ci, Co, taps);   // (ws is sized for the lean form's count)
        if (nsw < 1) nsw = 1;
        a.fat = 2; a.mj = mj; a.nj = nj;
        a.mtiles = Co / (16 * mj); a.ntiles = Ci / (16 * nj);
        a.chunk = ((rows + nsw - 1) / nsw + 4 * wbw - 1) / (4 * wbw) * (4 * wbw);
        a.nsplit = (int)((rows + a.chunk - 1) / a.chunk);
        a.part_b = db ? a.part + (int64_t)a.nsplit * Co * Ci * taps : nullptr;
        a.nblk = a.mtiles * a.ntiles * a.nsplit;
      }
    }
    defer->k1b->e[defer->k1b->n++] = a;
  } else {
    hipLaunchKernelGGL(pv_k1_wgrad_kernel, dim3((unsigned)a.nblk), dim3(256), 0, s, a);
    PV_LAUNCH_CHECK();
  }
  return pv_wgrad_finish(deferred ? defer : nullptr, a.part, a.nsplit, (int64_t)Co * Ci * taps, dw, a.part_b, Co, db, s);
}

int pv_k1_wgrad_flush(PvK1Batch* b, hipStream_t s) {
  if (!b || b->n == 0) return 0;
  K1Tab t{};
  t.n = b->n;
  int tot = 0;
  for (int k = 0; k < b->n; ++k) { t.e[k] = b->e[k]; t.blk0[k] = tot; tot += b->e[k].nblk; }
  t.blk0[b->n] = tot;
  b->n = 0;
  if (tot < 1) return 0;
  if (pv_exp_int("PV_K1_SKIP", 0)) return 0;          // (experiments build: what would the step cost without this launch?  wrong gradients)
  hipLaunchKernelGGL(pv_k1_wgrad_table_kernel, dim3((unsigned)tot), dim3(256), 0, s, t);
  PV_LAUNCH_CHECK();
  return 0;
}

// dw (Co, Ci) = g (rows, Co)^T in (rows, Ci); db (Co) = column sums of g (null: none)
int pv_k1_wgrad(const float* g, const float* in, int64_t rows, int Ci, int Co, float* dw, float* db, void* ws, int64_t ws_bytes,
                hipStream_t s, PvFinishList* defer, int up) {
  return k1_wgrad_launch(g, in, rows, 1, 1, Ci, Co, dw, db, ws, ws_bytes, s, defer, up);
}

// the kernel-3, padding-1 1-D convolution's weight gradient on the same kernel (three shifted, boundary-masked problems):
// dw (Co, Ci, 3), g (B, L, Co), in (B, L, Ci)
int pv_conv3_1d_wgrad_lean(const float* g, const float* in, int B, int L, int Ci, int Co, float* dw, float* db, void* ws,
                           int64_t ws_bytes, hipStream_t s, PvFinishList* defer) {
  return k1_wgrad_launch(g, in, (int64_t)B * L, L, 3, Ci, Co, dw, db, ws, ws_bytes, s, defer, 0);
}

// ---- test hooks (tests/test_gpu_conv_kernels.py) --------------------------------------------------------------------
extern "C" int pv_debug_k1(int what, const float* a0, const float* a1, const float* a2, float* o0, float* o1, long long rows,
                           int Ci, int Co, int act, const float* eg_y, void* ws, long long ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int up = what >> 2;                                                                       // (+4: the fused-upsample forms)
  what &= 3;
  if (what == 0) return pv_k1_fwd(a0, rows, Ci, a1, a2, o0, Co, act, s, up);                      // in, w, bias -> out
  if (what == 1) return pv_k1_dgrad(a0, rows, Co, a1, o0, Ci, eg_y, act, s, up);                  // g, w -> gin
  if (what == 2) return pv_k1_wgrad(a0, a1, rows, Ci, Co, o0, o1, ws, ws_bytes, s, nullptr, up);  // g, in -> dw, db
  if (what == 3)                                                                                  // kernel 3, 1-D: act = L
    return pv_conv3_1d_wgrad_lean(a0, a1, (int)(rows / act), act, Ci, Co, o0, o1, ws, ws_bytes, s, nullptr);
  return PV_EINVAL;
}
// the batched form: a kernel-1 and a kernel-3 (1-D, samples of L positions) weight gradient of the same (g, in) recorded in one
// PvK1Batch, one table launch, one reduction launch
extern "C" int pv_debug_k1_batch(const float* g, const float* in, long long rows, int L, int Ci, int Co, float* dw1, float* db1,
                                 float* dw3, float* db3, void* ws, long long ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PvFinishList fin{};
  PvK1Batch kb{};
  fin.base = (char*)ws; fin.cap = ws_bytes; fin.k1b = &kb;
  PV_TRY(pv_k1_wgrad(g, in, rows, Ci, Co, dw1, db1, nullptr, 0, s, &fin, 0));
  PV_TRY(pv_conv3_1d_wgrad_lean(g, in, (int)(rows / L), L, Ci, Co, dw3, db3, nullptr, 0, s, &fin));
  if (kb.n != 2) return PV_EINVAL;                   // (both must have been recorded)
  PV_TRY(pv_k1_wgrad_flush(&kb, s));
  return pv_wgrad_finish_all(&fin, s);
}
extern "C" long long pv_debug_k1_ws(long long rows, int Ci, int Co) { return k1_wg_ws(rows, Ci, Co, 3); }
