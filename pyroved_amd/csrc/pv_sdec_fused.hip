// pv_sdec_fused.hip — fused persistent forward+backward kernel of the spatial decoder
// (pyroved/nets/fc.py:155-237 sDecoderNet + coord_latent, with utils/coord.py:47-88's transform and
// utils/prob.py:25-29's likelihood fused in) for the default architecture hidden_dim_d = [128, 128], tanh.
//
// Why one kernel: the loss is a plain sum over pixels, so dL/d(logit) of a pixel is known as soon as its
// forward finishes; forward, loss, dgrad and wgrad of a row tile can run back to back with every
// activation resident on chip.  HBM traffic per step drops from ~1.2 GB (layer-by-layer) to the
// observations (4 B/pixel) + 20 B/pixel of per-row outputs; the kernel is bound by the fp32 matrix
// pipe (v_mfma_f32_16x16x4_f32, 157 TF peak), not by HBM.
//
// Mapping (one persistent 512-thread workgroup per CU, 8 waves = 2 per SIMD so one wave's tanh / loss
// VALU work overlaps its partner's MFMAs):
//   * rows (b, n) are flattened; a wave owns 16 consecutive rows ("unit") at a time, a workgroup 8 units.
//   * Every layer is computed TRANSPOSED, D[j][r] = sum_k W[j][k] h[r][k]: the weight is the MFMA A
//     operand (read from LDS: row-major, stride 132 floats, one ds_read_b128 feeds 4 MFMAs) and the
//     activation is the B operand.  The 16x16x4 C/D layout (lane = column r, 4 regs = rows 4q..4q+3)
//     is then exactly the B-operand layout of the next layer with k = 16*jb + 4*q + i, so activations
//     stay in registers from the coordinate layer to the logit and back down the dgrad chain —
//     no transposes, no LDS round trips, no workgroup barriers in forward/dgrad.
//   * wgrad contracts over rows, which live on different waves: each wave owns a 16x128 slice of dW1
//     and dW2 in accumulator registers for the whole kernel (64 VGPRs) and the workgroup exchanges
//     (dpre, h) through a 16-row LDS staging buffer, one unit at a time.
//   * Per-workgroup partial gradients are written once at the end and summed in a fixed order by
//     pv_sdec_fused_reduce (no float atomics: bit-reproducible).
// LDS: W1, W2 (2 x 66 KiB), staging (18 KiB), small vectors: 157.5 KiB of the CU's 160 KiB.
#include "pv_sdec_fused.h"
#include <stdlib.h>

#define LDW 132        // LDS row stride of the weight images (floats): conflict-free ds_read_b128 over 16 rows
#define LDST 144       // LDS row stride of the staging buffers: stride % 32 == 16 -> conflict-free operand reads
#define OFF_W1 0
#define OFF_W2 (FD_H * LDW)
#define OFF_STA (2 * FD_H * LDW)
#define OFF_STB (OFF_STA + FD_UNIT * LDST)
#define OFF_WC0 (OFF_STB + FD_UNIT * LDST)
#define OFF_WC1 (OFF_WC0 + FD_H)
#define OFF_BC (OFF_WC1 + FD_H)
#define OFF_WO (OFF_BC + FD_H)
#define OFF_B1 (OFF_WO + FD_H)
#define OFF_B2 (OFF_B1 + FD_H)
#define OFF_DWO (OFF_B2 + FD_H)
#define OFF_INFO (OFF_DWO + FD_WAVES * FD_H)
#define OFF_RED (OFF_INFO + 64)
#define FD_LDS_FLOATS (OFF_RED + 64)
#define FD_THREADS (64 * FD_WAVES)

#define LOG_SQRT_2PI 0.91893853320467274178f
#define BERN_EPS 1.1920928955078125e-07f

// keeps the machine scheduler from hoisting a whole layer's LDS operand loads ahead of its MFMAs
// (which would need hundreds of VGPRs); the partner wave on the SIMD hides the ds_read latency instead
#define FD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#ifdef FD_EXP_NO_LDS_OPERANDS   // experiment build only: MFMA weight operands come from one LDS word (no traffic)
#define FD_WADDR(x) 0
#else
#define FD_WADDR(x) (x)
#endif

// tanh(x) = 1 - 2 / (e^{2x} + 1) on the raw v_exp_f32 / v_rcp_f32 (1 ulp each): 5 VALU issues.
// On gfx950 the f32-input MFMA shares the FP32 ALUs with VALU work (measured: every VALU instruction placed
// between two v_mfma_f32_16x16x4_f32 adds ~3.5 cycles to the stream, scripts/ubench/mfma_f32.hip), so the
// activation's instruction count is paid in full.  Absolute error <= ~2e-7 (cancellation near 0 is absolute,
// not relative: harmless for the sums it feeds); FD_TANH_POLY adds an odd polynomial for |x| < 0.125 (+7 issues).
__device__ __forceinline__ float fd_tanh(float x) {
#ifdef FD_EXP_NO_TANH      // experiment build only: prices the VALU cost of the activations
  return x * 0.5f;
#endif
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);       // e^{2x}
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
#ifdef FD_TANH_POLY
  const float x2 = x * x;
  const float p = x * (1.0f + x2 * (-0.33333334f + x2 * 0.13333334f));
  return fabsf(x) < 0.125f ? p : t;
#else
  return t;
#endif
}

// D[j][r] = b[j] + sum_k W[j][k] in[r][k]   (transposed layer, see header); `out` is left PRE-activation
// except that, when ACT_HALF0, tanh is applied to out[0..3] while out[4..7] is still accumulating.
// The stream is 16 groups of (4 x ds_read_b128 -> 16 MFMAs); the next group's operands are fetched before the
// current group's MFMAs issue (explicit double buffer) and the scheduling fence keeps the compiler from hoisting
// more.  VALU work is placed INSIDE the groups so it issues in the shadow of this wave's own MFMAs (the two waves
// of a SIMD run in lock-step, so the partner cannot be relied on to cover it):
//   prep(jb) produces the input block in[jb] one group before its first use (the previous layer's activation),
//   and the finished half of the outputs is activated during the second half's groups.
template <bool ACT_HALF0, class Prep>
__device__ __forceinline__ void fd_layer_fwd(const float* __restrict__ Ws, const float* __restrict__ bs,
                                             f32x4 (&in)[8], f32x4 (&out)[8], int r, int q, Prep prep) {
  prep(0);
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) out[ob] = *reinterpret_cast<const f32x4*>(bs + 16 * ob + 4 * q);
  const float* wb = Ws + r * LDW + 4 * q;
  f32x4 a[2][4];
#pragma unroll
  for (int o = 0; o < 4; ++o) a[0][o] = *reinterpret_cast<const f32x4*>(wb + FD_WADDR(16 * o * LDW));
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int half = g >> 3, jb = g & 7, oh = 4 * half;
    if (g + 1 < 16) {
      const int hn = (g + 1) >> 3, jn = (g + 1) & 7;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        a[(g + 1) & 1][o] = *reinterpret_cast<const f32x4*>(wb + FD_WADDR(16 * (4 * hn + o) * LDW + 16 * jn));
    }
    FD_SCHED_FENCE();        // the prefetch must ISSUE before this group's MFMAs (else it is sunk behind them)
    if (half == 0 && jb + 1 < 8) prep(jb + 1);
    if (ACT_HALF0 && half == 1) {
      out[(2 * jb) >> 2][(2 * jb) & 3] = fd_tanh(out[(2 * jb) >> 2][(2 * jb) & 3]);
      out[(2 * jb + 1) >> 2][(2 * jb + 1) & 3] = fd_tanh(out[(2 * jb + 1) >> 2][(2 * jb + 1) & 3]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int o = 0; o < 4; ++o) out[oh + o] = MFMA(a[g & 1][o][i], in[jb][i], out[oh + o]);
    FD_SCHED_FENCE();
  }
}

// D[k][r] = sum_j W[j][k] dp[r][j]   (dgrad); 32 groups of (8 x ds_read_b32 -> 8 MFMAs), operands double-buffered.
// side(g) is independent VALU work issued in the shadow of group g's MFMAs.  No epilogue here.
template <class Side>
__device__ __forceinline__ void fd_layer_dgrad(const float* __restrict__ Ws, const f32x4 (&dp)[8],
                                               f32x4 (&out)[8], int r, int q, Side side) {
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) out[kb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const float* wb = Ws + 4 * q * LDW + r;
  float a[2][8];
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) a[0][kb] = wb[16 * kb];
#pragma unroll
  for (int g = 0; g < 32; ++g) {
    const int jb = g >> 2, i = g & 3;
    if (g + 1 < 32) {
      const int jn = (g + 1) >> 2, in_ = (g + 1) & 3;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) a[(g + 1) & 1][kb] = wb[(16 * jn + in_) * LDW + 16 * kb];
    }
    FD_SCHED_FENCE();
    side(g);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) out[kb] = MFMA(a[g & 1][kb], dp[jb][i], out[kb]);
    FD_SCHED_FENCE();
  }
}

// out *= 1 - h^2   (through the tanh that produced h)
__device__ __forceinline__ void fd_mul_dtanh(f32x4 (&out)[8], const f32x4 (&h)[8]) {
#pragma unroll
  for (int kb = 0; kb < 8; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[kb][i] *= 1.0f - h[kb][i] * h[kb][i];
}

// raw-instruction transcendental helpers (v_exp_f32 / v_log_f32 / v_rcp_f32: 1 ulp)
__device__ __forceinline__ float fd_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fd_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fd_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ void fd_stage_write(float* __restrict__ st, const f32x4 (&v)[8], int r, int q) {
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) *reinterpret_cast<f32x4*>(st + r * LDST + 16 * jb + 4 * q) = v[jb];
}

// dW[16*wave + ..][k] += sum over the 16 staged rows of dpre[row][16*wave + j'] * h[row][k]
__device__ __forceinline__ void fd_wgrad_consume(const float* __restrict__ stA, const float* __restrict__ stB,
                                                 f32x4 (&accW)[8], float& db, int wave, int r, int q) {
  const float* ab = stA + q * LDST + 16 * wave + r;
  const float* bb = stB + q * LDST + r;
  float a[2], bv[2][8];
  a[0] = ab[0];
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) bv[0][kb] = bb[16 * kb];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s + 1 < 4) {
      a[(s + 1) & 1] = ab[4 * (s + 1) * LDST];
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) bv[(s + 1) & 1][kb] = bb[4 * (s + 1) * LDST + 16 * kb];
    }
    FD_SCHED_FENCE();
    db += a[s & 1];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) accW[kb] = MFMA(a[s & 1], bv[s & 1][kb], accW[kb]);
    FD_SCHED_FENCE();
  }
}

__device__ __forceinline__ float fd_sum_q(float v) {      // over the 4 lanes l, l^16, l^32, l^48
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ __forceinline__ float fd_sum_r(float v) {      // over the 16 lanes of a row group
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

template <bool GRADS>
__global__ __launch_bounds__(FD_THREADS, 2) void pv_sdec_fused_kernel(PvFused f) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int g = blockIdx.x, G = gridDim.x;

  // ---- stage the weights in LDS (once per kernel) ----
  for (int idx = tid; idx < FD_H * (FD_H / 4); idx += FD_THREADS) {
    const int row = idx >> 5, c4 = idx & 31;
    *reinterpret_cast<f32x4*>(sm + OFF_W1 + row * LDW + 4 * c4) = reinterpret_cast<const f32x4*>(f.W1)[idx];
    *reinterpret_cast<f32x4*>(sm + OFF_W2 + row * LDW + 4 * c4) = reinterpret_cast<const f32x4*>(f.W2)[idx];
  }
  for (int j = tid; j < FD_H; j += FD_THREADS) {
    sm[OFF_WC0 + j] = f.Wc[j * f.cd];
    sm[OFF_WC1 + j] = f.cd == 2 ? f.Wc[j * 2 + 1] : 0.0f;
    sm[OFF_BC + j] = f.bc[j];
    sm[OFF_WO + j] = f.wo[j];
    sm[OFF_B1 + j] = f.b1[j];
    sm[OFF_B2 + j] = f.b2[j];
  }
  for (int j = tid; j < FD_WAVES * FD_H; j += FD_THREADS) sm[OFF_DWO + j] = 0.0f;
  __syncthreads();

  const float* W1s = sm + OFF_W1;
  const float* W2s = sm + OFF_W2;
  float* stA = sm + OFF_STA;
  float* stB = sm + OFF_STB;
  const float bo = f.bo[0];

  // persistent accumulators: this wave's 16x128 slices of dW1 / dW2, bias / coordinate-layer sums
  f32x4 accW1[8], accW2[8];
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) { accW1[kb] = f32x4{0, 0, 0, 0}; accW2[kb] = f32x4{0, 0, 0, 0}; }
  float db1 = 0.0f, db2 = 0.0f, aWc0 = 0.0f, aWc1 = 0.0f, ahz = 0.0f, dbo = 0.0f;
  int cur_b = -1;
  const int upb = f.N / FD_UNIT;                       // units per sample (N % 16 == 0)

  auto flush_hz = [&](int b) {
    // sample b's rows end (or the workgroup's do): publish this workgroup's partial dL/d(hz[b])
    const float t = fd_sum_q(ahz);
    const int64_t ub = (int64_t)b * upb;
    const int gfirst = (int)(((ub + 1) * G + f.units - 1) / f.units) - 1;   // first workgroup touching sample b
    if (q == 0) f.part_hz[((int64_t)b * f.kmax + (g - gfirst)) * FD_H + 16 * wave + r] = t;
  };

  const int64_t u_lo = (int64_t)g * f.units / G, u_hi = (int64_t)(g + 1) * f.units / G;
  for (int64_t u0 = u_lo; u0 < u_hi; u0 += FD_WAVES) {
    const int nact = (int)((u_hi - u0) < FD_WAVES ? (u_hi - u0) : FD_WAVES);
    const bool active = wave < nact;
    const int unit = (int)u0 + (active ? wave : 0);          // 32-bit index math: units < 2^31 / 16
    const int b = unit / upb, n = (unit - b * upb) * FD_UNIT + r;
    const int64_t row = (int64_t)unit * FD_UNIT + r;
    // three register tiles (16 rows x 128 cols each, 32 VGPRs) are rotated through the roles
    // h0 -> h1 -> h2/dpre2 -> dpre1 -> h0 (recomputed) -> dpre0 so that at most three are live
    f32x4 tA[8], tB[8], tC[8];
    float x0 = 0.0f, x1 = 0.0f, u0c = 0.0f, u1c = 0.0f, sc = 1.0f;
    const float* hzb = f.hz + (int64_t)b * FD_H;

    // coordinate layer, one 16-column block: h0[jb] = tanh(Wc x' + bc + hz[b])   (fc.py:226-237)
    auto coord_block = [&](f32x4 (&h0)[8], int jb) {
      const int j = 16 * jb + 4 * q;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(sm + OFF_WC0 + j);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(sm + OFF_WC1 + j);
      const f32x4 bc = *reinterpret_cast<const f32x4*>(sm + OFF_BC + j);
      const f32x4 hz = *reinterpret_cast<const f32x4*>(hzb + j);
#pragma unroll
      for (int i = 0; i < 4; ++i) h0[jb][i] = fd_tanh(w0[i] * x0 + w1[i] * x1 + bc[i] + hz[i]);
    };

    if (active) {
      // ---- x' = rotate/scale/translate(grid[n])   (coord.py:47-88) ----
      const float* t = f.tp + (int64_t)b * 8;
      if (f.cd == 2) {
        const float gx = f.grid[2 * n], gy = f.grid[2 * n + 1];
        u0c = gx * t[0] - gy * t[1];
        u1c = gx * t[1] + gy * t[0];
        sc = t[2];
        x0 = u0c * sc + t[3];
        x1 = u1c * sc + t[4];
      } else {
        u0c = f.grid[n];
        x0 = u0c + t[3];
      }
      // layer 1: its input h0 (tA) is produced block by block inside the MFMA stream; tB = pre-activation 1
      fd_layer_fwd<false>(W1s, sm + OFF_B1, tA, tB, r, q, [&](int jb) { coord_block(tA, jb); });
      // layer 2: activates its own input in place (tB -> h1) one block ahead; tC[0..3] = h2, tC[4..7] raw
      fd_layer_fwd<true>(W2s, sm + OFF_B2, tB, tC, r, q, [&](int jb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tB[jb][i] = fd_tanh(tB[jb][i]);
      });
#pragma unroll
      for (int ob = 4; ob < 8; ++ob)
#pragma unroll
        for (int i = 0; i < 4; ++i) tC[ob][i] = fd_tanh(tC[ob][i]);
      // ---- output layer + likelihood ----
      float part = 0.0f;
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(sm + OFF_WO + 16 * jb + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) part += tC[jb][i] * wv[i];
      }
      const float a = fd_sum_q(part) + bo;
      const float xv = f.x[f.x_units > 0 ? row % (f.x_units * FD_UNIT) : row];
      float ll, dlda, locv;
      if (f.lik == PV_LIK_BERNOULLI) {
        // torch Bernoulli(probs=sigmoid(a)).log_prob(x): clamp_probs -> logits -> -BCEWithLogits
        const float pr = fd_rcp(1.0f + fd_exp(-a));
        const float pc = fminf(fmaxf(pr, BERN_EPS), 1.0f - BERN_EPS);
        const float lg = fd_log(pc) - fd_log(1.0f - pc);
        ll = -(fmaxf(lg, 0.0f) - lg * xv + fd_log(1.0f + fd_exp(-fabsf(lg))));
        const float mask = (pr >= BERN_EPS && pr <= 1.0f - BERN_EPS) ? 1.0f : 0.0f;
        dlda = (fd_rcp(1.0f + fd_exp(-lg)) - xv) * mask;
        locv = pr;
      } else if (f.lik == PV_LIK_CBERNOULLI) {
        pv_cbern(a, xv, ll, dlda, locv);
      } else {
        const float pr = f.sigmoid_out ? fd_rcp(1.0f + fd_exp(-a)) : a;
        const float d = xv - pr;
        ll = -(d * d) / (2.0f * f.sig * f.sig) - fd_log(f.sig) - LOG_SQRT_2PI;
        dlda = -d / (f.sig * f.sig) * (f.sigmoid_out ? pr * (1.0f - pr) : 1.0f);
        locv = pr;
      }
      if (f.sw) dlda *= f.sw[row / f.N];          // (jiVAE) weight of the row's sample
      if (q == 0) {
        if (f.llrow) f.llrow[row] = ll;
        if (f.loc) f.loc[row] = locv;
      }
      if (GRADS) {
        if (q == 0) dbo += dlda;
        // d(wo) partial: sum over this unit's rows, accumulated in the wave's private LDS slot
        float* dwo = sm + OFF_DWO + wave * FD_H;
        if (!(f.ablate & 8))
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
          f32x4 tv;
#pragma unroll
          for (int i = 0; i < 4; ++i) tv[i] = fd_sum_r(dlda * tC[jb][i]);
          if (r == 0) {
            f32x4* p = reinterpret_cast<f32x4*>(dwo + 16 * jb + 4 * q);
            *p = *p + tv;
          }
        }
        // tC = dL/d(pre-activation of layer 2)
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(sm + OFF_WO + 16 * jb + 4 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) tC[jb][i] = dlda * wv[i] * (1.0f - tC[jb][i] * tC[jb][i]);
        }
      }
    }
    if (!GRADS) continue;

    // ---- wgrad of layer 2: exchange (dpre2 = tC, h1 = tB) one unit at a time ----
    const int nex = (f.ablate & 1) ? 0 : nact;
    for (int c = 0; c < nex; ++c) {
      if (wave == c) { fd_stage_write(stA, tC, r, q); fd_stage_write(stB, tB, r, q); }
      __syncthreads();
      fd_wgrad_consume(stA, stB, accW2, db2, wave, r, q);
      __syncthreads();
    }
    if (active && !(f.ablate & 4)) {
      fd_layer_dgrad(W2s, tC, tA, r, q, [](int) {});
      fd_mul_dtanh(tA, tB);                              // tA = dpre1 = (dpre2 W2) * (1 - h1^2)
      // tB = h0, recomputed (cheaper than keeping it live) in the shadow of the next dgrad's MFMAs
      fd_layer_dgrad(W1s, tA, tC, r, q, [&](int g) { if ((g & 3) == 0) coord_block(tB, g >> 2); });
      fd_mul_dtanh(tC, tB);                              // tC = dpre0 = (dpre1 W1) * (1 - h0^2)
    }
    // ---- wgrad of layer 1: exchange (dpre1 = tA, h0 = tB) ----
    for (int c = 0; c < nex; ++c) {
      if (wave == c) { fd_stage_write(stA, tA, r, q); fd_stage_write(stB, tB, r, q); }
      __syncthreads();
      fd_wgrad_consume(stA, stB, accW1, db1, wave, r, q);
      __syncthreads();
    }
    // ---- coordinate layer backward, row-local part: d(x') -> d(phi, scale, shift) per row ----
    if (active) {
      float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sm + OFF_WC0 + 16 * jb + 4 * q);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sm + OFF_WC1 + 16 * jb + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) { d0 += tC[jb][i] * w0[i]; d1 += tC[jb][i] * w1[i]; }
      }
      d0 = fd_sum_q(d0);
      d1 = fd_sum_q(d1);
      if (q == 0) {
        f.rowtp[row] = sc * (d1 * u0c - d0 * u1c);
        f.rowtp[f.M + row] = d0 * u0c + d1 * u1c;
        f.rowtp[2 * f.M + row] = d0;
        f.rowtp[3 * f.M + row] = d1;
      }
    }
    // ---- coordinate layer backward, cross-row part: dWc, dbc/dhz via staged dpre0 ----
    for (int c = 0; c < ((f.ablate & 2) ? 0 : nact); ++c) {
      if (wave == c) {
        fd_stage_write(stA, tC, r, q);
        if (q == 0) { sm[OFF_INFO + r] = x0; sm[OFF_INFO + 16 + r] = x1; }
        if (lane == 0) reinterpret_cast<int*>(sm + OFF_INFO)[32] = b;
      }
      __syncthreads();
      const int bc = reinterpret_cast<const int*>(sm + OFF_INFO)[32];     // workgroup-uniform
      if (bc != cur_b) {
        if (cur_b >= 0) flush_hz(cur_b);
        cur_b = bc;
        ahz = 0.0f;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float v = stA[(4 * s + q) * LDST + 16 * wave + r];
        ahz += v;
        aWc0 += v * sm[OFF_INFO + 4 * s + q];
        aWc1 += v * sm[OFF_INFO + 16 + 4 * s + q];
      }
      __syncthreads();
    }
  }
  if (!GRADS) return;

  // ---- publish this workgroup's partial gradients ----
  if (cur_b >= 0) flush_hz(cur_b);
  float* rec = f.part + (int64_t)g * FD_REC;
#pragma unroll
  for (int kb = 0; kb < 8; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // C/D layout: lane (col k' = r, q), reg i -> dW[16*wave + 4*q + i][16*kb + r]
      rec[(16 * wave + 4 * q + i) * FD_H + 16 * kb + r] = accW1[kb][i];
      rec[FD_H * FD_H + (16 * wave + 4 * q + i) * FD_H + 16 * kb + r] = accW2[kb][i];
    }
  const float t1 = fd_sum_q(db1), t2 = fd_sum_q(db2), tc0 = fd_sum_q(aWc0), tc1 = fd_sum_q(aWc1);
  if (q == 0) {
    rec[2 * FD_H * FD_H + 16 * wave + r] = t1;
    rec[2 * FD_H * FD_H + FD_H + 16 * wave + r] = t2;
    rec[2 * FD_H * FD_H + 2 * FD_H + 16 * wave + r] = tc0;
    rec[2 * FD_H * FD_H + 3 * FD_H + 16 * wave + r] = tc1;
  }
  const float tb = pv_wave_sum(dbo);
  if (lane == 0) sm[OFF_RED + wave] = tb;
  __syncthreads();
  if (tid < FD_H) {
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < FD_WAVES; ++w) v += sm[OFF_DWO + w * FD_H + tid];
    rec[2 * FD_H * FD_H + 4 * FD_H + tid] = v;
  }
  if (tid == 0) {
    float v = 0.0f;
    for (int w = 0; w < FD_WAVES; ++w) v += sm[OFF_RED + w];
    rec[2 * FD_H * FD_H + 5 * FD_H] = v;
  }
}

// ---- per-workgroup records -> flat gradient buffer (fixed summation order) ---------------------------
__global__ __launch_bounds__(256) void pv_sdec_fused_reduce_kernel(const float* __restrict__ part, int G_,
                                                                   float* __restrict__ Gr, PvFusedOffsets o, int cd,
                                                                   int dwo_slots) {
  __shared__ f32x4 sm[4][64];
  pv_sdec_fused_reduce_block(part, G_, Gr, o, cd, dwo_slots, blockIdx.x, sm);
}

int pv_sdec_fused_reduce(const float* part, int grid, float* G, const PvFusedOffsets& o, int cd, int dwo_slots,
                         hipStream_t s) {
  hipLaunchKernelGGL(pv_sdec_fused_reduce_kernel, dim3(PV_FUSED_REDUCE_BLOCKS), dim3(256), 0, s, part, grid, G, o, cd,
                     dwo_slots);
  PV_LAUNCH_CHECK();
  return 0;
}

// ---- host side -----------------------------------------------------------------------------------------
bool pv_sdec_fused_supported(const pv_ivae_plan* p) {
  if (!p || p->coord_dim < 1 || p->coord_dim > 2) return false;
  if (p->n_dec != 2 || p->n_pix % FD_UNIT != 0) return false;
  const pv_layer& c = p->fc_coord;
  const pv_layer& l = p->fc_latent;
  if (c.out_dim != FD_H || c.b_off < 0 || l.out_dim != FD_H) return false;
  for (int i = 0; i < 2; ++i) {
    const pv_layer& d = p->dec[i];
    if (d.in_dim != FD_H || d.out_dim != FD_H || d.act != PV_ACT_TANH || d.b_off < 0) return false;
    if (d.w_off % 4 != 0) return false;          // 16-byte aligned rows for the LDS staging loads
  }
  if (p->out.in_dim != FD_H || p->out.out_dim != 1 || p->out.b_off < 0) return false;
  return true;
}

static int fd_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;          // MI355X; also the answer on a build box without a GPU
  }
  return cus;
}

int pv_sdec_fused_grid(int64_t units) {
  const int cus = fd_num_cus();
  return (int)(units < cus ? (units < 1 ? 1 : units) : cus);
}

int pv_sdec_fused_kmax(int n_pix, int64_t units, int grid) {
  const int64_t upb = n_pix / FD_UNIT;
  return (int)((upb * grid) / (units > 0 ? units : 1)) + 2;
}

int pv_sdec_fused_launch(const PvFused& f_in, int grid, bool grads, hipStream_t s) {
  PvFused f = f_in;
  static const int ablate = pv_exp_int("PV_FD_ABLATE", 0);
  f.ablate = ablate;
  const size_t lds = FD_LDS_FLOATS * sizeof(float);
  PV_TRY(pv_set_dynamic_lds(reinterpret_cast<const void*>(&pv_sdec_fused_kernel<true>), (int)lds));
  PV_TRY(pv_set_dynamic_lds(reinterpret_cast<const void*>(&pv_sdec_fused_kernel<false>), (int)lds));
  if (grads)
    hipLaunchKernelGGL(pv_sdec_fused_kernel<true>, dim3(grid), dim3(FD_THREADS), lds, s, f);
  else
    hipLaunchKernelGGL(pv_sdec_fused_kernel<false>, dim3(grid), dim3(FD_THREADS), lds, s, f);
  PV_LAUNCH_CHECK();
  return 0;
}
