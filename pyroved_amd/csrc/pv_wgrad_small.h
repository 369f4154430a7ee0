// pv_wgrad_small.h — the one-tile-per-workgroup weight-gradient launch's device code (pv_wgrad.hip has the story), shared with the
// launch that runs it NEXT TO the record sums of the same step (pv_elementwise.hip: pv_rec_wgrad_kernel).
#pragma once
#include "pv_common.h"
#include "pv_kernels.h"

#define WG_WAVES 4
#define WG_CHUNK 64            // k's per register batch (16 MFMAs)
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct PvWgradSmall {
  PvGemm g[4];
  int tile_end[4];             // exclusive prefix sums of the problems' tile counts
  int n;
  // fused Adam (pv_common.h: PvAdamFuse): blocks >= tile_end[3] are guests that update every element outside the
  // launch's own outputs, rng[2i] / rng[2i+1] = problem i's weight / bias gradient range in the flat buffer
  int adam_on;
  PvAdamFuse ad;
  int64_t rng_lo[8], rng_hi[8];
  // one more guest (the last block) when fin_scalars is set: the step's loss scalars (pv_finish_scalars)
  const float* fin_llb; float* fin_scalars; const float* fin_kl_part; int fin_B, fin_n_part; float fin_beta;
  unsigned* tick;              // not null: a word the loss block increments — the step's hand-off generation (PvEncFold::coop_flags)
};
// fills w from the problems (validation as pv_wgrad_small's); returns the tile count or a negative error
int pv_wgrad_small_fill(PvWgradSmall& w, const PvGemm* gs, int n, const PvAdamFuse* adam, const PvFinishArgs* fin, int* guests);

// Block t of nblk (tiles, then Adam guests, then the loss-scalars block)
__device__ __forceinline__ void pv_wgrad_small_block(const PvWgradSmall& w, int t, int nblk, float (*part)[16][17], float (*rpart)[16]) {
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (w.fin_scalars && t == nblk - 1) {
    pv_finish_scalars_block(w.fin_llb, w.fin_B, w.fin_scalars, w.fin_kl_part, w.fin_n_part, w.fin_beta, &part[0][0][0]);
    if (w.tick && tid == 0) *w.tick = *w.tick + 1u;
    return;
  }
  if (w.adam_on && t >= w.tile_end[3]) {              // guest: Adam over everything this launch does not produce
    const PvAdamFuse& a = w.ad;
    const int64_t stride = (int64_t)(nblk - (w.fin_scalars ? 1 : 0) - w.tile_end[3]) * blockDim.x;
    for (int64_t i = (int64_t)(t - w.tile_end[3]) * blockDim.x + tid; i < a.n; i += stride) {
      bool own = false;
#pragma unroll
      for (int k = 0; k < 8; ++k) own = own || (i >= w.rng_lo[k] && i < w.rng_hi[k]);
      if (!own) pv_adam_update(a.p, a.g, a.m, a.v, i, a.g[i], a.b1, a.b2, a.eps, a.step_size, a.bc2_sqrt);
    }
    return;
  }
  int pi = 0;
  while (t >= w.tile_end[pi]) ++pi;
  if (pi > 0) t -= w.tile_end[pi - 1];
  const PvGemm& g = w.g[pi];
  const int nbs = (g.N + 15) / 16;
  const int mb = t / nbs, nb = t - mb * nbs;
  // MFMA operands: A lane (m = r, k = q), B lane (n = r, k = q); one instruction covers 4 k's
  const int m = 16 * mb + r, n = 16 * nb + r;
  const bool mok = m < g.M, nok = n < g.N;
  const float* ap = g.A + (int64_t)(mok ? m : 0) * g.a_rs;
  const float* bp = g.B + (int64_t)(nok ? n : 0) * g.b_cs;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float rs = 0.0f;
  float a[2][WG_CHUNK / 4], b[2][WG_CHUNK / 4];
  auto load_b = [&](int k0, float (&bv)[WG_CHUNK / 4]) {
    if (k0 + WG_CHUNK <= g.K) {
      // a whole batch inside K (every batch of the usual minibatches): a running pointer — the general form below spends a
      // 64-bit multiply-add (quarter rate) and a clamp per operand, ~1 k cycles in front of the loads of a latency-bound launch
      const float* pb = bp + (int64_t)(k0 + q) * g.b_rs;
      const int64_t sb = 4 * g.b_rs;
#pragma unroll
      for (int s = 0; s < WG_CHUNK / 4; ++s) {
        const float y = *pb;
        pb += sb;
        bv[s] = nok ? y : 0.0f;
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < WG_CHUNK / 4; ++s) {
      const int k = k0 + 4 * s + q;
      const int kc = k < g.K ? k : g.K - 1;
      const float y = bp[(int64_t)kc * g.b_rs];
      bv[s] = (k < g.K && nok) ? y : 0.0f;
    }
  };
  auto load_a = [&](int k0, float (&av)[WG_CHUNK / 4]) {
    if (k0 + WG_CHUNK <= g.K) {
      const float* pa = ap + (int64_t)(k0 + q) * g.a_cs;
      const int64_t sa = 4 * g.a_cs;
#pragma unroll
      for (int s = 0; s < WG_CHUNK / 4; ++s) {
        const float x = *pa;
        pa += sa;
        av[s] = mok ? x : 0.0f;
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < WG_CHUNK / 4; ++s) {
      const int k = k0 + 4 * s + q;
      const int kc = k < g.K ? k : g.K - 1;
      const float x = ap[(int64_t)kc * g.a_cs];
      av[s] = (k < g.K && mok) ? x : 0.0f;
    }
  };
  auto consume = [&](const float (&av)[WG_CHUNK / 4], const float (&bv)[WG_CHUNK / 4]) {
#pragma unroll
    for (int s = 0; s < WG_CHUNK / 4; ++s) {
      acc[s & 3] = MFMA4(av[s], bv[s], acc[s & 3]);
      rs += av[s];
    }
  };
  const int KS = WG_CHUNK * WG_WAVES;                         // k stride between a wave's batches
  const bool any = WG_CHUNK * wave < g.K;
  if (any) {
    load_a(WG_CHUNK * wave, a[0]);
    load_b(WG_CHUNK * wave, b[0]);
    for (int k0 = WG_CHUNK * wave; k0 < g.K; k0 += 2 * KS) {  // two register batches, the other one in flight
      const bool more1 = k0 + KS < g.K, more2 = k0 + 2 * KS < g.K;
      if (more1) { load_a(k0 + KS, a[1]); load_b(k0 + KS, b[1]); }
      consume(a[0], b[0]);
      if (more2) { load_a(k0 + 2 * KS, a[0]); load_b(k0 + 2 * KS, b[0]); }
      if (more1) consume(a[1], b[1]);
    }
  }
  const f32x4 c = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  // C/D layout: lane (n = r, q), reg i -> m = 16*mb + 4q + i
#pragma unroll
  for (int i = 0; i < 4; ++i) part[wave][4 * q + i][r] = c[i];
  rs = pv_sum_rows(rs);
  if (q == 0) rpart[wave][r] = rs;
  __syncthreads();
  {
    const int mm = tid >> 4, nn = tid & 15, mo = 16 * mb + mm, no = 16 * nb + nn;
    const PvAdamFuse& a = w.ad;
    if (mo < g.M && no < g.N) {
      const float c = (part[0][mm][nn] + part[1][mm][nn]) + (part[2][mm][nn] + part[3][mm][nn]);
      if (w.adam_on) pv_adam_update(a.p, a.g, a.m, a.v, (g.C - a.g) + (int64_t)mo * g.ldc + no, c, a.b1, a.b2, a.eps,
                                    a.step_size, a.bc2_sqrt);
      else g.C[(int64_t)mo * g.ldc + no] = c;
    }
    if (g.rowsumA && nb == 0 && tid < 16 && 16 * mb + tid < g.M) {
      const float c = (rpart[0][tid] + rpart[1][tid]) + (rpart[2][tid] + rpart[3][tid]);
      if (w.adam_on) pv_adam_update(a.p, a.g, a.m, a.v, (g.rowsumA - a.g) + 16 * mb + tid, c, a.b1, a.b2, a.eps,
                                    a.step_size, a.bc2_sqrt);
      else g.rowsumA[16 * mb + tid] = c;
    }
  }
}
