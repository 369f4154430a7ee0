// pv_encoder.hip — the encoder side of the SVI step as three compact kernels.
//   pv_enc_fwd   fcEncoderNet.forward (nets/fc.py:51-61) + Normal.rsample + log q(z|x) + log p(z) +
//                _split_latent (ivae.py:179-189, 217-221; base.py:97-119) + fc_latent(z) (fc.py:230)
//                = pv_enc_l1_kernel (first layer: the wide one, K = n_pix) + pv_enc_fwd_kernel (everything after)
//   pv_enc_dgrad the dgrad chain head -> hidden layers of the encoder's backward
// The encoder is < 1 % of the step's FLOPs but, as a chain of tiny dependent GEMMs, it is latency-bound.  The
// first layer (K = 784) is spread over (row blocks x column blocks) workgroups whose 4 waves split K and keep
// whole register batches of operands in flight; after it one workgroup carries 16 samples through the rest of
// the stack with the activations in LDS and every weight operand of a layer requested before its first MFMA.
// MFMA formulation as in pv_sdec_fused.hip (transposed layers, v_mfma_f32_16x16x4_f32, weights streamed from L2
// as the A operand).
// Supported: hidden widths <= 128 and multiples of 16, input width a multiple of 16, any activation but GELU;
// anything else takes the generic GEMM path of pv_plan.hip.
#include "pv_common.h"
#include "pv_kernels.h"
#include "pv_side.h"
#include <atomic>
#include <random>
#include <stdlib.h>

#define EN_ROWS 16
#define EN_LD 132
#define EN_THREADS 512
#define LOG_SQRT_2PI 0.91893853320467274178f
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float en_block_sum(float v, float* sm /* 8 floats */) {
  v = pv_wave_sum(v);
  pv_lds_barrier();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  pv_lds_barrier();
  return ((sm[0] + sm[1]) + (sm[2] + sm[3])) + ((sm[4] + sm[5]) + (sm[6] + sm[7]));
}

// one transposed layer for this wave's output blocks: D[j][r] = sum_k W[j][k] in[r][k]
//   in: global (GLOBAL_IN, row stride ldin) or LDS (stride EN_LD); K % 16 == 0; out rows j >= out_dim are zero
//   K <= 128: the block's whole weight operand (<= 8 float4 per lane) is requested before the first MFMA — and, for
//   the layers the kernel can see coming, before anything else the kernel does (en_load_w at its top)
__device__ __forceinline__ void en_load_w(const float* __restrict__ W, int K, int out_dim, int ob, int r, int q,
                                          f32x4 (&a)[8]) {
  const int j = 16 * ob + r;                       // this lane's A row
  const bool jok = j < out_dim;
  const float* wrow = W + (int64_t)(jok ? j : 0) * K + 4 * q;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const bool ok = jok && 16 * t < K;
    a[t] = *reinterpret_cast<const f32x4*>(wrow + (16 * t < K ? 16 * t : 0));
    if (!ok) a[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
}
__device__ __forceinline__ f32x4 en_mma(const f32x4 (&a)[8], int K, const float* __restrict__ in /* LDS */, int r, int q) {
  const float* irow = in + r * EN_LD + 4 * q;
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (16 * t < K) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(irow + 16 * t);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i & 1) acc1 = MFMA(a[t][i], b[i], acc1); else acc0 = MFMA(a[t][i], b[i], acc0);
      }
    }
  }
  return acc0 + acc1;
}
__device__ __forceinline__ f32x4 en_layer_block(const float* __restrict__ W, int K, int out_dim, int ob,
                                                const float* __restrict__ in /* LDS */, int r, int q) {
  f32x4 a[8];
  en_load_w(W, K, out_dim, ob, r, q, a);
  return en_mma(a, K, in, r, q);
}
__device__ __forceinline__ f32x4 en_load_bias(const float* bias, int out_dim, int ob, int q) {
  f32x4 b = {0.0f, 0.0f, 0.0f, 0.0f};
  if (bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 16 * ob + 4 * q + i;
      b[i] = j < out_dim ? bias[j] : 0.0f;
    }
  }
  return b;
}

#ifdef EN_TRACE
__device__ long long en_trace_p[16];
#define ENP_STAMP(k) do { if (COHERENT && bx == 0 && by == 0 && threadIdx.x == 0) en_trace_p[(k)] = (long long)__builtin_readcyclecounter(); } while (0)
extern "C" int pv_debug_read_enc_trace_p(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(en_trace_p), 16 * sizeof(long long)); }
#else
#define ENP_STAMP(k) do { } while (0)
#endif
// ---------------------------------------------------------------------------------------------
// first encoder layer: eact[0] = act(x W0^T + b0), one workgroup per (16 rows x 16 outputs), K split over 4 waves
#define L1_WAVES 4
#ifndef L1_STEPS
#define L1_STEPS 7             // k16-steps per register batch: at K = 784 a wave's 13 steps are two batches, both requested at once
#endif                         // (4: three to four batches, two in flight — one more memory round trip per tile; same sums)
// bx / by: output block / row block (by >= rb: guest workgroups, ny row blocks in all); the first 64 * L1_WAVES threads of the
// workgroup take part (NW = 4 or 8 waves split K).  Returns true when this workgroup wrote a tile of eact[0] (the merged launch then signals it).
template <bool COHERENT, int NW>
__device__ __forceinline__ bool enc_l1_body(const PvEncFwd& e, int bx, int by, int nx, int ny, float (*part)[EN_ROWS][17]) {
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = (e.B + EN_ROWS - 1) / EN_ROWS;
  if (by >= rb) {                                  // guest work: the decoder kernel's weight images (pv_fb_layout.h)
    const int64_t blk = (int64_t)(by - rb) * nx + bx;
    pv_fb_prep(e.prep, blk * (64 * NW) + tid, (int64_t)(ny - rb) * nx * (64 * NW), NW, wave);
    return false;
  }
  ENP_STAMP(0);
  const pv_layer l = e.enc[0];
  const int ob = bx, row0 = by * EN_ROWS, K = l.in_dim;
  const int rowc = min(row0 + r, e.B - 1);
  const int j = 16 * ob + r;
  const bool jok = j < l.out_dim;
  const float* wrow = e.params + l.w_off + (int64_t)(jok ? j : 0) * K + 4 * q;
  const float* xrow = e.x + (int64_t)rowc * e.ldx + 4 * q;
  // wave w takes k16-steps w, w + 4, w + 8, ...; a register batch is L1_STEPS of them (stride 64 * L1_STEPS k's)
  f32x4 a[2][L1_STEPS], b[2][L1_STEPS];
  auto load = [&](int k0, f32x4 (&av)[L1_STEPS], f32x4 (&bv)[L1_STEPS]) {
#pragma unroll
    for (int s = 0; s < L1_STEPS; ++s) {
      const int k = k0 + 16 * NW * s;
      const int kc = k < K ? k : 0;
      av[s] = *reinterpret_cast<const f32x4*>(wrow + kc);
      bv[s] = *reinterpret_cast<const f32x4*>(xrow + kc);
      if (k >= K || !jok) av[s] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  f32x4 acc[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
  auto consume = [&](const f32x4 (&av)[L1_STEPS], const f32x4 (&bv)[L1_STEPS]) {
#pragma unroll
    for (int s = 0; s < L1_STEPS; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i & 1] = MFMA(av[s][i], bv[s][i], acc[i & 1]);
  };
  const int KB = 16 * NW * L1_STEPS;         // k's per batch over the whole workgroup
  load(16 * wave, a[0], b[0]);
  ENP_STAMP(1);
  for (int k0 = 16 * wave; k0 < K; k0 += 2 * KB) {
    const bool more1 = k0 + KB < K, more2 = k0 + 2 * KB < K;
    if (more1) load(k0 + KB, a[1], b[1]);
    consume(a[0], b[0]);
    if (more2) load(k0 + 2 * KB, a[0], b[0]);
    if (more1) consume(a[1], b[1]);
  }
  ENP_STAMP(2);
  const f32x4 c = acc[0] + acc[1];
  // C/D layout: lane (batch row r, q), reg i -> output 16*ob + 4q + i
#pragma unroll
  for (int i = 0; i < 4; ++i) part[wave][r][4 * q + i] = c[i];
  pv_lds_barrier();
  {
    const int rr = tid >> 4, jj = tid & 15, jo = 16 * ob + jj, row = row0 + rr;
    if (tid < 256 && jo < l.out_dim && row < e.B) {
      float v = (part[0][rr][jj] + part[1][rr][jj]) + (part[2][rr][jj] + part[3][rr][jj]);
      if (NW == 8) v += (part[4][rr][jj] + part[5][rr][jj]) + (part[6][rr][jj] + part[7][rr][jj]);
      v += l.b_off >= 0 ? e.params[l.b_off + jo] : 0.0f;
      const float y = pv_act_fwd2(v, l.act);
      // COHERENT (the merged launch): a device-scope store — written through to where another XCD's workgroup of the SAME
      // launch can read it (a plain store may sit dirty in this XCD's L2 until the kernel ends)
      if (COHERENT) __hip_atomic_store(e.eact[0] + (int64_t)row * l.out_dim + jo, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else e.eact[0][(int64_t)row * l.out_dim + jo] = y;
    }
  }
  ENP_STAMP(3);
  return true;
}

__global__ __launch_bounds__(64 * L1_WAVES) void pv_enc_l1_kernel(PvEncFwd e) {
  __shared__ float part[L1_WAVES][EN_ROWS][17];
  enc_l1_body<false, L1_WAVES>(e, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y, part);
}


#ifdef EN_TRACE
__device__ long long en_trace[64];
#define EN_STAMP(k) do { if (rbk == 0 && threadIdx.x == 0) en_trace[(k)] = (long long)__builtin_readcyclecounter(); } while (0)
extern "C" int pv_debug_read_enc_trace(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(en_trace), (n > 64 ? 64 : n) * sizeof(long long));
}
#else
#define EN_STAMP(k) do { } while (0)
#endif

__device__ unsigned en_late_total;          // consumers of the merged launch that computed their own first-layer tiles
extern "C" long long pv_debug_enc_late_count() {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(en_late_total), sizeof v) != hipSuccess) return -1;
  return (long long)v;
}

// rbk: the workgroup's row block.  MERGED: the first layer runs in the SAME launch (pv_enc_kernel): everything that does not
// depend on it is requested first, then the workgroup waits for its row block's tiles (flags == e.gen), then reads them.
template <bool MERGED>
__device__ __forceinline__ void enc_fwd_body(const PvEncFwd& e, int rbk, float (*act)[EN_ROWS][EN_LD], float* sm) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int row0 = rbk * EN_ROWS;
  const bool rok = row0 + r < e.B;
  int cur = 0;
  EN_STAMP(0);
  // ---- the first layer's output (pv_enc_l1_kernel): requested before anything else — loads return in order, and the
  // first barrier waits only for this one (merged launch: after the wait below) ----
  const int w0_ = e.enc[0].out_dim;
  const bool l0ok = tid < EN_ROWS * (w0_ / 4);                       // (w0 <= 128: one float4 per thread)
  f32x4 l0v = {0.0f, 0.0f, 0.0f, 0.0f};
  if (!MERGED && l0ok) {
    const int rr = tid / (w0_ / 4), c4 = tid % (w0_ / 4);
    l0v = *reinterpret_cast<const f32x4*>(e.eact[0] + (int64_t)min(row0 + rr, e.B - 1) * w0_ + 4 * c4);
  }
  // ---- everything that does not depend on computed data is requested NOW (one memory latency for the whole kernel
  // instead of one per phase): this wave's weight / bias operands of hidden layer 1 and of the head, eps ----
  const bool pf1 = e.n_enc > 1 && 16 * wave < e.enc[1].out_dim;      // the wave's first block of layer 1
  const bool pfh = 16 * wave < e.head.out_dim;                        // the wave's first block of the head
  f32x4 w1[8], wh[8], b1v = {0.0f, 0.0f, 0.0f, 0.0f}, bhv = {0.0f, 0.0f, 0.0f, 0.0f};
  if (pf1) {
    en_load_w(e.params + e.enc[1].w_off, e.enc[1].in_dim, e.enc[1].out_dim, wave, r, q, w1);
    b1v = en_load_bias(e.enc[1].b_off >= 0 ? e.params + e.enc[1].b_off : nullptr, e.enc[1].out_dim, wave, q);
  }
  if (pfh) {
    en_load_w(e.params + e.head.w_off, e.head.in_dim, e.head.out_dim, wave, r, q, wh);
    bhv = en_load_bias(e.head.b_off >= 0 ? e.params + e.head.b_off : nullptr, e.head.out_dim, wave, q);
  }
  // fc_latent's row of this thread's output column (its column is the same in every pass when H0 divides the block)
  const bool pfz = e.hz != nullptr && EN_THREADS % e.H0 == 0;
  float wzp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (pfz) {
    const int lat_all = (e.z_dim - (e.coord_dim == 1 ? (e.has_t ? 1 : 0) : e.coord_dim == 2 ? e.has_r + 2 * e.has_t + e.has_s : 0)) +
                        e.c_dim + (e.K > 0 ? e.K : 0);
    const float* wz = e.Wz + (int64_t)(tid % e.H0) * lat_all;
#pragma unroll
    for (int i = 0; i < 4; ++i) wzp[i] = i < lat_all ? wz[i] : 0.0f;
  }
  float eps_pf = 0.0f;                               // thread t < 16*z_dim handles (row t / z, column t % z)
  if (tid < EN_ROWS * e.z_dim && row0 + tid / e.z_dim < e.B)
    eps_pf = e.eps[(int64_t)(row0 + tid / e.z_dim) * e.z_dim + tid % e.z_dim];
  if (MERGED) {
    // wait for the first layer's tiles of this row block: lane j of wave 0 polls flag j, a bounded number of times
    // (e.spin_limit polls of ~0.5 us).  HIP promises nothing about dispatch order: a consumer whose producers have not
    // published by then (not dispatched yet behind a full chip, a competing stream, a second rank on the GPU) computes its
    // row block's tiles ITSELF — the same arithmetic in the same order, so the same bits, whoever gets there first — and the
    // launch completes under any placement and any order (round 4; rounds 1-3 spun for seconds and then returned a NaN loss).
    // en_late_total counts the fallbacks (observability, tests: pv_debug_enc_late_count).
    const int cb = (w0_ + 15) >> 4;
    int late = 0;
    if (tid < cb) {
      const unsigned* f = e.flags + (int64_t)rbk * cb + tid;
      late = 1;
      for (int spin = 0; spin < e.spin_limit; ++spin) {
        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e.gen) { late = 0; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    if (__syncthreads_or(late) != 0) {                                // (workgroup-uniform)
      // (the producers' own K split — L1_WAVES waves — for the same summation order; the other waves pair the barrier inside)
      float (*part4)[EN_ROWS][17] = reinterpret_cast<float (*)[EN_ROWS][17]>(&act[0][0][0]);
      for (int bx = 0; bx < cb; ++bx) {
        if (tid < 64 * L1_WAVES) enc_l1_body<true, L1_WAVES>(e, bx, rbk, cb, (e.B + EN_ROWS - 1) / EN_ROWS, part4);
        else pv_lds_barrier();
        pv_lds_barrier();                                              // `part4` is free again
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every thread's write-through stores acknowledged
      pv_lds_barrier();
      if (tid == 0) atomicAdd(&en_late_total, 1u);
    }
    // Hand-off protocol (cdna_hip_programming.md guideline 16, form R1): the producers' tile stores are agent-scope
    // (write-through, sc1) stores, every storing thread drains them (s_waitcnt vmcnt(0)) before the workgroup barrier that
    // precedes ONE lane's relaxed agent-scope flag store; here ONE relaxed poll per flag, a barrier, then agent-scope (sc1)
    // loads, which bypass this CU's L1 and this XCD's possibly stale L2 lines — the guide's "sc1 loads may replace the
    // acquire only when the producer stored sc1".
    if (l0ok) {                                                       // device-scope loads: past this XCD's (possibly stale) L2 lines
      const int rr = tid / (w0_ / 4), c4 = tid % (w0_ / 4);
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(e.eact[0] + (int64_t)min(row0 + rr, e.B - 1) * w0_ + 4 * c4);
      const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      l0v[0] = __uint_as_float((unsigned)lo); l0v[1] = __uint_as_float((unsigned)(lo >> 32));
      l0v[2] = __uint_as_float((unsigned)hi); l0v[3] = __uint_as_float((unsigned)(hi >> 32));
    }
  }
  // ---- the first layer's output (pv_enc_l1_kernel) into LDS; rows past the batch repeat the last one ----
  {
    if (l0ok) *reinterpret_cast<f32x4*>(&act[0][tid / (w0_ / 4)][4 * (tid % (w0_ / 4))]) = l0v;
    pv_lds_barrier();
  }
  EN_STAMP(1);
  // ---- hidden layers 1.. ----
  for (int li = 1; li < e.n_enc; ++li) {
    const pv_layer l = e.enc[li];
    const float* W = e.params + l.w_off;
    const float* bias = l.b_off >= 0 ? e.params + l.b_off : nullptr;
    for (int ob = wave; 16 * ob < l.out_dim; ob += EN_THREADS / 64) {
      const bool pre = li == 1 && ob == wave;          // operands already in registers
      const f32x4 acc = pre ? en_mma(w1, l.in_dim, &act[cur][0][0], r, q)
                            : en_layer_block(W, l.in_dim, l.out_dim, ob, &act[cur][0][0], r, q);
      const f32x4 bv = pre ? b1v : en_load_bias(bias, l.out_dim, ob, q);
      // C/D layout: lane (col r, q), reg i -> output j = 16*ob + 4*q + i of row r
      f32x4 y;
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = pv_act_fwd2(acc[i] + bv[i], l.act);
      *reinterpret_cast<f32x4*>(&act[cur ^ 1][r][16 * ob + 4 * q]) = y;
      if (rok) *reinterpret_cast<f32x4*>(e.eact[li] + (int64_t)(row0 + r) * l.out_dim + 16 * ob + 4 * q) = y;
    }
    pv_lds_barrier();
    cur ^= 1;
  }
  EN_STAMP(2);
  // ---- head: [mu | softplus input] (fc11 | fc12) ----
  {
    const pv_layer l = e.head;
    const float* W = e.params + l.w_off;
    const float* bias = l.b_off >= 0 ? e.params + l.b_off : nullptr;
    for (int ob = wave; 16 * ob < l.out_dim; ob += EN_THREADS / 64) {
      const bool pre = ob == wave;
      const f32x4 acc = pre ? en_mma(wh, l.in_dim, &act[cur][0][0], r, q)
                            : en_layer_block(W, l.in_dim, l.out_dim, ob, &act[cur][0][0], r, q);
      const f32x4 bv = pre ? bhv : en_load_bias(bias, l.out_dim, ob, q);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * ob + 4 * q + i;
        if (j < l.out_dim) {
          const float v = acc[i] + bv[i];
          act[cur ^ 1][r][j] = v;
          if (rok) e.head_out[(int64_t)(row0 + r) * l.out_dim + j] = v;
        }
      }
    }
    pv_lds_barrier();
    cur ^= 1;
  }
  EN_STAMP(3);
  // ---- z = mu + softplus(s) * eps; sampled-KL terms ----
  const int zd = e.z_dim;
  float lp = 0.0f, lq = 0.0f;
  for (int t = tid; t < EN_ROWS * zd; t += EN_THREADS) {
    const int rr = t / zd, i = t % zd, row = row0 + rr;
    if (row < e.B) {
      const float mu = act[cur][rr][i], sp = act[cur][rr][zd + i];
      const float sig = pv_softplus(sp);
      const float ep = t == tid ? eps_pf : e.eps[(int64_t)row * zd + i];
      const float z = mu + sig * ep;
      e.z[(int64_t)row * zd + i] = z;
      e.z_scale[(int64_t)row * zd + i] = sig;
      if (e.z_loc_out) e.z_loc_out[(int64_t)row * zd + i] = mu;
      if (e.z_scale_out) e.z_scale_out[(int64_t)row * zd + i] = sig;
      const float d = z - mu;
      const float wb = e.w ? e.w[row] : 1.0f;
      lq += wb * (-(d * d) / (2.0f * (sig * sig)) - logf(sig) - LOG_SQRT_2PI);      // torch Normal.log_prob
      lp += wb * (-(z * z) / 2.0f - LOG_SQRT_2PI);
      act[cur ^ 1][rr][i] = z;                     // keep z for the split below
    }
  }
  // ---- jiVAE: alpha = softmax(class logits) (nets/fc.py:106); discrete KL terms; decoder row weights ----
  const int K = e.K;
  float lpd = 0.0f, lqd = 0.0f;
  if (K > 0 && tid < EN_ROWS && row0 + tid < e.B) {
    const int row = row0 + tid;
    const float* lg = &act[cur][tid][2 * zd];
    float mx = lg[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, lg[k]);
    float se = 0.0f;
    for (int k = 0; k < K; ++k) se += expf(lg[k] - mx);
    const float lse = logf(se);
    for (int k = 0; k < K; ++k) {
      const float la = lg[k] - mx - lse, a = expf(la);      // log_softmax, softmax
      e.alpha[(int64_t)row * K + k] = a;
      e.sw[(int64_t)k * e.B + row] = a;
      lqd += a * la;                                         // sum_k alpha log alpha
    }
    lpd = -logf((float)K);                                   // sum_k alpha log(1/K)
  }
  EN_STAMP(4);
  lp = en_block_sum(lp, sm);
  lq = en_block_sum(lq, sm);
  if (K > 0) {
    lpd = en_block_sum(lpd, sm);
    lqd = en_block_sum(lqd, sm);
  }
  if (tid == 0) {
    e.kl_part[2 * rbk] = e.beta * lp + e.beta_disc * lpd;
    e.kl_part[2 * rbk + 1] = e.beta * lq + e.beta_disc * lqd;
  }
  pv_lds_barrier();
  EN_STAMP(5);
  cur ^= 1;                                         // act[cur][rr][0..zd) = z
  // ---- _split_latent: transform parameters + decoder latent input (base.py:97-119, ivae.py:187-195) ----
  int coord = 0;
  if (e.coord_dim == 1) coord = e.has_t ? 1 : 0;
  else if (e.coord_dim == 2) coord = e.has_r + 2 * e.has_t + e.has_s;
  const int L = zd - coord, lat_in = L + e.c_dim + (K > 0 ? K : 0);
  const int KK = K > 0 ? K : 1;                     // decoder samples per input: (k, b) at row k*B + b
  if (tid < EN_ROWS && row0 + tid < e.B) {
    const int row = row0 + tid;
    const float* zb = &act[cur][tid][0];
    int idx = 0;
    float c = 1.0f, s = 0.0f, sc = 1.0f, tx = 0.0f, ty = 0.0f;
    if (e.coord_dim == 1) {
      if (e.has_t) { tx = zb[0] * e.tp0; idx = 1; }
    } else if (e.coord_dim == 2) {
      if (e.has_r) { const float phi = zb[idx++]; c = cosf(phi); s = sinf(phi); }
      if (e.has_t) { tx = zb[idx] * e.tp0; ty = zb[idx + 1] * e.tp1; idx += 2; }
      if (e.has_s) { sc = 1.0f + e.sc_prior * zb[idx++]; }
    }
    for (int k = 0; k < KK; ++k) {
      const int64_t srow = (int64_t)k * e.B + row;
      if (e.tp) {
        float* t = e.tp + srow * 8;
        t[0] = c; t[1] = s; t[2] = sc; t[3] = tx; t[4] = ty;
      }
      if (e.zy) {
        float* o = e.zy + srow * lat_in;
        for (int i = 0; i < L; ++i) o[i] = zb[coord + i];
        for (int i = 0; i < e.c_dim; ++i) o[L + i] = e.y[(int64_t)row * e.c_dim + i];
        for (int i = 0; i < K; ++i) o[L + e.c_dim + i] = i == k ? 1.0f : 0.0f;        // one-hot class (jivae.py:189)
      }
    }
  }
  EN_STAMP(6);
  // ---- hz = fc_latent(cat(z_content, y)) (no bias; fc.py:217,230) ----
  if (e.hz) {
    for (int t = tid; t < EN_ROWS * e.H0; t += EN_THREADS) {
      const int rr = t / e.H0, j = t % e.H0, row = row0 + rr;
      if (row >= e.B) continue;
      const float* wz = e.Wz + (int64_t)j * lat_in;
      float v = 0.0f;
      for (int i = 0; i < L; ++i) v += act[cur][rr][coord + i] * ((pfz && i < 4) ? wzp[i] : wz[i]);
      for (int i = 0; i < e.c_dim; ++i) v += e.y[(int64_t)row * e.c_dim + i] * wz[L + i];
      const float hsc = e.hz_scale != 0.0f ? e.hz_scale : 1.0f;     // (the 8-wave decoder kernel takes C * hz)
      if (K > 0) {
        for (int k = 0; k < K; ++k) e.hz[((int64_t)k * e.B + row) * e.H0 + j] = (v + wz[L + e.c_dim + k]) * hsc;
      } else {
        e.hz[(int64_t)row * e.H0 + j] = v * hsc;
      }
    }
  }
}

__global__ __launch_bounds__(EN_THREADS) void pv_enc_fwd_kernel(PvEncFwd e) {
  __shared__ __attribute__((aligned(16))) float act[2][EN_ROWS][EN_LD];
  __shared__ float sm[8];
  enc_fwd_body<false>(e, (int)blockIdx.x, act, sm);
}

// Both in ONE launch: workgroups [0, n1) are the first layer's tiles (and its guests), in row-block-major order; workgroups
// n1 + k carry row block k through the rest.  The second kind starts at once — its weights, biases and noise are requested
// while the first layer is still running, which hides the ~5 us every launch spends before its first operand is usable —
// and waits for its 8 producers through per-tile flags: a producer publishes e.gen (a value no earlier call used) after a
// device-scope release of its tile, the consumer polls with acquire loads.  Workgroups are dispatched in index order, so
// every producer is running or done before a consumer occupies a slot: no deadlock at any batch size.
__global__ __launch_bounds__(EN_THREADS) void pv_enc_kernel(PvEncFwd e, int nx, int ny) {
  __shared__ __attribute__((aligned(16))) float act[2][EN_ROWS][EN_LD];
  __shared__ float sm[8];
  const int n1 = nx * ny, id = (int)blockIdx.x;
  if (id < n1) {
    if (threadIdx.x >= 64 * L1_WAVES) return;                        // (four waves split K; all eight measured no faster)
    float (*part)[EN_ROWS][17] = reinterpret_cast<float (*)[EN_ROWS][17]>(&act[0][0][0]);
    const int bx = id % nx, by = id / nx;
    if (enc_l1_body<true, L1_WAVES>(e, bx, by, nx, ny, part)) {
      // every thread's (write-through) store of the tile has been acknowledged, then the flag: no cache-wide write-back
      // (a release fence here cost more than the launch it saves: 24 vs 21 us for the pair)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pv_lds_barrier();
      if (threadIdx.x == 0) __hip_atomic_store(e.flags + (int64_t)by * nx + bx, e.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  enc_fwd_body<true>(e, id - n1, act, sm);
}

bool pv_enc_compact_supported(const pv_ivae_plan* p) {
  if (p->n_enc < 1) return false;
  int in = p->n_pix + p->c_dim;
  if (in % 16 != 0) return false;
  for (int i = 0; i < p->n_enc; ++i) {
    const pv_layer& l = p->enc[i];
    if (l.in_dim != in || l.out_dim > 128 || l.out_dim % 16 != 0 || l.act == PV_ACT_GELU) return false;
    if (l.w_off % 4 != 0) return false;
    in = l.out_dim;
  }
  if (p->head.in_dim != in || p->head.out_dim > 128 || p->head.w_off % 4 != 0) return false;
  if (p->z_dim > 64 || p->discrete_dim > 64) return false;
  return true;
}

// a flag value no earlier call of this process used (and, with a random start, none a previous process is likely to have left
// in recycled device memory): producers publish it, consumers wait for it — no zero-initialised state anywhere
static unsigned enc_next_gen() {
  static std::atomic<unsigned> g{[] {
    std::random_device rd;
    return (unsigned)rd() | 1u;
  }()};
  unsigned v = g.fetch_add(1u);
  return v;
}

// Polls before a consumer computes its tiles itself (PvEncFwd::spin_limit, from the plan): 256 x ~0.5 us is 10-20x what the
// producers need on an idle GPU; the fallback is exact, so a short limit only costs redundant work.  PV_PLAN_ENC_NO_WAIT makes
// it 0: every consumer takes the fallback (the parity test of that path).  No process-wide state (ABI v15).

int pv_enc_fwd(const PvEncFwd& e, hipStream_t s) {
  const int rb = (e.B + EN_ROWS - 1) / EN_ROWS;
  const int cb = (e.enc[0].out_dim + 15) / 16;
  int extra = 0;                                   // rows of guest workgroups for the weight-image preparation
  if (e.prep.img) {
    const int64_t work = e.prep.nzero4 > 128 * 32 ? e.prep.nzero4 : 128 * 32;
    extra = (int)((work + (int64_t)cb * 64 * L1_WAVES - 1) / ((int64_t)cb * 64 * L1_WAVES));
    if (extra > 16) extra = 16;
  }
  static const int two = pv_exp_int("PV_ENC_TWO", 0) ? 1 : 0;     // (A/B in the experiments build; per plan: PV_PLAN_ENC_TWO_LAUNCH)
  if (e.flags && !two && !pv_stream_capturing(s)) {    // (a captured launch would replay its generation value: two launches then)
    PvEncFwd m = e;
    m.gen = enc_next_gen();
    if (m.spin_limit < 0) m.spin_limit = 256;
    hipLaunchKernelGGL(pv_enc_kernel, dim3((unsigned)(cb * (rb + extra) + rb)), dim3(EN_THREADS), 0, s, m, cb, rb + extra);
    PV_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(pv_enc_l1_kernel, dim3(cb, rb + extra), dim3(64 * L1_WAVES), 0, s, e);
  PV_LAUNCH_CHECK();
  hipLaunchKernelGGL(pv_enc_fwd_kernel, dim3(rb), dim3(EN_THREADS), 0, s, e);
  PV_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// encoder dgrad chain: edp[last] = (dhead Whead) * act'(eact[last]); edp[i-1] = (edp[i] W_i) * act'(eact[i-1])
__global__ __launch_bounds__(EN_THREADS) void pv_enc_dgrad_kernel(PvEncDgrad e) {
  __shared__ __attribute__((aligned(16))) float buf[2][EN_ROWS][EN_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
  if ((int)blockIdx.x == (e.B + EN_ROWS - 1) / EN_ROWS) {        // guest workgroup: the step's loss scalars
    pv_finish_scalars_block(e.fin_llb, e.B, e.fin_scalars, e.fin_kl_part, e.fin_n_part, e.fin_beta, &buf[0][0][0]);
    return;
  }
  const int row0 = blockIdx.x * EN_ROWS;
  const int ne = e.n_enc;
  int cur = 0;
  // ---- requested up front (one memory latency for the kernel): the wave's strided weight operand and activation
  // tile of the first MFMA layer, and the head-weight column of the thread's output in the first phase ----
  float a0[8][4];
  f32x4 h0 = {0.0f, 0.0f, 0.0f, 0.0f};
  const bool pf = ne > 1 && 16 * wave < e.enc[ne - 1].in_dim;
  if (pf) {
    const pv_layer l = e.enc[ne - 1];
    const float* wcol = e.params + l.w_off + 16 * wave + r;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * t + 4 * q + i;
        const float v = wcol[(int64_t)(j < l.out_dim ? j : 0) * l.in_dim];
        a0[t][i] = j < l.out_dim ? v : 0.0f;
      }
    if (row0 + r < e.B)
      h0 = *reinterpret_cast<const f32x4*>(e.eact[ne - 2] + (int64_t)(row0 + r) * e.enc[ne - 2].out_dim + 16 * wave + 4 * q);
  }
  {
    // from the head: K = 2*z_dim is tiny -> plain FMAs
    const pv_layer hd = e.head;
    const pv_layer ll = e.enc[ne - 1];
    const float* Wh = e.params + hd.w_off;
    const bool fastk = EN_THREADS % ll.out_dim == 0;        // then a thread keeps its column k in every pass
    float whp[16];
#pragma unroll
    for (int o = 0; o < 16; ++o)
      whp[o] = (fastk && o < hd.out_dim) ? Wh[(int64_t)o * hd.in_dim + tid % ll.out_dim] : 0.0f;
    for (int t = tid; t < EN_ROWS * ll.out_dim; t += EN_THREADS) {
      const int rr = t / ll.out_dim, k = t % ll.out_dim, row = row0 + rr;
      float v = 0.0f;
      if (row < e.B) {
        const float* dh = e.dhead + (int64_t)row * hd.out_dim;
        const float hv = e.eact[ne - 1][(int64_t)row * ll.out_dim + k];
        float dv[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) dv[o] = o < hd.out_dim ? dh[o] : 0.0f;
#pragma unroll
        for (int o = 0; o < 16; ++o) v += dv[o] * (fastk ? whp[o] : (o < hd.out_dim ? Wh[(int64_t)o * hd.in_dim + k] : 0.0f));
        for (int o = 16; o < hd.out_dim; ++o) v += dh[o] * Wh[(int64_t)o * hd.in_dim + k];
        v *= pv_act_grad2(hv, 0.0f, ll.act);
        e.edp[ne - 1][(int64_t)row * ll.out_dim + k] = v;
      }
      buf[cur][rr][k] = v;
    }
    pv_lds_barrier();
  }
  for (int li = ne - 1; li > 0; --li) {
    const pv_layer l = e.enc[li];          // edp[li] (16 x l.out_dim) in buf[cur]; produce edp[li-1] (16 x l.in_dim)
    const pv_layer lp = e.enc[li - 1];
    const float* W = e.params + l.w_off;
    for (int kb = wave; 16 * kb < l.in_dim; kb += EN_THREADS / 64) {
      const float* wcol = W + 16 * kb + r;                 // A lane (k' = r, q): W[j][16*kb + k']
      const bool pre = pf && li == ne - 1 && kb == wave;   // operands requested at the top of the kernel
      float a[8][4];                                       // out_dim <= 128: the whole operand in flight at once
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = 16 * t + 4 * q + i;
          if (pre) { a[t][i] = a0[t][i]; continue; }
          const float v = wcol[(int64_t)(j < l.out_dim ? j : 0) * l.in_dim];
          a[t][i] = j < l.out_dim ? v : 0.0f;
        }
      f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (16 * t < l.out_dim) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(&buf[cur][r][16 * t + 4 * q]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i & 1) acc1 = MFMA(a[t][i], b[i], acc1); else acc0 = MFMA(a[t][i], b[i], acc0);
          }
        }
      }
      const f32x4 acc = acc0 + acc1;
      const int row = row0 + r;
      f32x4 y = {0.0f, 0.0f, 0.0f, 0.0f};
      if (row < e.B) {
        const f32x4 h = pre ? h0 : *reinterpret_cast<const f32x4*>(e.eact[li - 1] + (int64_t)row * lp.out_dim + 16 * kb + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = acc[i] * pv_act_grad2(h[i], 0.0f, lp.act);
        *reinterpret_cast<f32x4*>(e.edp[li - 1] + (int64_t)row * lp.out_dim + 16 * kb + 4 * q) = y;
      }
      *reinterpret_cast<f32x4*>(&buf[cur ^ 1][r][16 * kb + 4 * q]) = y;
    }
    pv_lds_barrier();
    cur ^= 1;
  }
}

int pv_enc_dgrad(const PvEncDgrad& e, hipStream_t s) {
  hipLaunchKernelGGL(pv_enc_dgrad_kernel, dim3((e.B + EN_ROWS - 1) / EN_ROWS + (e.fin_scalars ? 1 : 0)),
                     dim3(EN_THREADS), 0, s, e);
  PV_LAUNCH_CHECK();
  return 0;
}
