// pv_dec1d.hip — a 1-D convolutional decoder stack (nets/conv.py:190-262 Upsampler over Conv1d: kernel-3 convolutions with
// an activation, UpsampleBlocks = nearest 2x interpolation + kernel-1 convolution, the kernel-1 output layer; VED's
// im2spec decoder, models/ved.py:96-106) as ONE forward launch and ONE input-gradient launch.
//
// Such a stack is a chain of ten tiny GEMMs per sample (a layer is 16-128 positions x 32-128 channels: 0.4 GFLOP at batch
// 256), each 5-12 us as a launch of its own — launch, cold start and drain, not arithmetic.  Here a workgroup (8 waves)
// carries ONE sample through the whole chain: every activation of the sample (a few thousand floats per layer, zero halo rows
// for the kernel-3 padding) stays in LDS until the end, a wave owns 16 output channels x 16 positions at a time, the weights
// stream from L2 straight into the A operand of v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: an exact fp32 FMA chain,
// like the kernel-1 family they replace) from a layout written once per step by pv_conv_wprep_table in MFMA fragment order
// (kind 8; the input-gradient form with taps flipped and channel roles swapped).  Every layer's output / gradient goes to
// global memory in one burst at the end, in the layouts the step's other kernels use: the weight gradients
// (pv_conv_k1.hip's recorded batch) read them.  latent_to_features (the Linear in front of the stack) and the observation
// likelihood behind it can ride in the same launches.
//
// Conventions as in pv_convstack.h: op i maps a[i] -> a[i+1]; an UpsampleBlock arrives as CONV k1 (no activation) followed
// by UPSAMPLE2 (the convolution first: both linear, half the positions) and is one step here (rows stored twice forward,
// g[2p] + g[2p+1] backward); the gradient written for op i is dL/d(a[i]) with the producing convolution's activation
// derivative applied (= dL/d(pre-activation) of op i-1: what its weight gradient wants).
#include "pv_common.h"
#include "pv_conv.h"
#include "pv_dec1d.h"
#include "pv_side.h"
#include <stdlib.h>

#define D1_THREADS 512
#define D1_WAVES (D1_THREADS / 64)
#define D1_LDS_MAX (150 * 1024)                      // bytes of LDS a sample's activations may take (all of them stay resident)
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

struct D1Op {
  const float* w;        // tiled (kind 8): MFMA fragment order; [tap][N][K] when K is not a multiple of 16 (plain loops)
  const float* e;        // epilogue operand: forward the bias (null: none); backward the step's INPUT activation (B, L, N) when
                         // its producer's activation derivative applies (null: none)
  float* out;            // forward: the step's output (B, Lout, N); backward: dL/d(input) (B, L, N)
  int K, N;              // contraction width, output width
  int L;                 // positions the convolution runs over
  int taps;              // 3 (zero padding 1) or 1
  int act;               // forward: this convolution's activation; backward: the producer's (through e)
  int up;                // forward: every output row stored twice (Lout = 2 L); backward: the incoming gradient has 2 L rows
  int li, lo, lt;        // LDS float offsets: input rows, output rows, (backward, up) the row-pair sums
  int G;                 // chunks per group of the contraction loop (4: short contractions; 8: the operand requests stay a group of 8 ahead)
};
struct D1Args {
  D1Op op[PV_D1_MAXOPS];
  int n, B;
  const float* in;       // forward: a[0] (B, L0, C0); backward: dL/d(a[n]) (B, Ln, Cn)
  int L0, C0;
  // latent_to_features in the same launch (zd > 0).  Forward: a[0] = bias + z wt is computed here and written to a0_out;
  // backward: dz[b][k] = <dL/d(a[0])[b], wt[k]> from the last step's result
  const float* z; const float* l2f_wt; const float* l2f_b; float* a0_out; float* dz;
  int zd;
  // the head in the same launches (h_head != null): forward z from [mu | softplus input] and eps, backward dhead
  const float* h_head; const float* h_eps; float* h_z; float* h_zs; float* h_zlo; float* h_zso; float* h_kl; float* h_dhead;
  int h_ldh; float h_beta;
  const float* h_part; const float* h_bias; float* h_hout; int h_nseg;     // forward: head from its partial sums (pv_convhead_fwd_partials)
  // forward: the observation likelihood of the last step's result (y != null)
  const float* y; float* loc; float* dlda; float* llb;
  int lik, sigmoid_out; float sig;
};

__device__ __forceinline__ int d1_pitch(int C) { return C + 4; }
static int d1_rows_floats(int L, int C) { return (L + 2) * (C + 4); }

// rows 0 and L + 1 of a buffer (the zero padding of a kernel-3 convolution)
__device__ __forceinline__ void d1_zero_halo(float* buf, int L, int C) {
  const int P = d1_pitch(C);
  for (int e = threadIdx.x; e < 2 * C; e += D1_THREADS) buf[(e < C ? 0 : (L + 1)) * P + (e < C ? e : e - C)] = 0.0f;
}

// A wave's tile: 16 output channels x 16 positions, contraction in chunks of 16 (chunk c = tap * (K / 16) + t: columns
// 16 t + 4 q .. + 3 of tap `tap`), the chunk count padded to groups of 4 or 8 with ZERO weights (pv_conv_wprep_table kind
// 8 writes the padding) — a rolled loop over groups with a fixed body and no per-chunk conditions: four 1 KB-contiguous
// operand loads off one address register (the next group's requested before the current group's MFMAs; the FIRST group
// of a step's tile requested before the previous step's closing barrier), four LDS reads, sixteen MFMAs.  The instruction
// count is what this kernel is bound by (a fully unrolled form with per-chunk selects ran 8x slower than its MFMAs), so
// everything that is uniform — chunk counters, LDS offsets, the padding's clamp — lives in scalar registers.
// Nothing is STORED to global memory while the chain runs: every activation of the sample stays in LDS and goes out in one
// burst at the end.
#define D1_GMAX 8
__host__ __device__ __forceinline__ int d1_group(int taps, int K) { return taps * (K >> 4) >= 12 ? 8 : 4; }
__host__ __device__ __forceinline__ int d1_nckp_of(int taps, int K) { const int G = d1_group(taps, K); return (taps * (K >> 4) + G - 1) / G * G; }
__device__ __forceinline__ int d1_nckp(const D1Op& o) { return d1_nckp_of(o.taps, o.K); }
// tile -> (output-channel block, position block); tile is wave-uniform and small: a scalar loop instead of a division
__device__ __forceinline__ void d1_split(int tile, int nco, int& ob, int& lb) {
  ob = tile; lb = 0;
  while (ob >= nco) { ob -= nco; ++lb; }
}
// this lane's operand pointer of the tile's channel block ob, group 0
__device__ __forceinline__ const float* d1_wl(const D1Op& o, int ob, int lane) {
  return o.w + ((int64_t)ob * d1_nckp(o) * 64 + lane) * 4;
}
template <int G>
__device__ __forceinline__ void d1_ldg(const float* wl, f32x4 (&a)[D1_GMAX]) {
#pragma unroll
  for (int j = 0; j < G; ++j) a[j] = *reinterpret_cast<const f32x4*>(wl + j * 256);
}
__device__ __forceinline__ void d1_ldg_op(const D1Op& o, const float* wl, f32x4 (&a)[D1_GMAX]) {
  if (o.G == 8) d1_ldg<8>(wl, a); else d1_ldg<4>(wl, a);
}

// the tile's contraction; cu: group 0 of its operands (already requested).  acc[i] <-> output channel 16 ob + 4 q + i at
// position 16 lb + r; B from LDS rows l + tap (halo offset included)
template <int G>
__device__ __forceinline__ f32x4 d1_mma(const D1Op& o, const float* __restrict__ in, int lb, int r, int q, const float* wl,
                                        f32x4 (&cu)[D1_GMAX]) {
  const int kc = o.K >> 4, nck = o.taps * kc, ng = d1_nckp(o) / G;
  const int P = d1_pitch(o.K);
  const float* irow = in + (16 * lb + r + (o.taps == 3 ? 0 : 1)) * P + 4 * q;      // (tap 0 of three reads position l - 1 = row l)
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  int c = 0, t = 0, off = 0;                           // (uniform) chunk, its column block, its LDS offset in floats
  f32x4 bn[4];                                         // the NEXT four chunks' B operands: requested four chunks ahead of their MFMAs
  auto read4 = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bn[j] = *reinterpret_cast<const f32x4*>(irow + off);
      if (c + 1 < nck) {                               // (the zero padding re-reads the last real chunk's rows: finite values)
        ++c; off += 16;
        if (++t == kc) { t = 0; off += P - 16 * kc; }
      }
    }
  };
  read4();
#pragma nounroll
  for (int g = 0; g < ng; ++g) {
    f32x4 nx[D1_GMAX];
    wl += G * 256;
    if (g + 1 < ng) d1_ldg<G>(wl, nx);
#pragma unroll
    for (int h = 0; h < G; h += 4) {
      f32x4 bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = bn[j];
      if (g + 1 < ng || h + 4 < G) read4();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc0 = MFMA4(cu[h + j][0], bv[j][0], acc0); acc1 = MFMA4(cu[h + j][1], bv[j][1], acc1);
        acc0 = MFMA4(cu[h + j][2], bv[j][2], acc0); acc1 = MFMA4(cu[h + j][3], bv[j][3], acc1);
      }
    }
    if (g + 1 < ng) {
#pragma unroll
      for (int j = 0; j < G; ++j) cu[j] = nx[j];
    }
  }
  return acc0 + acc1;
}

// the epilogue operand of a tile: forward the bias of the lane's 4 output channels, backward the input activation at the
// lane's position (zeros when there is none or the lane's channels are not a whole float4)
template <bool BWD>
__device__ __forceinline__ f32x4 d1_ev(const D1Op& o, int b, int ob, int lb, int r, int q) {
  const int co = 16 * ob + 4 * q, l = 16 * lb + r;
  f32x4 ev = {0.0f, 0.0f, 0.0f, 0.0f};
  if (o.e && co + 3 < o.N && (o.N & 3) == 0)
    ev = *reinterpret_cast<const f32x4*>(BWD ? o.e + ((int64_t)b * o.L + l) * o.N + co : o.e + co);
  return ev;
}

// element index -> row: the divisors (channel counts, a quarter of them) are powers of two in every stock decoder — a shift behind a
// uniform branch instead of a 30-instruction division in front of every element of every layer's staging / epilogue loops
__device__ __forceinline__ int d1_div(int e, int n) { return (n & (n - 1)) == 0 ? e >> (31 - __clz(n)) : e / n; }

// contraction narrower than 16 (the output layer's input gradient: K = output channels of the whole stack): plain FMAs
template <bool BWD>
__device__ __forceinline__ void d1_small_k(const D1Op& o, const float* __restrict__ in, float* __restrict__ outb, const float* y) {
  const int Pi = d1_pitch(o.K), Po = d1_pitch(o.N);
  for (int e = threadIdx.x; e < o.L * o.N; e += D1_THREADS) {
    const int l = d1_div(e, o.N), n = e - l * o.N;
    float v = 0.0f;
    for (int tap = 0; tap < o.taps; ++tap)
      for (int k = 0; k < o.K; ++k)
        v += o.w[((int64_t)tap * o.N + n) * o.K + k] * in[(l + tap + (o.taps == 3 ? 0 : 1)) * Pi + k];
    if (BWD) { if (y) v *= pv_act_grad2(y[(int64_t)l * o.N + n], 0.0f, o.act); }
    else v = pv_act_fwd2(v + (o.e ? o.e[n] : 0.0f), o.act);
    outb[(l + 1) * Po + n] = v;
  }
}

__device__ __forceinline__ int d1_tiles(const D1Op& o) { return ((o.N + 15) >> 4) * (o.L >> 4); }
__device__ __forceinline__ bool d1_mfma_step(const D1Op& o) { return (o.K & 15) == 0; }

#ifdef D1_TRACE
__device__ long long d1_trace[2][64];
#define D1_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) d1_trace[BWD ? 1 : 0][(k)] = (long long)__builtin_readcyclecounter(); } while (0)
extern "C" int pv_debug_read_d1_trace(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(d1_trace), sizeof(long long) * 128); }
#else
#define D1_STAMP(k) do { } while (0)
#endif
// finer stamps inside two chosen steps (1: a 128 -> 128 kernel-3 step, 8: a 32 -> 32 kernel-1 step): slots 32 + 8 s + j
#define D1_FINE(j) do { if (i == 1) D1_STAMP(32 + (j)); else if (i == 8) D1_STAMP(40 + (j)); } while (0)

// One step of a sample.  BWD = false: out = act(conv(in) + bias), rows stored twice for a fused upsample; BWD = true:
// out = conv^T(g) * act'(y) (the incoming gradient summed over row pairs first for a fused upsample).  pre: group 0 of the
// operands of this wave's first tile of THIS step (requested during the previous one); refilled here for the next step.
template <bool BWD>
__device__ __forceinline__ void d1_step(const D1Args& A, int i, int b, float* lds, int wave, int lane, f32x4 (&pre)[D1_GMAX], f32x4& pev) {
  const D1Op& o = A.op[i];
  const int tid = threadIdx.x, r = lane & 15, q = lane >> 4;
  const bool last = i + 1 == A.n;
  if (BWD && o.up) {                                   // the nearest upsample's backward: g[p] = g[2p] + g[2p + 1]
    const int P = d1_pitch(o.K);
    const float* gi = lds + o.li;
    float* go = lds + o.lt;
    for (int e = tid; e < o.L * o.K; e += D1_THREADS) {
      const int l = d1_div(e, o.K), c = e - l * o.K;
      go[(l + 1) * P + c] = gi[(2 * l + 1) * P + c] + gi[(2 * l + 2) * P + c];
    }
    d1_zero_halo(go, o.L, o.K);
    pv_lds_barrier();
  }
  const float* in = lds + ((BWD && o.up) ? o.lt : o.li);
  float* ob_ = lds + o.lo;
  const int up = BWD ? 0 : o.up;
  const int Lout = up ? 2 * o.L : o.L, Po = d1_pitch(o.N);
  D1_FINE(0);
  d1_zero_halo(ob_, Lout, o.N);
  D1_FINE(1);
  if (!d1_mfma_step(o)) {
    d1_small_k<BWD>(o, in, ob_, BWD && o.e ? o.e + (int64_t)b * o.L * o.N : nullptr);
  } else {
    const int nco = (o.N + 15) >> 4, ntile = d1_tiles(o);
    for (int tile = wave; tile < ntile; tile += D1_WAVES) {
      int ob, lb;
      d1_split(tile, nco, ob, lb);
      const int co = 16 * ob + 4 * q, l = 16 * lb + r;
      const bool vec = co + 3 < o.N && (o.N & 3) == 0;
      const float* wl = d1_wl(o, ob, lane);
      if (tile != wave) { d1_ldg_op(o, wl, pre); pev = d1_ev<BWD>(o, b, ob, lb, r, q); }   // (a further tile of the step: on demand)
      const f32x4 ev = pev;
      D1_FINE(2);
      const f32x4 acc = o.G == 8 ? d1_mma<8>(o, in, lb, r, q, wl, pre) : d1_mma<4>(o, in, lb, r, q, wl, pre);
#ifdef D1_TRACE
      asm volatile("s_nop 0" :: "v"(acc[0]));
#endif
      D1_FINE(3);
      f32x4 v;
      if (vec && pv_act_is_lin(o.act)) {               // the common case branch-free: y > 0 ? y : slope y, derivative 1 or slope
        const float slope = pv_act_slope(o.act);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (BWD) v[k] = acc[k] * ((o.e == nullptr || ev[k] > 0.0f) ? 1.0f : slope);
          else { const float y = acc[k] + ev[k]; v[k] = y > 0.0f ? y : y * slope; }
        }
      } else
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (BWD) {
          float d = 1.0f;
          if (o.e) d = pv_act_grad2(vec ? ev[k] : (co + k < o.N ? o.e[((int64_t)b * o.L + l) * o.N + co + k] : 0.0f), 0.0f, o.act);
          v[k] = acc[k] * d;
        } else {
          const float bias = vec ? ev[k] : (o.e && co + k < o.N ? o.e[co + k] : 0.0f);
          v[k] = pv_act_fwd2(acc[k] + bias, o.act);
        }
      }
      if (vec) {
        if (up) {
          *reinterpret_cast<f32x4*>(&ob_[(2 * l + 1) * Po + co]) = v;
          *reinterpret_cast<f32x4*>(&ob_[(2 * l + 2) * Po + co]) = v;
        } else {
          *reinterpret_cast<f32x4*>(&ob_[(l + 1) * Po + co]) = v;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (co + k >= o.N) continue;
          for (int d = 0; d < (up ? 2 : 1); ++d) ob_[((up ? 2 * l + d : l) + 1) * Po + co + k] = v[k];
        }
      }
    }
  }
  D1_FINE(4);
  if (!last && d1_mfma_step(A.op[i + 1]) && wave < d1_tiles(A.op[i + 1])) {      // the next step's first requests, before the barrier
    const D1Op& nx = A.op[i + 1];
    int ob, lb;
    d1_split(wave, (nx.N + 15) >> 4, ob, lb);
    d1_ldg_op(nx, d1_wl(nx, ob, lane), pre);
    pev = d1_ev<BWD>(nx, b, ob, lb, r, q);
  }
  D1_FINE(5);
  pv_lds_barrier();
  D1_FINE(6);
}

// BWD = false: A.in = a[0], op[j] = stack op j.  BWD = true: A.in = dL/d(a[n]); op[j] = the stack's steps in reverse, K = that
// convolution's output channels, N = its input channels, out = dL/d(its input)
template <bool BWD>
__global__ __launch_bounds__(D1_THREADS) void pv_dec1d_kernel(D1Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x4 pre[D1_GMAX], pev = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    D1_STAMP(0);
    if (d1_mfma_step(A.op[0]) && wave < d1_tiles(A.op[0])) {
      int ob, lb;
      d1_split(wave, (A.op[0].N + 15) >> 4, ob, lb);
      d1_ldg_op(A.op[0], d1_wl(A.op[0], ob, lane), pre);
      pev = d1_ev<BWD>(A.op[0], b, ob, lb, lane & 15, lane >> 4);
    }
    D1_STAMP(48);
    {                                                  // the sample's input -> LDS rows 1 .. L0 (any width)
      float* dst = lds + A.op[0].li;
      const int P = d1_pitch(A.C0);
      const float* src = A.in + (int64_t)b * A.L0 * A.C0;
      if (!BWD && A.zd > 0) {                          // latent_to_features: the input is computed, not read (C0 % 16 == 0)
        const int c4n = A.C0 >> 2;
        const int64_t F = (int64_t)A.L0 * A.C0;
        float zv[8];
        // requested now, used after the head's finish and the sample: this thread's first latent_to_features element (bias and
        // up to four weight rows) — its round trip runs under the finish's instead of after it
        const int c4n0 = A.C0 >> 2;
        const bool pref = A.zd <= 4 && tid < A.L0 * c4n0;
        f32x4 pw[4], pb4 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (pref) {
          const int l = d1_div(tid, c4n0), c = 4 * (tid - l * c4n0);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < A.zd) pw[k] = *reinterpret_cast<const f32x4*>(A.l2f_wt + (int64_t)k * ((int64_t)A.L0 * A.C0) + (int64_t)l * A.C0 + c);
          if (A.l2f_b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pb4[j] = A.l2f_b[(int64_t)(c + j) * A.L0 + l];
          }
        }
        const float* hfin = nullptr;                   // the conv head's finish: bias + partial sums in segment order
        if (A.h_head && A.h_part) {
          // one partial sum per thread into LDS (the first step's output rows: free until the staging barrier), then thread j adds
          // output j's in segment order — one memory round trip instead of h_nseg dependent ones in every thread (round 5: this
          // stage 26.7 k -> 17.4 k cycles of the launch's 120 k; the same chain, the same bits)
          float* hs = lds + A.op[0].lo;
          const int np = A.h_nseg * A.h_ldh;           // (<= D1_THREADS: pv_dec1d_fwd checks)
          if (tid < np) hs[tid] = A.h_part[(int64_t)b * np + tid];
          pv_lds_barrier();
          if (tid < A.h_ldh) {
            float v = A.h_bias ? A.h_bias[tid] : 0.0f;
            for (int sg = 0; sg < A.h_nseg; ++sg) v += hs[sg * A.h_ldh + tid];
            hs[np + tid] = v;
            A.h_hout[(int64_t)b * A.h_ldh + tid] = v;
          }
          pv_lds_barrier();
          hfin = hs + np;
        }
        D1_STAMP(49);
        if (A.h_head) {
          // the reparameterised sample: component k by thread k (one copy of the softplus / log chain in the code — the launch runs
          // it once, cold: eight predicated copies cost more in instruction fetch than in arithmetic), the two log-density sums
          // by thread 0 in component order, every thread reads z back from LDS
          float* hz = lds + A.op[0].lo + D1_THREADS + 16;          // [0, 8): z, [8, 16): log q terms, [16, 24): log p terms
          if (tid < A.zd) {
            const int k = tid;
            float mu, sp;
            if (hfin) { mu = hfin[k]; sp = hfin[A.zd + k]; }
            else { mu = A.h_head[(int64_t)b * A.h_ldh + k]; sp = A.h_head[(int64_t)b * A.h_ldh + A.zd + k]; }
            const float sig = pv_softplus(sp), ep = A.h_eps[(int64_t)b * A.zd + k];
            const float zz = mu + sig * ep, d = zz - mu;
            hz[k] = zz;
            hz[8 + k] = -(d * d) / (2.0f * (sig * sig)) - logf(sig) - 0.91893853320467274178f;      // torch Normal.log_prob
            hz[16 + k] = -(zz * zz) / 2.0f - 0.91893853320467274178f;
            A.h_z[(int64_t)b * A.zd + k] = zz; A.h_zs[(int64_t)b * A.zd + k] = sig;
            if (A.h_zlo) A.h_zlo[(int64_t)b * A.zd + k] = mu;
            if (A.h_zso) A.h_zso[(int64_t)b * A.zd + k] = sig;
          }
          pv_lds_barrier();
          if (tid == 0) {
            float lp = 0.0f, lq = 0.0f;
            for (int k = 0; k < A.zd; ++k) { lq += hz[8 + k]; lp += hz[16 + k]; }
            A.h_kl[2 * b] = A.h_beta * lp; A.h_kl[2 * b + 1] = A.h_beta * lq;
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) { const float t = hz[k < A.zd ? k : 0]; zv[k] = k < A.zd ? t : 0.0f; }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) zv[k] = k < A.zd ? A.z[(int64_t)b * A.zd + k] : 0.0f;
        }
        D1_STAMP(50);
        for (int e = tid; e < A.L0 * c4n; e += D1_THREADS) {
          const int l = d1_div(e, c4n), c = 4 * (e - l * c4n);
          f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
          if (pref && e == tid) {                      // (the requested element: the same chain from registers)
            v = pb4;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (k < A.zd) v += zv[k] * pw[k];
          } else {
            if (A.l2f_b) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = A.l2f_b[(int64_t)(c + j) * A.L0 + l];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
              if (k < A.zd) v += zv[k] * *reinterpret_cast<const f32x4*>(A.l2f_wt + (int64_t)k * F + (int64_t)l * A.C0 + c);
          }
          *reinterpret_cast<f32x4*>(&dst[(l + 1) * P + c]) = v;
        }
      } else if ((A.C0 & 3) == 0) {
        const int c4n = A.C0 >> 2;
        for (int e = tid; e < A.L0 * c4n; e += D1_THREADS) {
          const int l = d1_div(e, c4n), c4 = e - l * c4n;
          *reinterpret_cast<f32x4*>(&dst[(l + 1) * P + 4 * c4]) = *reinterpret_cast<const f32x4*>(src + (int64_t)l * A.C0 + 4 * c4);
        }
      } else {
        for (int e = tid; e < A.L0 * A.C0; e += D1_THREADS) {
          const int l = d1_div(e, A.C0), c = e - l * A.C0;
          dst[(l + 1) * P + c] = src[e];
        }
      }
      D1_STAMP(51);
      d1_zero_halo(dst, A.L0, A.C0);
    }
    pv_lds_barrier();
    D1_STAMP(1);
#pragma nounroll
    for (int i = 0; i < A.n; ++i) {
      d1_step<BWD>(A, i, b, lds, wave, lane, pre, pev);
      D1_STAMP(2 + i);
    }
    // every step's result: LDS -> global in one burst
    for (int i = 0; i < A.n; ++i) {
      const D1Op& o = A.op[i];
      const int Lout = (!BWD && o.up) ? 2 * o.L : o.L, Po = d1_pitch(o.N);
      const float* src = lds + o.lo;
      float* dst = o.out + (int64_t)b * Lout * o.N;
      if ((o.N & 3) == 0) {
        const int c4n = o.N >> 2;
        for (int e = tid; e < Lout * c4n; e += D1_THREADS) {
          const int l = d1_div(e, c4n), c4 = e - l * c4n;
          *reinterpret_cast<f32x4*>(dst + (int64_t)l * o.N + 4 * c4) = *reinterpret_cast<const f32x4*>(&src[(l + 1) * Po + 4 * c4]);
        }
      } else {
        for (int e = tid; e < Lout * o.N; e += D1_THREADS) {
          const int l = d1_div(e, o.N), c = e - l * o.N;
          dst[e] = src[(l + 1) * Po + c];
        }
      }
    }
    if (!BWD && A.y) {                                 // log p(y | output): elements over threads, waves and then 8 partials in order
      const D1Op& o = A.op[A.n - 1];
      const int Lout = o.up ? 2 * o.L : o.L, Po = d1_pitch(o.N), per = Lout * o.N;
      const float* src = lds + o.lo;
      float acc = 0.0f;
      for (int e = tid; e < per; e += D1_THREADS) {
        const int l = d1_div(e, o.N), c = e - l * o.N;
        float ll, d, lv;
        pv_lik_one(src[(l + 1) * Po + c], A.y[(int64_t)b * per + e], A.lik, A.sigmoid_out, A.sig, ll, d, lv);
        if (A.loc) A.loc[(int64_t)b * per + e] = lv;
        if (A.dlda) A.dlda[(int64_t)b * per + e] = d;
        acc += ll;
      }
      acc = pv_wave_sum(acc);
      float* red = lds + o.lt;                         // (the row-pair scratch: unused by the forward)
      if (lane == 0) red[wave] = acc;
      pv_lds_barrier();
      if (tid == 0) {
        float v = 0.0f;
        for (int w = 0; w < D1_WAVES; ++w) v += red[w];
        A.llb[b] = v;
      }
    }
    if (!BWD && A.zd > 0) {                            // the computed input goes out too (the first layer's weight gradient reads it)
      const int P = d1_pitch(A.C0), c4n = A.C0 >> 2;
      const float* src = lds + A.op[0].li;
      float* dst = A.a0_out + (int64_t)b * A.L0 * A.C0;
      for (int e = tid; e < A.L0 * c4n; e += D1_THREADS) {
        const int l = d1_div(e, c4n), c4 = e - l * c4n;
        *reinterpret_cast<f32x4*>(dst + (int64_t)l * A.C0 + 4 * c4) = *reinterpret_cast<const f32x4*>(&src[(l + 1) * P + 4 * c4]);
      }
    }
    if (BWD && A.zd > 0) {                             // dz[b][k] = <g0, wt[k]>: threads over elements, waves and then 8 partials in order
      const D1Op& o = A.op[A.n - 1];
      const int Po = d1_pitch(o.N), c4n = o.N >> 2;
      const float* g0 = lds + o.lo;
      const int64_t F = (int64_t)o.L * o.N;
      float acc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
      for (int e = tid; e < o.L * c4n; e += D1_THREADS) {
        const int l = d1_div(e, c4n), c = 4 * (e - l * c4n);
        const f32x4 g = *reinterpret_cast<const f32x4*>(&g0[(l + 1) * Po + c]);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < A.zd) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(A.l2f_wt + (int64_t)k * F + (int64_t)l * o.N + c);
            acc[k] += (g[0] * w[0] + g[1] * w[1]) + (g[2] * w[2] + g[3] * w[3]);
          }
      }
      pv_lds_barrier();                                // (the row-pair scratch is free: every step is done)
      float* red = lds + o.lt;                         // [8 waves][8]
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = pv_wave_sum(acc[k]);
        if (lane == 0) red[wave * 8 + k] = v;
      }
      pv_lds_barrier();
      if (tid < A.zd) {
        float v = 0.0f;
        for (int w = 0; w < D1_WAVES; ++w) v += red[w * 8 + tid];
        A.dz[(int64_t)b * A.zd + tid] = v;
        if (A.h_head) {                                // head backward (pv_head_bwd_elem, no coordinates / weights)
          const int64_t e = (int64_t)b * A.zd + tid;
          const float zz = A.h_z[e], sig = A.h_zs[e], ep = A.h_eps[e];
          const float sp = A.h_head[(int64_t)b * A.h_ldh + A.zd + tid];
          const float g = v + A.h_beta * zz;           // d(-ll - beta log p(z)) / dz
          const float dsig = g * ep - A.h_beta / sig;  // + beta d(log q) / d(sigma) (total derivative)
          const float sgm = sp > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-sp));
          A.h_dhead[(int64_t)b * A.h_ldh + tid] = g;
          A.h_dhead[(int64_t)b * A.h_ldh + A.zd + tid] = dsig * sgm;
        }
      }
    }
    D1_STAMP(2 + A.n + 1);
    pv_lds_barrier();                                  // (the next sample reuses the buffers)
  }
}

bool dec1d_on() {                                      // (per plan: PV_PLAN_NO_DEC1D; PV_NO_DEC1D=1 in the experiments build)
  static const bool on = !pv_exp_int("PV_NO_DEC1D", 0);
  return on;
}

bool lin_act(int act) { return act == PV_ACT_NONE || act == PV_ACT_RELU || act == PV_ACT_LRELU || act == PV_ACT_TANH ||
                               act == PV_ACT_SIGMOID || act == PV_ACT_SOFTPLUS; }     // (derivative through the output alone)


// the steps of the stack: a CONV, or a CONV k1 without activation + the UPSAMPLE2 after it
struct Step { int conv; int up; int L; };
int steps_of(const pv_op* ops, int n, int L0, int C0, Step* st) {
  int ns = 0, L = L0, C = C0;
  if ((C0 & 15) != 0 || (L0 & 15) != 0) return -1;
  for (int i = 0; i < n; ++i) {
    const pv_op& o = ops[i];
    if (o.kind != PV_OP_CONV || (o.ksize != 1 && o.ksize != 3) || o.cin != C || !lin_act(o.act)) return -1;
    if ((o.cin & 15) != 0 || (L & 15) != 0 || o.cout < 1 || ns >= PV_D1_MAXOPS) return -1;
    const bool last = i + 1 == n;
    if (!last && (o.cout & 15) != 0) return -1;        // (only the output layer may be narrow)
    int up = 0;
    if (!last && ops[i + 1].kind == PV_OP_UPSAMPLE2) {
      if (o.ksize != 1 || o.act != PV_ACT_NONE) return -1;
      up = 1;
    }
    st[ns++] = Step{i, up, L};
    C = o.cout;
    if (up) { L *= 2; ++i; }
  }
  return ns;
}

}  // namespace

bool pv_dec1d_enabled() { return dec1d_on(); }

static int64_t lay_out(D1Args& A, bool bwd);
static bool build(D1Args& A, bool bwd, const float* params, const pv_op* ops, int n, const float* wt, int B, int L0, int C0,
                  float* const* a, const float* g_out, float* const* gown);

bool pv_dec1d_supported(const pv_op* ops, int n, int nd, int L0, int C0) {
  if (nd != 1 || n < 1) return false;
  const int lim = pv_device_lds_limit();               // (gfx950: 160 KB; another device falls back to the layer launches)
  if (lim > 0 && lim < D1_LDS_MAX) return false;
  D1Args A;
  for (int bwd = 0; bwd < 2; ++bwd) {                  // a sample's activations (either direction) fit the LDS
    if (!build(A, bwd != 0, nullptr, ops, n, nullptr, 1, L0, C0, nullptr, nullptr, nullptr)) return false;
    if (lay_out(A, bwd != 0) * (int64_t)sizeof(float) > D1_LDS_MAX) return false;
  }
  return true;
}

// floats of one orientation's tiling: N rows (padded to 16 in fragment order), contraction C per tap
static int64_t wt_size(int N, int C, int KK) {
  if ((C & 15) != 0) return pv_align_up((int64_t)N * C * KK, 64);
  return (int64_t)((N + 15) / 16) * d1_nckp_of(KK, C) * 256;          // (chunks padded to groups)
}
static int64_t wt_size(const pv_op& o, int flip) { return flip ? wt_size(o.cin, o.cout, o.ksize) : wt_size(o.cout, o.cin, o.ksize); }

int64_t pv_dec1d_wt_floats(const pv_op* ops, int n) {
  int64_t f = 0;
  for (int i = 0; i < n; ++i)
    if (ops[i].kind == PV_OP_CONV) f += wt_size(ops[i], 0) + wt_size(ops[i], 1);
  return f;
}

// wt: pv_dec1d_wt_floats floats; entries for pv_conv_wprep_table appended to e[ne...] (kind 8, both orientations of every conv)
void pv_dec1d_wt_entries(const float* params, const pv_op* ops, int n, float* wt, PvWprepEntry* e, int& ne) {
  int64_t off = 0;
  for (int i = 0; i < n; ++i) {
    if (ops[i].kind != PV_OP_CONV) continue;
    for (int flip = 0; flip < 2; ++flip) {
      PvWprepEntry& E = e[ne++];
      E.w = params + ops[i].w_off; E.dst = reinterpret_cast<char*>(wt + off);
      E.Co = ops[i].cout; E.Ci = ops[i].cin; E.KK = ops[i].ksize; E.flip = flip; E.kind = 8; E.pad_ = 0; E.start = E.total = 0;
      off += wt_size(ops[i], flip);
    }
  }
}

static const float* wt_of(const pv_op* ops, int upto, const float* wt, int flip) {
  int64_t off = 0;
  for (int i = 0; i < upto; ++i)
    if (ops[i].kind == PV_OP_CONV) off += wt_size(ops[i], 0) + wt_size(ops[i], 1);
  return wt + off + (flip ? wt_size(ops[upto], 0) : 0);
}

static unsigned d1_grid(int B) { return (unsigned)(B < 2048 ? B : 2048); }

// LDS float offsets of a direction's steps (every step's result stays resident): returns the total in floats
static int64_t lay_out(D1Args& A, bool bwd) {
  int64_t off = 0;
  auto take = [&](int L, int C) { const int64_t o = off; off += (d1_rows_floats(L, C) + 3) / 4 * 4; return (int)o; };
  int in0 = take(A.L0, A.C0), prev = in0;
  int tmax = 0;
  for (int j = 0; j < A.n; ++j) {
    D1Op& d = A.op[j];
    d.li = prev;
    const int Lout = (!bwd && d.up) ? 2 * d.L : d.L;
    d.lo = take(Lout, d.N);
    prev = d.lo;
    if (bwd && d.up && d1_rows_floats(d.L, d.K) > tmax) tmax = d1_rows_floats(d.L, d.K);
  }
  if (tmax < 64) tmax = 64;                            // (also the 8 x 8 partials of the latent gradient)
  const int lt = (int)off;                             // one shared buffer for the row-pair sums
  off += (tmax + 3) / 4 * 4;
  for (int j = 0; j < A.n; ++j) A.op[j].lt = lt;
  // (forward staging scratch from the first step's output rows on: D1_THREADS partial sums, 16 head outputs, 24 sample terms)
  if (!bwd && A.n > 0 && off < A.op[0].lo + D1_THREADS + 48) off = A.op[0].lo + D1_THREADS + 48;
  return off;
}

static bool build(D1Args& A, bool bwd, const float* params, const pv_op* ops, int n, const float* wt, int B, int L0, int C0,
                  float* const* a, const float* g_out, float* const* gown) {
  Step st[PV_D1_MAXOPS];
  const int ns = steps_of(ops, n, L0, C0, st);
  if (ns <= 0) return false;
  A = D1Args{};
  A.n = ns; A.B = B;
  if (!bwd) {
    A.in = a ? a[0] : nullptr; A.L0 = L0; A.C0 = C0;
    for (int j = 0; j < ns; ++j) {
      const pv_op& o = ops[st[j].conv];
      D1Op& d = A.op[j];
      d.w = wt ? wt_of(ops, st[j].conv, wt, 0) : nullptr;
      d.e = (params && o.b_off >= 0) ? params + o.b_off : nullptr;
      d.out = a ? a[st[j].conv + 1 + st[j].up] : nullptr;   // (a fused pair writes the upsampled tensor; the one between is never written)
      d.K = o.cin; d.N = o.cout; d.L = st[j].L; d.taps = o.ksize; d.act = o.act; d.up = st[j].up;
      d.G = d1_group(d.taps, d.K);
    }
  } else {
    A.in = g_out;
    const Step& l = st[ns - 1];
    A.L0 = l.up ? 2 * l.L : l.L; A.C0 = ops[l.conv].cout;
    for (int j = 0; j < ns; ++j) {
      const Step& t = st[ns - 1 - j];
      const pv_op& o = ops[t.conv];
      D1Op& d = A.op[j];
      d.w = wt ? wt_of(ops, t.conv, wt, 1) : nullptr;
      d.out = gown ? gown[t.conv] : nullptr;
      const int pact = t.conv > 0 && ops[t.conv - 1].kind == PV_OP_CONV ? ops[t.conv - 1].act : PV_ACT_NONE;
      d.e = (pact != PV_ACT_NONE && a) ? a[t.conv] : nullptr;
      d.K = o.cout; d.N = o.cin; d.L = t.L; d.taps = o.ksize; d.act = pact; d.up = t.up;
      d.G = d1_group(d.taps, d.K);
    }
  }
  return true;
}

template <bool BWD>
static int launch(D1Args& A, hipStream_t s) {
  const int64_t lds = lay_out(A, BWD) * (int64_t)sizeof(float);
  if (lds > D1_LDS_MAX) return PV_EINVAL;
  const void* fn = reinterpret_cast<const void*>(&pv_dec1d_kernel<BWD>);
  PV_TRY(pv_set_dynamic_lds(fn, D1_LDS_MAX));           // (per device)
  // (backward: the recorded weight gradients fork off this launch; forward: the step's loss scalars, when the caller armed an event)
  PV_LAUNCH_FORK(pv_dec1d_kernel<BWD>, dim3(d1_grid(A.B)), dim3(D1_THREADS), (size_t)lds, s, A);
  PV_LAUNCH_CHECK();
  return 0;
}

int pv_dec1d_fwd(const float* params, const pv_op* ops, int n, const float* wt, int B, int L0, int C0, float* const* a, hipStream_t s,
                 const PvD1L2f* l2f, const PvD1Lik* lk, const PvD1Head* hd) {
  D1Args A;
  if (!build(A, false, params, ops, n, wt, B, L0, C0, a, nullptr, nullptr)) return PV_EINVAL;
  if (hd) {
    if (!l2f || !hd->head || !hd->eps || !hd->z || !hd->z_scale || !hd->kl_part) return PV_EINVAL;
    A.h_head = hd->head; A.h_eps = hd->eps; A.h_z = hd->z; A.h_zs = hd->z_scale; A.h_zlo = hd->z_loc_out; A.h_zso = hd->z_scale_out;
    A.h_kl = hd->kl_part; A.h_ldh = hd->ldh; A.h_beta = hd->beta;
    if (hd->part) {
      if (!hd->head_out || hd->nseg < 1 || hd->ldh != 2 * l2f->zd) return PV_EINVAL;
      if ((int64_t)hd->nseg * hd->ldh > D1_THREADS) return PV_EINVAL;    // (one partial sum per thread of the staging pass)
      A.h_part = hd->part; A.h_bias = hd->bias; A.h_hout = hd->head_out; A.h_nseg = hd->nseg;
    }
  }
  if (lk) {
    if (!lk->y || !lk->llb) return PV_EINVAL;
    A.y = lk->y; A.loc = lk->loc; A.dlda = lk->dlda; A.llb = lk->llb; A.lik = lk->lik; A.sigmoid_out = lk->sigmoid_out; A.sig = lk->sig;
  }
  if (l2f) {
    if (!pv_dec1d_l2f_ok(l2f->zd) || (!l2f->z && !hd) || !l2f->wt) return PV_EINVAL;
    A.z = l2f->z; A.l2f_wt = l2f->wt; A.l2f_b = l2f->bias; A.a0_out = a[0]; A.zd = l2f->zd;
  }
  return launch<false>(A, s);
}

// g_out = dL/d(a[n]) (B, Ln, Cn); gown[i] <- dL/d(a[i]) for every conv op i (with act'(a[i]) of a producing convolution applied)
int pv_dec1d_bwd(const pv_op* ops, int n, const float* wt, int B, int L0, int C0, float* const* a, const float* g_out,
                 float* const* gown, hipStream_t s, const PvD1L2f* l2f, const PvD1Head* hd) {
  D1Args A;
  if (!build(A, true, nullptr, ops, n, wt, B, L0, C0, a, g_out, gown)) return PV_EINVAL;
  if (hd) {
    if (!l2f || !hd->head || !hd->eps || !hd->z || !hd->z_scale || !hd->dhead) return PV_EINVAL;
    A.h_head = hd->head; A.h_eps = hd->eps; A.h_z = hd->z; A.h_zs = hd->z_scale; A.h_dhead = hd->dhead; A.h_ldh = hd->ldh;
    A.h_beta = hd->beta;
  }
  if (l2f) {
    if (!pv_dec1d_l2f_ok(l2f->zd) || !l2f->wt || !l2f->dz || (A.op[A.n - 1].N & 3) != 0) return PV_EINVAL;
    A.l2f_wt = l2f->wt; A.dz = l2f->dz; A.zd = l2f->zd;
  }
  return launch<true>(A, s);
}
