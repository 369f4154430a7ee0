// pv_gemm.hip — generic fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), with the
// fused epilogues the Linear(+activation) layers of pyroVED's fc nets need (nets/fc.py:307-324).
//
// This is the GENERAL path: any layer width, any of the reference's activations, any batch.
// The default-architecture hot loop (sDecoderNet 128-128, tanh) has its own fused persistent
// kernel in pv_sdec_fused.hip; this file serves the encoder, the heads, non-default hidden sizes,
// and is the layer-by-layer cross-check of the fused kernel.
//
// Tiling: 64x64 output tile per 256-thread workgroup (4 waves, 2x2, one 32x32 MFMA block each),
// K advanced 16 at a time through LDS (k-major images, row stride 66 floats so both the k-contiguous
// and the m-contiguous global layouts stage without >2-way ds_write conflicts and every MFMA operand
// read is a conflict-free ds_read_b32 of 32 consecutive floats per half-wave).  Operands are
// described by (row stride, col stride) so NT (forward), NN (dgrad) and TN (wgrad) share the kernel.
// fp32 MFMA == an fmaf chain bit-for-bit, so results are exact-fp32-class (no TF32 on gfx950).
#include "pv_common.h"
#include <stdlib.h>

#define GT 64      // tile edge
#define GK 16      // k per stage
#define GLD 66     // LDS row stride (floats)

// implicit im2col: element (row = (b, y, x), j = ci*KK + tap) of the never-materialised patch matrix.  The two
// halves of the address are decoded separately so that whichever is fixed for a thread over the whole k-loop (the
// patch row in the forward, the patch column in the wgrad) is decoded once per tile.
struct ConvGeom { int H, W, C, nd; };
struct ConvRow { int64_t base; int y, x; bool ok; };        // base = ((b*H + y)*W + x)*C
struct ConvCol { int off, dy, dx; bool ok; };               // off = (dy*W + dx)*C + ci
__device__ __forceinline__ ConvRow g_conv_row(const ConvGeom& cg, int64_t row, bool ok) {
  ConvRow r;
  r.x = (int)(row % cg.W);
  const int64_t ry = row / cg.W;
  r.y = (int)(ry % cg.H);
  r.base = row * cg.C;
  r.ok = ok;
  return r;
}
__device__ __forceinline__ ConvCol g_conv_col(const ConvGeom& cg, int j, bool ok) {
  const int KK = cg.nd == 2 ? 9 : 3;
  const int ci = j / KK, t = j - ci * KK;
  ConvCol c;
  c.dy = (cg.nd == 2 ? t / 3 : t) - 1;
  c.dx = cg.nd == 2 ? t % 3 - 1 : 0;
  c.off = (c.dy * cg.W + c.dx) * cg.C + ci;
  c.ok = ok;
  return c;
}
__device__ __forceinline__ float g_conv_at(const float* __restrict__ in, const ConvGeom& cg, const ConvRow& r,
                                           const ConvCol& c) {
  const int yy = r.y + c.dy, xx = r.x + c.dx;
  if (!r.ok || !c.ok || yy < 0 || yy >= cg.H || xx < 0 || xx >= cg.W) return 0.0f;
  return in[r.base + c.off];
}

template <bool KCONTIG>   // true: the k index is the contiguous one in global memory
__device__ __forceinline__ void g_load(const float* __restrict__ P, int64_t rs, int64_t cs, int x0, int xlim,
                                       int k0, int klim, bool vec, int t, float (&r)[4]) {
  // tile is 64 (x: m or n) by 16 (k).  P(x,k) = P[x*rs + k*cs]
  if (KCONTIG) {
    const int x = x0 + (t >> 2), k = k0 + (t & 3) * 4;
    const float* p = P + (int64_t)x * rs + k;
    if (x < xlim && k + 3 < klim && vec) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = (x < xlim && k + i < klim) ? p[(int64_t)i * cs] : 0.0f;
    }
  } else {
    const int k = k0 + (t >> 4), x = x0 + (t & 15) * 4;
    const float* p = P + (int64_t)k * cs + (int64_t)x * rs;
    if (k < klim && x + 3 < xlim && vec) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = (k < klim && x + i < xlim) ? p[(int64_t)i * rs] : 0.0f;
    }
  }
}

template <bool KCONTIG>
__device__ __forceinline__ void g_store_lds(float (*S)[GLD], int t, const float (&r)[4]) {
  if (KCONTIG) {
    const int x = t >> 2, k = (t & 3) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) S[k + i][x] = r[i];
  } else {
    const int k = t >> 4, x = (t & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) S[k][x + i] = r[i];
  }
}

struct GemmK {
  PvGemm g;
  int k_chunk;
  float* part;     // != null: raw partial sums part[z][M][N]
  float* part_rs;  // != null: partial row sums of A, part_rs[z][M]
  int a_vec, b_vec;
};

template <bool AK, bool BK, bool CA = false, bool CB = false>   // CA / CB: implicit im2col A / B operand
__device__ __forceinline__ void gemm_tile(const GemmK& p, float (*As)[GLD], float (*Bs)[GLD], int bx, int by, int bz) {
  const PvGemm& g = p.g;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = bx * GT, n0 = by * GT;
  const int kbeg = bz * p.k_chunk;
  const int kend = min(g.K, kbeg + p.k_chunk);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  const bool do_rs = g.rowsumA != nullptr && by == 0 && t < GT;
  float rs = 0.0f;

  float ra[4], rb[4];
  const ConvGeom cg{g.cH, g.cW, g.cC, g.cnd};
  // conv operands: A (forward / dgrad-as-convolution): thread = (row m0 + t/4 fixed, 4 consecutive patch columns per
  // k-step); B (wgrad): thread = (4 consecutive patch columns n0 + 4*(t%16).. fixed, one patch row per k-step)
  ConvRow arow{};
  ConvCol bcol[4] = {};
  if (CA) arow = g_conv_row(cg, m0 + (t >> 2), m0 + (t >> 2) < g.M);
  if (CB) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bcol[i] = g_conv_col(cg, n0 + (t & 15) * 4 + i, n0 + (t & 15) * 4 + i < g.N);
  }
  auto load_a = [&](int k0) {
    if (CA) {
      const int k = k0 + (t & 3) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = g_conv_at(g.A, cg, arow, g_conv_col(cg, k + i, k + i < kend));
    } else {
      g_load<AK>(g.A, g.a_rs, g.a_cs, m0, g.M, k0, kend, p.a_vec, t, ra);
    }
  };
  auto load_b = [&](int k0) {
    if (CB) {
      const int k = k0 + (t >> 4);
      const ConvRow br = g_conv_row(cg, k, k < kend);
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[i] = g_conv_at(g.B, cg, br, bcol[i]);
    } else {
      g_load<BK>(g.B, g.b_cs, g.b_rs, n0, g.N, k0, kend, p.b_vec, t, rb);
    }
  };
  if (kbeg < kend) { load_a(kbeg); load_b(kbeg); }
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    g_store_lds<AK>(As, t, ra);
    g_store_lds<BK>(Bs, t, rb);
    __syncthreads();
    if (k0 + GK < kend) { load_a(k0 + GK); load_b(k0 + GK); }
    if (do_rs) {
#pragma unroll
      for (int k = 0; k < GK; ++k) rs += As[k][t];
    }
    const int i = lane & 31, kk = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < GK / 2; ++ks) {
      const float a = As[2 * ks + kk][wm * 32 + i];
      const float b = Bs[2 * ks + kk][wn * 32 + i];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  if (do_rs && m0 + t < g.M) {
    if (p.part_rs) p.part_rs[(int64_t)bz * g.M + m0 + t] = rs;
    else g.rowsumA[m0 + t] = rs;
  }
  // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int n = n0 + wn * 32 + (lane & 31);
  if (n >= g.N) return;
  const float bias = (g.bias && !p.part) ? g.bias[n] : 0.0f;
  const bool simple = pv_act_is_lin(g.act) && (!g.aux || pv_act_is_lin(g.act_aux));      // (pv_common.h: epilogue helpers)
  const bool lin_f = g.act == PV_ACT_NONE && g.aux != nullptr;                            // input-gradient form: only act_aux' to apply
  const float slope = pv_act_slope(g.act), gslope = pv_act_slope(g.act_aux);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m >= g.M) continue;
    float v = acc[r];
    if (p.part) {
      p.part[((int64_t)bz * g.M + m) * g.N + n] = v;
    } else {
      v += bias;
      if (g.pre) g.pre[(int64_t)m * g.ldc + n] = v;
      const float y = g.aux ? g.aux[(int64_t)m * g.ldaux + n] : 0.0f;
      if (simple) v = (v > 0.0f ? v : v * slope) * (g.aux ? (y > 0.0f ? 1.0f : gslope) : 1.0f);
      else if (lin_f) v *= pv_act_grad2(y, g.auxpre ? g.auxpre[(int64_t)m * g.ldaux + n] : 0.0f, g.act_aux);   // (tanh' / sigmoid' inline)
      else v = pv_act_pair_slow(v, g.act, g.aux != nullptr, y, g.auxpre ? g.auxpre[(int64_t)m * g.ldaux + n] : 0.0f, g.act_aux);
      g.C[(int64_t)m * g.ldc + n] = v;
    }
  }
}

template <bool AK, bool BK, bool CA = false, bool CB = false>
__global__ __launch_bounds__(256) void pv_gemm_kernel(GemmK p) {
  __shared__ float As[GK][GLD];
  __shared__ float Bs[GK][GLD];
  gemm_tile<AK, BK, CA, CB>(p, As, Bs, blockIdx.x, blockIdx.y, blockIdx.z);
}

// sums the split-K partials in ascending split order (deterministic) and applies the epilogue
__global__ __launch_bounds__(256) void pv_gemm_finish_kernel(GemmK p, int splits) {
  const PvGemm& g = p.g;
  const int64_t total = (int64_t)g.M * g.N;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int m = (int)(e / g.N), n = (int)(e % g.N);
    // ascending split order (deterministic); four independent chains keep 8+ loads in flight per thread
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
    int z = 0;
    for (; z + 3 < splits; z += 4) {
      v0 += p.part[(int64_t)z * total + e];
      v1 += p.part[(int64_t)(z + 1) * total + e];
      v2 += p.part[(int64_t)(z + 2) * total + e];
      v3 += p.part[(int64_t)(z + 3) * total + e];
    }
    for (; z < splits; ++z) v0 += p.part[(int64_t)z * total + e];
    float v = (v0 + v1) + (v2 + v3);
    if (g.bias) v += g.bias[n];
    if (g.pre) g.pre[(int64_t)m * g.ldc + n] = v;
    v = pv_act_fwd(v, g.act);
    if (g.aux) {
      const float y = g.aux[(int64_t)m * g.ldaux + n];
      const float pr = g.auxpre ? g.auxpre[(int64_t)m * g.ldaux + n] : 0.0f;
      v *= pv_act_grad(y, pr, g.act_aux);
    }
    g.C[(int64_t)m * g.ldc + n] = v;
  }
  if (p.part_rs) {
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < g.M; m += (int64_t)gridDim.x * 256) {
      float v = 0.0f;
      for (int z = 0; z < splits; ++z) v += p.part_rs[(int64_t)z * g.M + m];
      g.rowsumA[m] = v;
    }
  }
}

// many splits, few outputs (wgrads of small layers over ~1e6 rows): the loop above is a chain of dependent-latency loads
// in a handful of workgroups.  Here a workgroup takes 4 outputs x 64 split slices and meets in a fixed-order LDS tree.
__global__ __launch_bounds__(256) void pv_gemm_finish_deep_kernel(GemmK p, int splits) {
  __shared__ float sm[64][4];
  const PvGemm& g = p.g;
  const int64_t total = (int64_t)g.M * g.N, nrs = p.part_rs ? g.M : 0;
  const int o = threadIdx.x & 3, sl = threadIdx.x >> 2;
  const int64_t i = (int64_t)blockIdx.x * 4 + o;          // [0, total): C entries; [total, total + M): row sums
  float v = 0.0f;
  if (i < total) {
    for (int z = sl; z < splits; z += 64) v += p.part[(int64_t)z * total + i];
  } else if (i < total + nrs) {
    for (int z = sl; z < splits; z += 64) v += p.part_rs[(int64_t)z * g.M + (i - total)];
  }
  sm[sl][o] = v;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (sl < w) sm[sl][o] += sm[sl + w][o];
    __syncthreads();
  }
  if (sl != 0) return;
  v = sm[0][o];
  if (i < total) {
    const int m = (int)(i / g.N), n = (int)(i % g.N);
    if (g.bias) v += g.bias[n];
    if (g.pre) g.pre[(int64_t)m * g.ldc + n] = v;
    v = pv_act_fwd(v, g.act);
    if (g.aux) {
      const float y = g.aux[(int64_t)m * g.ldaux + n];
      const float pr = g.auxpre ? g.auxpre[(int64_t)m * g.ldaux + n] : 0.0f;
      v *= pv_act_grad(y, pr, g.act_aux);
    }
    g.C[(int64_t)m * g.ldc + n] = v;
  } else if (i < total + nrs) {
    g.rowsumA[i - total] = v;
  }
}

int pv_gemm_pick_splits(int M, int N, int K) {
  // aim for >= ~512 workgroups (2 per CU) when the contraction is long enough to split; a split costs a
  // second (finish) launch, ~6 us on the stream, so short contractions are never split
  const int64_t tiles = (int64_t)((M + GT - 1) / GT) * ((N + GT - 1) / GT);
  static const int few_env = pv_exp_int("PV_GEMM_NO_FEWSPLIT", 0) ? 0 : 1;
  if (few_env && tiles < 64 && K >= 256 && K <= 1024) {
    // a handful of tiles over a medium contraction (a 256 x 787 -> 128 encoder layer: 8 workgroups walking 25 stages, 24 us):
    // at least 128 of k per split, ~128 workgroups in all — the walk shrinks to 4-5 stages, the finish launch costs ~4 us
    int64_t s = (128 + tiles - 1) / tiles;
    const int64_t maxs = K / 128;
    if (s > maxs) s = maxs;
    return (int)(s < 1 ? 1 : s);
  }
  if (tiles >= 512 || K <= 1024) return 1;
  int64_t s = (512 + tiles - 1) / tiles;
  const int64_t maxs = (K + 4 * GK - 1) / (4 * GK);     // at least 64 of k per split
  if (s > maxs) s = maxs;
  if (s > 1024) s = 1024;
  return (int)(s < 1 ? 1 : s);
}

int pv_gemm(const PvGemm& g, int splits, void* ws, int64_t ws_bytes, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0) return 0;
  if (g.K <= 0) return PV_EINVAL;
  GemmK p;
  p.g = g;
  if (splits < 1) splits = 1;
  int k_chunk = (g.K + splits - 1) / splits;
  k_chunk = (k_chunk + GK - 1) / GK * GK;           // keep split boundaries on stage boundaries
  splits = (g.K + k_chunk - 1) / k_chunk;
  p.k_chunk = k_chunk;
  p.part = nullptr;
  p.part_rs = nullptr;
  if (splits > 1) {
    const int64_t need = (int64_t)splits * g.M * (g.N + (g.rowsumA ? 1 : 0)) * (int64_t)sizeof(float);
    if (!ws || ws_bytes < need) return PV_EWS;        // (callers size their scratch with gemm_ws_need = the same pick_splits)
    p.part = (float*)ws;
    if (g.rowsumA) p.part_rs = p.part + (int64_t)splits * g.M * g.N;
  }
  const bool ak = (g.a_cs == 1), bk = (g.b_rs == 1);
  // 16-byte vector loads need an aligned base and a stride that keeps rows aligned
  auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  p.a_vec = al(g.A) && (ak ? (g.a_rs % 4 == 0) : (g.a_rs == 1 && g.a_cs % 4 == 0));
  p.b_vec = al(g.B) && (bk ? (g.b_cs % 4 == 0) : (g.b_cs == 1 && g.b_rs % 4 == 0));
  if (g.conv_a && g.K > g.cC * 9) return PV_EINVAL;
  dim3 grid((g.M + GT - 1) / GT, (g.N + GT - 1) / GT, splits);
  if (grid.y > 65535) return PV_EINVAL;
  if (g.conv_a || g.conv_b) {
    // convolution operands: forward / dgrad-as-convolution (A = patches, k-contiguous; B = weights, k-contiguous)
    // and wgrad (A = dpre^T, B = patches with the patch row on k)
    if (g.conv_a && !g.conv_b && ak && bk) hipLaunchKernelGGL((pv_gemm_kernel<true, true, true, false>), grid, dim3(256), 0, s, p);
    else if (g.conv_b && !g.conv_a && !ak && !bk) hipLaunchKernelGGL((pv_gemm_kernel<false, false, false, true>), grid, dim3(256), 0, s, p);
    else return PV_EINVAL;
  } else if (ak && bk) hipLaunchKernelGGL((pv_gemm_kernel<true, true>), grid, dim3(256), 0, s, p);
  else if (ak && !bk) hipLaunchKernelGGL((pv_gemm_kernel<true, false>), grid, dim3(256), 0, s, p);
  else if (!ak && bk) hipLaunchKernelGGL((pv_gemm_kernel<false, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((pv_gemm_kernel<false, false>), grid, dim3(256), 0, s, p);
  PV_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t total = (int64_t)g.M * g.N;
    if (splits >= 32 && total + g.M <= 4 * 16384) {
      hipLaunchKernelGGL(pv_gemm_finish_deep_kernel, dim3((unsigned)((total + g.M + 3) / 4)), dim3(256), 0, s, p, splits);
      PV_LAUNCH_CHECK();
      return 0;
    }
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pv_gemm_finish_kernel, dim3(blocks), dim3(256), 0, s, p, splits);
    PV_LAUNCH_CHECK();
  }
  return 0;
}

// ---- deterministic reductions ----------------------------------------------------------------
__global__ __launch_bounds__(256) void pv_reduce_partials_kernel(const float* __restrict__ part, int nparts,
                                                                 int64_t stride, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = 0.0f;
    for (int q = 0; q < nparts; ++q) v += part[(int64_t)q * stride + i];
    out[i] = v;
  }
}

// many partials, few outputs: 4 outputs x 64 partial-slices per workgroup, fixed-order LDS tree
__global__ __launch_bounds__(256) void pv_reduce_partials_deep_kernel(const float* __restrict__ part, int nparts,
                                                                      int64_t stride, float* __restrict__ out,
                                                                      int64_t n) {
  __shared__ float sm[64][4];
  const int o = threadIdx.x & 3, sl = threadIdx.x >> 2;
  const int64_t i = (int64_t)blockIdx.x * 4 + o;
  float v = 0.0f;
  if (i < n)
    for (int q = sl; q < nparts; q += 64) v += part[(int64_t)q * stride + i];
  sm[sl][o] = v;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (sl < w) sm[sl][o] += sm[sl + w][o];
    __syncthreads();
  }
  if (sl == 0 && i < n) out[i] = sm[0][o];
}

int pv_reduce_partials(const float* part, int nparts, int64_t stride, float* out, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  if (nparts > 32 && n <= 4 * 65535) {
    hipLaunchKernelGGL(pv_reduce_partials_deep_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, part, nparts,
                       stride, out, n);
    PV_LAUNCH_CHECK();
    return 0;
  }
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pv_reduce_partials_kernel, dim3(blocks), dim3(256), 0, s, part, nparts, stride, out, n);
  PV_LAUNCH_CHECK();
  return 0;
}

// column sums of x[M,N] (bias gradients): row-chunked partials, then the ordered reduction above
#define CS_ROWS 256
__global__ __launch_bounds__(256) void pv_colsum_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int N,
                                                        float* __restrict__ part) {
  __shared__ float sm[256];
  const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS;
  const int64_t r1 = r0 + CS_ROWS < M ? r0 + CS_ROWS : M;
  // narrow matrices: split the rows of the chunk over 256/width thread groups, combine in fixed order
  const int width = N > 128 ? 256 : (N > 64 ? 128 : 64);
  const int groups = 256 / width;
  const int c = threadIdx.x % width, rg = threadIdx.x / width;
  for (int n0 = 0; n0 < N; n0 += width) {
    const int n = n0 + c;
    float v = 0.0f;
    if (n < N)
      for (int64_t r = r0 + rg; r < r1; r += groups) v += x[r * ldx + n];
    __syncthreads();
    sm[threadIdx.x] = v;
    __syncthreads();
    if (rg == 0 && n < N) {
      float a = sm[c];
      for (int g = 1; g < groups; ++g) a += sm[g * width + c];
      part[(int64_t)blockIdx.x * N + n] = a;
    }
  }
}

int64_t pv_colsum_ws(int64_t M, int N) { return ((M + CS_ROWS - 1) / CS_ROWS) * (int64_t)N * (int64_t)sizeof(float); }

int pv_colsum(const float* x, int64_t ldx, int64_t M, int N, float* out, void* ws, int64_t ws_bytes, hipStream_t s) {
  if (N <= 0) return 0;
  const int64_t chunks = (M + CS_ROWS - 1) / CS_ROWS;
  if (chunks <= 0) return PV_EINVAL;
  if (!ws || ws_bytes < pv_colsum_ws(M, N)) return PV_EWS;
  hipLaunchKernelGGL(pv_colsum_kernel, dim3((unsigned)chunks), dim3(256), 0, s, x, ldx, M, N, (float*)ws);
  PV_LAUNCH_CHECK();
  return pv_reduce_partials((const float*)ws, (int)chunks, N, out, N, s);
}
