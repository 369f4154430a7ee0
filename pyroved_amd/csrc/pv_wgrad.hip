// pv_wgrad.hip — weight gradients of the small Linear layers around the spatial decoder (encoder, fc_latent):
//     dW[m][n] = sum_k A(m,k) B(k,n),   A = dpre^T (stored [K][M], k = minibatch sample),  B = layer input [K][N]
//     db[m]    = sum_k A(m,k)
// (torch.nn.Linear backward: grad_weight = grad_output^T input, grad_bias = grad_output.sum(0).)
// These problems have a SHORT contraction (K = minibatch, a few hundred) and a wide output (128 x 784 for the
// first encoder layer): an LDS-tiled 64x64 GEMM gives ~30 workgroups each looping over K behind barriers, i.e.
// latency-bound at ~24 us.  Here a workgroup owns one 16x16 output tile and its 4 waves split K in 64-k register
// batches (wave w takes batches w, w+4, ...): at K = 256 every operand of the whole launch is requested at once,
// one memory latency in all.  Operands stream from L2 straight into registers and feed v_mfma_f32_16x16x4_f32
// (fp32 in, fp32 accumulate) on four independent accumulators; the 4 partial tiles meet in LDS.  Up to 4
// problems share one launch (~470 workgroups for the iVAE 28x28 step, several resident per CU).
#include "pv_wgrad_small.h"

__global__ __launch_bounds__(64 * WG_WAVES) void pv_wgrad_small_kernel(PvWgradSmall w) {
  __shared__ float part[WG_WAVES][16][17];
  __shared__ float rpart[WG_WAVES][16];
  pv_wgrad_small_block(w, (int)blockIdx.x, (int)gridDim.x, part, rpart);
}

// gs[i]: plain wgrad problems (no bias / activation / aux epilogue); any strides, any M, N, K >= 1
int pv_wgrad_small_fill(PvWgradSmall& w, const PvGemm* gs, int n, const PvAdamFuse* adam, const PvFinishArgs* fin, int* guests_out) {
  if (n < 1 || n > 4) return PV_EINVAL;
  w = PvWgradSmall{};
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (gs[i].bias || gs[i].aux || gs[i].pre || gs[i].act != PV_ACT_NONE || gs[i].K < 1) return PV_EINVAL;
    w.g[i] = gs[i];
    tiles += ((gs[i].M + 15) / 16) * ((gs[i].N + 15) / 16);
    w.tile_end[i] = tiles;
  }
  for (int i = n; i < 4; ++i) w.tile_end[i] = tiles;
  w.n = n;
  int guests = 0;
  if (adam) {
    // every output must be a dense block of the flat gradient buffer (the caller guarantees it: nn.Linear weights/biases)
    for (int i = 0; i < n; ++i) {
      const int64_t c0 = gs[i].C - adam->g;
      if (c0 < 0 || c0 + (int64_t)gs[i].M * gs[i].N > adam->n || gs[i].ldc != gs[i].N) return PV_EINVAL;
      w.rng_lo[2 * i] = c0; w.rng_hi[2 * i] = c0 + (int64_t)gs[i].M * gs[i].N;
      if (gs[i].rowsumA) {
        const int64_t r0 = gs[i].rowsumA - adam->g;
        if (r0 < 0 || r0 + gs[i].M > adam->n) return PV_EINVAL;
        w.rng_lo[2 * i + 1] = r0; w.rng_hi[2 * i + 1] = r0 + gs[i].M;
      }
    }
    w.adam_on = 1; w.ad = *adam;
    guests = (int)((adam->n + 1023) / 1024);
    if (guests > 256) guests = 256;
    if (guests < 1) guests = 1;
  }
  if (fin && fin->scalars) {
    w.fin_llb = fin->llb; w.fin_B = fin->B; w.fin_scalars = fin->scalars; w.fin_kl_part = fin->kl_part;
    w.fin_n_part = fin->n_part; w.fin_beta = fin->beta;
    guests += 1;
  }
  *guests_out = guests;
  return tiles;
}

int pv_wgrad_small(const PvGemm* gs, int n, hipStream_t s, const PvAdamFuse* adam, const PvFinishArgs* fin) {
  PvWgradSmall w;
  int guests = 0;
  const int tiles = pv_wgrad_small_fill(w, gs, n, adam, fin, &guests);
  if (tiles < 0) return tiles;
  hipLaunchKernelGGL(pv_wgrad_small_kernel, dim3(tiles + guests), dim3(64 * WG_WAVES), 0, s, w);
  PV_LAUNCH_CHECK();
  return 0;
}
