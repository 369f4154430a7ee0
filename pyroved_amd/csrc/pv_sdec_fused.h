// pv_sdec_fused.h — the fused persistent spatial-decoder forward+backward kernel (pv_sdec_fused.hip).
#pragma once
#include "pv_common.h"

#define FD_H 128               // hidden width the kernel is specialised for (hidden_dim_d = [128, 128])
#define FD_UNIT 16             // rows per wave tile (one 16x16x4 MFMA column block)
#define FD_WAVES 8             // waves per workgroup (2 per SIMD)
#define FD_REC (2 * FD_H * FD_H + 14 * FD_H)  // floats of one per-workgroup partial-gradient record:
// [dW1 HxH | dW2 HxH | db1 H | db2 H | dWc0 H | dWc1 H | dwo H | dbo (1, padded to H) | 8 spare slots x H]
// (the spare slots are summed into dwo by the reduction when it is called with dwo_slots = 1; both kernels now
//  sum their d(wo) in LDS and leave the slots unused)
#define FB_WIMG_BYTES (4 * FD_H * FD_H * 2)    // bf16x3 kernel: global copy of its four LDS weight images

struct PvFused {
  const float* x;        // (M) observations, M = B*N rows (b, n)
  const float* grid;     // (N, cd)
  const float* tp;       // (B, 8) cos, sin, scale, tx, ty
  const float* hz;       // (B, H) fc_latent(z)
  const float *Wc, *bc;  // coord_latent.fc_coord (H, cd), (H)
  const float *W1, *b1;  // decoder.fc_layers.0 (H, H), (H)
  const float *W2, *b2;  // decoder.fc_layers.2
  const float *wo, *bo;  // decoder.out (1, H), (1)
  float* llrow;          // (M) log-likelihood per row
  float* loc;            // (M) decoder output or null
  float* rowtp;          // (4, M) per-row d(phi), d(scale), d(tx), d(ty)
  float* part_hz;        // (B * kmax, H) partial sums of dL/d(hz), zero-filled by the caller
  float* part;           // (G, FD_REC) per-workgroup partial gradients
  void* wimg;            // bf16x3 kernel only: FB_WIMG_BYTES of pre-split weight images (pv_sdec_fused_bf16_prep)
  int64_t M;             // rows
  int64_t units;         // M / FD_UNIT
  int N, cd, B, lik, sigmoid_out, kmax;
  int ablate;            // profiling only (env PV_FD_ABLATE): 1 skip wgrad exchanges, 2 skip coord-layer exchange,
                         // 4 skip dgrad, 8 skip d(wo) reduction  -> wrong gradients, used to price the phases
  float sig;
};

// true when the plan's architecture is the one the fused kernel is specialised for
bool pv_sdec_fused_supported(const pv_ivae_plan* p);
// workgroups the kernel runs with (<= number of CUs, <= units)
int pv_sdec_fused_grid(int64_t units);
// max workgroups that can touch one sample's rows
int pv_sdec_fused_kmax(int n_pix, int64_t units, int grid);
// launches the kernel (grads = false: forward + likelihood only)
int pv_sdec_fused_launch(const PvFused& f, int grid, bool grads, hipStream_t s);
// same interface, bf16 split-precision ("bf16x3") matrix math (pv_sdec_fused_bf16.hip); _prep must run first on the
// same stream, once per parameter state: it writes f.wimg and, with grads, zero-fills f.part_hz
int pv_sdec_fused_bf16_prep(const PvFused& f, bool grads, hipStream_t s);
int pv_sdec_fused_bf16_launch(const PvFused& f, int grid, bool grads, hipStream_t s);
// sums the per-workgroup records (ascending workgroup order) into the flat gradient buffer
struct PvFusedOffsets { int64_t W1, b1, W2, b2, Wc, wo, bo; };
int pv_sdec_fused_reduce(const float* part, int grid, float* G, const PvFusedOffsets& o, int cd, int dwo_slots,
                         hipStream_t s);
