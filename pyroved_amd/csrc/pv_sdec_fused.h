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
#define PV_RS_W 8               // floats per slot of PvFused::part_rs (5 used)
#define FB_WIMG_BYTES (4 * FD_H * FD_H * 2 + 256)   // split-precision kernels: global copy of the four LDS weight images (+ the fp16 modes' scales)

struct PvFused {
  const float* x;        // (M) observations, M = B*N rows (b, n)
  const float* grid;     // (N, cd)
  const float* tp;       // (B, 8) cos, sin, scale, tx, ty
  const float* hz;       // (B, H) fc_latent(z), multiplied by hz_scale when that is non-zero
  float hz_scale;        // what the producer of hz already multiplied it by (pv_sdec_fused_w8.hip wants 2 log2(e) * hz)
  const float *Wc, *bc;  // coord_latent.fc_coord (H, cd), (H)
  const float *W1, *b1;  // decoder.fc_layers.0 (H, H), (H)
  const float *W2, *b2;  // decoder.fc_layers.2
  const float *wo, *bo;  // decoder.out (1, H), (1)
  float* llrow;          // (M) log-likelihood per row (null: not wanted — forward-only decode)
  float* loc;            // (M) decoder output or null
  float* rowtp;          // (4, M) per-row d(phi), d(scale), d(tx), d(ty)
  float* part_hz;        // (B * kmax, H) partial sums of dL/d(hz), zero-filled by the caller
  float* part_rs;        // (round 6) (B * kmax, PV_RS_W) or null.  Not null (training launches of the kernels that support it):
                         //   instead of writing llrow / rowtp per ROW the kernel publishes, in the slot of part_hz's scheme, the
                         //   sums over a wave's rows of a sample of {ll, d(phi), d(scale), d(tx), d(ty)} — pv_latent_bwd_reduce then
                         //   adds kmax slots per sample where it loaded and block-reduced 5 x N rows; zero-filled with part_hz
                         //   (it FOLLOWS part_hz's B * kmax * H floats in memory)
  float* part;           // (G, FD_REC) per-workgroup partial gradients
  float* dhz_out;        // (round 6) (B, H) or null.  Not null (the launch that hosts the guide: one image per workgroup): the workgroup
                         //   sums its waves' dL/d(hz) partials itself and writes the image's dL/d(hz) here — pv_latent_bwd_reduce then
                         //   reads it instead of adding kmax slots of part_hz (PvLatentBwd::dhz_ready)
  float* dzc_out;        // (round 6) (B, lat_in) or null, with dhz_out: ... and the image's dL/dz = dL/d(hz) Wz (PvLatentBwd::dzc_in)
  const float* Wz;       //   coord_latent.fc_latent.weight (H, lat_in) and lat_in, for dzc_out (the 4-wave kernels; the hosting 8-wave
  int lat_in;            //   launch has them in PvEncFold).  dhz_out / dzc_out are only set when every workgroup's unit range is exactly
                         //   ONE sample (grid == B, units == B * N / 16, x_units == 0): the caller's promise, re-checked by the kernel
  const float* sw;       // per-sample weight of dL/dlogit (jiVAE: alpha[b][k] of sample (k, b)); null: 1
  int64_t x_units;       // > 0: the observations repeat every x_units units (jiVAE: B*N/16; x is (B, N)); 0: x is (M)
  void* wimg;            // bf16x3 kernel only: FB_WIMG_BYTES of pre-split weight images (pv_sdec_fused_bf16_prep)
  void* park;            // 8-wave split-precision kernel only: pv_sdec_fused_w8x3_park_bytes(grid) of per-wave parking slots
  int64_t M;             // rows
  int64_t units;         // M / FD_UNIT
  int N, cd, B, lik, sigmoid_out, kmax;
  int ablate;            // profiling only (env PV_FD_ABLATE): 1 skip wgrad exchanges, 2 skip coord-layer exchange,
                         // 4 skip dgrad, 8 skip d(wo) reduction  -> wrong gradients, used to price the phases
  int qswap;             // the weight images are in the q-swapped column order (pv_fb_layout.h; set by the 4-wave kernel's launcher)
  float sig;
  int sel;               // pv_ivae_plan.dec_kernel: which build of the decoder kernel runs (0: by problem size; pv_sdec_fused_bf16.hip)
  int dl_exp;            // fp16 modes (pv_sdec_fused_bf16.hip): exponent bias of the per-row dL/dlogit factor folded into the
                         // staged activations: 2^dl_exp * |dL/dlogit| should sit around 1 .. 2^8 (Bernoulli 4; Gaussian: -log2(1 / sig^2))
};

// ---- the guide folded into the decoder launch (round 5; pv_sdec_fused_w8.hip) -------------------------------------------------
// When a workgroup's unit range is a whole number of images (batch a multiple of the grid: BASELINE's batch 256 on 256 CUs is one
// image per workgroup) the workgroup can run its images' guide ITSELF in the launch's prologue — fcEncoderNet.forward (nets/fc.py:
// 51-61) as fp32 matrix-vector products straight from the L2-resident weights, the reparameterised sample and its sampled-KL terms
// (models/ivae.py:204-221), _split_latent (models/base.py:97-119) and fc_latent(z) — and convert the decoder's two hidden weight
// matrices into its own LDS images: the encoder launch (18.7 us of dependent latency for 51 MFLOP), its hand-off protocol and the
// global weight-image copy all disappear from the step.  Everything the backward launches read is written exactly where the
// encoder launch would have put it.
struct PvEncFold {
  const float* params;
  pv_layer enc0, enc1, head;       // two hidden layers (widths 128) + the merged [mu | softplus input] head
  const float* x; int64_t ldx;     // (B, ldx) observations (the encoder's input: x.view(B, N))
  const float* eps;                // (B, z_dim)
  float* eact0; float* eact1;      // hidden activations (B, 128)
  float* head_out;                 // (B, 2 z_dim)
  float* z; float* z_scale; float* z_loc_out; float* z_scale_out;
  float* tp;                       // (B, 8) cos, sin, scale, tx, ty
  float* kl_part;                  // (B, 2): beta * log p(z_b), beta * log q(z_b | x_b)
  float* hz; const float* Wz;      // (B, 128) fc_latent(z content) * C ; fc_latent.weight (128, lat_in)
  int lat_in, z_dim, coord_dim, has_r, has_t, has_s;
  float tp0, tp1, sc_prior, beta;
  int img_per_wg;                  // images per workgroup (B / grid): 1
  // (round 6, third cut) chain != 0: the workgroup also runs its image's latent backward (models/ivae.py's guide differentiated:
  // head backward from dL/dz and the sampled-KL terms) and the encoder's input-gradient chain in the launch's EPILOGUE — it holds
  // the image's row sums, dL/d(hz) and dL/dz there already (PvFused::part_rs / dhz_out / dzc_out) — so that the step's closing launch
  // has no per-sample work left (pv_elementwise.hip: pv_rec_wgrad_kernel).  Needs head.out_dim <= 16.
  int chain;
  float* dhead; int ldh;           // (B, ldh) dL/d[mu | softplus input]
  float* edp0; float* edp1;        // (B, 128) dL/dpre of the two hidden layers
  float* llb;                      // (B) the image's log-likelihood
  // (round 6, fourth cut) coop != 0 (with chain, grid == 256): the guide's first layer is shared among the 32 workgroups of a group
  // (pv_sdec_fused_w8.hip).  coop_flags: grid + 1 words of any content — [g] workgroup g's hand-off tag, [grid] a word that the
  // step's closing launch increments (PvWgradSmall::tick), so that no launch's tag repeats an earlier one's
  int coop;
  unsigned* coop_flags;
};
// A kernel (PvFused f, PvEncFold e) whose EPILOGUE alone needs `e` reads it there through this pointer into the kernarg segment, made
// opaque at the point of use: named directly, the compiler fetches the fields at kernel entry and carries them — spilled — through
// the tile loop (pv_sdec_fused_bf16.hip: 51 -> 92 spilled SGPRs and +3 us on the launch; with the pointer: 42).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) PvEncFold* PvEncFoldArg;
__device__ __forceinline__ PvEncFoldArg pv_kernarg_fold() {
  const __attribute__((address_space(4))) char* k = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  PvEncFoldArg p = (PvEncFoldArg)(k + ((sizeof(PvFused) + alignof(PvEncFold) - 1) / alignof(PvEncFold)) * alignof(PvEncFold));
  asm volatile("" : "+s"(p));
  return p;
}
#else
typedef const PvEncFold* PvEncFoldArg;
__device__ inline PvEncFoldArg pv_kernarg_fold() { return nullptr; }      // (host pass of the same source: never called)
#endif
// the same guide as a launch of its own, one workgroup per image (pv_guide_img.hip; round 6): for the plans the fold cannot take.
// prep != null: guest workgroups write the decoder's weight images / clear its dL/d(hz) slots; hz_mul: what hz leaves multiplied by
#define PV_GUIDE_IMG_MAX_BATCH 384     // (measured, scripts/ab_guide_img.py: -7..-10 % of the step at batch 64, -6..-9 % at 128, -2..-4 % at 256, +-0 at 512)
struct PvFbPrep;
bool pv_guide_img_ok(const PvEncFold& e, int B);
int pv_guide_img_launch(const PvEncFold& e, const PvFbPrep* prep, float hz_mul, int B, hipStream_t s);
// whether the launch of (f, grads) can host the guide of `p`'s encoder (the caller checks the encoder's architecture)
bool pv_sdec_fused_w8_fold_ok(const PvFused& f, int grid);
// ... and whether the launcher will pick that kernel for (f, x3) at all
bool pv_sdec_fused_fold_ok(const PvFused& f, int grid, bool x3);

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
// x3: split precision (pv_sdec_fused_bf16.hip) or plain bf16 operands (pv_sdec_fused_w8.hip, whose images are
// pre-scaled by 2 log2(e), once a workgroup has enough units to fill its 8-wave tiles; the 4-wave plain-bf16 kernel
// for smaller problems; PV_W8=0 / 1 in the environment forces one)
int pv_sdec_fused_bf16_prep(const PvFused& f, bool grads, bool x3, hipStream_t s);
// ... or, as arguments for a kernel that hosts the preparation (pv_fb_layout.h: pv_fb_prep)
struct PvFbPrep;
PvFbPrep pv_sdec_fused_bf16_prep_args(const PvFused& f, bool grads, bool x3);
bool pv_sdec_fused_w8_qswap();          // the 8-wave plain-bf16 kernel's images are in the q-swapped column order (pv_fb_layout.h)
int pv_sdec_fused_bf16_launch(const PvFused& f, int grid, bool grads, bool x3, hipStream_t s, const PvEncFold* fold = nullptr,
                              const PvEncFold* chain = nullptr);      // chain: the 4-wave kernels' own-sample epilogue (PvEncFold::chain)
int pv_sdec_fused_w8_launch(const PvFused& f, int grid, bool grads, hipStream_t s, const PvEncFold* fold = nullptr);
// the 8-wave split-precision kernel (pv_sdec_fused_w8x3.hip; images pre-scaled by 2 log2(e) like the plain 8-wave kernel's)
int pv_sdec_fused_w8x3_launch(const PvFused& f, int grid, bool grads, hipStream_t s, int waves);   // waves: 8 or 4
int64_t pv_sdec_fused_w8x3_park_bytes(int grid);
// the 8-wave fp16 kernel of the fp32-class path (pv_sdec_fused_w8h.hip; training launches; prep mode 2); ds: dL/dpre split
int pv_sdec_fused_w8h_launch(const PvFused& f, int grid, bool ds, hipStream_t s);
// bytes of PvFused::park the launch of (x3, units) needs (0: the kernel that will run has no parking slots)
int64_t pv_sdec_fused_bf16_park_bytes(bool x3, int64_t units, int grid, int sel);
// waves per workgroup that publish a dL/d(hz) slot (sizes part_hz)
int pv_sdec_fused_bf16_waves(bool x3, int64_t units, int sel);
// the record format (PV_REC_*) the training launch of (x3, units, sel) writes
int pv_sdec_fused_bf16_record_fmt(bool x3, int64_t units, int sel);
// whether `sel` (pv_ivae_plan.dec_kernel) names a decoder-kernel build this library contains for the fused mode
bool pv_sdec_fused_sel_valid(int fused, int sel);
// sums the per-workgroup records (ascending workgroup order) into the flat gradient buffer
struct PvFusedOffsets { int64_t W1, b1, W2, b2, Wc, wo, bo; };
int pv_sdec_fused_reduce(const float* part, int grid, float* G, const PvFusedOffsets& o, int cd, int dwo_slots,
                         hipStream_t s);

// ---- the reduction's workgroup body (256 threads), shared with the launch that also hosts pv_latent_bwd -------
// 64 float4 outputs x 4 slices of the workgroup range per 256-thread block (a wave reads 1 KB per record, eight
// records in flight); slices combined in fixed order
#define PV_FUSED_REDUCE_BLOCKS (((2 * FD_H * FD_H + 5 * FD_H + 1 + 3) / 4 + 63) / 64)
// (round 6) Record FORMATS of the two HxH weight-gradient partials — a private contract between a decoder kernel's epilogue and
// this reduction; everything from float 2*H*H on (bias / coordinate / output-layer vectors) is the same in all of them:
//   PV_REC_ROWMAJOR  [m][row][col] fp32 (rounds 1-5; the f32-MFMA kernel and the experiments builds): a lane of the C/D layout
//                    holds ONE column of four rows, so every store instruction wrote 4 x 64-byte segments — 128 (64) scattered
//                    4-byte store instructions per wave, 6-12 k cycles of store issue per workgroup at the end of the launch;
//   PV_REC_LANE_F32  LANE-NATIVE fp32 (the 4-wave kernels of pv_sdec_fused_bf16.hip): chunk (16 bytes)
//                    [m][wave][s][kb][lane] = the lane's accumulator block accW_m[s][kb] as it sits in registers — rows
//                    16 (2 wave + s) + 4 q + i (i = 0..3), column 16 kb + r: ONE 1 KB-contiguous store instruction per block;
//   PV_REC_LANE_BF16 lane-native PACKED (the 8-wave throughput kernel, pv_sdec_fused_w8.hip): chunk [wave][s][o][lane] =
//                    {W1 rows (i0, i1), W1 rows (i2, i3), W2 rows (i0, i1), W2 rows (i2, i3)} as bf16 pairs — rows
//                    32 jp + 16 (s ^ kh) + 4 q + i, column 64 kh + 16 o + r (jp = wave >> 1, kh = wave & 1) — in the record's first H*H
//                    floats: half the bytes (the weights they update were rounded to bf16 in the forward anyway; the reader sums in
//                    fp32) and 8 store instructions per wave.
// The reduction reads 16 bytes per thread and record either way and writes each output element once; per element the records are
// summed in the same fixed order as before (four slices of ascending workgroups): the fp32 formats give the same bits.
#define PV_REC_ROWMAJOR 0
#define PV_REC_LANE_F32 1
#define PV_REC_LANE_BF16 2
#define PV_FUSED_REDUCE_VEC_BLOCKS (((5 * FD_H + 1 + 3) / 4 + 63) / 64)
#define PV_FUSED_REDUCE_MAT_BLOCKS_F32 ((2 * FD_H * FD_H / 4) / 64)                     // 128
#define PV_FUSED_REDUCE_MAT_BLOCKS_BF16 ((2 * (FD_H / 2) * FD_H / 4) / 64)              // 64
__host__ __device__ inline int pv_fused_reduce_mat_blocks(int fmt) {
  return fmt == PV_REC_LANE_BF16 ? PV_FUSED_REDUCE_MAT_BLOCKS_BF16 : PV_FUSED_REDUCE_MAT_BLOCKS_F32;
}
__host__ __device__ inline int pv_fused_reduce_blocks(int fmt) {
  return fmt == PV_REC_ROWMAJOR ? PV_FUSED_REDUCE_BLOCKS : pv_fused_reduce_mat_blocks(fmt) + PV_FUSED_REDUCE_VEC_BLOCKS;
}
// one block = 64 chunks = ONE accumulator block of one wave (thread c = lane c of that wave), four slices of the workgroup range
// a finished gradient element: into the flat gradient, or (ad: pv_ivae_step's optimizer riding in the reducing launch) straight
// through torch.optim.Adam's update of its parameter — the element's gradient slot is then left zeroed (pv_common.h: pv_adam_update)
#ifndef PV_RED_UNROLL
#define PV_RED_UNROLL 8          // records in flight per thread of a reducing block
#endif
struct PvRecAdam { PvAdamFuse a; int on; };       // (by value: the address of a kernel argument would put it in scratch memory)
__device__ __forceinline__ void pv_rec_out(float* __restrict__ Gr, int idx, float v, const PvRecAdam& ad) {
  if (ad.on) pv_adam_update(ad.a.p, ad.a.g, ad.a.m, ad.a.v, (Gr - ad.a.g) + idx, v, ad.a.b1, ad.a.b2, ad.a.eps, ad.a.step_size, ad.a.bc2_sqrt);
  else Gr[idx] = v;
}
template <int FMT>
__device__ __forceinline__ void pv_sdec_fused_reduce_block_lane(const float* __restrict__ part, int G_, float* __restrict__ Gr,
                                                                const PvFusedOffsets& o, int block, f32x4 (*sm)[64],
                                                                const PvRecAdam& ad = PvRecAdam{}) {
  const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int per = (G_ + 3) / 4;
  const int w0 = sl * per, w1 = min(G_, w0 + per);
  const float* pbase = part + 4 * (block * 64 + c);
  f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f}, b = {0.0f, 0.0f, 0.0f, 0.0f};
  if (FMT == PV_REC_LANE_BF16) {
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll PV_RED_UNROLL
    for (int w = w0; w < w1; ++w) {
      const u32x4_ v = *reinterpret_cast<const u32x4_*>(pbase + (int64_t)w * FD_REC);
      a[0] += __uint_as_float(v[0] << 16); a[1] += __uint_as_float(v[0] & 0xffff0000u);
      a[2] += __uint_as_float(v[1] << 16); a[3] += __uint_as_float(v[1] & 0xffff0000u);
      b[0] += __uint_as_float(v[2] << 16); b[1] += __uint_as_float(v[2] & 0xffff0000u);
      b[2] += __uint_as_float(v[3] << 16); b[3] += __uint_as_float(v[3] & 0xffff0000u);
    }
  } else {
#pragma unroll PV_RED_UNROLL
    for (int w = w0; w < w1; ++w) a += *reinterpret_cast<const f32x4*>(pbase + (int64_t)w * FD_REC);
  }
  sm[sl][c] = a;
  __syncthreads();
  const f32x4 ta = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
  f32x4 tb = b;
  if (FMT == PV_REC_LANE_BF16) {
    __syncthreads();
    sm[sl][c] = b;
    __syncthreads();
    tb = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
  }
  if (sl != 0) return;
  const int r = c & 15, q = c >> 4;
  if (FMT == PV_REC_LANE_BF16) {
    const int oo = block & 3, s_ = (block >> 2) & 1, wave = block >> 3, jp = wave >> 1, kh = wave & 1;
    const int row0 = 32 * jp + 16 * (s_ ^ kh) + 4 * q, col = 64 * kh + 16 * oo + r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pv_rec_out(Gr, o.W1 + (row0 + i) * FD_H + col, ta[i], ad);
      pv_rec_out(Gr, o.W2 + (row0 + i) * FD_H + col, tb[i], ad);
    }
  } else {
    const int kb = block & 7, s_ = (block >> 3) & 1, wave = (block >> 4) & 3, m = block >> 6;
    const int row0 = 16 * (2 * wave + s_) + 4 * q, col = 16 * kb + r;
    const int dst = m == 0 ? o.W1 : o.W2;
#pragma unroll
    for (int i = 0; i < 4; ++i) pv_rec_out(Gr, dst + (row0 + i) * FD_H + col, ta[i], ad);
  }
}
__device__ __forceinline__ void pv_sdec_fused_reduce_block(const float* __restrict__ part, int G_,
                                                           float* __restrict__ Gr, const PvFusedOffsets& o, int cd,
                                                           int dwo_slots, int block, f32x4 (*sm)[64], int fmt = PV_REC_ROWMAJOR,
                                                           const PvRecAdam& ad = PvRecAdam{}) {
  const int HH = FD_H * FD_H;
  const int total = 2 * HH + 5 * FD_H + 1;          // the record is padded well past this: whole float4s are readable
  const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
  if (fmt != PV_REC_ROWMAJOR && block < pv_fused_reduce_mat_blocks(fmt)) {
    if (fmt == PV_REC_LANE_BF16) pv_sdec_fused_reduce_block_lane<PV_REC_LANE_BF16>(part, G_, Gr, o, block, sm, ad);
    else pv_sdec_fused_reduce_block_lane<PV_REC_LANE_F32>(part, G_, Gr, o, block, sm, ad);
    return;
  }
  // (lane-native formats: the blocks behind the matrices' take the vectors, which sit at float 2*H*H in every format)
  const int e = fmt != PV_REC_ROWMAJOR ? 2 * HH + ((block - pv_fused_reduce_mat_blocks(fmt)) * 64 + c) * 4 : (block * 64 + c) * 4;
  const int per = (G_ + 3) / 4;
  const int w0 = sl * per, w1 = min(G_, w0 + per);
  f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
  if (e < total) {
    const bool slots = dwo_slots && e >= 2 * HH + 4 * FD_H && e < 2 * HH + 5 * FD_H;   // d(wo): 8 per-wave slots
    if (slots) {
      for (int w = w0; w < w1; ++w) {
        const float* p8 = part + (int64_t)w * FD_REC + 2 * HH + 6 * FD_H + (e - (2 * HH + 4 * FD_H));
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = 0; k < FD_WAVES; ++k) a += *reinterpret_cast<const f32x4*>(p8 + k * FD_H);
        v += a;
      }
    } else {
      const float* p = part + e;
#pragma unroll PV_RED_UNROLL
      for (int w = w0; w < w1; ++w) v += *reinterpret_cast<const f32x4*>(p + (int64_t)w * FD_REC);
    }
  }
  sm[sl][c] = v;
  __syncthreads();
  if (sl != 0 || e >= total) return;
  v = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ei = e + i;
    if (ei >= total) break;
    if (ei < HH) pv_rec_out(Gr, o.W1 + ei, v[i], ad);
    else if (ei < 2 * HH) pv_rec_out(Gr, o.W2 + (ei - HH), v[i], ad);
    else {
      const int k = ei - 2 * HH, seg = k / FD_H, j = k % FD_H;
      if (seg == 0) pv_rec_out(Gr, o.b1 + j, v[i], ad);
      else if (seg == 1) pv_rec_out(Gr, o.b2 + j, v[i], ad);
      else if (seg == 2) pv_rec_out(Gr, o.Wc + j * cd, v[i], ad);
      else if (seg == 3) { if (cd == 2) pv_rec_out(Gr, o.Wc + j * 2 + 1, v[i], ad); }
      else if (seg == 4) pv_rec_out(Gr, o.wo + j, v[i], ad);
      else pv_rec_out(Gr, o.bo, v[i], ad);
    }
  }
}


