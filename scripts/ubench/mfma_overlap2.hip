// Microbenchmark (round 3, second pass): which instruction classes issue UNDER a matrix instruction on gfx950.
// Every filler is inline asm (no SLP packing, exact opcode); order pinned: MFMA, F fillers, MFMA, ...
// Reported: shader-clock ticks per (MFMA + F fillers), from the MAX end - MIN start over all waves of workgroup 0
// (with two waves per SIMD the older wave wins arbitration, so one wave's own duration says nothing), and the
// event-timed figure at an assumed 2.4 GHz beside it.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

enum { K_ADD2 = 0, K_FMA3, K_FMA1, K_ADD1, K_EXP, K_RCP, K_CVT, K_PKMUL, K_PKFMA, K_AND, K_PERM, K_DSR128, K_DSTR, K_DSW64,
       K_SALU, K_MOV, K_N };
static const char* kname[] = {"v_add_f32 v,v,v", "v_fma_f32 v,v,v,v", "v_fma_f32 v,v,-2,1", "v_add_f32 v,1.0,v", "v_exp_f32",
                              "v_rcp_f32", "v_cvt_pk_bf16_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_and_b32", "v_perm_b32",
                              "ds_read_b128", "ds_read_b64_tr_b16", "ds_write_b64", "s_add_u32", "v_mov_b32"};

template <int KIND>
__device__ __forceinline__ void filler(float& a, float& b, float& c, float (&p)[2], f32x4& ld, unsigned lp, int& sa) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  if (KIND == K_ADD2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  else if (KIND == K_FMA3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  else if (KIND == K_FMA1) asm volatile("v_fma_f32 %0, %0, -2.0, 1.0" : "+v"(a));
  else if (KIND == K_ADD1) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a));
  else if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
  else if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
  else if (KIND == K_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  else if (KIND == K_PKMUL) { f2 t = {p[0], p[1]}; f2 u = {b, c}; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(t) : "v"(u)); p[0] = t[0]; p[1] = t[1]; }
  else if (KIND == K_PKFMA) { f2 t = {p[0], p[1]}; f2 u = {b, c}; asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(t) : "v"(u)); p[0] = t[0]; p[1] = t[1]; }
  else if (KIND == K_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
  else if (KIND == K_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  else if (KIND == K_DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(lp));
  else if (KIND == K_DSTR) { f2 t; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(t) : "v"(lp)); ld[0] = t[0]; ld[1] = t[1]; }
  else if (KIND == K_DSW64) { f2 t = {b, c}; asm volatile("ds_write_b64 %0, %1" :: "v"(lp), "v"(t)); }
  else if (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sa));
  else if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
}

template <int SHAPE, int F, int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
  __shared__ f32x4 lds[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = f32x4{1, 2, 3, 4};
  __syncthreads();
  constexpr int NACC = SHAPE == 0 ? 8 : 4;
  f32x4 a4[8];
  f32x16 a16[4];
  for (int i = 0; i < 8; ++i) a4[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) a16[i][j] = 0.0f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  float v[16], pk[8][2];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  for (int i = 0; i < 8; ++i) { pk[i][0] = 0.5f + i; pk[i][1] = 0.25f + i; }
  float m1 = 1.0001f + 1e-9f * threadIdx.x, m2 = 0.5f;
  asm volatile("" : "+v"(m1), "+v"(m2));
  f32x4 ld = {0, 0, 0, 0};
  const unsigned lp = (unsigned)(size_t)(lds + (threadIdx.x & 511)) ;
  int sa = 0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (SHAPE == 0) a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a4[i], 0, 0, 0);
        else a16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a16[i], 0, 0, 0);
        FENCE();
#pragma unroll
        for (int j = 0; j < F; ++j) {
          const int s = (i * F + j) & 15;
          filler<KIND>(v[s], m1, m2, pk[s & 7], ld, lp + 16 * 512 * (s & 3), sa);
        }
        FENCE();
      }
    }
    if (KIND == K_DSR128 || KIND == K_DSTR || KIND == K_DSW64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = ld[0] + ld[1] + (float)sa;
  for (int i = 0; i < 8; ++i) s += a4[i][0] + a4[i][3] + pk[i][0] + pk[i][1];
  for (int i = 0; i < 4; ++i) s += a16[i][0] + a16[i][15];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int SHAPE, int F, int KIND>
void run(float* d, long long* dc) {
  constexpr int NACC = SHAPE == 0 ? 8 : 4;
  const int iters = 500;
  double res[2][2];
  for (int threads = 256; threads <= 512; threads += 256) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, F, KIND>), dim3(256), dim3(threads), 0, 0, d, 10, dc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, F, KIND>), dim3(256), dim3(threads), 0, 0, d, iters, dc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[16];
    (void)hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
    long long lo = h[0], hi = h[1];
    for (int w = 0; w < threads / 64; ++w) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
    const int wps = threads / 256;
    const double groups = (double)iters * 4.0 * NACC * wps;             // MFMA groups issued per SIMD
    res[wps - 1][0] = (double)(hi - lo) / groups;
    res[wps - 1][1] = (ms - 0.004) * 1e-3 * 2.4e9 / groups;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  const double base = SHAPE == 0 ? 16.0 : 32.0;
  printf("%-9s F=%-2d %-20s | 1 wave/SIMD: %6.1f ticks (%6.1f ev)  -> %+5.1f per filler | 2 waves/SIMD: %6.1f ticks (%6.1f ev) per MFMA-group per SIMD -> %+5.1f per filler\n",
         SHAPE == 0 ? "16x16x32" : "32x32x16", F, kname[KIND], res[0][0], res[0][1], F ? (res[0][0] - base) / F : 0.0, res[1][0],
         res[1][1], F ? (res[1][0] - base) / F : 0.0);
}

template <int KIND>
void sweep(float* d, long long* dc) {
  run<0, 1, KIND>(d, dc); run<0, 2, KIND>(d, dc); run<0, 4, KIND>(d, dc); run<0, 8, KIND>(d, dc);
  run<1, 2, KIND>(d, dc); run<1, 4, KIND>(d, dc); run<1, 8, KIND>(d, dc); run<1, 16, KIND>(d, dc);
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  long long* dc; (void)hipMalloc(&dc, 256);
  run<0, 0, 0>(d, dc);
  run<1, 0, 0>(d, dc);
  sweep<K_ADD2>(d, dc); sweep<K_FMA3>(d, dc); sweep<K_FMA1>(d, dc); sweep<K_ADD1>(d, dc); sweep<K_EXP>(d, dc);
  sweep<K_RCP>(d, dc); sweep<K_CVT>(d, dc); sweep<K_PKMUL>(d, dc); sweep<K_PKFMA>(d, dc); sweep<K_AND>(d, dc);
  sweep<K_PERM>(d, dc); sweep<K_DSR128>(d, dc); sweep<K_DSTR>(d, dc); sweep<K_DSW64>(d, dc); sweep<K_SALU>(d, dc);
  sweep<K_MOV>(d, dc);
  return 0;
}
