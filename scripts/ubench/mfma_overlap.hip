// Microbenchmark (round 3): how many independent single-issue instructions hide under one matrix instruction, by MFMA
// shape (16x16x32: 4 passes, 32x32x16: 8 passes; same FLOP rate), filler type and waves per SIMD.  Order is pinned with
// scheduling barriers: MFMA, then F fillers, repeated.  Cycles are s_memtime ticks of wave 0 / workgroup 0 (shader clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// KIND: 0 v_fma_f32, 1 v_exp_f32, 2 ds_read_b128, 3 v_cvt_pk_bf16_f32, 4 v_pk_mul_f32
template <int SHAPE, int F, int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
  __shared__ f32x4 lds[1024];
  lds[threadIdx.x] = f32x4{1, 2, 3, 4};
  lds[threadIdx.x + 512] = f32x4{1, 2, 3, 4};
  __syncthreads();
  constexpr int NACC = SHAPE == 0 ? 8 : 4;
  f32x4 a4[8];
  f32x16 a16[4];
  for (int i = 0; i < 8; ++i) a4[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) a16[i][j] = 0.0f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  float m1 = 1.0001f + 1e-9f * threadIdx.x, m2 = 0.5f;
  asm volatile("" : "+v"(m1), "+v"(m2));
  f32x4 ld = {0, 0, 0, 0};
  const f32x4* lp = lds + (threadIdx.x & 511);
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
          if (KIND == 0) v[s] = __builtin_fmaf(v[s], m1, m2);
          else if (KIND == 1) v[s] = __builtin_amdgcn_exp2f(v[s]);
          else if (KIND == 2) { f32x4 t = lp[(s & 1) * 512]; asm volatile("" :: "v"(t)); }
          else if (KIND == 3) {
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            bf2 h; h[0] = (__bf16)v[s]; h[1] = (__bf16)v[(s + 1) & 15];
            v[s] = __builtin_bit_cast(float, h);
          } else {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 t = {v[s & 14], v[(s & 14) + 1]};
            t = t * f2{m1, m1};
            v[s & 14] = t[0]; v[(s & 14) + 1] = t[1];
          }
        }
        FENCE();
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = ld[0];
  for (int i = 0; i < 8; ++i) s += a4[i][0] + a4[i][3];
  for (int i = 0; i < 4; ++i) s += a16[i][0] + a16[i][15];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int SHAPE, int F, int KIND>
void run(float* d, long long* dc) {
  static const char* kn[] = {"v_fma_f32", "v_exp_f32", "ds_read_b128", "v_cvt_pk_bf16", "v_pk_mul_f32"};
  constexpr int NACC = SHAPE == 0 ? 8 : 4;
  const int iters = 1000;
  for (int threads = 256; threads <= 512; threads += 256) {
    hipLaunchKernelGGL((k<SHAPE, F, KIND>), dim3(256), dim3(threads), 0, 0, d, 10, dc);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<SHAPE, F, KIND>), dim3(256), dim3(threads), 0, 0, d, iters, dc);
    (void)hipDeviceSynchronize();
    long long h = 0;
    (void)hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / (iters * 4.0 * NACC);                 // cycles per (MFMA + F fillers) of ONE wave
    const double flop = SHAPE == 0 ? 16384.0 : 32768.0;
    const int wps = threads / 256;
    // per-SIMD cycles per 16 KFLOP of matrix work (16x16x32 = 1 unit; 32x32x16 = 2 units), both waves counted
    printf("%-9s F=%d %-14s waves/SIMD=%d  %.1f cyc per MFMA-group per wave;  %.1f cyc/SIMD per 16x16x32-equivalent (floor 16)\n",
           SHAPE == 0 ? "16x16x32" : "32x32x16", F, kn[KIND], wps, per, per / wps / (flop / 16384.0));
  }
}

template <int KIND>
void sweep(float* d, long long* dc) {
  run<0, 0, KIND>(d, dc); run<0, 1, KIND>(d, dc); run<0, 2, KIND>(d, dc); run<0, 3, KIND>(d, dc); run<0, 4, KIND>(d, dc);
  run<0, 6, KIND>(d, dc); run<0, 8, KIND>(d, dc);
  run<1, 0, KIND>(d, dc); run<1, 2, KIND>(d, dc); run<1, 4, KIND>(d, dc); run<1, 6, KIND>(d, dc); run<1, 8, KIND>(d, dc);
  run<1, 12, KIND>(d, dc); run<1, 16, KIND>(d, dc);
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  long long* dc; (void)hipMalloc(&dc, 64);
  sweep<0>(d, dc);
  sweep<1>(d, dc);
  sweep<2>(d, dc);
  sweep<3>(d, dc);
  sweep<4>(d, dc);
  return 0;
}
