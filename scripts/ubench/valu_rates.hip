// Microbenchmark: issue cost per wave-instruction of the VALU classes the fused decoder kernel leans on, with one and
// two waves per SIMD, and of an MFMA stream next to a VALU-only partner wave (do the pipes overlap ACROSS waves?).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define N 32

// KIND: 0 v_exp_f32, 1 v_rcp_f32, 2 v_fma_f32, 3 v_pk_fma_f32, 4 v_cvt_pk_bf16_f32, 5 v_add_f32, 6 v_pk_mul_f32,
//       7 tanh chain staged (exp, add, rcp, fma per element), 8 v_lshlrev (int)
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  float v[N];
  for (int i = 0; i < N; ++i) v[i] = 0.001f * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
    } else if (KIND == 1) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_rcpf(v[i]);
    } else if (KIND == 2) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
    } else if (KIND == 3) {
#pragma unroll
      for (int i = 0; i < N; i += 2) {
        f32x2 t = {v[i], v[i + 1]};
        t = t * 1.0001f + 0.5f;
        v[i] = t[0]; v[i + 1] = t[1];
      }
    } else if (KIND == 4) {
#pragma unroll
      for (int i = 0; i < N; i += 2) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        bf2 h; h[0] = (__bf16)v[i]; h[1] = (__bf16)v[i + 1];
        v[i] = __builtin_bit_cast(float, h);
      }
    } else if (KIND == 5) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = v[i] + 0.5f;
    } else if (KIND == 6) {
#pragma unroll
      for (int i = 0; i < N; i += 2) {
        f32x2 t = {v[i], v[i + 1]};
        t = t * 1.0001f;
        v[i] = t[0]; v[i + 1] = t[1];
      }
    } else if (KIND == 7) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = v[i] + 1.0f;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_rcpf(v[i]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = 1.0f - 2.0f * v[i];
      __builtin_amdgcn_sched_barrier(0);
    } else if (KIND == 8) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[i]) << 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int i = 0; i < N; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves 0-3 (one per SIMD) run MFMAs, waves 4-7 run VALU (MODE 1), transcendentals (MODE 2) or nothing useful (MODE 0)
template <int MODE>
__global__ __launch_bounds__(512) void mix(float* out, int iters, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  float s = 0;
  long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = MFMA(a, b, acc[i]);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  } else if (MODE > 0) {
    float v[N];
    for (int i = 0; i < N; ++i) v[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = MODE == 1 ? __builtin_fmaf(v[i], 1.0001f, 0.5f) : __builtin_amdgcn_exp2f(v[i]);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int i = 0; i < N; ++i) s += v[i];
  }
  long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int threads, float* d, int per_iter) {
  const int iters = 2000, grid = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(threads), 0, 0, d, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc_simd = ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * (threads / 256));   // per instruction per SIMD (at 2.4 GHz)
  printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cyc per wave-instruction per SIMD (@2.4GHz)\n", name, threads / 256, ms, cyc_simd);
}

template <int MODE>
void runmix(const char* name, float* d, long long* dc) {
  const int iters = 2000;
  hipLaunchKernelGGL((mix<MODE>), dim3(256), dim3(512), 0, 0, d, 10, dc);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((mix<MODE>), dim3(256), dim3(512), 0, 0, d, iters, dc);
  (void)hipDeviceSynchronize();
  long long h[8];
  (void)hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-28s mfma wave: %.1f cyc/MFMA   partner wave: %.2f cyc per instruction\n", name, (double)h[0] / (iters * 32.0),
         MODE ? (double)h[4] / (iters * (double)N) : 0.0);
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  long long* dc; (void)hipMalloc(&dc, 64);
  for (int th = 256; th <= 512; th += 256) {
    run<0>("v_exp_f32", th, d, N);
    run<1>("v_rcp_f32", th, d, N);
    run<2>("v_fma_f32", th, d, N);
    run<3>("v_pk_fma_f32", th, d, N / 2);
    run<4>("v_cvt_pk_bf16_f32", th, d, N / 2);
    run<5>("v_add_f32", th, d, N);
    run<6>("v_pk_mul_f32", th, d, N / 2);
    run<7>("tanh x32 (4 instr/elem)", th, d, 4 * N);
    run<8>("v_lshlrev_b32", th, d, N);
  }
  runmix<0>("mfma alone", d, dc);
  runmix<1>("mfma + fma partner", d, dc);
  runmix<2>("mfma + exp partner", d, dc);
  return 0;
}
