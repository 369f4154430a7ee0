// Microbenchmark: v_mfma_f32_16x16x32_bf16 under the fused kernel's structure: does VALU work overlap it?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

template <int NACC, int VALU>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = MFMA(a, b, acc[i]);
#pragma unroll
        for (int j = 0; j < VALU; ++j) v[(i + j) & 7] = v[(i + j) & 7] * 1.0001f + 0.5f;
      }
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int VALU>
void run(const char* name, int threads, float* d) {
  const int iters = 4000, grid = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, VALU>), dim3(grid), dim3(threads), 0, 0, d, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, VALU>), dim3(grid), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)grid * (threads / 64) * iters * 16.0 * NACC;
  const double tf = mfmas * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 16 * NACC * (threads / 256));
  printf("%-24s threads=%d  %.3f ms  %.0f TF (%.0f%% of 2500)  %.1f cyc/MFMA/SIMD\n", name, threads, ms, tf, 100 * tf / 2500, cyc);
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  run<8, 0>("acc8 valu0", 256, d);
  run<8, 0>("acc8 valu0", 512, d);
  run<4, 0>("acc4 valu0", 512, d);
  run<8, 1>("acc8 valu1", 512, d);
  run<8, 2>("acc8 valu2", 512, d);
  run<8, 3>("acc8 valu3", 512, d);
  run<8, 4>("acc8 valu4", 512, d);
  run<8, 6>("acc8 valu6", 512, d);
  run<8, 8>("acc8 valu8", 512, d);
  run<8, 1>("acc8 valu1", 256, d);
  run<8, 2>("acc8 valu2", 256, d);
  run<8, 3>("acc8 valu3", 256, d);
  run<8, 4>("acc8 valu4", 256, d);
  run<8, 6>("acc8 valu6", 256, d);
  run<8, 8>("acc8 valu8", 256, d);
  return 0;
}
