// Microbenchmark: v_mfma_f32_16x16x4_f32 issue rate under the fused kernel's structure.
// variants: waves per SIMD (1/2), accumulators in flight, VALU filler per MFMA, LDS operand reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NACC, int VALU, bool LDSOP>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  __shared__ float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = 0.001f * i;
  __syncthreads();
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float aa = a;
      if (LDSOP) aa = sm[(threadIdx.x * 4 + g * 64 + it) & 4095];
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = MFMA(aa, b, acc[i]);
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

template <int NACC, int VALU, bool LDSOP>
void run(const char* name, int threads, float* d) {
  const int iters = 2000, grid = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, VALU, LDSOP>), dim3(grid), dim3(threads), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, VALU, LDSOP>), dim3(grid), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)grid * (threads / 64) * iters * 16.0 * NACC;
  const double tf = mfmas * 2048.0 / (ms * 1e-3) / 1e12;
  printf("%-40s threads=%d  %.3f ms  %.1f TF (%.0f%% of 157.3)\n", name, threads, ms, tf, 100 * tf / 157.3);
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  run<4, 0, false>("acc4 valu0", 256, d);
  run<4, 0, false>("acc4 valu0", 512, d);
  run<8, 0, false>("acc8 valu0", 256, d);
  run<8, 0, false>("acc8 valu0", 512, d);
  run<2, 0, false>("acc2 valu0", 512, d);
  run<1, 0, false>("acc1 valu0", 512, d);
  run<8, 1, false>("acc8 valu1", 512, d);
  run<8, 2, false>("acc8 valu2", 512, d);
  run<8, 3, false>("acc8 valu3", 512, d);
  run<8, 4, false>("acc8 valu4", 512, d);
  run<8, 6, false>("acc8 valu6", 512, d);
  run<8, 3, false>("acc8 valu3", 256, d);
  run<8, 6, false>("acc8 valu6", 256, d);
  run<8, 0, true>("acc8 valu0 ldsop", 512, d);
  run<8, 3, true>("acc8 valu3 ldsop", 512, d);
  return 0;
}
