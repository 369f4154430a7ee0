// Microbenchmark (round 3): v_mfma_f32_16x16x32_bf16 issue rate vs the number of independent accumulators a wave rotates
// through (1 = fully dependent chain), one and two waves per SIMD.  Event-timed, cycles at an assumed 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 48 / NACC; ++g)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(float* d) {
  const int iters = 2000;
  for (int threads = 256; threads <= 512; threads += 256) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC>), dim3(256), dim3(threads), 0, 0, d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC>), dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = ms * 1e-3 * 2.4e9 / ((double)iters * (48 / NACC) * NACC);
    printf("accumulators=%d waves/SIMD=%d: %.1f cycles per MFMA per wave, %.1f per MFMA per SIMD\n", NACC, threads / 256, per_wave,
           per_wave / (threads / 256));
  }
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  run<1>(d); run<2>(d); run<3>(d); run<4>(d); run<6>(d); run<8>(d);
  return 0;
}
