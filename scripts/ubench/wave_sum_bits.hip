#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include "../../pyroved_amd/csrc/pv_common.h"
__global__ void k(const float* in, float* o1, float* o2) {
  float v = in[threadIdx.x];
  o1[threadIdx.x] = pv_wave_sum(v);
  float w = v;
  for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
  o2[threadIdx.x] = w;
}
int main() {
  float h[64], a[64], b[64]; float *d, *x, *y;
  srand(1); for (int i = 0; i < 64; ++i) h[i] = (float)rand() / RAND_MAX * 1000.f - 300.f + 1e-3f * i;
  hipMalloc(&d, 256); hipMalloc(&x, 256); hipMalloc(&y, 256);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, x, y);
  hipMemcpy(a, x, 256, hipMemcpyDeviceToHost); hipMemcpy(b, y, 256, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) if (memcmp(&a[i], &b[i], 4)) ++bad;
  printf("mismatching lanes: %d (sum %.9g vs %.9g)\n", bad, a[0], b[0]);
  return bad;
}
