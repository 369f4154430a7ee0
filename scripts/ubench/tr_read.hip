// Probe: semantics of ds_read_b64_tr_b16 (gfx950) — what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4_ lds_short4;

__global__ void k(int* out, int stride) {
  __shared__ __attribute__((aligned(16))) short sm[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) sm[i] = (short)i;     // value = its own element index
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, q = l >> 4;
  // lane i of 16-lane group q supplies &M[row = 4q + i/4][col = 4*(i%4)], row stride `stride` elements
  short* p = sm + (4 * q + i / 4) * stride + 4 * (i % 4);
  short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

int main() {
  int* d; (void)hipMalloc(&d, 64 * 4 * 4);
  int h[256];
  for (int stride : {16, 72}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d\n", stride);
    for (int l = 0; l < 64; l += 1) {
      if (l % 16 < 3 || l % 16 == 15) {
        printf(" lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
        printf("\n");
      }
    }
  }
  return 0;
}
