// LDS-DMA (global_load_lds_dwordx4) into LDS offsets above 64 KB: builtin and inline-asm (M0) forms.
// build: hipcc --offload-arch=gfx950 -O3 glds_hi.hip -o glds_hi ; run: ./glds_hi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* src, float* dst, int base) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // each wave copies 4 KB: 4 instructions of 1 KB
  for (int c = 0; c < 4; ++c) {
    const int off = (wave * 4 + c) * 1024;
    const char* g = reinterpret_cast<const char*>(src) + off + lane * 16;
    if (MODE == 0) {
      __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(sm + base + off), 16, 0, 0);
    } else {
      const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(sm + base + off));
      glds16_asm(g, l);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float* s = reinterpret_cast<const float*>(sm + base);
  for (int i = tid; i < 4096; i += 256) dst[i] = s[i];
}

int main() {
  const int n = 4096;
  std::vector<float> h(n), o(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *e;
  hipMalloc(&d, n * 4); hipMalloc(&e, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  const int lds = 160 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int mode = 0; mode < 2; ++mode)
    for (int base : {0, 32768, 65536, 69632, 100352, 139264}) {
      hipMemset(e, 0xff, n * 4);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), lds, 0, d, e, base);
      else hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), lds, 0, d, e, base);
      hipError_t err = hipDeviceSynchronize();
      hipMemcpy(o.data(), e, n * 4, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < n; ++i) bad += (o[i] != h[i]);
      printf("mode %d base %6d: %s (bad %d) err %d\n", mode, base, bad ? "MISMATCH" : "ok", bad, (int)err);
    }
  return 0;
}
