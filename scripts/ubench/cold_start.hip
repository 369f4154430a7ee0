// What does the first load of a launch cost right after the launch that wrote the data?  Kernel A (256 workgroups x 512 threads, like the
// decoder) writes `n` floats; kernel B (256 x 256) follows on the same stream: thread 0 of every workgroup times (s_memtime) a first load
// of A's output, a second load of another line of the same 4 KB, a third from another 2 MB region, a fourth far away in a buffer nobody
// touched, and a repeat of the first.  hipcc --offload-arch=gfx950 -O3 cold_start.hip -o /tmp/cold && /tmp/cold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void writer(float* a, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a[i] = (float)i;
}
__device__ __forceinline__ long long tick() { return (long long)__builtin_readcyclecounter(); }
__global__ void reader(const float* a, const float* cold, long n, long long* out, float* sink) {
  if (threadIdx.x != 0) return;
  const long base = (long)blockIdx.x * (n / gridDim.x);
  long long t[7];
  float acc = 0.0f;
  t[0] = tick();
  float v = __builtin_nontemporal_load(a + base); acc += v; asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  t[1] = tick();
  v = a[base + 512]; acc += v; asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // same 4 KB, another line
  t[2] = tick();
  v = a[(base + (n >> 1)) % n]; acc += v; asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // another region of A's output
  t[3] = tick();
  v = cold[(long)blockIdx.x * 1048576]; acc += v; asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // untouched buffer, 4 MB apart
  t[4] = tick();
  v = a[base + 32]; acc += v; asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the first line again
  t[5] = tick();
  for (int k = 0; k < 5; ++k) out[blockIdx.x * 8 + k] = t[k + 1] - t[k];
  sink[blockIdx.x] = acc;
}
int main() {
  const long n = 1 << 20;                    // 4 MB, like the decoder's per-row outputs
  float *a, *cold, *sink; long long* out;
  CHK(hipMalloc(&a, n * 4)); CHK(hipMalloc(&cold, (size_t)1 << 30)); CHK(hipMalloc(&sink, 4096)); CHK(hipMalloc(&out, 256 * 8 * 8));
  CHK(hipMemset(cold, 0, (size_t)1 << 30));
  long long h[256 * 8];
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL(writer, dim3(256), dim3(512), 0, 0, a, n);
    hipLaunchKernelGGL(reader, dim3(256), dim3(256), 0, 0, a, cold, n, out, sink);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    double s[5] = {0, 0, 0, 0, 0}; long long mx[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < 256; ++b) for (int k = 0; k < 5; ++k) { s[k] += h[b * 8 + k]; if (h[b * 8 + k] > mx[k]) mx[k] = h[b * 8 + k]; }
    printf("rep %d: mean / max cycles over 256 workgroups: first load %.0f / %lld, same 4 KB %.0f / %lld, other region %.0f / %lld, untouched buffer %.0f / %lld, first line again %.0f / %lld\n",
           rep, s[0] / 256, mx[0], s[1] / 256, mx[1], s[2] / 256, mx[2], s[3] / 256, mx[3], s[4] / 256, mx[4]);
  }
  return 0;
}
