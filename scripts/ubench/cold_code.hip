// cold_code.hip — what does straight-line code cost the FIRST time a launch runs it?  Each workgroup (8 waves) runs a block of
// N independent 8-byte VALU instructions twice (a loop of two trips over the same code) and stamps each trip: trip 0 fetches
// the code through the instruction cache cold (every workgroup of the launch at once), trip 1 runs it warm.
//   hipcc --offload-arch=gfx950 -O3 cold_code.hip -o cold_code && ./cold_code
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define REP512(x) REP8(REP64(x))
template <int KB>   // KB * 1024 bytes of code = KB * 128 instructions
__global__ __launch_bounds__(512) void k(long long* out, float* sink) {
  float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;
  long long t[3];
  for (int trip = 0; trip < 2; ++trip) {
    t[trip] = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < KB / 4; ++i) {     // 512 instructions = 4 KB per REP512 (v_fmac_f32 with a 32-bit encoding would be 4 bytes: use VOP3)
      REP512(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    }
    asm volatile("" : "+v"(d));
  }
  t[2] = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[3 * blockIdx.x] = t[1] - t[0]; out[3 * blockIdx.x + 1] = t[2] - t[1]; }
  if (a == 12345.0f) sink[0] = a + d;
}
template <int KB> void run(long long* d, float* s) {
  hipLaunchKernelGGL(k<KB>, dim3(256), dim3(512), 0, 0, d, s);
  hipDeviceSynchronize();
  long long h[3 * 256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double c0 = 0, c1 = 0;
  for (int i = 0; i < 256; ++i) { c0 += h[3 * i]; c1 += h[3 * i + 1]; }
  printf("%3d KB of code (%5d instructions): cold trip %8.0f cycles, warm trip %8.0f cycles  (mean over 256 workgroups; per instruction %.2f / %.2f)\n",
         KB, KB * 128, c0 / 256, c1 / 256, c0 / 256 / (KB * 128), c1 / 256 / (KB * 128));
}
int main() {
  long long* d; float* s;
  hipMalloc(&d, 3 * 256 * sizeof(long long)); hipMalloc(&s, 4);
  for (int rep = 0; rep < 2; ++rep) { run<4>(d, s); run<8>(d, s); run<16>(d, s); run<32>(d, s); run<64>(d, s); }
  return 0;
}
