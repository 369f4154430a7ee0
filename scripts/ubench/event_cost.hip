// What does a fork point cost the stream that records it?  A chain of N short kernels on stream A, with after every kernel
// (a) nothing, (b) hipEventRecord + hipStreamWaitEvent(B) (+ a kernel on B), (c) the kernel launched through
// hipExtLaunchKernelGGL with the event as its stop event + hipStreamWaitEvent(B) (+ a kernel on B).
//   hipcc --offload-arch=gfx950 -O2 event_cost.hip -o event_cost && ./event_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>
__global__ void spin(float* p, int n) {
  float v = p[threadIdx.x];
  for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
int main() {
  float *a, *b;
  hipMalloc(&a, 4096); hipMalloc(&b, 4096);
  hipMemset(a, 0, 4096); hipMemset(b, 0, 4096);
  hipStream_t A, B;
  hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
  int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithPriority(&B, hipStreamNonBlocking, lo);
  const int N = 40;
  hipEvent_t ev[N];
  for (int i = 0; i < N; ++i) hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
  hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(t0, A);
      for (int i = 0; i < N; ++i) {
        if (mode == 2 || mode == 4) hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, A, nullptr, ev[i], 0, a, 2000);
        else hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, A, a, 2000);
        if (mode == 1 || mode == 3) hipEventRecord(ev[i], A);
        if (mode >= 1) hipStreamWaitEvent(B, ev[i], 0);
        if (mode == 3 || mode == 4) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, B, b, 2000);
      }
      hipEventRecord(t1, A);
      hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, t0, t1);
      if (rep == 2) printf("mode %d (%s): %.2f us per kernel on A\n", mode,
                           mode == 0 ? "plain chain" : mode == 1 ? "record + wait, B idle" : mode == 2 ? "ext-launch stop event + wait, B idle"
                           : mode == 3 ? "record + wait + kernel on B" : "ext-launch stop event + wait + kernel on B", ms * 1e3 / N);
    }
  }
  // ordering check of the stop-event fork: A runs a LONG kernel that ends by writing a flag value; B (waiting on the stop event)
  // copies the flag.  A stale copy = the wait did not hold.
  int bad = 0;
  for (int it = 0; it < 50; ++it) {
    hipMemsetAsync(a, 0, 4096, A); hipMemsetAsync(b, 0, 4096, A);
    hipDeviceSynchronize();
    hipExtLaunchKernelGGL(spin, dim3(1), dim3(256), 0, A, nullptr, ev[it % N], 0, a, 200000 + 1000 * it);
    hipError_t rc = hipGetLastError();
    if (rc != hipSuccess) { printf("ext launch: %s\n", hipGetErrorString(rc)); return 1; }
    rc = hipStreamWaitEvent(B, ev[it % N], 0);
    if (rc != hipSuccess) { printf("wait: %s\n", hipGetErrorString(rc)); return 1; }
    hipMemcpyAsync(b, a, 1024, hipMemcpyDeviceToDevice, B);
    hipStreamSynchronize(B);
    float hb[256]; hipMemcpy(hb, b, 1024, hipMemcpyDeviceToHost);
    hipDeviceSynchronize();
    float ha[256]; hipMemcpy(ha, a, 1024, hipMemcpyDeviceToHost);
    if (hb[0] != ha[0] || ha[0] == 0.0f) ++bad;
  }
  printf("stop-event ordering: %d of 50 stale\n", bad);
  return 0;
}
