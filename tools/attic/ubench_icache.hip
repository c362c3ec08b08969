// Micro-benchmark: cost of running LONG STRAIGHT-LINE code once per wave (cold
// instruction cache after every kernel launch) vs. the same number of instructions in a loop.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_icache.hip -o tools/ubench_icache.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N, int PASSES>
__global__ void straight(float* out, float a, float b) {
  float x = threadIdx.x * a, y = b;
  for (int p = 0; p < PASSES; ++p) {
#pragma unroll
    for (int i = 0; i < N; ++i) { x = x * y + (float)(i + 1); y = y * x + a; }   // 2 N distinct v_fma (different literals)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
__global__ void looped(float* out, float a, float b, int n) {
  float x = threadIdx.x * a, y = b;
#pragma unroll 1
  for (int i = 0; i < n; ++i) { x = x * y + 1.5f; y = y * x + a; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / 50;
}
int main() {
  float* out; (void)hipMalloc(&out, 1 << 24);
  for (int G : {256, 1024, 4096}) {
    printf("grid %d x 64 threads:\n", G);
    printf("  loop      1000 iters (2000 fma)        : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(looped, dim3(G), dim3(64), 0, 0, out, 1.0001f, 0.5f, 1000); }));
    printf("  straight  2000 fma (16 KB), 1 pass      : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((straight<1000, 1>), dim3(G), dim3(64), 0, 0, out, 1.0001f, 0.5f); }));
    printf("  straight  2000 fma (16 KB), 2 passes    : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((straight<1000, 2>), dim3(G), dim3(64), 0, 0, out, 1.0001f, 0.5f); }));
    printf("  straight  500 fma (4 KB), 1 pass        : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((straight<250, 1>), dim3(G), dim3(64), 0, 0, out, 1.0001f, 0.5f); }));
    printf("  straight  500 fma (4 KB), 4 passes      : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((straight<250, 4>), dim3(G), dim3(64), 0, 0, out, 1.0001f, 0.5f); }));
    printf("  loop      250 iters (500 fma)           : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(looped, dim3(G), dim3(64), 0, 0, out, 1.0001f, 0.5f, 250); }));
  }
  return 0;
}
