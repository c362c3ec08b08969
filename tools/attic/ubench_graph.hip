// Micro-benchmark: 11 dependent tiny kernels, launched one by one vs replayed as a hipGraph.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_graph.hip -o tools/ubench_graph.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k(unsigned* p, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) p[i] = p[i ? i - 1 : 0] + 1; }
int main() {
  unsigned* p; (void)hipMalloc(&p, 256); (void)hipMemset(p, 0, 256);
  hipStream_t st; (void)hipStreamCreate(&st);
  const int N = 11, reps = 200;
  auto launch_all = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, st, p, i); };
  for (int i = 0; i < 5; ++i) launch_all();
  (void)hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { launch_all(); (void)hipStreamSynchronize(st); }
  auto t1 = std::chrono::steady_clock::now();
  printf("stream launches : %.1f us per %d-kernel chain (incl. sync)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, N);
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  launch_all();
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 5; ++i) (void)hipGraphLaunch(ge, st);
  (void)hipStreamSynchronize(st);
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st); }
  t1 = std::chrono::steady_clock::now();
  printf("graph replay    : %.1f us per %d-kernel chain (incl. sync)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, N);
  return 0;
}
