// Micro-benchmark: how long does the GPU need to START and RETIRE workgroups?
// An (almost) empty kernel -- every wave reads one word and, if it is non-zero (never),
// writes -- is launched with G workgroups of T threads (optionally with an LDS
// allocation); 200 back-to-back launches, time per launch.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_dispatch.hip -o /tmp/ubench_dispatch && /tmp/ubench_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS>
__global__ void k(const unsigned* flag, unsigned* out) {
  __shared__ unsigned sm[LDS > 0 ? LDS : 1];
  if (LDS > 0) sm[threadIdx.x] = threadIdx.x;
  if (flag[0] != 0u) out[blockIdx.x * blockDim.x + threadIdx.x] = (LDS > 0) ? sm[(threadIdx.x + 1) % LDS] : 1u;
}
__global__ void knoload(unsigned* out, unsigned never) {
  if (never == 12345u) out[blockIdx.x * blockDim.x + threadIdx.x] = 1u;
}
int main() {
  unsigned *flag, *out;
  hipMalloc(&flag, 256); hipMemset(flag, 0, 256);
  hipMalloc(&out, 1 << 26);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 200;
  printf("%8s %6s %6s | %9s %12s\n", "blocks", "thr", "lds", "us/launch", "ns/workgroup");
  for (int variant = 0; variant < 3; ++variant)
    for (int T : {64, 256, 512, 1024})
      for (int G : {1, 256, 1024, 2048, 4096, 8192, 16384, 65536}) {
        if ((long long)G * T > (1 << 24)) continue;
        auto launch = [&]() {
          if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(T), 0, 0, flag, out);
          else if (variant == 1) hipLaunchKernelGGL(k<2560>, dim3(G), dim3(T), 0, 0, flag, out);
          else hipLaunchKernelGGL(knoload, dim3(G), dim3(T), 0, 0, out, 0u);
        };
        for (int i = 0; i < 5; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%8d %6d %6s | %9.2f %12.2f\n", G, T, variant == 0 ? "0" : variant == 1 ? "10KB" : "noload", ms * 1e3 / reps, ms * 1e6 / reps / G);
      }
  return 0;
}
