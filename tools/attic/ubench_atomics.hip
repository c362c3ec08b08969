// Micro-benchmark: rate of scattered 32-bit atomic adds on a table in device memory, by scope and with / without the
// returned value -- the operation that bounds nms_ginsert_kernel.  Agent scope is what the library uses
// (any workgroup may hit any cell); workgroup scope is executed by the L2 of the XCD the wave runs on and would only be
// CORRECT if every workgroup touching a cell ran on one XCD -- measured here to see what an XCD-partitioned table could buy.
// hipcc --offload-arch=gfx950 -O3 tools/attic/ubench_atomics.hip -o gpurun_out/ubench_atomics && ./gpurun_out/ubench_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int SCOPE, bool RET>
__global__ void k(uint32_t *tab, uint32_t mask, int per_thread, uint32_t *sink) {
  uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
  for (int i = 0; i < per_thread; ++i) {
    x = x * 1664525u + 1013904223u;
    const uint32_t a = (x >> 8) & mask;
    if (RET) acc += __hip_atomic_fetch_add(tab + a, 1u, __ATOMIC_RELAXED, SCOPE);
    else __hip_atomic_fetch_add(tab + a, 1u, __ATOMIC_RELAXED, SCOPE);
  }
  if (RET && acc == 0xFFFFFFFFu) sink[0] = acc;
}
template <int SCOPE, bool RET> void run(const char *name, uint32_t *tab, uint32_t cells, uint32_t *sink) {
  const int blocks = 1024, threads = 256, per = 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SCOPE, RET>), dim3(blocks), dim3(threads), 0, 0, tab, cells - 1, per, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<SCOPE, RET>), dim3(blocks), dim3(threads), 0, 0, tab, cells - 1, per, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ops = 10.0 * blocks * threads * per;
  printf("%-44s table %8u cells: %7.1f atomics / ns (%.1f us per launch of %.1f M)\n", name, cells, ops / (ms * 1e6), ms * 100.0, ops / 10 / 1e6);
}
int main() {
  uint32_t *tab, *sink; hipMalloc(&tab, (size_t)(1u << 24) * 4); hipMalloc(&sink, 64); hipMemset(tab, 0, (size_t)(1u << 24) * 4);
  for (uint32_t cells : {1u << 14, 1u << 16, 1u << 20, 1u << 24}) {
    run<__HIP_MEMORY_SCOPE_AGENT, true>("agent scope, returning", tab, cells, sink);
    run<__HIP_MEMORY_SCOPE_AGENT, false>("agent scope, no return", tab, cells, sink);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, true>("workgroup scope (XCD-local L2), returning", tab, cells, sink);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, false>("workgroup scope (XCD-local L2), no return", tab, cells, sink);
  }
  return 0;
}
