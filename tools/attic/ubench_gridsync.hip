// Micro-benchmark: cost of a cooperative-groups grid barrier (hipLaunchCooperativeKernel).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_gridsync.hip -o tools/ubench_gridsync.bin
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ void k(unsigned* p, int nsync) {
  cg::grid_group g = cg::this_grid();
  unsigned v = 0;
  for (int i = 0; i < nsync; ++i) {
    if (threadIdx.x == 0) p[blockIdx.x] = v + i;
    g.sync();
    v += p[(blockIdx.x + 1) % gridDim.x];
  }
  if (threadIdx.x == 0) p[blockIdx.x] = v;
}
int main() {
  unsigned* p; (void)hipMalloc(&p, 1 << 20); (void)hipMemset(p, 0, 1 << 20);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int blocks : {64, 256, 1024}) for (int threads : {256, 1024}) {
    if (blocks * threads > 256 * 2048) continue;
    float t[2];
    int idx = 0;
    for (int nsync : {1, 21}) {
      void* args[] = {&p, &nsync};
      for (int w = 0; w < 3; ++w) (void)hipLaunchCooperativeKernel((void*)k, dim3(blocks), dim3(threads), args, 0, 0);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      for (int r = 0; r < 50; ++r) (void)hipLaunchCooperativeKernel((void*)k, dim3(blocks), dim3(threads), args, 0, 0);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&t[idx++], e0, e1);
    }
    hipError_t err = hipGetLastError();
    printf("blocks %4d x %4d threads: launch+1 sync %.2f us, per extra grid.sync %.2f us  (%s)\n", blocks, threads, t[0] * 1e3 / 50, (t[1] - t[0]) * 1e3 / 50 / 20, hipGetErrorString(err));
  }
  return 0;
}
