// Micro-benchmark: issue rate of the VALU ops the broad-phase scan uses.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench_valu && ./gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned subsat(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
template <int OP>
__global__ void k(unsigned* out, int iters, unsigned seed) {
  unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 17;
  unsigned s = seed;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (OP == 0) { a0 = subsat(a0, s); a1 = subsat(a1, s); a2 = subsat(a2, s); a3 = subsat(a3, s); a4 = subsat(a4, s); a5 = subsat(a5, s); a6 = subsat(a6, s); a7 = subsat(a7, s); }
      if (OP == 1) { a0 = a0 - s; a1 = a1 | s; a2 = a2 - s; a3 = a3 | s; a4 = a4 - s; a5 = a5 | s; a6 = a6 - s; a7 = a7 | s; }
      if (OP == 2) { a0 = min(a0, s); a1 = min(a1, a0); a2 = min(a2, a1); a3 = min(a3, a2); a4 = min(a4, a3); a5 = min(a5, a4); a6 = min(a6, a5); a7 = min(a7, a6); }
      if (OP == 3) { float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1); f0 = f0 * 1.0001f + f1; a0 = __float_as_uint(f0); f1 = f1 * 1.0001f + f0; a1 = __float_as_uint(f1);
                     float f2 = __uint_as_float(a2), f3 = __uint_as_float(a3); f2 = f2 * 1.0001f + f3; a2 = __float_as_uint(f2); f3 = f3 * 1.0001f + f2; a3 = __float_as_uint(f3);
                     float f4 = __uint_as_float(a4), f5 = __uint_as_float(a5); f4 = f4 * 1.0001f + f5; a4 = __float_as_uint(f4); f5 = f5 * 1.0001f + f4; a5 = __float_as_uint(f5);
                     float f6 = __uint_as_float(a6), f7 = __uint_as_float(a7); f6 = f6 * 1.0001f + f7; a6 = __float_as_uint(f6); f7 = f7 * 1.0001f + f6; a7 = __float_as_uint(f7); }
      s += 0x10001u;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP> void run(const char* name, int waves_per_simd) {
  unsigned* out; hipMalloc(&out, 256 * 4 * 64 * 8 * 4 * 4);
  int iters = 4000; int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD per block
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(out, 10, 1); hipDeviceSynchronize();
  hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, iters, 1); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double winstr = (double)blocks * 4 * iters * 16 * 8;     // wave-instructions
  double per_simd = winstr / 1024.0;                         // per SIMD
  printf("%-14s waves/SIMD %d : %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) { run<0>("v_pk_sub_u16", w); run<1>("v_sub/v_or u32", w); run<2>("v_min_u32 dep", w); run<3>("v_fma_f32", w); }
  return 0;
}
