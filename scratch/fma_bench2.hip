// Straight-line vs looped VALU code: is instruction fetch a limit for long unrolled bodies?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CH, int UNROLL>
__global__ __launch_bounds__(256) void fma_kernel(float* out, int iters, float a, float b) {
  float acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) acc[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[i] = __builtin_fmaf(acc[i], 1.0f + 1e-6f * (float)(u * CH + i), b);   // distinct literal per instr
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH, int UNROLL> void run(const char* name, int blocks, int total_per_wave) {
  float* out; (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int iters = total_per_wave / (UNROLL * CH);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((fma_kernel<CH, UNROLL>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((fma_kernel<CH, UNROLL>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  double winstr = (double)blocks * 4 * iters * UNROLL * CH;
  printf("%-28s blocks=%d iters=%d: %.1f us, per SIMD %.3f instr/ns\n", name, blocks, iters, ms * 1e3, winstr / 1024 / (ms * 1e6));
  (void)hipFree(out);
}
int main() {
  run<8, 8>("loop body 64", 3072, 1024);
  run<8, 128>("straight-line 1024", 3072, 1024);
  run<8, 8>("loop body 64 x4096", 3072, 4096);
  run<8, 512>("straight-line 4096", 3072, 4096);
  run<8, 128>("body 1024 looped 4x", 3072, 4096);
  return 0;
}
