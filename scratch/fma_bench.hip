// VALU issue-rate calibration: N independent FMA chains per lane, fully unrolled inner body.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CH>
__global__ __launch_bounds__(256) void fma_kernel(float* out, int iters, float a, float b) {
  float acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) acc[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH> void run(const char* name, int blocks) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 512;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(fma_kernel<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(fma_kernel<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  double winstr = (double)blocks * 4 * iters * 8 * CH;  // wave-instructions
  printf("%s blocks=%d: %.1f us, %.2f G wave-instr/s, per SIMD %.3f instr/ns\n", name, blocks, ms * 1e3, winstr / ms / 1e6,
         winstr / 1024 / (ms * 1e6));
  hipFree(out);
}
int main() {
  run<1>("chains=1", 256 * 8); run<4>("chains=4", 256 * 8); run<8>("chains=8", 256 * 8); run<8>("chains=8", 256 * 4);
  run<8>("chains=8 1wave/simd", 256);
  return 0;
}
