// Loop-body size sweep: at what body size does instruction fetch start to limit VALU issue?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CH, int UNROLL, bool LIT>
__global__ __launch_bounds__(256) void fma_kernel(float* out, int iters, float a, float b) {
  float acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) acc[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int i = 0; i < CH; ++i)
        acc[i] = LIT ? __builtin_fmaf(acc[i], 1.0f + 1e-6f * (float)(u * CH + i), b)   // 8-byte v_fmaak/v_fmamk
                     : __builtin_fmaf(acc[i], a, b);                                      // 4-byte v_fmac / 8-byte v_fma
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH, int UNROLL, bool LIT> void run(int blocks, int total_per_wave) {
  float* out; (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int iters = total_per_wave / (UNROLL * CH);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((fma_kernel<CH, UNROLL, LIT>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((fma_kernel<CH, UNROLL, LIT>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  double winstr = (double)blocks * 4 * iters * UNROLL * CH;
  printf("body %4d instr lit=%d blocks=%d: %.1f us, per SIMD %.3f instr/ns\n", UNROLL * CH, (int)LIT, blocks, ms * 1e3, winstr / 1024 / (ms * 1e6));
  (void)hipFree(out);
}
int main() {
  run<8, 8, true>(3072, 8192); run<8, 16, true>(3072, 8192); run<8, 32, true>(3072, 8192); run<8, 64, true>(3072, 8192);
  run<8, 128, true>(3072, 8192); run<8, 256, true>(3072, 8192);
  run<8, 8, false>(3072, 8192); run<8, 32, false>(3072, 8192); run<8, 128, false>(3072, 8192); run<8, 512, false>(3072, 8192);
  run<8, 128, true>(1024, 8192); run<8, 128, true>(256, 8192);
  return 0;
}
