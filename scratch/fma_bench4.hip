// Operand-shape test: FMA forms at equal count, registers only, loop body ~256 instr.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s0, float s1, float s2, float s3) {
  float x[32], acc[8];
  f2 xp[16], accp[4];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = threadIdx.x * 0.001f + i;
#pragma unroll
  for (int i = 0; i < 16; ++i) xp[i] = f2{x[2 * i], x[2 * i + 1]};
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = i;
#pragma unroll
  for (int i = 0; i < 4; ++i) accp[i] = f2{(float)i, (float)i + 0.5f};
  const float sw[4] = {s0, s1, s2, s3};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (MODE == 0) {          // v_fmac v_acc, s_w, v_x : two VGPR sources
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(x[(u + i * 3) & 31], sw[(u + i) & 3], acc[i]);
      } else if (MODE == 1) {   // fma(acc, s, s): one VGPR source
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(acc[i], sw[(u + i) & 3], sw[(u + i + 1) & 3]);
      } else {                  // packed: v_pk_fma_f32, 2 FMAs per instruction (4 instr here = 8 FMAs)
#pragma unroll
        for (int i = 0; i < 4; ++i) accp[i] = __builtin_elementwise_fma(xp[(u + i * 3) & 15], f2{sw[(u + i) & 3], sw[(u + i) & 3]}, accp[i]);
      }
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) r += accp[i].x + accp[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, int fma_per_iter, int blocks) {
  float* out; (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int iters = 64;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.01f, 0.99f, 0.5f, 0.25f);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.01f, 0.99f, 0.5f, 0.25f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  double fmas = (double)blocks * 4 * iters * fma_per_iter;   // wave-level FMAs
  printf("%-28s blocks=%d: %.1f us, %.3f wave-FMA/ns/SIMD\n", name, blocks, ms * 1e3, fmas / 1024 / (ms * 1e6));
  (void)hipFree(out);
}
int main() {
  for (int blocks : {3072, 1024}) {
    run<0>("fmac v,s,v (2 VGPR src)", 256, blocks);
    run<1>("fma v,s,s (1 VGPR src)", 256, blocks);
    run<2>("pk_fma (2 FMA/instr)", 256, blocks);
  }
  return 0;
}
