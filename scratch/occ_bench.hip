// VALU issue rate vs resident waves per SIMD on gfx950: how many waves does a SIMD need to reach 1 instr / 2 cycles?
#include <hip/hip_runtime.h>
#include <stdio.h>
enum { FMAC_LIT, FMA_MIX_V, FMA_MIX_S, FMAC_SGPR, MIN3, CVT_PK, MIXED };
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s0) {
  float acc[16], x[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 0.001f + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 1.0f + 1e-7f * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == FMAC_LIT) asm volatile("v_fmac_f32_e32 %0, 0x3f8ccccd, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == FMA_MIX_V) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == FMA_MIX_S) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "s"(s0));
        if (KIND == FMAC_SGPR) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[i]) : "s"(s0), "v"(x[(i + u) & 7]));
        if (KIND == MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == CVT_PK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == MIXED) {
          if (i % 4 == 3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
          else asm volatile("v_fmac_f32_e32 %0, 0x3f8ccccd, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        }
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, int blocks, int iters) {
  float* out; (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  double winstr = (double)blocks * 4 * iters * 64;
  printf("%-12s waves/SIMD=%d: %8.1f us  %.3f wave-instr/ns/SIMD\n", name, blocks / 256, ms * 1e3, winstr / 1024 / (ms * 1e6));
  (void)hipFree(out);
}
#define ALL(KIND) for (int w = 1; w <= 8; w = w < 4 ? w + 1 : w * 2) run<KIND>(#KIND, 256 * w, 16384 / w);
int main() {
  ALL(FMAC_LIT) ALL(FMA_MIX_V) ALL(FMA_MIX_S) ALL(FMAC_SGPR) ALL(MIN3) ALL(CVT_PK) ALL(MIXED)
  return 0;
}
