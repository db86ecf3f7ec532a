// Issue rate of straight-line code vs the size of the loop body on gfx950 (64 KB instruction cache per 2 CUs):
// what does a phase of ~70-90 KB of unrolled code cost against the same instructions inside a loop that fits?
#include <hip/hip_runtime.h>
#include <stdio.h>
// 16 instructions per R16: 8-byte VOP3 (v_fma_f32) or 4-byte VOP2 (v_fmac_f32_e32)
#define I3(a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[a]) : "v"(x[b]), "v"(x[(b + 3) & 7]));
#define I2(a, b) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[a]) : "v"(x[b]), "v"(x[(b + 3) & 7]));
#define R16(I) I(0,0) I(1,1) I(2,2) I(3,3) I(4,4) I(5,5) I(6,6) I(7,7) I(8,1) I(9,2) I(10,3) I(11,4) I(12,5) I(13,6) I(14,7) I(15,0)
#define R128(I) R16(I) R16(I) R16(I) R16(I) R16(I) R16(I) R16(I) R16(I)
template <int CHUNKS, bool VOP3>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float acc[16], x[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 0.001f + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 1.0f + 1e-7f * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      if (VOP3) { R128(I3) } else { R128(I2) }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHUNKS, bool VOP3> void run(int waves_per_simd) {
  int blocks = 256 * waves_per_simd;
  int iters = (1 << 15) / CHUNKS / waves_per_simd;     // 4 M instructions per SIMD in every configuration
  float* out; (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<CHUNKS, VOP3>), dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<CHUNKS, VOP3>), dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double winstr = (double)blocks * 4 * iters * CHUNKS * 128;
  double per_simd = winstr / 1024;
  printf("%s body %4d KB  waves/SIMD=%d: %8.1f us  %.3f instr/ns/SIMD  %.2f B/ns/SIMD\n", VOP3 ? "VOP3" : "VOP2",
         CHUNKS * (VOP3 ? 1 : 1) * 128 * (VOP3 ? 8 : 4) / 1024, waves_per_simd, ms * 1e3, per_simd / (ms * 1e6),
         per_simd * (VOP3 ? 8 : 4) / (ms * 1e6));
  (void)hipFree(out);
}
#define BOTH(C) run<C, true>(1); run<C, true>(2); run<C, false>(2);
int main() {
  BOTH(1) BOTH(4) BOTH(16) BOTH(32) BOTH(48) BOTH(64) BOTH(96) BOTH(128) BOTH(256)
  return 0;
}
