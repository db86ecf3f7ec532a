// Store-pattern test: 75.5 MB written as (a) 3 x 16 B per lane at 48-B lane stride (strip rows),
// (b) 3 instructions each writing 1 KB contiguous per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint4* out, size_t n16, uint4 val) {
  // each wave handles 3 KB chunks
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const size_t nw = (size_t)gridDim.x * 4;
  for (size_t c = wave; c * 192 < n16; c += nw) {
    uint4* base = out + c * 192;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) base[lane * 3 + j] = val;       // lane stride 48 B
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j) base[j * 64 + lane] = val;      // contiguous 1 KB per instruction
    }
  }
}
template <int MODE> void run(const char* name, int blocks) {
  const size_t bytes = 75497472, n16 = bytes / 16;
  uint4* out; (void)hipMalloc(&out, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, n16, make_uint4(1, 2, 3, 4));
  (void)hipEventRecord(e0);
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, n16, make_uint4(1, 2, 3, 4));
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  printf("%-34s blocks=%d: %.1f us, %.2f TB/s\n", name, blocks, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
}
int main() {
  for (int b : {1024, 2048, 6144}) { run<0>("48-B lane stride (3 x 16 B)", b); run<1>("contiguous 1 KB per instruction", b); }
  return 0;
}
