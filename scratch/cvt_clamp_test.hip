#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, unsigned* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 < n) {
    float a = in[2 * i], b = in[2 * i + 1];
    unsigned r;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    out[i] = r;
  }
}
int main() {
  float h[8] = {-0.5f, 0.25f, 1.5f, 0.99999f, 1.0004f, 0.0f, __builtin_nanf(""), 0.333333f};
  float* d; unsigned* o; hipMalloc(&d, 32); hipMalloc(&o, 16); hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, 8); unsigned r[4]; hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i) printf("%08x\n", r[i]);
  return 0;
}
