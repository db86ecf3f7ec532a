// What does v_cvt_pk_u8_f32 do with fractions, out-of-range values and NaN - under the default rounding mode and with
// the single-precision rounding mode of the MODE register set to round-toward-zero?  (isp_mega_cam.h, phase D.)
//   hipcc --offload-arch=gfx950 -O2 scratch/cvt_pk_u8_test.hip -o build/cvt_pk_u8_test && build/cvt_pk_u8_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* x, unsigned* out_rne, unsigned* out_rtz, unsigned* ref, int n) {
  int i = threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  unsigned a = 0, b = 0;
  asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(a) : "v"(v));
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\tv_cvt_pk_u8_f32 %0, %1, 0, %0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "+v"(b) : "v"(v));
  out_rne[i] = a; out_rtz[i] = b;
  float c = fminf(fmaxf(v, 0.f), 255.f);
  ref[i] = (unsigned)c;
}
int main() {
  const float h[] = {0.f, 0.49f, 0.5f, 0.51f, 0.999f, 1.0f, 1.5f, 2.5f, 2.7f, 3.5f, 126.5f, 127.5f, 254.5f, 254.9f, 255.f, 255.4f, 255.5f, 256.f, 300.f, 1e9f, -0.4f, -0.6f, -3.f, NAN, INFINITY, -INFINITY, 99.99999f, 100.f};
  const int n = sizeof(h) / sizeof(float);
  float* d; unsigned *a, *b, *r;
  hipMalloc(&d, sizeof(h)); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&r, n * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, b, r, n);
  unsigned ha[64], hb[64], hr[64];
  hipMemcpy(ha, a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hr, r, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    printf("%14g  default %3u  rtz %3u  clamp+trunc %3u%s\n", h[i], ha[i], hb[i], hr[i], hb[i] == hr[i] ? "" : "   <-- rtz differs");
    bad += hb[i] != hr[i];
  }
  printf("rtz mode == clamp + truncation: %s\n", bad ? "NO" : "yes");
  return 0;
}
