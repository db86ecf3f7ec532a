// Section microbenchmarks: the strip code of the tile kernel in isolation (registers only), looped,
// to find the achievable instruction issue rate of each section at a given occupancy.
#include "../taichi_image_amd/csrc/isp_tile.h"
#include <stdio.h>
void mi_set_error(const char*, ...) {}
using namespace tile;

template <int MODE, int LDSBYTES>
__global__ __launch_bounds__(256) void sec_kernel(Params p, float* out, int iters) {
  __shared__ float pad[LDSBYTES / 4];
  if (threadIdx.x == 9999) pad[0] = 1.f;
  float win[6][12];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 12; ++b) win[a][b] = (float)((threadIdx.x * 7 + a * 12 + b) & 255) * (1.0f / 256.0f);
  float sink = 0.f;
  ReinhardK rk; rk.map_key = p.la; rk.ei = 0.37f; rk.mean3[0] = rk.mean3[1] = rk.mean3[2] = 0.4f; rk.la = 1.f; rk.ca = 0.f;
  StatsAcc st; st.init();
  float vmin = 1e9f, vmax = -1e9f;
  for (int it = 0; it < iters; ++it) {
    static_for<0, 2>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      float v[24];
      static_for<0, 8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int KIDX = (i & 1) + 2 * (k & 1);
        float acc[3];
        if (MODE != 2) accumulate<KIDX, true, i, k>(p.wq, win, acc);
        else { acc[0] = win[i][k]; acc[1] = win[i + 1][k + 1]; acc[2] = win[i + 2][k + 2]; }
        v[3 * k] = acc[0]; v[3 * k + 1] = acc[1]; v[3 * k + 2] = acc[2];
      });
      if (MODE == 0) {            // accumulate + min/max (pass 0)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          vmin = fminf(vmin, fminf(v[3 * k], fminf(v[3 * k + 1], v[3 * k + 2])));
          vmax = fmaxf(vmax, fmaxf(v[3 * k], fmaxf(v[3 * k + 1], v[3 * k + 2])));
        }
      } else {                    // MODE 1: full pass-2 style epilogue; MODE 2: epilogue only
#pragma unroll
        for (int j = 0; j < 24; ++j) v[j] = clamp01(v[j]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float t[3], q[3];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) t[ch] = norm01((float)cast_out<half_t>(v[3 * k + ch]), p.in_scale, p.out_scale);
          reinhard_px<true>(t, rk, q);
          vmin = fminf(vmin, fminf(q[0], fminf(q[1], q[2])));
          vmax = fmaxf(vmax, fmaxf(q[0], fmaxf(q[1], q[2])));
        }
      }
    });
    // perturb the window so iterations are not hoisted
#pragma unroll
    for (int b = 0; b < 12; ++b) { win[0][b] += vmin * 1e-9f; win[3][b] += vmax * 1e-9f; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = vmin + vmax + sink;
}

template <int MODE, int LDSBYTES> void run(const char* name, int instr_guess) {
  Params p = {}; set_weights(p); p.in_scale = 0.01f; p.out_scale = 1.1f; p.la = 0.8f;
  float* out; (void)hipMalloc(&out, 3072 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 16;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((sec_kernel<MODE, LDSBYTES>), dim3(3072), dim3(256), 0, 0, p, out, iters);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((sec_kernel<MODE, LDSBYTES>), dim3(3072), dim3(256), 0, 0, p, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("%-34s lds=%6d: %.1f us per launch = %.2f us per strip-iteration-pass-equivalent (x3072 blocks)\n", name, LDSBYTES, ms * 1e3, ms * 1e3 / iters);
  (void)hipFree(out);
}
int main() {
  run<0, 1024>("accumulate+minmax occ max", 0);
  run<0, 40000>("accumulate+minmax 4 blk/CU", 0);
  run<0, 53000>("accumulate+minmax 3 blk/CU", 0);
  run<1, 1024>("accumulate+reinhard epi occ max", 0);
  run<1, 40000>("accumulate+reinhard epi 4 blk/CU", 0);
  run<2, 1024>("reinhard epilogue only occ max", 0);
  run<2, 40000>("reinhard epilogue only 4 blk/CU", 0);
  return 0;
}
