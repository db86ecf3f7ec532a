// Cost per instruction class on gfx950 at 1 and 2 waves per SIMD (what a 256-VGPR kernel gets).
#include <hip/hip_runtime.h>
#include <stdio.h>
enum { FMA, MOV, PK_FMA, PK_MUL, PK_ADD, MAXF, MIN3, CNDMASK, CVT_F32_F16, CVT_PK, LOGF, RCPF, EXPF, ANDB, LSHR, PERM, MOV_DPP,
       FMAMK, MUL, ADD, TRANS_FMA_MIX, LOG_DEP, MED3, CVT_SDWA, PK_FMA_F16, MAX_PK_F16, FMA_DEP1, FMA_DEP2, FMA_DEP4, MUL_DEP1, CNDMASK_E64, NKIND };
const char* names[] = {"v_fma_f32", "v_mov_b32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_max_f32_e32", "v_min3_f32", "v_cndmask_e32",
  "v_cvt_f32_f16", "v_cvt_pk_f16_f32", "v_log_f32", "v_rcp_f32", "v_exp_f32", "v_and_b32", "v_lshrrev_b32", "v_perm_b32", "v_mov_dpp",
  "v_fmamk_f32", "v_mul_f32_e32", "v_add_f32_e32", "1 log + 3 fma interleaved (per 4)", "v_log dependent chain + fma", "v_med3_f32", "v_cvt_f32_f16_sdwa",
  "v_pk_fma_f16", "v_pk_max_f16", "v_fma_f32 chain: every instr depends on previous", "v_fma_f32 2 chains", "v_fma_f32 4 chains", "v_mul_f32_e32 chain", "v_cndmask_b32_e64 (sgpr pair mask)"};
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float acc[16], x[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pa[8], px[4];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 0.001f + i + 1.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = 1.0f + 1e-7f * (threadIdx.x + i); pa[i] = f2{acc[i], acc[i + 8]}; }
#pragma unroll
  for (int i = 0; i < 4; ++i) px[i] = f2{x[i], x[i + 4]};
  unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)iters;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int j = (i + u) & 7, j2 = (i + 3) & 7;
        if (KIND == FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == MOV) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(acc[i]) : "v"(x[j]));
        if (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i & 7]) : "v"(px[j & 3]), "v"(px[j2 & 3]));
        if (KIND == PK_MUL) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(pa[i & 7]) : "v"(px[j & 3]), "v"(px[j2 & 3]));
        if (KIND == PK_ADD) asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(pa[i & 7]) : "v"(px[j & 3]), "v"(px[j2 & 3]));
        if (KIND == MAXF) asm volatile("v_max_f32_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == CNDMASK) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == CVT_F32_F16) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "+v"(acc[i]) : "v"(x[j]));
        if (KIND == CVT_SDWA) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(acc[i]) : "v"(x[j]));
        if (KIND == CVT_PK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == LOGF) asm volatile("v_log_f32_e32 %0, %1" : "+v"(acc[i]) : "v"(x[j]));
        if (KIND == RCPF) asm volatile("v_rcp_f32_e32 %0, %1" : "+v"(acc[i]) : "v"(x[j]));
        if (KIND == EXPF) asm volatile("v_exp_f32_e32 %0, %1" : "+v"(acc[i]) : "v"(x[j]));
        if (KIND == ANDB) asm volatile("v_and_b32_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == LSHR) asm volatile("v_lshrrev_b32_e32 %0, 3, %1" : "+v"(acc[i]) : "v"(x[j]));
        if (KIND == PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]), "v"(x[(i + 5) & 7]));
        if (KIND == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(x[j]));
        if (KIND == FMAMK) asm volatile("v_fmamk_f32 %0, %1, 0x3f8ccccd, %2" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == MUL) asm volatile("v_mul_f32_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == ADD) asm volatile("v_add_f32_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == MAX_PK_F16) asm volatile("v_pk_max_f16 %0, %1, %2" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == FMA_DEP1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == FMA_DEP2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i & 1]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == FMA_DEP4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(x[j]), "v"(x[j2]));
        if (KIND == MUL_DEP1) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(acc[0]) : "v"(x[j]));
        if (KIND == CNDMASK_E64) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]), "s"(mask));
        if (KIND == TRANS_FMA_MIX) {
          if (i % 4 == 0) asm volatile("v_log_f32_e32 %0, %1" : "+v"(acc[i]) : "v"(x[j]));
          else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        }
        if (KIND == LOG_DEP) {
          if (i % 4 == 0) asm volatile("v_log_f32_e32 %0, %0" : "+v"(acc[0]));
          else if (i % 4 == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(x[j]), "v"(x[j2]));
          else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[j]), "v"(x[j2]));
        }
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += pa[i].x + pa[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> void run(int w) {
  int blocks = 256 * w, iters = 8192 / w;
  float* out; (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double per_simd = (double)blocks * 4 * iters * 64 / 1024;
  printf("%-40s waves/SIMD=%d: %.3f instr/ns/SIMD = %.2f cycles at 2.1 GHz\n", names[KIND], w, per_simd / (ms * 1e6), 2.1 * ms * 1e6 / per_simd);
  (void)hipFree(out);
}
template <int K> struct All { static void go() { run<K>(1); run<K>(2); All<K + 1>::go(); } };
template <> struct All<NKIND> { static void go() {} };
int main() { All<0>::go(); return 0; }
