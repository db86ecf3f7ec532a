// VALU issue rate by encoding / operand kind on gfx950 (wave64): which instruction forms run at
// 1 wave-instr per 2.4 cycles and which at 1 per 4?
#include <hip/hip_runtime.h>
#include <stdio.h>
enum { FMAC_VOP2, FMA_VOP3, FMAC_SGPR, FMA_SGPR, PK_FMA, MUL_VOP2, ADD_VOP2, FMA_VOP3_DIST, CVT, MAX_VOP2, MED3, FMAC_DEP, FMAC_LIT, FMA_VOP3_3SRC, MOV, MUL_SGPR, ADD_SGPR, AND, LSHR, BFE, LSHL_OR, PERM, CVT_F32_U32, CVT_F32_F16, CVT_UB0, MIN3, CNDMASK, ADD_U32, MAD_U24, LOGF, EXPF, RCPF, MIN_VOP2, FMAMK, MUL_LIT, MAX_SELF, PK_MUL, PK_ADD, CVT_PKRTZ, MOV_DPP, FMA_MIX, FMA_MIX_HI, FMA_MIX_LIT, PK_MIN_F16, PK_FMA_F16, CVT_PK_F16, OR_LIT, FMAAK, SUB_LIT, MIN_LIT, MUL_INL, FMAC_INL, CND_VCC, CND_SGPR, CMP_VCC, CMP_SGPR, CMP_CND, MIN_U32, MAX_I32, MIN3_U32, CVT_F16_SDWA, BFI, LSHL_ADD, ADD3, AND_OR, XAD, CMP_CND3, CND_S64, CND_S64_ALT };
template <int KIND, int BODY>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s0, float s1) {
  float acc[8], x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = threadIdx.x * 0.001f + i; x[i] = 1.0f + 1e-7f * (threadIdx.x + i); }
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pa[8], px[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) pa[i] = f2{acc[i], acc[i] + 1};
#pragma unroll
  for (int i = 0; i < 4; ++i) px[i] = f2{x[i], x[i + 4]};
  unsigned long long mask = __builtin_amdgcn_ballot_w64(threadIdx.x & 1), mask2 = __builtin_amdgcn_ballot_w64(threadIdx.x & 2);
  asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(acc[0]), "v"(x[1]) : "vcc");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < BODY / 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == FMAC_VOP2) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == FMA_VOP3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == FMA_VOP3_3SRC) asm volatile("v_fma_f32 %0, %1, %2, %3" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]), "v"(x[(i + 5) & 7]));
        if (KIND == FMAC_SGPR) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[i]) : "s"(s0), "v"(x[(i + u) & 7]));
        if (KIND == FMA_SGPR) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(s0), "v"(x[(i + u) & 7]));
        if (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i]) : "v"(px[(i + u) & 3]), "v"(px[(i + 1) & 3]));
        if (KIND == MUL_VOP2) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == ADD_VOP2) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == MAX_VOP2) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == CVT) asm volatile("v_cvt_f16_f32_e32 %0, %0" : "+v"(acc[i]));
        if (KIND == FMAC_DEP) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[0]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == FMAC_LIT) asm volatile("v_fmac_f32_e32 %0, 0x3f8ccccd, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == MUL_SGPR) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(acc[i]) : "s"(s0));
        if (KIND == ADD_SGPR) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(acc[i]) : "s"(s0));
        if (KIND == AND) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == LSHR) asm volatile("v_lshrrev_b32_e32 %0, 3, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == BFE) asm volatile("v_bfe_u32 %0, %1, 4, 12" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %1, 8, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == PERM) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == CVT_F32_U32) asm volatile("v_cvt_f32_u32_e32 %0, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == CVT_F32_F16) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == CVT_UB0) asm volatile("v_cvt_f32_ubyte0_e32 %0, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == CNDMASK) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == ADD_U32) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == MAD_U24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == LOGF) asm volatile("v_log_f32_e32 %0, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == EXPF) asm volatile("v_exp_f32_e32 %0, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == RCPF) asm volatile("v_rcp_f32_e32 %0, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == MIN_VOP2) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == FMAMK) asm volatile("v_fmamk_f32 %0, %1, 0x3d800000, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == MUL_LIT) asm volatile("v_mul_f32_e32 %0, 0x3d800000, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == MAX_SELF) asm volatile("v_max_f32_e32 %0, %0, %0" : "+v"(acc[i]));
        if (KIND == PK_MUL) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(pa[i]) : "v"(px[(i + u) & 3]));
        if (KIND == PK_ADD) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(pa[i]) : "v"(px[(i + u) & 3]));
        if (KIND == CVT_PKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == FMA_MIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == FMA_MIX_HI) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == FMA_MIX_LIT) asm volatile("v_fma_mix_f32 %0, %1, 0.5, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == PK_MIN_F16) asm volatile("v_pk_min_f16 %0, %1, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == CVT_PK_F16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == OR_LIT) asm volatile("v_or_b32_e32 %0, 0x4b000000, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == FMAAK) asm volatile("v_fmaak_f32 %0, %1, %0, 0x3d800000" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == SUB_LIT) asm volatile("v_subrev_f32_e32 %0, 0x4b000000, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == MIN_LIT) asm volatile("v_min_f32_e32 %0, 1.0, %0" : "+v"(acc[i]));
        if (KIND == MUL_INL) asm volatile("v_mul_f32_e32 %0, 0.5, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == FMAC_INL) asm volatile("v_fmac_f32_e32 %0, 0.5, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == CND_VCC) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(x[(i + u) & 7]) : );
        if (KIND == CMP_VCC) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(acc[i]), "v"(x[(i + u) & 7]) : "vcc");
        if (KIND == CMP_CND) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(x[(i + u) & 7]) : "vcc");
        if (KIND == MIN_U32) asm volatile("v_min_u32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == MAX_I32) asm volatile("v_max_i32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == MIN3_U32) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == CVT_F16_SDWA) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == BFI) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
        if (KIND == ADD3) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == AND_OR) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == XAD) asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(x[(i + 3) & 7]));
        if (KIND == CMP_CND3) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %0, %1, %0, vcc\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(x[(i + u) & 7]) : "vcc");
        if (KIND == CND_S64) { float tmp; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(tmp) : "v"(acc[i]), "v"(x[(i + u) & 7]), "s"(mask)); acc[i] = tmp; }
        if (KIND == CND_S64_ALT) { float tmp; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(tmp) : "v"(acc[i]), "v"(x[(i + u) & 7]), "s"((i & 1) ? mask : mask2)); acc[i] = tmp; }
        if (KIND == MOV) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(acc[i]) : "v"(x[(i + u) & 7]));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i] + pa[i].x + pa[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)((mask ^ mask2) & 1);
}
template <int KIND, int BODY> void run(const char* name, int blocks, int total) {
  float* out; (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int iters = total / BODY;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<KIND, BODY>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<KIND, BODY>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  double winstr = (double)blocks * 4 * iters * BODY;
  printf("%-34s body=%4d blocks=%d: %8.1f us  %.3f wave-instr/ns/SIMD  (%.2f cycles/instr @2.4GHz)\n", name, BODY, blocks, ms * 1e3,
         winstr / 1024 / (ms * 1e6), 2.4 / (winstr / 1024 / (ms * 1e6)));
  (void)hipFree(out);
}
#define ALL(KIND) run<KIND, 64>(#KIND, 2048, 16384); run<KIND, 64>(#KIND " 4 waves/SIMD", 1024, 16384);
int main() {
  ALL(CND_VCC) ALL(CMP_CND) ALL(CMP_CND3) ALL(CND_S64) ALL(CND_S64_ALT) ALL(MOV)
  return 0;
}
